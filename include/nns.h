/*
 * nns.h — C ABI of the MI355X-native brute-force nearest-neighbour engine.
 *
 * This is the drop-in boundary for the reference's "distance matrix + argmin"
 * path (sty-hhh/NNS-CUDA, V0-V9).  Every entry point is plain C: pointers and
 * sizes, no C++/torch types.  The shared library is libnns_mi355x.so
 * (nns-cuda_amd/csrc, hand-written HIP for gfx950).
 *
 * Reference interface replaced (file:line are into the reference tree):
 *   - vN::cudaCall(int k, int m, int n, float *s_points, float *r_points,
 *                  int **results)                     core.cu:23-29 (V0) and the
 *     identical signatures at core.cu:123, 177, 258, 350, 427, 538, 634, 761,
 *     965; selected through the function pointer of main.cu:7, called at
 *     main.cu:74.                                  -> nns_search_f32()
 *   - the per-GPU shard body of V8/V9 (core.cu:778-829, 982-1033): upload a
 *     contiguous ref shard, transpose (mat_inv_kernel core.cu:293-306), run the
 *     fused distance+argmin kernel.                -> nns_index_create() +
 *                                                     nns_index_search()
 *   - the V7/V8/V9 second-stage merge (core.cu:675-696, 832-852, 1036-1056).
 *                                                  -> nns_keys_min() /
 *                                                     an RCCL min all-reduce on
 *                                                     the packed keys.
 *   - utils.h CHECK (print + exit(1), utils.h:16-26): the C ABI never exits; it
 *     returns a status code (the C++ shim mi355x::cudaCall reproduces the
 *     print-and-exit behaviour, see nns_cudacall.hpp).
 *
 * Semantics (identical to the reference's V0, core.cu:31-52): for every query
 * i, the index j in [0,n) minimising the fp32 value
 *     sum_{t=0..k-1, t ascending} fl( fl(q[i][t] - r[j][t])^2 )
 * (un-contracted IEEE fp32, starting from 0), the LOWEST index winning exact
 * ties; a NaN or +INF distance is never selected; a row with no selectable
 * distance returns index 0.  Indices are bit-exact against V0; the returned
 * distance is that same fp32 value (bit-equal, tolerance 0 ulp).
 *
 * Layouts: queries s_points[m][k], refs r_points[n][k], row-major fp32
 * (core.cu:41); results int32[m], 0-based global ref index.
 *
 * Threading: one caller thread per nns_index, and searches of ONE index must not
 * overlap in time (its workspaces, incl. the exact kernel's merge accumulator, belong
 * to the index: use one index per stream).  The whole-call entry points are
 * re-entrant: the process-wide caches behind them (workspace pool, the small-call
 * scratch, RCCL communicators) are internally synchronised, and a call that finds the
 * scratch busy takes the plain path.  All device work of the split API is enqueued on
 * the caller's HIP stream and is asynchronous unless stated.
 *
 * Manners towards the host application:
 *   - no entry point calls hipDeviceSynchronize(): destroy, workspace regrow and the whole-call exits hand
 *     their device blocks back to the library's pool behind an EVENT on the stream that used them, read-outs
 *     (nns_index_stats, nns_index_near_ties) wait for the index's own stream, and the whole-call entry points
 *     run their kernels on a non-blocking stream of the library.  Kernels the application has running on
 *     other streams are never waited for by the library.  (The stream an index last worked on must still exist
 *     when the index is destroyed; if it does not, destroy falls back to waiting for the device.  The HIP runtime
 *     multiplexes streams onto a few hardware queues: work that lands on the queue of a long-running foreign kernel
 *     runs behind it whatever a library does.)
 *   - every entry point that selects a device restores the caller's current device before it returns.
 */
#ifndef NNS_MI355X_H
#define NNS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNS_VERSION_MAJOR 0
#define NNS_VERSION_MINOR 1

/* status codes */
enum {
    NNS_OK = 0,
    NNS_ERR_INVALID = 1,   /* bad argument (k,m,n <= 0, null pointer, ...) */
    NNS_ERR_HIP = 2,       /* a HIP runtime call failed: see nns_last_error() */
    NNS_ERR_NOMEM = 3,     /* host or device allocation failed */
    NNS_ERR_NODEVICE = 4,  /* no gfx950 device visible */
    NNS_ERR_UNSUPPORTED = 5
};

/* path selection flags (nns_index_create / nns_search_f32_ex) */
enum {
    NNS_PATH_AUTO = 0,   /* MFMA filter for 8 <= k <= 1024 (bf16 points: 32..1024) and >= 64 queries (small 8- / 16-D
                          * problems excepted), exact kernels otherwise */
    NNS_PATH_EXACT = 1,  /* exact per-pair kernels only (V1..V9 arithmetic re-expressed) */
    NNS_PATH_MFMA = 2,   /* -2*Q*R^T MFMA filter + exact re-rank (k padded to the tile K) */
    NNS_PATH_MASK = 3,
    NNS_PROFILE = 16,    /* record HIP-event timings per stage (adds syncs at read-out) */
    NNS_MULTI_VIRTUAL = 32, /* nns_search_f32_multi: allow more shards than GPUs (rehearsal) */
    NNS_FILTER_BF16 = 128, /* OPT-IN, fp32 points only, k <= 1024: run the MFMA filter on the centred points
                          * rounded to bf16 (v_mfma_f32_16x16x32_bf16: 16x the fp32 MFMA rate) with a margin tau
                          * widened by the rounding bound 2^-6 |x'||y'|, then re-rank the candidates with V0's fp32
                          * arithmetic on the ORIGINAL fp32 points as usual.  Indices and distances are the same
                          * bits as without the flag (the filter only decides which refs are re-ranked); queries
                          * whose candidate lists overflow fall back to the exact scan.  NNS_PATH_AUTO
                          * takes this filter by itself only for 256 < k <= 1024, where no fp32 tile exists (the
                          * alternative is the VALU scan); it is not what bench.py measures for the fp32
                          * configurations. */
    NNS_REFS_SOA = 64,   /* the reference points are given dimension-major, r[t * n + j] (a dense [k][n]
                          * array: the layout v4::mat_inv_kernel produces, core.cu:293-306, :327) instead
                          * of r[j * k + t]; queries stay [m][k].  The library transposes once on the
                          * device into a copy it owns. */
    NNS_RECORDS_PER_REF = 512, /* MFMA filter: keep the candidate records per SCORE (the form long reference streams
                          * use) also on short streams, where AUTO records ref TILES (a lane's two best tiles, or one
                          * record per tile within its threshold) and lets K5 evaluate the tile's rows — same results
                          * either way; lets tests and A/B runs drive both forms at any size */
    NNS_MULTI_FORCE_COLLECTIVE = 256 /* nns_search_*_multi, for tests: no single-GPU shortcut — even ONE shard runs the
                          * thread-per-GPU body, ncclCommInitAll and the grouped ncclAllReduce (core.cu:965-1057's
                          * shape), so that branch can be executed on a one-GPU box (a 1-rank all-reduce) */
};

/*
 * Packed (distance, index) key: (fp32 bits of the V0 distance << 32) | index.
 * Distances are >= +0, so unsigned (and, below 2^63, signed) integer order is
 * (distance, then index) lexicographic order: min over keys == V0's argmin
 * rule at every reduction level, including an RCCL ncclMin
 * all-reduce on int64/uint64.  NNS_KEY_NONE marks "no selectable distance"
 * (V0 leaves minSum = INFINITY, index = 0: core.cu:34-35).
 */
typedef uint64_t nns_key;
#define NNS_KEY_NONE 0x7F80000000000000ull

/* Most points one set (queries of a search, refs of an index or whole call) may hold: 2^31 - 2^20.  The reference's
 * sizes and indices are `int` (core.cu:24-26); this library keeps int32 indices and leaves 2^20 of headroom below
 * 2^31 for padded tile images and range ends.  Larger m or n: NNS_ERR_INVALID (shard the refs with index_base). */
#define NNS_MAX_POINTS 0x7FF00000

/* opaque handle: one device-resident, prepared shard of reference points */
typedef struct nns_index nns_index;

/* per-search statistics.  NNS_PROFILE adds the *_ms fields (HIP events on the caller's stream):
 * averages over the searches since the previous nns_index_stats call (the last 32 at most), so
 * a caller can run many refresh + search steps back to back and read them once. */
typedef struct nns_stats {
    int path;              /* NNS_PATH_EXACT or NNS_PATH_MFMA actually taken */
    int k_tile;            /* K of the MFMA tile (k padded up), 0 on the exact path */
    int splits;            /* ref-range splits of the filter grid */
    int ambiguous;         /* queries re-ranked by the exact scan (filter margin < tau) */
    int nonfinite;         /* 1 if NaN/INF/huge inputs forced the exact path */
    float prep_refs_ms;    /* K2 on refs (index create) */
    float prep_queries_ms; /* K2 on queries */
    float filter_ms;       /* K3 MFMA filter */
    float finalize_ms;     /* K5 merge + exact distance of winners */
    float rerank_ms;       /* exact scan of ambiguous queries */
    float exact_ms;        /* exact path kernels (K1) */
    float total_ms;        /* all device work of a search */
    int multi_candidate;   /* queries whose filter margin was below tau: K5 chose among > 1 candidates with
                            * V0's arithmetic (nns_index_near_ties lists them) */
} nns_stats;

/* ---- whole-call drop-ins (host pointers; alloc + H2D + kernels + D2H) ------ */

/* Replaces vN::cudaCall (core.cu:23-29): *results is malloc()'d (m ints) and
 * owned by the caller (free()).  Uses device 0.  Returns NNS_OK or an error
 * (never exits). */
int nns_search_f32(int k, int m, int n, const float *s_points,
                   const float *r_points, int **results);

/* Same search into caller-provided host buffers; dist_out may be NULL.
 * num_shards > 1 splits the refs into contiguous ceil(n/num_shards) ranges
 * (the V8/V9 split rule, core.cu:781-791) searched one after another on the
 * one device and merged with nns_keys_min — the single-GPU rehearsal of the
 * multi-GPU path.  flags: NNS_PATH_*. */
int nns_search_f32_ex(int k, int m, int n, const float *s_points,
                      const float *r_points, int *idx_out, float *dist_out,
                      int num_shards, unsigned flags, int device);

/* The V8/V9 analogue (core.cu:761-853, 965-1057): refs sharded contiguously over
 * num_devices GPUs (<= 0: all visible) by one host thread per GPU, per-GPU packed
 * keys combined with ONE RCCL min all-reduce (uint64, ncclMin) over xGMI; falls back
 * to one GPU for small problems exactly as the reference does (core.cu:775-777:
 * n <= min(2^18, 1024 m)).  Result = V0's, for every m (the reference's own merge
 * is wrong for m > 1).  The RCCL communicators of a device set are created
 * on the first call and cached for the life of the process (nns_shutdown() destroys
 * them); concurrent multi calls of one process are serialised around the collective.
 * NNS_REFS_SOA is accepted (a shard is then a column range of the [k][n] array).  m or n above
 * NNS_MAX_POINTS: NNS_ERR_INVALID, before anything is allocated.  The caller's current device is
 * restored on return. */
int nns_search_f32_multi(int k, int m, int n, const float *s_points,
                         const float *r_points, int *idx_out, float *dist_out,
                         int num_devices, unsigned flags);
int nns_search_bf16_multi(int k, int m, int n, const uint16_t *s_points,
                          const uint16_t *r_points, int *idx_out, float *dist_out,
                          int num_devices, unsigned flags);
/* Diagnostic: the number of ranks of the last grouped RCCL all-reduce an nns_search_*_multi call of this process
 * completed (0: none yet, or the keys were merged through the host instead). */
int nns_multi_last_exchange_ranks(void);

/* ---- split API (device-resident buffers, caller's stream) ------------------ */

/* Prepare a shard of n reference points r_dev[n][k] (device memory, fp32 AoS)
 * for searching.  index_base is added to every returned index (the shard's
 * offset into the global ref set, core.cu:827-829).  r_dev must stay valid and
 * unchanged for the life of the index (the exact re-rank reads the original
 * values).  stream: a hipStream_t (NULL = default stream). */
int nns_index_create(nns_index **out, int device, int k, int n,
                     const float *r_dev, int64_t index_base, unsigned flags,
                     void *stream);
int nns_index_destroy(nns_index *ix);

/* Re-run the reference pre-pass (K2) on the same buffer, e.g. after the caller
 * rewrote r_dev in place, or to time it. */
int nns_index_refresh(nns_index *ix, void *stream);

/* keys_dev[i] = packed (V0 distance, index_base + argmin) of query i over this
 * shard, or NNS_KEY_NONE.  q_dev[m][k] fp32 AoS in device memory. */
int nns_index_search(nns_index *ix, int m, const float *q_dev,
                     nns_key *keys_dev, void *stream);

/* bf16 points (config C5: bf16 inputs, fp32 accumulate).  Arrays hold raw bf16 bit
 * patterns (uint16_t), same [points][k] layout.  Semantics: V0's arithmetic on the
 * bf16 values widened to fp32 (the reference has no bf16 code; SURVEY 8c defines the
 * oracle this way).  MFMA filter for 32 <= k <= 1024 (v_mfma_f32_16x16x32_bf16 up to 256,
 * v_mfma_f32_32x32x16_bf16 beyond), exact kernels otherwise.  An index and its queries must have the same dtype. */
int nns_index_create_bf16(nns_index **out, int device, int k, int n,
                          const uint16_t *r_dev, int64_t index_base,
                          unsigned flags, void *stream);
int nns_index_search_bf16(nns_index *ix, int m, const uint16_t *q_dev,
                          nns_key *keys_dev, void *stream);
int nns_search_bf16_ex(int k, int m, int n, const uint16_t *s_points,
                       const uint16_t *r_points, int *idx_out, float *dist_out,
                       int num_shards, unsigned flags, int device);

/* nns_index_search + nns_keys_unpack in one call for the single-shard case (what the reference's
 * cudaCall hands back is indices): keys_dev[m] as above AND idx_dev[m] = index of each key (0 for
 * NNS_KEY_NONE, as V0), dist_dev (optional) = its fp32 distance.  The low-dimensional exact kernel
 * writes all three in its one launch; the other paths append the unpack kernel.  q_dev has the
 * index's dtype (fp32, or bf16 bit patterns). */
int nns_index_search_indices(nns_index *ix, int m, const void *q_dev, nns_key *keys_dev,
                             int *idx_dev, float *dist_dev, void *stream);

int nns_index_stats(nns_index *ix, nns_stats *out);

/* The queries of the LAST nns_index_search on the MFMA path whose answer was NOT settled by the
 * filter alone: more than one reference lay within the proof margin tau of the filter's minimum and
 * K5 decided among them with V0's exact arithmetic (queries sent to the exact scan are counted by
 * nns_stats.ambiguous instead).  *count_out = how many; up to `cap` query numbers are copied into
 * ids_out (host memory, any order).  Synchronises the device.  These are the queries a parity check
 * at sizes beyond the oracle's reach should verify in full (SURVEY 8d's "filter margin" gate). */
int nns_index_near_ties(nns_index *ix, int *ids_out, int cap, int *count_out);

/* inout[i] = min(inout[i], other[i]) : the cross-shard merge operator. */
int nns_keys_min(nns_key *inout_dev, const nns_key *other_dev, int m, void *stream);

/* idx_dev[i] = index of keys_dev[i] (0 for NNS_KEY_NONE, as V0); dist_dev
 * (optional) = its fp32 distance (+INF for NNS_KEY_NONE). */
int nns_keys_unpack(const nns_key *keys_dev, int m, int *idx_dev,
                    float *dist_dev, void *stream);

/* Deterministic synthetic clouds: dev[i] = u24(splitmix64(seed, offset+i)) * 2^-24
 * in [0,1) — bit-identical to oracle/v0_oracle.c:nns_rng_fill on the CPU. */
int nns_fill_uniform(float *dev, size_t count, uint64_t seed, uint64_t offset,
                     void *stream);

/* Diagnostic: one 32x32 tile through the filter's MFMA k-order.  a[32][kt],
 * b[32][kt], c0[32], out[32][32] are HOST fp32 buffers; out[i][j] = the MFMA
 * accumulation of sum_t a[i][t] * b[j][t] seeded with c0[i].  bf16 = 0:
 * v_mfma_f32_32x32x2_f32 (compared by the tests with a host fmaf() chain);
 * bf16 = 1: v_mfma_f32_32x32x16_bf16, bf16 = 2: four 16x16 tiles of v_mfma_f32_16x16x32_bf16
 * (the bf16 filter's shape and lane mapping), on the values cast to bf16 (compared with
 * fp64).  These are the error models behind the filter's proof margin tau. */
int nns_selftest_mfma(int kt, int bf16, const float *a, const float *b, const float *c0,
                      float *out);
/* Diagnostic (host only, no device needed): the launch geometry the MFMA filter would use for a k-D search of
 * m queries over n refs.  out[0..11] = {tile depth kt, bf16 operands, fp32 points rounded to bf16 operands,
 * candidate lists per query, m_pad, n_pad, ring slots in total, ref-range splits (grid.y), slots per split,
 * query groups (grid.x), refs per ring slot, queries per workgroup}; with out_len >= 14 also {lanes of a query
 * share thresholds, records per ref tile} (the short-stream forms).  NNS_ERR_UNSUPPORTED beyond the deepest
 * tile.  Lets CPU tests check the planner's invariants (coverage, padding, whole blocks per split). */
int nns_plan_filter(int k, int m, int n, int bf16_points, unsigned flags, int *out, int out_len);
/* Diagnostic (host only, no device needed): the launch geometry of the EXACT path (the reference's V1-V7 kernels,
 * core.cu:58-696) for a k-D search of m queries over n refs.  out[0..5] = {kernel: 0 K1a (lane = query, exact), 1 K1f
 * (k <= 3 from 2^27 pairs: vector-ALU filter + V0 re-rank in the same launch), 2 K1b (lane = ref), 3 K1c (<= 4 queries:
 * the HBM-streaming form); query tiles (grid.x); ref ranges (grid.y, 0 = chosen at launch); refs per range; waves per
 * workgroup; queries per workgroup}.  refs_aligned: the refs are 16-byte aligned; have_workspace: the merge workspace
 * exists (without it K1a / K1f run one ref range per query tile).  Lets CPU tests check the planner's invariants. */
int nns_plan_exact(int k, int m, int n, int refs_aligned, int have_workspace, int *out, int out_len);
/* Diagnostic: what the filter's slow path does when the lanes that carry one query share their record
 * thresholds (short ref streams): out64[l] = min of in64 over the lanes l ^ 32 (tile16 = 0: 32x32 MFMA tiles)
 * or l ^ 16, l ^ 32, l ^ 48 (tile16 = 1: 16x16 tiles) — through the very row-swap instructions the kernel
 * uses.  HOST buffers of 64 floats.  A wrong lane pairing would hand a query another query's threshold. */
int nns_selftest_lane_share(int tile16, const float *in64, float *out64);
/* Diagnostic (host only): the constants of the proof margin tau(a) = c0 + c1 * max(a + x2, 0) the
 * filter and K5 use for a query of squared norm qnorm2 against refs of maximum squared norm ymax2 at
 * tile depth kt; mode 0 fp32 operands, 1 bf16 points, 2 fp32 points rounded to bf16 operands.
 * out3 = {c0, c1, x2}.  Lets the tests hold the measured MFMA error against the model. */
int nns_tau_consts(int kt, float qnorm2, float ymax2, int mode, float *out3);

/* ---- the exchange of the one-process-per-GPU form ----------------------------
 * What V8/V9's gather + host re-rank (core.cu:821-852, 1025-1056) becomes when every GPU
 * has its own process (bench.py --gpus N under torch.distributed.run): each rank searches
 * its contiguous ref shard (nns_index_create with index_base = shard offset, core.cu:781-791,
 * :827-829) and the ranks combine their packed keys with ONE
 * ncclAllReduce(ncclUint64, ncclMin) over xGMI.  Same call site as nns_search_*_multi.
 *   rank 0:     nns_comm_unique_id(id, sizeof id)      (RCCL's ncclUniqueId, 128 bytes)
 *   caller:     carries the id bytes to every rank (MPI, a file, torch.distributed, ...)
 *   every rank: nns_comm_create(&c, id, sizeof id, nranks, rank, device)   (collective)
 *               nns_comm_allreduce_min(c, keys_dev, m, stream)            (async, in place)
 *               nns_comm_destroy(c)
 * librccl is dlopen()ed on first use; NNS_ERR_UNSUPPORTED if it cannot be loaded. */
#define NNS_COMM_ID_BYTES 128
typedef struct nns_comm nns_comm;
int nns_comm_unique_id(void *id_out, size_t id_bytes);
int nns_comm_create(nns_comm **out, const void *id, size_t id_bytes, int nranks,
                    int rank, int device);
int nns_comm_size(nns_comm *c);   /* ranks RCCL reports for the communicator (0 on error) */
int nns_comm_allreduce_min(nns_comm *c, nns_key *keys_dev, int m, void *stream);
int nns_comm_destroy(nns_comm *c);

/* ---- misc ------------------------------------------------------------------ */
int nns_device_count(void);
const char *nns_strerror(int status);
const char *nns_last_error(void); /* thread-local detail of the last failure */
int nns_version(void);            /* major * 1000 + minor */
/* Explicit replacement for the reference's hidden WarmUP static (ten V9 calls before main(),
 * core.cu:1900-1933): runs one tiny search through every kernel family (exact lane-per-query and
 * lane-per-ref, every fp32 and bf16 tile depth of the MFMA filter) on `device`, so that
 * code-object loading, the filter's LDS opt-in and the first pool allocations are paid here and
 * not inside a timed call.  Optional: every entry point works without it. */
int nns_warmup(int device);
/* The library parks freed device workspaces (queries/refs staging, tile images, candidate
 * lists) in a per-device pool instead of returning them to the runtime on every call — the
 * reference allocates and frees on every cudaCall (core.cu:793-802).  nns_trim() gives all
 * parked blocks back and returns the number of bytes released.  NNS_POOL_BYTES in the
 * environment caps what the pool may hold (default 16 GiB; 0 disables pooling). */
size_t nns_trim(void);
/* Destroys the cached RCCL communicators of nns_search_*_multi and trims the pool: call once
 * when the process is done with the library (optional; indexes and nns_comm handles are the
 * caller's to destroy). */
int nns_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* NNS_MI355X_H */
