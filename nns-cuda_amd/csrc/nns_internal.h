// nns_internal.h — shared declarations of the gfx950 nearest-neighbour kernels.
// (internal; the public boundary is include/nns.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <type_traits>
#include "nns.h"

namespace nns {

// ---- error plumbing ---------------------------------------------------------
void set_error(const char *fmt, ...);

#define NNS_HIP(call)                                                          \
    do {                                                                       \
        hipError_t nns_e_ = (call);                                            \
        if (nns_e_ != hipSuccess) {                                            \
            ::nns::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,     \
                             hipGetErrorString(nns_e_));                       \
            return NNS_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

#define NNS_TRY(call)                                                          \
    do {                                                                       \
        int nns_s_ = (call);                                                   \
        if (nns_s_ != NNS_OK) return nns_s_;                                   \
    } while (0)

// ---- pooled device workspaces (dev_pool.hip) -----------------------------------
hipError_t pool_alloc(void **out, size_t bytes);
template <class T> static inline hipError_t pool_alloc(T **out, size_t bytes) { return pool_alloc((void **)out, bytes); }
void pool_free(void *ptr);     // caller has synchronised the work that used ptr
// ptr may still be in use by work already enqueued on `st`: reusable once an event recorded there has fired (no host wait)
void pool_free_after(void *ptr, hipStream_t st);
void pool_free_after(void *const *ptrs, int count, hipStream_t st);   // one event for all of them; nulls skipped
size_t pool_trim();            // give every parked block back to the runtime
// the library's own non-blocking streams (whole-call entry points); nullptr if none can be created
hipStream_t lib_stream_acquire();
void lib_stream_release(hipStream_t s);

// Every C-ABI entry point that selects a device restores the caller's current device on return (the reference's
// V8/V9 leave whatever cudaSetDevice they called last, core.cu:996; a library must not).
struct DeviceScope {
    int saved = -1;
    DeviceScope()
    {
        if (hipGetDevice(&saved) != hipSuccess) {
            (void)hipGetLastError();
            saved = -1;
        }
    }
    ~DeviceScope()
    {
        int cur = -1;
        if (saved >= 0 && hipGetDevice(&cur) == hipSuccess && cur != saved) (void)hipSetDevice(saved);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// "the H2D copy that just returned (synchronous, legacy default stream) happens before whatever is enqueued on
// `st` next": an explicit event edge.  This runtime's pageable hipMemcpy returns after the device-side DMA, so the
// edge is already true here; CUDA documents the opposite for pageable sources and nothing in HIP's API promises it.
int order_after_default_stream(hipStream_t st);
void stager_release();         // free the small whole calls' pinned scratch (nns_api.hip)
// the overlapped upload of a contiguous host ref range (nns_api.hip), shared with nns_search_*_multi's shards
bool upload_overlap_pays(int k, int64_t m, int64_t n, int bf16, unsigned flags, size_t rbytes);
int search_range_overlapped(int device, int k, int m, int n, const void *q_d, const void *r_host, char *r_d, int bf16,
                            int64_t base, unsigned flags, nns_key *keys, nns_key *keys_tmp);

// (the sum in 64 bits: a + b - 1 passes 2^31 for a point count near NNS_MAX_POINTS and a large divisor — that is how
//  a K1a plan for n = 2^31 - 2^20 once came out with a negative split count)
static inline int divup(int a, int b) { return (int)(((int64_t)a + b - 1) / b); }
static inline int64_t divup64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- packed keys --------------------------------------------------------------
// (fp32 bits << 32) | index; distances are >= +0 so integer order == (distance,
// index) lexicographic order == V0's rule (reference core.cu:44).
__device__ __forceinline__ nns_key pack_key(float d, uint32_t idx)
{
    return ((nns_key)__float_as_uint(d) << 32) | (nns_key)idx;
}

// V0 never selects +INF or NaN (strict '>' against minSum = INFINITY).
__device__ __forceinline__ nns_key make_key(float best, int64_t idx)
{
    return (best < __builtin_inff()) ? pack_key(best, (uint32_t)idx) : (nns_key)NNS_KEY_NONE;
}

// ---- V0 arithmetic, spelled so that nothing can contract it -------------------
// reference core.cu:41-42: diff = q - r; tempSum += diff * diff  (fp32, no FMA)
__device__ __forceinline__ float v0_step(float sum, float q, float r)
{
    const float diff = __fsub_rn(q, r);
    return __fadd_rn(sum, __fmul_rn(diff, diff));
}

// LDS-DMA (global_load_lds_*): 64 lanes x {16, 4} bytes from per-lane global addresses to
// LDS at M0 + lane * size, no VGPR destination.  Inline asm on purpose: through the
// builtin, hipcc (ROCm 7.2) treats every later ds_read as possibly aliasing the
// in-flight DMA and drains it with s_waitcnt vmcnt(0) right after the issue.  The asm
// form is invisible to that pass; completion is tracked by OUR vmcnt waits + the slot
// barrier.  M0 is compiler-reserved: saved and restored inside the same statement.
__device__ __forceinline__ void dma16(const void *g, unsigned lds_byte)
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);   // wave-uniform by construction: keep it scalar
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_byte)
                 : "memory");
}
__device__ __forceinline__ void dma4(const void *g, unsigned lds_byte)
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_byte)
                 : "memory");
}

// ---- tile-image geometry (K2 output, K3 input) ---------------------------------
// A "block" is 32 points.  Its image is [KT/8][2][32][4] floats:
//   img[b][h][i][e] = value(point i, dim 8b + 4h + e)
// i.e. exactly the order in which the 64 lanes of a wave (lane = 32h + i) read
// float4 #b, so one ds_read_b128 / global_load_dwordx4 per lane is lane-linear
// (1 KiB contiguous per wave-instruction, bank-conflict free), and the four
// floats are the operands of MFMA k-steps 4b..4b+3 for that lane
// (v_mfma_f32_32x32x2_f32: lane supplies A[i = lane&31][k = lane>>5]).
constexpr int kBlockPts = 32;

// bf16 filter MFMA shape: 1 = v_mfma_f32_16x16x32_bf16 (the product: the chip clocks ~14 % higher
// on it under load), 0 = v_mfma_f32_32x32x16_bf16 (A/B builds).  Fixes the bf16 tile-image order
// (prep_kernels.hip) and the lanes a query's candidate lists live on (finalize.hip).
#ifndef NNS_BF16_TILE16
#define NNS_BF16_TILE16 1
#endif

struct FilterGeom {
    int bf16;             // 1: bf16 operands (K4), 0: fp32 operands (K3)
    int lpq;              // lanes (= private candidate lists) per query and split: 2, or 4 with 16x16 tiles
    int mixed;            // 1: fp32 points through the bf16 filter (NNS_FILTER_BF16); bf16 = 1 then too
    int kt;               // K of the tile (k padded up with zeros)
    int m_pad;            // queries padded to the workgroup's query count
    int n_pad;            // refs padded to a whole ring slot
    int total_slots;      // n_pad / (32 * kSlotBlocks)
    int splits;           // grid.y: contiguous ref ranges
    int slots_per_split;
    int qgroups;          // grid.x
    int slot_pts;         // refs per ring slot
    int share_thr;        // short ref streams: a query's lanes share their record thresholds
    int tile_rec;         // short ref streams: candidate entries are (tile minimum, first ref of the lane's rows of that
                          // tile) — K5 evaluates all of the lane's rows — instead of (score, ref).  1: appended behind the
                          // threshold test (16 x 16 tiles); 2: the lane's two best tiles + its third-best minimum,
                          // tracked branch-free (32 x 32 tiles)
};

// One candidate of the filter: score s = |y'|^2 - 2 x'.y' and shard-local ref index.
struct __attribute__((aligned(8))) CandEntry {
    float s;
    int j;
};
// capacity of one lane's private candidate list (per split, query, lane half).  The list
// is a RING: a lane collects ~ln(refs it sees) records, and when more than kCandCap were
// appended (monotone inputs) the oldest entry is overwritten — provably harmless when its
// score is above the lane's current threshold (thresholds only shrink), otherwise the
// overflow bit is set and the query is re-ranked by the exact scan instead.
constexpr int kCandCap = 64;                 // power of two
constexpr int kCandOverflow = 1 << 30;       // bit of the per-list count word
constexpr int kCandCountMask = kCandOverflow - 1;

// tau(a) = c0 + c1 * max(a + x2, 0): the margin within which a filter score cannot be
// ordered against V0's fp32 distances (derivation: finalize.hip).  Shared by K3/K4 (which
// widen it by 0.2 % so their candidate sets are supersets) and K5.
struct TauConsts {
    float c0, c1, x2;
};

// mode: 0 fp32 operands, 1 bf16 points (operands exact), 2 fp32 points ROUNDED to bf16 operands
__host__ __device__ inline TauConsts tau_consts(int kt, float qnorm2, float ymax2, int mode)
{
    const bool bf16 = mode != 0;
    const double u = 5.9604644775390625e-08;   // 2^-24
    const double X2 = (double)qnorm2 * (1.0 + 4.0 * u);
    const double Y2 = (double)ymax2 * (1.0 + 4.0 * u);
    const double X = sqrt(X2), Y = sqrt(Y2);
    const double gk = (kt + 2) * u / (1.0 - (kt + 2) * u);   // V0's own rounding (k+1 per term)
    double e3, e2;
    if (!bf16) {
        // fp32 MFMA = k-ordered fmaf chain of kt steps seeded with the rounded norm
        e3 = gk * (Y2 + 2.0 * X * Y) + 2.0 * u * Y2;
        e2 = 2.5 * u * (X + Y) * (X + Y);   // x' = fl(x - c), y' = fl(y - c)
    } else {
        // bf16 products are exact in fp32; the accumulation order/rounding inside
        // v_mfma_f32_32x32x16_bf16 is not documented: allow 2u per add, kt + kt/16 adds
        const double gf = 2.0 * (kt + kt / 16 + 2) * u / (1.0 - 2.0 * (kt + kt / 16 + 2) * u);
        e3 = gf * (Y2 + 2.0 * X * Y) + 2.0 * u * Y2;
        e2 = 0.0;                            // no centring on the bf16 path
        if (mode == 2) {
            // operands x^ = rn_bf16(x'), y^ = rn_bf16(y') with |x^ - x'| <= 2^-8 |x'| (8-bit significand):
            // |x^.y^ - x'.y'| <= (2 * 2^-8 + 2^-16) sum |x'_t||y'_t| <= 2^-7 (1 + 2^-9) X Y, doubled by the
            // factor -2 of the score; plus the centring term of the fp32 path (x' = fl(x - c)), and the
            // accumulation bound above on the slightly larger rounded magnitudes
            e3 = e3 * (1.0 + 0x1p-6) + 0x1p-6 * (1.0 + 0x1p-8) * X * Y;
            e2 = 2.5 * u * (X + Y) * (X + Y);
        }
    }
    const double c1 = 2.0 * gk / (1.0 - gk) * 1.001;
    // er: the fp32 rounding of the THRESHOLD SUM itself.  K5 forms fl(a + tau_fl(a)) and the filter
    // fl(t + fl(1.002 tau_fl(t))): one rounding of a sum whose magnitude is set by the score, not by tau —
    // |a| <= max(X^2, Y^2 + 2XY) + e3 <= (X + Y)^2 + e3 — so it can pull the threshold down by u (|a| + tau),
    // up to 1 / (2 (K + 2)) of tau itself (2.8 % at K = 16): more than the 0.1 % factors below absorb.  It gets a
    // term of its own (2u instead of u: the evaluation of tau_fl — three more roundings, relative — and the
    // rounding of the term itself ride on the spare u and on the 1.001 factors; finalize.hip, "fp32 evaluation").
    const double er = 2.0 * u * ((X + Y) * (X + Y) + e3 + e2);
    const double c0 = (2.0 + c1) * (e3 + e2) * 1.001 + er + 1e-30;
    TauConsts t;
    t.c0 = (float)(c0 * (1.0 + 1e-6));
    t.c1 = (float)(c1 * (1.0 + 1e-6));
    t.x2 = (float)(X2 * (1.0 + 1e-6));
    return t;
}

__host__ __device__ inline float tau_of(const TauConsts &t, float a)
{
    const float d = a + t.x2;
    return t.c0 + t.c1 * (d > 0.0f ? d : 0.0f);
}

// The record threshold a filter lane derives from a score a it has seen: a + 1.002 tau(a) (a hair
// wider than K5's own tau so that the lists are supersets of what K5 needs).  Monotone in a, so the
// threshold of a minimum is the minimum of the thresholds.  (The filter's slow path, tighten, spells the
// same expression with its constants in registers.)
__host__ __device__ inline float record_threshold(const TauConsts &t, float a)
{
    const float d = a + t.x2;
    return a + (t.c0 + t.c1 * (d > 0.0f ? d : 0.0f)) * 1.002f;
}

// device-side scalars shared between kernels of one index
struct DevScalars {
    unsigned r_maxabs_bits;  // max |r| bits (NaN/INF/huge detection)
    unsigned ymax2_bits;     // max centred squared norm over refs
    // the three per-search words are adjacent: one 12-byte memset resets them
    unsigned q_maxabs_bits;  // max |q| bits
    int amb_count;           // number of ambiguous queries (sent to the exact scan)
    int multi_count;         // number of queries K5 decided among MORE THAN ONE candidate within tau
    unsigned pad[3];
};

// ---- launchers (each .hip file owns its kernels) --------------------------------
// exact_kernels.hip
int launch_keys_fill(nns_key *keys, int m, nns_key value, hipStream_t st);
int launch_keys_min(nns_key *inout, const nns_key *other, int m, hipStream_t st);
int launch_keys_unpack(const nns_key *keys, int m, int *idx, float *dist, hipStream_t st);
int launch_fill_uniform(float *dev, size_t count, uint64_t seed, uint64_t offset, hipStream_t st);
// exact search of all m queries (K1a lane=query when k is small, else K1b)
// ws: workspace of ws_keys keys (exact_workspace_keys) for K1a's in-kernel second stage; ws_fresh: it was
// (re)allocated since the last launch (its arrival counters are then zeroed once; they re-arm themselves).
// idx_out / dist_out (optional): also write the unpacked indices / distances (K1a: same launch).
size_t exact_workspace_keys(int k, int m, int n);
// the exact path's launch geometry for a shape (host only: nns_plan_exact)
int exact_plan(int k, int m, int n, bool aligned, bool have_ws, int *v6);
int launch_exact_search(int k, int m, int n, const float *q, const float *r,
                        int64_t index_base, nns_key *keys, nns_key *ws, size_t ws_keys, bool ws_fresh,
                        int *idx_out, float *dist_out, hipStream_t st);
// exact scan of the queries listed in qlist[0 .. *qcount) (device memory); keys
// of listed queries must hold NNS_KEY_NONE on entry (atomic-min merge).
int launch_exact_listed(int k, int n, const float *q, const float *r,
                        const int *qlist, const int *qcount, int max_listed,
                        int64_t index_base, nns_key *keys, hipStream_t st);

int launch_exact_search_bf16(int k, int m, int n, const uint16_t *q, const uint16_t *r,
                             int64_t index_base, nns_key *keys, hipStream_t st);
int launch_exact_listed_bf16(int k, int n, const uint16_t *q, const uint16_t *r, const int *qlist,
                             const int *qcount, int max_listed, int64_t index_base, nns_key *keys,
                             hipStream_t st);

// prep_kernels.hip (K2)
int prep_workspace_bytes(int kt, size_t *bytes);
int launch_prep_mean(int k, int kt, int n, const float *r, double *partial_ws,
                     float *mean, unsigned *maxabs_bits, hipStream_t st);
int launch_prep_image(int k, int kt, int npts, int npts_pad, const float *pts,
                      const float *mean, float scale, float pad_norm,
                      float *img, float *norms, unsigned *max_norm_bits,
                      unsigned *maxabs_bits, hipStream_t st, bool out_bf16 = false);

// bf16 points (raw uint16 bits) -> bf16 tile image [blk][16][64 lanes][8 bf16], value * scale
// (scale = 1 or -2, exact), fp32 norms of the UNcentred points, max-|v| word
// dimension-major [k][n] -> point-major [n][k] (the inverse of the reference's mat_inv_kernel,
// core.cu:293-306); esz = 4 (fp32) or 2 (bf16 bits)
int launch_soa_to_aos(int k, int n, const void *src, void *dst, int esz, hipStream_t st);
// order: 0 = 32x32x16 operands, 1 = 16x16x32 operands (fragment 8 * tile + k-step)
int launch_prep_image_bf16(int order, int kt, int k, int npts, int npts_pad, const uint16_t *pts, float scale,
                           float pad_norm, void *img, float *norms, unsigned *max_norm_bits,
                           unsigned *maxabs_bits, hipStream_t st);

// filter_mfma.hip (K3 fp32 / K4 bf16)
int filter_plan(int k, int m, int n, bool bf16, FilterGeom *g, bool mixed = false, bool per_ref = false);
int launch_filter(const FilterGeom &g, const void *qimg, const void *rimg, const float *rnorm,
                  const float *qnorm, const DevScalars *scal, CandEntry *lists, int *counts,
                  hipStream_t st);
int launch_mfma_selftest(int kt, int bf16, const float *a, const float *b, const float *c0, float *out,
                         hipStream_t st);
int launch_lane_share_selftest(int t16, const float *in, float *out, hipStream_t st);

// finalize.hip (K5)
int launch_finalize(const FilterGeom &g, int k, int m, int n, const void *q, const void *r,
                    const CandEntry *lists, const int *counts, const float *qnorm,
                    DevScalars *scal, int64_t index_base, nns_key *keys, int *amb_list,
                    int *multi_list, hipStream_t st);

}  // namespace nns
