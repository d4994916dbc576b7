// exact_kernels.hip — K1: exact per-pair distance + argmin kernels (gfx950 VALU),
// and the small packed-key utilities (K5's merge operator).
//
// These re-express the reference's V1..V7 "distance + argmin" kernels
// (core.cu:58-122 get_dis_kernel/get_min_kernel, core.cu:215-257, 307-349,
// 589-633 fused cudaCallKernel) with V0's exact arithmetic (core.cu:38-44):
// fp32 diff, un-contracted mul then add, t ascending, strict '>' so the lowest
// index wins ties.  The distance matrix is never materialised and
// every reduction level compares packed (distance, index) keys, so the result is
// bit-identical to V0 by construction.
//
// Two geometries:
//   K1a  lane = query   (many queries, small k):  the query sits in k VGPRs, the
//        refs are wave-uniform and arrive through scalar loads straight from the
//        AoS array (12 B per ref at k = 3: no transpose needed, unlike the
//        reference's mat_inv_kernel core.cu:293-306); grid = query tiles x ref
//        splits (the V7 idea, core.cu:662, without its host second stage).
//   K1b  lane = ref     (few queries or large k): a tile of queries in LDS read
//        by broadcast, each lane walks its own ref row; used for the exact
//        re-rank of ambiguous queries behind the MFMA filter and for m ~ 1.
#ifdef NNS_K1A_STAMPS
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#endif
#include "nns_internal.h"

namespace nns {

// ---------------------------------------------------------------------------
// key utilities
// ---------------------------------------------------------------------------
__global__ void keys_fill_kernel(nns_key *keys, int m, nns_key v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) keys[i] = v;
}

__global__ void keys_min_kernel(nns_key *inout, const nns_key *other, int m)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        nns_key a = inout[i], b = other[i];
        inout[i] = b < a ? b : a;
    }
}

__global__ void keys_unpack_kernel(const nns_key *keys, int m, int *idx, float *dist)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        nns_key kx = keys[i];
        idx[i] = (int)(uint32_t)(kx & 0xFFFFFFFFull);  // NNS_KEY_NONE -> 0, as V0
        if (dist) dist[i] = __uint_as_float((uint32_t)(kx >> 32));
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// same stream as oracle/v0_oracle.c:nns_rng_fill
__global__ void fill_uniform_kernel(float *out, size_t count, uint64_t s, uint64_t offset)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        uint64_t h = mix64(s ^ ((offset + i) * 0xD1342543DE82EF95ull));
        out[i] = (float)(h >> 40) * (1.0f / 16777216.0f);
    }
}

int launch_keys_fill(nns_key *keys, int m, nns_key value, hipStream_t st)
{
    hipLaunchKernelGGL(keys_fill_kernel, dim3(divup(m, 256)), dim3(256), 0, st, keys, m, value);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}
int launch_keys_min(nns_key *inout, const nns_key *other, int m, hipStream_t st)
{
    hipLaunchKernelGGL(keys_min_kernel, dim3(divup(m, 256)), dim3(256), 0, st, inout, other, m);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}
int launch_keys_unpack(const nns_key *keys, int m, int *idx, float *dist, hipStream_t st)
{
    hipLaunchKernelGGL(keys_unpack_kernel, dim3(divup(m, 256)), dim3(256), 0, st, keys, m, idx, dist);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}
int launch_fill_uniform(float *dev, size_t count, uint64_t seed, uint64_t offset, hipStream_t st)
{
    // host-side first mix, exactly as the oracle does (s = mix64(seed))
    uint64_t x = seed + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) return NNS_OK;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dev, count, x, offset);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// ---------------------------------------------------------------------------
// wave64 / workgroup key reduction
// ---------------------------------------------------------------------------
__device__ __forceinline__ nns_key wave_min_key(nns_key v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
        nns_key o = ((nns_key)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// ---------------------------------------------------------------------------
// K1a: lane = query, wave-uniform refs
// ---------------------------------------------------------------------------
// K1a geometry: a lane holds QPL queries in registers; a workgroup of NW waves (all NW waves hold the SAME
// 64 x QPL queries) walks its ref range in chunks of CH refs, the waves round-robin.  K <= 4: the workgroup's
// queries and double-buffered tiles of the range come into LDS by LDS-DMA (the next tile in flight while the
// current one is walked) and a chunk is read by BROADCAST ds_read_b128 (all lanes, same address: one LDS access,
// no bank conflict).  K >= 8: a chunk is 32 floats — the wave fetches it itself with scalar loads (wave-uniform
// address) and the refs feed the VALU as SGPR operands.  Workgroups are 8 waves (four per CU) unless the problem
// has so few queries that the cut would need more than 256 ref ranges (k1a_plan).
//
// The inner loop is branch-free: per pair the 3K - 1 V0 operations (the first add of the chain is
// 0 + x = x, exact), per chunk a min tree and TWO conditional moves — the lane's best distance and the
// first ref of the CHUNK it came from.  Keys travelling through the merges carry (distance, chunk start):
// integer order is still (distance, then lowest index) because chunks are disjoint ascending ranges;
// the exact index inside the winning chunk is recovered once per QUERY by whoever writes the final key
// (write_final).  (Recovering it inside the loop behind a wave-uniform "any lane improved" branch was
// taken by almost every chunk of a short ref range — a lane improves on its c-th chunk with probability
// ~1/c and a wave has 64 x QPL of them — and cost 25 % more VALU instructions: PMC, 12.2 per pair vs 9.)
//
// Merges, all inside the one launch (V7 does its second stage on the host, core.cu:675-696):
//   * the NW waves of a workgroup: packed keys through LDS;
//   * the ref splits of a query tile (grid.y; the V7 idea, core.cu:662): a returning 64-bit atomicMin per
//     query into an accumulator the index owns (memory-side atomics: the 8 XCDs' L2s are not coherent
//     with each other inside a kernel, and a release / acquire fence pair per workgroup — L2 write-back +
//     invalidate — cost 3x the kernel, measured), then one arrival counter per query tile; the LAST
//     workgroup to arrive exchanges the accumulator back to NNS_KEY_NONE (read + re-arm in one RMW), recovers
//     the exact indices, writes the final keys (and, optionally, the unpacked indices / distances).
//     MEMORY MODEL: all atomics are relaxed, agent scope — there is no release-acquire edge, and none is needed,
//     because every cross-workgroup access of the protocol is a RETURNING read-modify-write on hipMalloc'd memory:
//       (1) on gfx950 an agent-scope RMW is not performed in the issuing XCD's L2 (eight L2s that are not coherent
//           with each other could not make it atomic) but at the memory side of the fabric, one point per address:
//           the RMWs on one address are totally ordered there, and none of them reads or leaves a cached copy;
//       (2) the value of a returning RMW comes back only after it has been performed, and the wave WAITS for it
//           (the asm statements that consume the returned value: s_waitcnt vmcnt(0)) before the workgroup barrier,
//           behind which ONE lane adds to the counter: every min of a workgroup is performed before its add is;
//       (3) the workgroup whose add returns splits - 1 therefore issues its exchanges after every other
//           workgroup's add — hence every min of the query tile — has been performed; the exchanges are RMWs
//           themselves, so what they return is the memory-side value whatever any L1 / L2 holds.
//     Not relied on: dispatch order, workgroup -> XCD placement, cache states, or ordering between DIFFERENT
//     addresses beyond (2).  (MI355X guide, 'inter-workgroup visibility': "8-B agent atomics both sides" is one of
//     the valid hand-off forms.)  A release-acquire fetch_add on the counter alone adds buffer_wbl2 sc1 +
//     buffer_inv sc1 to every workgroup: measured in profiles/r03_ab_k1a_counter.txt.
#ifndef NNS_K1A_QPL
#define NNS_K1A_QPL 2
#endif
// refs through the scalar cache (K >= 8) or through an LDS tile (K <= 4): see the kernel
template <int K>
constexpr bool kK1aScalarRefs = K >= 8;
constexpr int K1A_QPL = NNS_K1A_QPL;   // queries per lane
#ifndef NNS_K1A_LDS_FLOATS
#define NNS_K1A_LDS_FLOATS 4096
#endif
constexpr int K1A_LDS_FLOATS = NNS_K1A_LDS_FLOATS;   // 16 KiB ref tile
#ifndef NNS_K1A_WAVES
#define NNS_K1A_WAVES 8192   // target number of waves in the grid (8 per SIMD)
#endif
#ifndef NNS_K1A_MAXNW
#define NNS_K1A_MAXNW 16     // waves per workgroup, at most
#endif
#ifndef NNS_K1A_NW
#define NNS_K1A_NW 8         // waves per workgroup, normally (k1a_plan)
#endif
#ifndef NNS_K1A_MIN_REFS_PER_WAVE
#define NNS_K1A_MIN_REFS_PER_WAVE 16
#endif
constexpr int K1A_MAXNW = NNS_K1A_MAXNW;

template <int K>
struct K1aChunk {
    static constexpr int value = K <= 4 ? 8 : (K <= 8 ? 4 : 2);   // CH * K floats, multiple of 8
};

// the cross-split stage of K1a (device pointers into the index's exact workspace)
struct K1aMerge {
    nns_key *acc;       // [m] accumulator, NNS_KEY_NONE between launches
    int *cnt;           // [qtiles] arrivals per query tile, zero between launches
    int splits;
    int *idx_out;       // optional fused unpack of the final keys
    float *dist_out;
#ifdef NNS_K1A_STAMPS
    unsigned long long *stamps;   // diagnostic builds only: [workgroup][16] s_memrealtime stamps (100 MHz)
#endif
};
#ifdef NNS_K1A_STAMPS
#define K1A_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (threadIdx.x == 0 && mg.stamps)                                                                   \
            mg.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define K1A_STAMP(i)
#endif

template <int K>
__device__ __forceinline__ void write_final(const K1aMerge &mg, nns_key *keys, int qi, nns_key key,
                                            const float (&qrow)[K], const float *__restrict__ r, int n,
                                            int64_t index_base)
{
    constexpr int CH = K1aChunk<K>::value;
    if (key != (nns_key)NNS_KEY_NONE) {
        // recompute the winning chunk's CH distances with V0's arithmetic (all loads issued together: one
        // memory latency) and keep the first that equals the winning distance
        const float best = __uint_as_float((uint32_t)(key >> 32));
        const int cstart = (int)((int64_t)(uint32_t)(key & 0xFFFFFFFFull) - index_base);
        float rv[CH][K];
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
            const int j = cstart + cc < n ? cstart + cc : n - 1;   // (clamped: the loads stay unconditional)
#pragma unroll
            for (int t = 0; t < K; ++t) rv[cc][t] = r[(size_t)j * K + t];
        }
        int bidx = cstart;
#pragma unroll
        for (int cc = CH - 1; cc >= 0; --cc) {
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < K; ++t) sum = v0_step(sum, qrow[t], rv[cc][t]);
            if (sum == best && cstart + cc < n) bidx = cstart + cc;   // descending: the lowest index wins
        }
        key = pack_key(best, (uint32_t)(index_base + bidx));
    }
    keys[qi] = key;
    if (mg.idx_out) {
        mg.idx_out[qi] = (int)(uint32_t)(key & 0xFFFFFFFFull);   // NNS_KEY_NONE -> 0, as V0
        if (mg.dist_out) mg.dist_out[qi] = __uint_as_float((uint32_t)(key >> 32));
    }
}

// MAXNW = 8 / 16: the same code under two launch bounds — with the grid, block size and ISA unchanged, the C2
// launch of 8-wave workgroups runs 2.5 us (5 %) faster when the kernel is DECLARED for at most 512 threads than
// for 1024 (measured A/B on one device; the only difference in the code object is max_flat_workgroup_size)
template <int K, int MAXNW>
__global__ __launch_bounds__(64 * MAXNW) void exact_lane_query_kernel(
    int m, int n, int refs_per_split, const float *__restrict__ q,
    const float *__restrict__ r, int64_t index_base, nns_key *__restrict__ keys, const K1aMerge mg)
{
    constexpr int CH = K1aChunk<K>::value;
    constexpr int NF = CH * K;                       // floats per chunk (multiple of 8)
    // refs per LDS tile: a multiple of 16 chunks, so that up to 16 waves get the same number of chunks of a full tile
    constexpr int TILE = K1A_LDS_FLOATS / K / (16 * CH) * (16 * CH);
    constexpr int QW = 64 * K1A_QPL;                 // queries per workgroup
    __shared__ __attribute__((aligned(16))) float sref[(kK1aScalarRefs<K> ? 1 : 2) * TILE * K];   // two tile buffers
    // the waves' packed keys for the merge at the end: K <= 4 re-uses the tile buffers (>= 16 KiB, idle by then)
    constexpr bool kOwnKeys = kK1aScalarRefs<K>;
    __shared__ nns_key wkeys_own[kOwnKeys ? MAXNW : 1][QW];
    nns_key(*const wkeys)[QW] = kOwnKeys ? wkeys_own : reinterpret_cast<nns_key(*)[QW]>(sref);
    static_assert(kK1aScalarRefs<K> || sizeof(sref) >= sizeof(nns_key) * MAXNW * QW, "merge keys fit the tile buffers");
    __shared__ __attribute__((aligned(16))) float sq[kK1aScalarRefs<K> ? 4 : QW * K];   // the workgroup's queries (K <= 4)
    __shared__ int s_last;
    const int nthreads = blockDim.x, nw = nthreads >> 6;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = blockIdx.y * refs_per_split;   // (< n)
    // (64-bit: j0 + refs_per_split passes 2^31 for the last range of a very large n)
    const int64_t j1l = (int64_t)j0 + refs_per_split;
    const int j1 = j1l > n ? n : (int)j1l;

    K1A_STAMP(0);
#ifdef NNS_K1A_STAMPS
    if ((threadIdx.x & 63) == 0 && mg.stamps)
        mg.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
#endif
    int qi[K1A_QPL];
    float qv[K1A_QPL][K];
    float best[K1A_QPL];
    int bchunk[K1A_QPL];   // first ref of the chunk the best distance came from
#pragma unroll
    for (int u = 0; u < K1A_QPL; ++u) {
        qi[u] = blockIdx.x * QW + u * 64 + lane;
        // (K <= 4: the workgroup's queries come through LDS, below)
        if constexpr (kK1aScalarRefs<K>) {
#pragma unroll
            for (int t = 0; t < K; ++t) qv[u][t] = (qi[u] < m) ? q[(size_t)qi[u] * K + t] : 0.0f;
        }
        best[u] = __builtin_inff();
        bchunk[u] = j0;
    }

    // K >= 8: refs straight from global memory through the scalar cache: the wave's chunks are wave-uniform
    // addresses -> s_load_dwordx8/x16, the refs arrive in SGPRs and feed the VALU as scalar operands; no LDS tile,
    // no barriers.  Two chunk buffers ping-pong so that the next chunk's loads are in flight while the current one
    // is walked.  (Measured: 4096 x 4096 x 16 49.9 -> 41.5 us, 1024 x 4096 x 8 15.8 -> 13.6 us; at K = 3 the LDS tile
    // below is 5 % faster: 52.1 vs 54.6 us at C2 — so the choice is by K.)
    if constexpr (kK1aScalarRefs<K>) {
        auto walk = [&](const float (&cf)[NF], int cbase) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < K1A_QPL; ++u) {
                float d[CH];
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    float sum = 0.0f;
#pragma unroll
                    for (int t = 0; t < K; ++t) sum = v0_step(sum, qv[u][t], cf[cc * K + t]);
                    d[cc] = sum;
                }
                float cmin = d[0];
#pragma unroll
                for (int cc = 1; cc < CH; ++cc) cmin = fminf(cmin, d[cc]);
                const bool imp = cmin < best[u];
                best[u] = imp ? cmin : best[u];
                bchunk[u] = imp ? cbase : bchunk[u];
            }
        };
        auto fetch = [&](float (&cf)[NF], int c) __attribute__((always_inline)) {
            const float *p = r + (size_t)__builtin_amdgcn_readfirstlane(c) * K;   // wave-uniform
#pragma unroll
            for (int e = 0; e < NF; ++e) cf[e] = p[e];
        };
        {
            const int stride = nw * CH;
            const int full_end = j0 + (j1 - j0) / CH * CH;     // chunks [c, c + CH) with c + CH <= full_end are whole
            int c = j0 + wave * CH;
            float ca[NF], cb[NF];
            if (c + CH <= full_end) fetch(ca, c);
            while (c + CH <= full_end) {
                const int c1 = c + stride;
                if (c1 + CH <= full_end) fetch(cb, c1);
                walk(ca, c);
                if (c1 + CH > full_end) { c = c1; break; }
                const int c2 = c1 + stride;
                if (c2 + CH <= full_end) fetch(ca, c2);
                walk(cb, c1);
                c = c2;
            }
            // the ragged last chunk of the range (n not a multiple of CH): one wave, refs one by one
            if (full_end < j1 && wave == ((full_end - j0) / CH) % nw) {
                float ct[NF];
#pragma unroll
                for (int cc = 0; cc < CH; ++cc)
#pragma unroll
                    for (int t = 0; t < K; ++t) ct[cc * K + t] = full_end + cc < j1 ? r[(size_t)(full_end + cc) * K + t] : __builtin_nanf("");
                walk(ct, full_end);
            }
        }
    } else {
        // Tiles of the ref range through a DOUBLE-BUFFERED LDS image filled by LDS-DMA (global_load_lds_dword:
        // no registers, no wait at the issue): tile i + 1 is in flight while tile i is walked, ONE barrier per
        // tile.
        // copy `nvalid` floats src -> dst (LDS), then pad up to `npad` with `pad`: 16-byte DMA pieces where source
        // and count allow (1 KiB per wave-instruction; the DMA path takes ~100 cycles per INSTRUCTION whatever
        // its width, and a C2 launch starts with 150 of them per CU), dword pieces for an unaligned source and
        // for the last, partial piece (never a byte beyond src + nvalid is read)
        auto lds_fill = [&](const float *src, float *dst, int nvalid, int npad, float pad) __attribute__((always_inline)) {
            const unsigned dst_lds = (unsigned)(uintptr_t)dst;
            const int n4 = (((uintptr_t)src & 15) == 0) ? (nvalid & ~3) : 0;   // (wave-uniform)
            for (int e0 = wave * 256; e0 < n4; e0 += nthreads * 4) {
                const int e = e0 + lane * 4;
                if (e < n4) dma16(src + e, dst_lds + (unsigned)e0 * 4u);
            }
            for (int e0 = n4 + wave * 64; e0 < npad; e0 += nthreads) {
                const int e = e0 + lane;
                if (e < nvalid) dma4(src + e, dst_lds + (unsigned)e0 * 4u);
                else if (e < npad) dst[e] = pad;
            }
        };
        auto stage = [&](int t0, int buf) __attribute__((always_inline)) {
            const int cnt = (j1 - t0) < TILE ? (j1 - t0) : TILE;
            // a ragged last chunk is padded with NaN coordinates: its distances are NaN, never selected
            lds_fill(r + (size_t)t0 * K, sref + buf * (TILE * K), cnt * K, (cnt + CH - 1) / CH * CH * K, __builtin_nanf(""));
        };
        // The workgroup's 64 x QPL queries, ONCE, through LDS as well (all its waves hold the same queries: read
        // per wave from global memory they were 16 x (waves) x splits redundant, 12-byte-strided requests)
        {
            const int64_t qbase = (int64_t)blockIdx.x * QW * K, qleft = (int64_t)m * K - qbase;
            lds_fill(q + qbase, sq, qleft < QW * K ? (int)qleft : QW * K, QW * K, 0.0f);
        }
        stage(j0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of the queries and of tile 0 have landed
#ifdef NNS_K1A_STAMPS
        if (lane == 0 && mg.stamps)
            mg.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + 32 + wave] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();                                   // everyone's have
        K1A_STAMP(1);
#pragma unroll
        for (int u = 0; u < K1A_QPL; ++u)
#pragma unroll
            for (int t = 0; t < K; ++t) qv[u][t] = sq[(u * 64 + lane) * K + t];
        int buf = 0;
        for (int t0 = j0; t0 < j1; t0 += TILE, buf ^= 1) {
            const int cnt = (j1 - t0) < TILE ? (j1 - t0) : TILE;
            const bool more = t0 + TILE < j1;
            if (more) stage(t0 + TILE, buf ^ 1);   // (that buffer's tile was finished before the last barrier)
            const float *tile = sref + buf * (TILE * K);
            for (int c = wave * CH; c < cnt; c += nw * CH) {
                float cf[NF];
                const float4 *src = reinterpret_cast<const float4 *>(tile + c * K);   // uniform address
#pragma unroll
                for (int e = 0; e < NF / 4; ++e) {
                    const float4 v = src[e];
                    cf[4 * e + 0] = v.x;
                    cf[4 * e + 1] = v.y;
                    cf[4 * e + 2] = v.z;
                    cf[4 * e + 3] = v.w;
                }
#pragma unroll
                for (int u = 0; u < K1A_QPL; ++u) {
                    float d[CH];
#pragma unroll
                    for (int cc = 0; cc < CH; ++cc) {
                        float sum = 0.0f;
#pragma unroll
                        for (int t = 0; t < K; ++t) sum = v0_step(sum, qv[u][t], cf[cc * K + t]);
                        d[cc] = sum;
                    }
                    float cmin = d[0];
#pragma unroll
                    for (int cc = 1; cc < CH; ++cc) cmin = fminf(cmin, d[cc]);   // NaN-ignoring min
                    const bool imp = cmin < best[u];                             // false for NaN / INF; strict: first chunk wins
                    best[u] = imp ? cmin : best[u];
                    bchunk[u] = imp ? t0 + c : bchunk[u];
                }
            }
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of the next tile have landed
                __syncthreads();                                   // everyone's have; everyone is done with this tile
                K1A_STAMP(2 + ((t0 - j0) / TILE < 4 ? (t0 - j0) / TILE : 4));
            }
        }
    }
    // ---- the workgroup's waves: packed keys through LDS --------------------------------------------
    K1A_STAMP(7);
    if constexpr (!kOwnKeys) __syncthreads();   // every wave is done with the last tile: its buffer becomes wkeys
#pragma unroll
    for (int u = 0; u < K1A_QPL; ++u) wkeys[wave][u * 64 + lane] = make_key(best[u], index_base + bchunk[u]);
    __syncthreads();
    K1A_STAMP(8);
    // Wave w finishes the queries of register slots u = w, w + nw, ... (< QPL): query blockIdx.x * QW +
    // u * 64 + lane, whose coordinates it holds in qv[u] (for the index recovery of write_final).
    if (mg.splits <= 1) {
#pragma unroll
        for (int u = 0; u < K1A_QPL; ++u)
            if (u % nw == wave && qi[u] < m) {   // (wave-uniform test)
                nns_key mine = NNS_KEY_NONE;
                for (int w = 0; w < nw; ++w) {
                    const nns_key o = wkeys[w][u * 64 + lane];
                    mine = o < mine ? o : mine;
                }
                write_final<K>(mg, keys, qi[u], mine, qv[u], r, n, index_base);
            }
        return;
    }
    // ---- the ref splits of this query tile: memory-side atomicMin, last arriver finishes ------------
#pragma unroll
    for (int u = 0; u < K1A_QPL; ++u)
        if (u % nw == wave && qi[u] < m) {
            nns_key mine = NNS_KEY_NONE;
            for (int w = 0; w < nw; ++w) {
                const nns_key o = wkeys[w][u * 64 + lane];
                mine = o < mine ? o : mine;
            }
            const nns_key old = __hip_atomic_fetch_min(&mg.acc[qi[u]], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));   // returning atomic: waiting for its value = it has been performed
        }
    __syncthreads();                   // every atomic of the workgroup is done before the arrival is counted
    K1A_STAMP(9);
    if (threadIdx.x == 0) {
#ifdef NNS_K1A_COUNTER_ACQREL   // (A/B builds: a release-acquire edge on the counter — buffer_wbl2 sc1 + buffer_inv sc1 per workgroup)
        const int old = __hip_atomic_fetch_add(&mg.cnt[blockIdx.x], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        const int old = __hip_atomic_fetch_add(&mg.cnt[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        s_last = old == mg.splits - 1;
    }
    __syncthreads();
    K1A_STAMP(10);
    if (!s_last) return;               // (workgroup-uniform)
#pragma unroll
    for (int u = 0; u < K1A_QPL; ++u)
        if (u % nw == wave && qi[u] < m) {
            // read + re-arm in ONE returning RMW: like the mins it is performed at the memory side, so what it returns
            // does not depend on the state of any XCD's L2 or this CU's L1 (see the memory-model note above)
            const nns_key v = __hip_atomic_exchange(&mg.acc[qi[u]], (nns_key)NNS_KEY_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            write_final<K>(mg, keys, qi[u], v, qv[u], r, n, index_base);
        }
    if (threadIdx.x == 0) __hip_atomic_store(&mg.cnt[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    K1A_STAMP(11);
}

// geometry of a K1a launch
struct K1aPlan {
    int qtiles, nw, splits, per;
};
template <int K>
static K1aPlan k1a_plan(int m, int n)
{
    constexpr int CH = K1aChunk<K>::value;
    K1aPlan p;
    p.qtiles = divup(m, 64 * K1A_QPL);
    // waves per workgroup: up to the target wave count with ONE ref range per query tile, each wave with at least
    // 4 chunks of work per LDS tile; normally at most 8 (four workgroups per CU: one's head / tail / barrier waits are another's issue slots;
    // 16-wave workgroups measured 4 % slower at C2 and 30 % slower on the 16-D scalar-ref form), 16 only where the
    // finer cut would need more than 256 ref ranges per query tile (few queries: m = 100 x n = 100000)
    int nw = 1, splits = 1;
    for (int cap = NNS_K1A_NW; cap <= K1A_MAXNW; cap *= 2) {
        nw = divup(NNS_K1A_WAVES, p.qtiles);
        if (nw > cap) nw = cap;
        while (nw > 1 && (int64_t)nw * 4 * CH > n) nw >>= 1;
        if (nw < 1) nw = 1;
        // ref splits: the rest of the way to the target, >= 16 refs per wave (small problems are launch-bound: a
        // finer cut spreads 1024 x 4096 x 3 over 128 workgroups instead of 32: 12 -> 8 us; 512 x 8192 x 16: 46 -> 26)
        splits = divup(NNS_K1A_WAVES, p.qtiles * nw);
        const int max_splits = divup(n, (NNS_K1A_MIN_REFS_PER_WAVE) * nw);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        if (splits <= 256) break;
    }
    p.nw = nw;
    if (splits > 65535) splits = 65535;
    int per = divup(n, splits);
    per = divup(per, CH) * CH;
    p.per = per;
    p.splits = divup(n, per);
    return p;
}

// workspace of the cross-split merge, in keys (8-byte units): the accumulator + the counters
template <int K>
static size_t k1a_workspace_keys(int m, int n)
{
    const K1aPlan p = k1a_plan<K>(m, n);
    if (p.splits <= 1) return 0;
    return (size_t)m + ((size_t)p.qtiles * sizeof(int) + 7) / 8 + 1;
}

// K1f (below): the filter + exact re-rank form of this search for k <= 3
#ifndef NNS_K1F_CH
#define NNS_K1F_CH 16
#endif
#ifndef NNS_K1F_QPL
#define NNS_K1F_QPL 2
#endif
constexpr int K1F_CH = NNS_K1F_CH;      // refs per chunk (16 or 32)
constexpr int K1F_QPL = NNS_K1F_QPL;    // queries per lane
#ifndef NNS_K1F_TILE
#define NNS_K1F_TILE 1024
#endif
constexpr int K1F_TILE = NNS_K1F_TILE;  // refs per LDS tile: 16 KiB of float4
template <int K>
__global__ void lowdim_filter_kernel(int m, int n, int refs_per_split, const float *__restrict__ q, const float *__restrict__ r,
                                     int64_t index_base, nns_key *__restrict__ keys, const K1aMerge mg);
#ifndef NNS_K1F_MIN_PAIRS
// K1f against K1a by size (tools/probe_k1f_crossover.py, profiles/r03_k1f_crossover.txt): 2^26 pairs 19.8 vs 18.4 us and
// 17.8 vs 15.4 (its fixed 18 us of staging, merges and re-rank), 2^27 26.1 vs 26.8, 2^28 36.3 vs 43.1, 2^30 121 vs 153
#define NNS_K1F_MIN_PAIRS ((int64_t)1 << 27)
#endif
#ifndef NNS_K1F_MIN_PER
#define NNS_K1F_MIN_PER 512
#endif
#ifndef NNS_K1F_WAVES
#define NNS_K1F_WAVES 4096   // target number of waves in a K1f grid (4 per SIMD)
#endif
template <int K>
static bool k1f_wanted(int m, int n, const K1aPlan &p)
{
#ifdef NNS_K1F_OFF   // (A/B builds)
    return false;
#else
    return K <= 3 && p.nw == 8 && p.per >= NNS_K1F_MIN_PER && (int64_t)m * n >= NNS_K1F_MIN_PAIRS;
#endif
}

template <int K>
static int launch_k1a(int m, int n, const float *q, const float *r, int64_t base,
                      nns_key *keys, nns_key *ws, size_t ws_keys, bool ws_fresh, int *idx_out, float *dist_out,
                      hipStream_t st)
{
    K1aPlan p = k1a_plan<K>(m, n);
    K1aMerge mg{};
    mg.idx_out = idx_out;
    mg.dist_out = dist_out;
    if (p.splits > 1) {
        if (!ws || ws_keys < k1a_workspace_keys<K>(m, n)) {
            // no workspace (allocation failed): one ref range per query tile — slower, still one launch
            p.splits = 1;
            p.per = divup(n, K1aChunk<K>::value) * K1aChunk<K>::value;
        } else {
            mg.acc = ws;
            mg.cnt = reinterpret_cast<int *>(ws + m);
            // accumulator and counters re-arm themselves; a fresh (or re-laid-out) workspace is armed once
            if (ws_fresh) {
                NNS_TRY(launch_keys_fill(mg.acc, m, NNS_KEY_NONE, st));
                NNS_HIP(hipMemsetAsync(mg.cnt, 0, (size_t)p.qtiles * sizeof(int), st));
            }
        }
    }
    mg.splits = p.splits;
    // k <= 3, eight-wave workgroups with at least a few chunks per wave, enough pairs to pay for the re-rank stage:
    // the filter + exact re-rank form (K1f).  Same grid, same merge workspace; ranges are whole 16-ref chunks.
    bool use_f = false;
    int per_f = 0;
    if constexpr (K <= 3) {
        if (k1f_wanted<K>(m, n, p)) {
            // ~110 registers at k = 3: four waves per SIMD, two workgroups per CU — cut for ONE round of them; a
            // workgroup holds 64 x K1F_QPL queries (the counters of the merge workspace are per 128 queries: enough)
            const int qtf = divup(m, 64 * K1F_QPL);
            int want = divup(NNS_K1F_WAVES, qtf * 8);
            const int maxs = divup(n, NNS_K1F_MIN_PER);
            if (want > maxs) want = maxs;
            if (want < 1) want = 1;
            const int per = divup(divup(n, want), K1F_CH) * K1F_CH;
            const int splits = divup(n, per);
            if (splits <= 1 || mg.acc) {   // (the merge workspace does not depend on the cut)
                use_f = true;
                per_f = per;
                p.qtiles = qtf;
                p.splits = splits;
                mg.splits = splits;
            }
        }
    }
#ifdef NNS_K1A_STAMPS
    static unsigned long long *stamps_dev = nullptr;
    const size_t nwg = (size_t)p.qtiles * p.splits;
    const bool stamp = getenv("NNS_K1A_STAMPS") != nullptr && nwg <= 65536;
    if (stamp && !stamps_dev) NNS_HIP(hipMalloc(&stamps_dev, 65536 * 64 * sizeof(unsigned long long)));
    mg.stamps = stamp ? stamps_dev : nullptr;
    if (stamp) NNS_HIP(hipMemsetAsync(stamps_dev, 0, nwg * 64 * sizeof(unsigned long long), st));
#endif
    if (use_f) {
        if constexpr (K <= 3)
            hipLaunchKernelGGL((lowdim_filter_kernel<K>), dim3(p.qtiles, p.splits), dim3(512), 0, st, m, n, per_f, q, r, base, keys, mg);
    } else if (p.nw <= 8)
        hipLaunchKernelGGL((exact_lane_query_kernel<K, 8>), dim3(p.qtiles, p.splits), dim3(64 * p.nw), 0, st,
                           m, n, p.per, q, r, base, keys, mg);
    else
        hipLaunchKernelGGL((exact_lane_query_kernel<K, K1A_MAXNW>), dim3(p.qtiles, p.splits), dim3(64 * p.nw), 0, st,
                           m, n, p.per, q, r, base, keys, mg);
    NNS_HIP(hipGetLastError());
#ifdef NNS_K1A_STAMPS
    if (stamp) {
        NNS_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(nwg * 64);
        NNS_HIP(hipMemcpy(h.data(), stamps_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < nwg; ++w) t0 = std::min(t0, h[w * 64]);
        fprintf(stderr, "k1a stamps: %d qtiles x %d splits x %d waves, per %d (x10 ns since the first entry; min / median / max over workgroups)\n", p.qtiles, p.splits, p.nw, p.per);
        for (int i = 0; i < 48; ++i) {
            std::vector<unsigned long long> v;
            for (size_t w = 0; w < nwg; ++w) if (h[w * 64 + i]) v.push_back(h[w * 64 + i] - t0);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "  stamp %2d: n %6zu  min %6llu  med %6llu  max %6llu\n", i, v.size(), v.front(), v[v.size() / 2], v.back());
        }
    }
#endif
    return NNS_OK;
}

// ---------------------------------------------------------------------------
// K1f: K1a's geometry as FILTER + EXACT RE-RANK on the vector ALU (k <= 3)
// ---------------------------------------------------------------------------
// V0's sub / mul / add cannot fuse, so the exact kernel above pays 3k - 1 = 8 instructions per pair at
// k = 3 — 0.60 of the non-FMA vector rate is where it sits at C2.  But only the WINNER needs V0's arithmetic.  With
// x' = q - c, y' = r - c (c = the first ref of the workgroup's range: clouds far from the origin keep their
// resolution), the score  s(i, j) = |y'_j|^2 - 2 x'_i . y'_j  has the same argmin over j as the squared distance and is
// K FMAs per pair on operands prepared once per ref: the workgroup stages its tiles as (y'_0, y'_1, y'_2, |y'|^2) — one
// broadcast ds_read_b128 per ref — and a lane walks them for its two queries with 3 v_fma per pair.  Record
// collection is the filter's branch-free form 2 (filter_mfma.hip): per CHUNK of 16 refs the minimum of the lane's 16
// scores (v_min3 tree) is inserted into the lane's sorted best three chunk minima (v_med3), the chunks of the best two
// remembered: 4.1 vector instructions per pair instead of 9.1.  At the end of the range the workgroup merges its waves'
// entries per query (LDS), takes a = the smallest chunk minimum, and — tau(a) bounding |s + |x'|^2 - V0's distance| from
// both sides exactly as for the MFMA filter (finalize.hip; here an FMA chain of K steps behind an FMA-evaluated norm:
// tau_consts' fp32 model with kt = 8 covers it) —
//   * evaluates V0's own arithmetic on the 2 x 16 refs of the best two chunks, 16 lanes side by side, reading the
//     ORIGINAL coordinates: the (distance, index)-lexicographic minimum is V0's answer for this range, ties included,
//     because every ref whose score is within tau(a) of a lies in one of those chunks —
//   * unless the THIRD-best chunk minimum is within tau(a) too (three chunks within the filter's resolution: duplicates,
//     lattices): then the whole workgroup scans the range for that query with V0's arithmetic (lane = ref), and if more
//     than a few queries need it, or a ref is NaN / INF / huge (a score could overflow), every lane walks the range for
//     its own queries exactly as K1a does.  Slower on such inputs, never different.
// The cross-split stage is K1a's (returning atomic mins on exact keys, last workgroup writes), minus the index recovery.
// Measured (profiles/r03_ab_k1f.txt, r03_k1f_timeline.txt): C2 49.6 -> 40.3 us.  The walk is 5.2 instructions per pair
// (3 FMAs, 0.5 v_min3, the insert, one broadcast ds_read_b128 + its wait per ref and wave) and runs at the LDS's pace:
// with half the reads (diagnostic build) the launch takes 34 us.  Tried and slower: four queries per lane (fewer reads
// per pair, 121-137 registers: one or two workgroups per CU), 32-ref chunks, 64 registers with four workgroups per CU,
// and refs handed to the lanes by v_readlane from one lane-linear read (4 scalar moves per ref cost 4 / (3 QPL) of the
// FMAs' issue slots).
#ifndef NNS_K1F_MAX_AMB
#define NNS_K1F_MAX_AMB 8       // ambiguous queries a workgroup scans one by one before it falls back as a whole
#endif

// minimum of a packed key over the 16 lanes of each DPP row, on every lane of the row (row rotations: no LDS round trip)
template <int CTRL>
__device__ __forceinline__ nns_key key_min_dpp(nns_key v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
    const nns_key o = ((nns_key)(uint32_t)hi << 32) | (uint32_t)lo;
    return o < v ? o : v;
}
__device__ __forceinline__ nns_key row_min_key(nns_key v)
{
    v = key_min_dpp<0x128>(v);   // row_ror:8
    v = key_min_dpp<0x124>(v);   // row_ror:4
    v = key_min_dpp<0x122>(v);   // row_ror:2
    return key_min_dpp<0x121>(v);   // row_ror:1
}

// tau(a) of the VALU filter in fp32, rounded up everywhere: tau_consts(kt = 8, mode 0) with (X + Y)^2 <= 2 (X^2 + Y^2):
// c0 <= 62.2 u (X^2 + Y^2), c1 <= 20.1 u
__device__ __forceinline__ float k1f_tau(float a, float x2, float y2)
{
    const float d = a + x2;
    return 3.8185e-6f * (x2 + y2) + 1.9074e-6f * (d > 0.0f ? d : 0.0f);   // 2^-18 x 1.001, 2^-19
}

template <int K>
__global__ __launch_bounds__(512) void lowdim_filter_kernel(
    int m, int n, int refs_per_split, const float *__restrict__ q,
    const float *__restrict__ r, int64_t index_base, nns_key *__restrict__ keys, const K1aMerge mg)
{
    static_assert(K >= 1 && K <= 3, "a ref is one float4: up to three centred coordinates + the norm");
    constexpr int CH = K1F_CH, TILE = K1F_TILE;
    constexpr int QPL = K1F_QPL, QW = 64 * QPL;
    static_assert(CH == 16 || CH == 32, "the re-rank stage evaluates a chunk with 16 or 32 lanes");
    constexpr int MAXNW = 8, nthreads = 512, nw = 8;   // always eight waves (launch_k1a sends smaller problems to K1a)
    constexpr int SREF_N = 2 * TILE > 5 * MAXNW * QW / 4 ? 2 * TILE : 5 * MAXNW * QW / 4;
    __shared__ __attribute__((aligned(16))) float4 sref[SREF_N];          // two tile buffers; the merge entries afterwards
    __shared__ __attribute__((aligned(16))) float sq[QW * K];              // the workgroup's queries, original coordinates
    __shared__ float s_ymax[MAXNW];
    __shared__ nns_key s_fkey[QW];      // the workgroup's exact key per query
    __shared__ int s_cand[QW][2];       // the two chunks to evaluate
    __shared__ int s_amb[QW];           // ambiguous queries (list)
    __shared__ int s_namb, s_bad, s_last;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = blockIdx.y * refs_per_split;   // (< n)
    const int64_t j1l = (int64_t)j0 + refs_per_split;
    const int j1 = j1l > n ? n : (int)j1l;
    if (threadIdx.x == 0) {
        s_namb = 0;
        s_bad = 0;
    }
    // the centre: the range's first ref (wave-uniform loads)
    float cen[K];
#pragma unroll
    for (int t = 0; t < K; ++t) cen[t] = r[(size_t)j0 * K + t];

    // queries -> LDS by LDS-DMA (as K1a): 16-byte pieces where source and count allow, dword pieces otherwise
    {
        const int64_t qbase = (int64_t)blockIdx.x * QW * K, qleft = (int64_t)m * K - qbase;
        const float *src = q + qbase;
        const int nvalid = qleft < QW * K ? (int)qleft : QW * K;
        const unsigned dst_lds = (unsigned)(uintptr_t)sq;
        const int n4 = (((uintptr_t)src & 15) == 0) ? (nvalid & ~3) : 0;
        for (int e0 = wave * 256; e0 < n4; e0 += nthreads * 4) {
            const int e = e0 + lane * 4;
            if (e < n4) dma16(src + e, dst_lds + (unsigned)e0 * 4u);
        }
        for (int e0 = n4 + wave * 64; e0 < QW * K; e0 += nthreads) {
            const int e = e0 + lane;
            if (e < nvalid) dma4(src + e, dst_lds + (unsigned)e0 * 4u);
            else if (e < QW * K) sq[e] = 0.0f;
        }
    }
    // ---- staging: refs t0 .. t0 + TILE of the range -> (y', |y'|^2) in tile buffer buf -------------------------
    // The loads are issued BEFORE the walk of the current tile (stage_load: K dwords per ref, lane = ref: 12-byte
    // stride, every byte of the lines used), the arithmetic and the LDS stores follow it (stage_store): two refs per
    // lane and tile, 2 K registers across the walk.
    constexpr int SPT = TILE / nthreads;
    float sv[SPT][K];
    float ymax = 0.0f;
    bool bad = false;
    auto stage_load = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int j = t0 + i * nthreads + (int)threadIdx.x;
#pragma unroll
            for (int t = 0; t < K; ++t) sv[i][t] = j < j1 ? r[(size_t)j * K + t] : 0.0f;
        }
    };
    auto stage_store = [&](int t0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int e = i * nthreads + (int)threadIdx.x;
            float4 o = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());   // padding: score +INF, never a minimum
            if (t0 + e < j1) {
                float yc[3] = {0.0f, 0.0f, 0.0f};
                float nrm = 0.0f;
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    const float v = sv[i][t];
                    bad = bad || !(fabsf(v) < 1e17f);          // NaN, INF, or a square that could overflow
                    yc[t] = __fsub_rn(v, cen[t]);
                    nrm = __builtin_fmaf(yc[t], yc[t], nrm);
                }
                ymax = fmaxf(ymax, nrm);
                o = make_float4(yc[0], yc[1], yc[2], nrm);
            }
            sref[buf * TILE + e] = o;
        }
    };

    K1A_STAMP(0);
    stage_load(j0);
    stage_store(j0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of the queries have landed
    __syncthreads();
    K1A_STAMP(1);
    // lane state: -2 x' of its two queries; best three chunk minima, the chunks of the best two
    float x2[QPL][K], xn[QPL];
    float m1[QPL], m2[QPL], m3[QPL];
    int c1[QPL], c2[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        xn[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float xc = __fsub_rn(sq[(u * 64 + lane) * K + t], cen[t]);
            xn[u] = __builtin_fmaf(xc, xc, xn[u]);
            x2[u][t] = -2.0f * xc;
        }
        m1[u] = m2[u] = m3[u] = __builtin_inff();
        c1[u] = c2[u] = j0;
    }
    int buf = 0;
    for (int t0 = j0; t0 < j1; t0 += TILE, buf ^= 1) {
        const int cnt = (j1 - t0) < TILE ? (j1 - t0) : TILE;
        const bool more = t0 + TILE < j1;
        if (more) stage_load(t0 + TILE);
        const float4 *tile = sref + buf * TILE;
#if defined(NNS_DIAG) && defined(NNS_K1F_ABLATE) && (NNS_K1F_ABLATE & 1)   // timing experiment: no walk (results are wrong)
        if (m < 0)
#endif
        for (int c = wave * CH; c < cnt; c += nw * CH) {
            // the chunk in pieces of 8 refs (32-ref chunks: a ROLLED loop — unrolled, hipcc walks a query over the whole
            // chunk at a time and keeps all its refs in registers, 128 of them: one workgroup per CU)
            float tmq[QPL];
#pragma unroll
            for (int u = 0; u < QPL; ++u) tmq[u] = __builtin_inff();
#ifndef NNS_K1F_PS
#define NNS_K1F_PS 8      // refs per piece
#endif
#ifndef NNS_K1F_UNROLL
#define NNS_K1F_UNROLL (CH <= 16 && NNS_K1F_PS == 8 ? 2 : 1)
#endif
            constexpr int PS = NNS_K1F_PS;
            static_assert(PS == 4 || PS == 8, "a piece is 4 or 8 refs");
            constexpr int UNR = NNS_K1F_UNROLL;
#pragma unroll UNR
            for (int piece = 0; piece < CH / PS; ++piece) {
                float4 rf[PS];
#pragma unroll
#if defined(NNS_DIAG) && defined(NNS_K1F_HALFREADS)   // timing experiment: half the LDS reads, the same arithmetic (results are wrong)
                for (int e = 0; e < PS; ++e) rf[e] = tile[c + PS * piece + (e & 3)];
#else
                for (int e = 0; e < PS; ++e) rf[e] = tile[c + PS * piece + e];   // uniform address: broadcast reads
#endif
#pragma unroll
                for (int u = 0; u < QPL; ++u) {
                    float sc[PS];
#pragma unroll
                    for (int e = 0; e < PS; ++e) {
                        float a = rf[e].w;
                        a = __builtin_fmaf(x2[u][0], rf[e].x, a);
                        if constexpr (K > 1) a = __builtin_fmaf(x2[u][1], rf[e].y, a);
                        if constexpr (K > 2) a = __builtin_fmaf(x2[u][2], rf[e].z, a);
                        sc[e] = a;
                    }
                    if constexpr (PS == 8) {
                        const float g0 = fminf(fminf(sc[0], sc[1]), sc[2]), g1 = fminf(fminf(sc[3], sc[4]), sc[5]);
                        tmq[u] = fminf(fminf(fminf(g0, g1), fminf(sc[6], sc[7])), tmq[u]);
                    } else {
                        tmq[u] = fminf(fminf(fminf(sc[0], sc[1]), sc[2]), fminf(sc[3], tmq[u]));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < QPL; ++u) {
                const float tm = tmq[u];
                // sorted insert (the filter's record form 2): strict compares, equal minima fill the next rank
                const float o1 = m1[u], o2 = m2[u];
                const bool lt1 = tm < o1, lt2 = tm < o2;
                m3[u] = __builtin_amdgcn_fmed3f(tm, o2, m3[u]);
                m2[u] = __builtin_amdgcn_fmed3f(tm, o1, o2);
                m1[u] = fminf(o1, tm);
                c2[u] = lt1 ? c1[u] : (lt2 ? t0 + c : c2[u]);
                c1[u] = lt1 ? t0 + c : c1[u];
            }
        }
        if (more) {
            stage_store(t0 + TILE, buf ^ 1);   // (that buffer's tile was finished before the last barrier)
            __syncthreads();
            K1A_STAMP(2 + ((t0 - j0) / TILE < 4 ? (t0 - j0) / TILE : 4));
        }
    }
    K1A_STAMP(7);
    // ---- the workgroup's waves: entries through LDS (the tile buffers are idle now) ---------------------------------
    {   // non-finite / huge refs anywhere in the range, and the largest norm
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
        if (lane == 0) s_ymax[wave] = ymax;
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) s_bad = 1;
    }
    __syncthreads();
    float *em1 = reinterpret_cast<float *>(sref);            // [nw][QW] each
    float *em2 = em1 + MAXNW * QW, *em3 = em2 + MAXNW * QW;
    int *ec1 = reinterpret_cast<int *>(em3 + MAXNW * QW), *ec2 = ec1 + MAXNW * QW;
    static_assert(sizeof(sref) >= 5 * 4 * MAXNW * QW, "merge entries fit the tile buffers");
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        const int e = wave * QW + u * 64 + lane;
        em1[e] = m1[u];
        em2[e] = m2[u];
        em3[e] = m3[u];
        ec1[e] = c1[u];
        ec2[e] = c2[u];
    }
    __syncthreads();
    float y2 = 0.0f;
    for (int w = 0; w < nw; ++w) y2 = fmaxf(y2, s_ymax[w]);
    const bool wg_bad = s_bad != 0;
    // wave w finishes the queries of register slots u = w, w + nw, ... (its lanes hold their |x'|^2)
#pragma unroll
    for (int u = 0; u < QPL; ++u)
        if (u % nw == wave) {   // (wave-uniform)
            const int ql = u * 64 + lane;
            float a1 = __builtin_inff(), a2 = a1, a3 = a1;
            int b1 = j0, b2 = j0;
            auto insert = [&](float tm, int ch) __attribute__((always_inline)) {
                const float o1 = a1, o2 = a2;
                const bool lt1 = tm < o1 || (tm == o1 && ch < b1), lt2 = tm < o2 || (tm == o2 && ch < b2);   // (chunk order among equals)
                a3 = __builtin_amdgcn_fmed3f(tm, o2, a3);
                a2 = __builtin_amdgcn_fmed3f(tm, o1, o2);
                a1 = fminf(o1, tm);
                b2 = lt1 ? b1 : (lt2 ? ch : b2);
                b1 = lt1 ? ch : b1;
            };
#pragma unroll
            for (int w = 0; w < nw; ++w) {   // (unrolled: all 5 nw entries are read before the dependent inserts)
                const int e = w * QW + ql;
                insert(em1[e], ec1[e]);
                insert(em2[e], ec2[e]);
                a3 = fminf(a3, em3[e]);   // (a third minimum has no chunk: it only ever decides "ambiguous")
            }
            const float thr = a1 + k1f_tau(a1, xn[u] * 1.00001f, y2 * 1.00001f);
            s_cand[ql][0] = b1;
            s_cand[ql][1] = b2;
            s_fkey[ql] = NNS_KEY_NONE;
            // three chunks within the filter's resolution (or a threshold that is not a number): scan the range
            if (!wg_bad && !(a3 > thr) && blockIdx.x * QW + ql < m) s_amb[atomicAdd(&s_namb, 1)] = ql;
        }
    __syncthreads();
    K1A_STAMP(8);
    const int namb = s_namb;
    const bool whole = wg_bad || namb > NNS_K1F_MAX_AMB;   // (workgroup-uniform)
#if defined(NNS_DIAG) && defined(NNS_K1F_ABLATE) && (NNS_K1F_ABLATE & 2)   // timing experiment: no re-rank (results are wrong)
    if (m < 0)
#endif
    if (!whole) {
        // ---- V0 on the refs of the two best chunks of every query: 16 lanes per chunk, 2 queries per wave pass ------
        // (CH = 16: 2 queries per wave pass, 16 lanes per chunk; CH = 32: 1 query per pass, 32 lanes per chunk)
        constexpr int QPP = 64 / (2 * CH);   // queries per wave pass
        constexpr int NPASS = QW / QPP / nw;  // passes per wave
        static_assert(QW % (QPP * nw) == 0, "whole passes");
        // every pass's coordinates are loaded first (one memory latency for all of them), then evaluated
        float rr[NPASS][K];
        int jj[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int it = wave + ps * nw;
            const int ql = QPP * it + (lane / (2 * CH)), slot = (lane / CH) & 1;
            const int j = s_cand[ql][slot] + (lane & (CH - 1));
            const bool dup = slot == 1 && s_cand[ql][1] == s_cand[ql][0];   // (fewer than two chunks seen)
            jj[ps] = (j < j1 && !dup) ? j : -1;
#pragma unroll
            for (int t = 0; t < K; ++t) rr[ps][t] = jj[ps] >= 0 ? r[(size_t)j * K + t] : 0.0f;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int it = wave + ps * nw;
            const int ql = QPP * it + (lane / (2 * CH));
            nns_key key = NNS_KEY_NONE;
            if (jj[ps] >= 0) {
                float sum = 0.0f;
#pragma unroll
                for (int t = 0; t < K; ++t) sum = v0_step(sum, sq[ql * K + t], rr[ps][t]);
                key = make_key(sum, index_base + jj[ps]);
            }
            key = row_min_key(key);   // the 16 lanes of a row; then the rows of the query's two chunks
#pragma unroll
            for (int off = CH; off >= 16; off >>= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)key, off, 64), hi = __shfl_xor((uint32_t)(key >> 32), off, 64);
                const nns_key o = ((nns_key)hi << 32) | lo;
                key = o < key ? o : key;
            }
            if ((lane & (2 * CH - 1)) == 0) s_fkey[ql] = key;
        }
        __syncthreads();
        // ---- ambiguous queries, one by one: the whole workgroup scans the range (lane = ref) ---------------------
        for (int ai = 0; ai < namb; ++ai) {
            const int ql = s_amb[ai];
            float qr[K];
#pragma unroll
            for (int t = 0; t < K; ++t) qr[t] = sq[ql * K + t];
            nns_key key = NNS_KEY_NONE;
            for (int j = j0 + (int)threadIdx.x; j < j1; j += nthreads) {
                float sum = 0.0f;
#pragma unroll
                for (int t = 0; t < K; ++t) sum = v0_step(sum, qr[t], r[(size_t)j * K + t]);
                const nns_key kj = make_key(sum, index_base + j);
                key = kj < key ? kj : key;
            }
            key = wave_min_key(key);
            if (lane == 0) atomicMin(reinterpret_cast<unsigned long long *>(&s_fkey[ql]), (unsigned long long)key);   // (LDS)
        }
        if (namb) __syncthreads();
    } else {
        // ---- the range as K1a walks it: every lane, its own queries, V0's arithmetic, refs by scalar loads --------
        float best[QPL];
        int bj[QPL];
        float qv[QPL][K];
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            best[u] = __builtin_inff();
            bj[u] = j0;
#pragma unroll
            for (int t = 0; t < K; ++t) qv[u][t] = sq[(u * 64 + lane) * K + t];
        }
        for (int c = j0 + wave * 8; c < j1; c += nw * 8) {
            float rv[8][K];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int j = __builtin_amdgcn_readfirstlane(c + cc < j1 ? c + cc : j1 - 1);   // wave-uniform, clamped
#pragma unroll
                for (int t = 0; t < K; ++t) rv[cc][t] = r[(size_t)j * K + t];
            }
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
#pragma unroll
                for (int u = 0; u < QPL; ++u) {
                    float sum = 0.0f;
#pragma unroll
                    for (int t = 0; t < K; ++t) sum = v0_step(sum, qv[u][t], rv[cc][t]);
                    const bool imp = sum < best[u] && c + cc < j1;   // strict, ascending j: the first minimum
                    best[u] = imp ? sum : best[u];
                    bj[u] = imp ? c + cc : bj[u];
                }
        }
        nns_key(*wk)[QW] = reinterpret_cast<nns_key(*)[QW]>(sref);   // (the entries have been consumed: barrier above)
#pragma unroll
        for (int u = 0; u < QPL; ++u) wk[wave][u * 64 + lane] = make_key(best[u], index_base + bj[u]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < QPL; ++u)
            if (u % nw == wave) {
                nns_key mine = NNS_KEY_NONE;
                for (int w = 0; w < nw; ++w) {
                    const nns_key o = wk[w][u * 64 + lane];
                    mine = o < mine ? o : mine;
                }
                s_fkey[u * 64 + lane] = mine;
            }
        __syncthreads();
    }
    K1A_STAMP(9);
    // ---- this range's exact keys: straight out, or through the cross-split stage (K1a's protocol) ---------------
    auto put = [&](int qi, nns_key key) __attribute__((always_inline)) {
        keys[qi] = key;
        if (mg.idx_out) {
            mg.idx_out[qi] = (int)(uint32_t)(key & 0xFFFFFFFFull);   // NNS_KEY_NONE -> 0, as V0
            if (mg.dist_out) mg.dist_out[qi] = __uint_as_float((uint32_t)(key >> 32));
        }
    };
    if (mg.splits <= 1) {
        for (int ql = threadIdx.x; ql < QW; ql += nthreads)
            if (blockIdx.x * QW + ql < m) put(blockIdx.x * QW + ql, s_fkey[ql]);
        return;
    }
    for (int ql = threadIdx.x; ql < QW; ql += nthreads)
        if (blockIdx.x * QW + ql < m) {
            const nns_key old = __hip_atomic_fetch_min(&mg.acc[blockIdx.x * QW + ql], s_fkey[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));   // returning atomic: waiting for its value = it has been performed
        }
    __syncthreads();                   // every atomic of the workgroup is done before the arrival is counted
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(&mg.cnt[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == mg.splits - 1;
    }
    __syncthreads();
    K1A_STAMP(10);
    if (!s_last) return;               // (workgroup-uniform)
    for (int ql = threadIdx.x; ql < QW; ql += nthreads)
        if (blockIdx.x * QW + ql < m) {
            const int qi = blockIdx.x * QW + ql;
            const nns_key v = __hip_atomic_exchange(&mg.acc[qi], (nns_key)NNS_KEY_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            put(qi, v);
        }
    if (threadIdx.x == 0) __hip_atomic_store(&mg.cnt[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    K1A_STAMP(11);
}

// ---------------------------------------------------------------------------
// K1b: lane = ref, query tile in LDS (broadcast reads)
// ---------------------------------------------------------------------------
// element access: fp32 as is; bf16 (raw uint16 bits) widened exactly to fp32
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t *p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4(const uint16_t *p)
{
    const uint2 v = *reinterpret_cast<const uint2 *>(p);   // 4 bf16
    float4 o;
    o.x = __uint_as_float(v.x << 16);
    o.y = __uint_as_float(v.x & 0xFFFF0000u);
    o.z = __uint_as_float(v.y << 16);
    o.w = __uint_as_float(v.y & 0xFFFF0000u);
    return o;
}

template <int QT, int VEC, typename T>
__global__ __launch_bounds__(256) void exact_lane_ref_kernel(
    int k, int n, const T *__restrict__ q, const T *__restrict__ r,
    const int *__restrict__ qlist, const int *__restrict__ qcount, int mq,
    int64_t index_base, nns_key *__restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];  // [QT][k]
    __shared__ nns_key wkeys[4][QT];
    const int nq = qlist ? *qcount : mq;
    const int tid = threadIdx.x;

    for (int g = blockIdx.y; g * QT < nq; g += gridDim.y) {
        __syncthreads();
        for (int e = tid; e < QT * k; e += 256) {
            const int u = e / k, t = e - u * k;
            const int qslot = g * QT + u;
            float v = 0.0f;
            if (qslot < nq) {
                const int qi = qlist ? qlist[qslot] : qslot;
                v = ld1(q + (size_t)qi * k + t);
            }
            sq[e] = v;
        }
        __syncthreads();

        float best[QT];
        int bidx[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            best[u] = __builtin_inff();
            bidx[u] = 0;
        }
        for (int64_t jl = (int64_t)blockIdx.x * 256 + tid; jl < n; jl += (int64_t)gridDim.x * 256) {   // (64-bit: the stride may carry past 2^31)
            const int j = (int)jl;
            const T *rj = r + (size_t)j * k;
            float sum[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) sum[u] = 0.0f;
            if (VEC == 4) {
                for (int t = 0; t < k; t += 4) {
                    const float4 rv = ld4(rj + t);
#pragma unroll
                    for (int u = 0; u < QT; ++u) {
                        const float4 qv = *reinterpret_cast<const float4 *>(&sq[u * k + t]);
                        float s = sum[u];
                        s = v0_step(s, qv.x, rv.x);
                        s = v0_step(s, qv.y, rv.y);
                        s = v0_step(s, qv.z, rv.z);
                        s = v0_step(s, qv.w, rv.w);
                        sum[u] = s;
                    }
                }
            } else {
                for (int t = 0; t < k; ++t) {
                    const float rv = ld1(rj + t);
#pragma unroll
                    for (int u = 0; u < QT; ++u) sum[u] = v0_step(sum[u], sq[u * k + t], rv);
                }
            }
#pragma unroll
            for (int u = 0; u < QT; ++u)
                if (best[u] > sum[u]) {   // strict: first (lowest j) minimum of this lane
                    best[u] = sum[u];
                    bidx[u] = j;
                }
        }
        // lanes -> wave -> workgroup -> global, always on packed keys
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            nns_key key = wave_min_key(make_key(best[u], index_base + bidx[u]));
            if (lane == 0) wkeys[wave][u] = key;
        }
        __syncthreads();
        if (tid < QT) {
            nns_key key = wkeys[0][tid];
            for (int w = 1; w < 4; ++w) key = wkeys[w][tid] < key ? wkeys[w][tid] : key;
            const int qslot = g * QT + tid;
            if (qslot < nq) {
                const int qi = qlist ? qlist[qslot] : qslot;
                atomicMin((unsigned long long *)&keys[qi], (unsigned long long)key);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1b, tiled: the same lane-per-ref scan with the refs staged through LDS so that every
// global load instruction reads whole 128-byte lines (8 rows x 8 x 16 B) instead of one
// 16-byte fragment from each of 64 rows.  Per wave and step: a sub-tile of 64 rows x 8
// 16-byte chunks is loaded coalesced, written to a padded LDS image (row stride 144 B:
// conflict-free ds_read_b128 with row = lane), read back one row per lane, and folded into
// the QT running V0 chains (t ascending across steps).  The next sub-tile's global loads are
// in flight while the current one is consumed.  Only the wave's own LDS region is touched
// between the two workgroup barriers of a query group, so no barrier is needed per step.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void widen4(const float4 &raw, const float *, float (&o)[4])
{
    o[0] = raw.x; o[1] = raw.y; o[2] = raw.z; o[3] = raw.w;
}
__device__ __forceinline__ void widen8(const float4 &raw, float (&o)[8])   // 8 bf16 -> 8 fp32
{
    const unsigned w[4] = {__float_as_uint(raw.x), __float_as_uint(raw.y), __float_as_uint(raw.z), __float_as_uint(raw.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[2 * e] = __uint_as_float(w[e] << 16);
        o[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
    }
}

template <int QT, typename T>
__global__ __launch_bounds__(256) void exact_lane_ref_tiled_kernel(
    int k, int n, const T *__restrict__ q, const T *__restrict__ r,
    const int *__restrict__ qlist, const int *__restrict__ qcount, int mq,
    int64_t index_base, nns_key *__restrict__ keys)
{
    constexpr int EPC = 16 / sizeof(T);          // elements per 16-byte chunk (4 fp32 / 8 bf16)
    constexpr int ROWB = 8 * 16 + 16;            // padded LDS row: 8 chunks + 16 B
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float *sq = smem_f;                                            // [QT][k] fp32 queries
    char *stile = reinterpret_cast<char *>(smem_f + QT * k);       // [4 waves][64 rows][ROWB]
    __shared__ nns_key wkeys[4][QT];
    const int nq = qlist ? *qcount : mq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char *mytile = stile + wave * (64 * ROWB);
    const int kc = k / EPC;                       // 16-byte chunks per row
    const char *rbytes = reinterpret_cast<const char *>(r);
    const size_t row_bytes = (size_t)k * sizeof(T);

    for (int g = blockIdx.y; g * QT < nq; g += gridDim.y) {
        __syncthreads();
        for (int e = tid; e < QT * k; e += 256) {
            const int u = e / k, t = e - u * k;
            const int qslot = g * QT + u;
            float v = 0.0f;
            if (qslot < nq) {
                const int qi = qlist ? qlist[qslot] : qslot;
                v = ld1(q + (size_t)qi * k + t);
            }
            sq[e] = v;
        }
        __syncthreads();

        float best[QT];
        int bidx[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            best[u] = __builtin_inff();
            bidx[u] = 0;
        }
        // this wave's rows: groups of 64, strided over all waves of the grid
        const int wave_global = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
        for (int64_t row0l = (int64_t)wave_global * 64; row0l < n; row0l += (int64_t)nwaves * 64) {   // (64-bit: see above)
            const int row0 = (int)row0l;
            float sum[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) sum[u] = 0.0f;
            const int j = row0 + lane;
            // load mapping: cpr chunks per row per step (8, or fewer for short rows); instruction i
            // covers rows rpi*i .. rpi*i + rpi-1, lane -> (row rpi*i + lane/cpr, chunk lane%cpr), so a
            // wave-instruction always reads 64 x 16 B in whole-row segments
            const int cshift = kc >= 8 ? 3 : (kc >= 4 ? 2 : (kc >= 2 ? 1 : 0));
            const int cpr = 1 << cshift, rpi = 64 >> cshift;
            const int lrow = lane >> cshift, lch = lane & (cpr - 1);
            float4 cur[8], nxt[8];
            auto fetch = [&](float4 (&buf)[8], int c0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = row0 + rpi * i + lrow;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < cpr && rr < n && c0 + lch < kc)
                        v = *reinterpret_cast<const float4 *>(rbytes + (size_t)rr * row_bytes + (size_t)(c0 + lch) * 16);
                    buf[i] = v;
                }
            };
            fetch(cur, 0);
            for (int c0 = 0; c0 < kc; c0 += cpr) {
                if (c0 + cpr < kc) fetch(nxt, c0 + cpr);
                // stage: my fragments -> LDS image of this wave
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i < cpr) *reinterpret_cast<float4 *>(mytile + (rpi * i + lrow) * ROWB + lch * 16) = cur[i];
                __builtin_amdgcn_wave_barrier();
                const int nc = (kc - c0) < cpr ? (kc - c0) : cpr;
                for (int ch = 0; ch < nc; ++ch) {
                    const float4 raw = *reinterpret_cast<const float4 *>(mytile + lane * ROWB + ch * 16);
                    float rv[EPC];
                    if constexpr (EPC == 4) widen4(raw, nullptr, reinterpret_cast<float (&)[4]>(rv));
                    else widen8(raw, reinterpret_cast<float (&)[8]>(rv));
                    const int t0 = (c0 + ch) * EPC;
#pragma unroll
                    for (int u = 0; u < QT; ++u) {
                        float sacc = sum[u];
#pragma unroll
                        for (int e = 0; e < EPC; e += 4) {
                            const float4 qv = *reinterpret_cast<const float4 *>(&sq[u * k + t0 + e]);   // broadcast
                            sacc = v0_step(sacc, qv.x, rv[e + 0]);
                            sacc = v0_step(sacc, qv.y, rv[e + 1]);
                            sacc = v0_step(sacc, qv.z, rv[e + 2]);
                            sacc = v0_step(sacc, qv.w, rv[e + 3]);
                        }
                        sum[u] = sacc;
                    }
                }
                __builtin_amdgcn_wave_barrier();   // all rows read before the image is overwritten
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
            }
            if (j < n) {
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    if (best[u] > sum[u]) {   // strict: first (lowest j) minimum of this lane
                        best[u] = sum[u];
                        bidx[u] = j;
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            nns_key key = wave_min_key(make_key(best[u], index_base + bidx[u]));
            if (lane == 0) wkeys[wave][u] = key;
        }
        __syncthreads();
        if (tid < QT) {
            nns_key key = wkeys[0][tid];
            for (int w = 1; w < 4; ++w) key = wkeys[w][tid] < key ? wkeys[w][tid] : key;
            const int qslot = g * QT + tid;
            if (qslot < nq) {
                const int qi = qlist ? qlist[qslot] : qslot;
                atomicMin((unsigned long long *)&keys[qi], (unsigned long long)key);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1c: the HBM-bound shapes — a handful of queries (m <= 4) against a long ref stream.
// ---------------------------------------------------------------------------
// The reference driver's m = 1 samples (main.cu:39-42) and SURVEY 8(f1)'s 1 x 1 M x 16: n * k * 4 bytes read once, a few
// flops per byte — the stream from HBM is the whole cost (V7 splits such a search over G blocks and finishes on the
// host: core.cu:655-696).  What K1b's lane-per-row form does wrong on short rows: with 64-byte rows a float4 load
// per lane touches 64 different rows, i.e. an eighth of each 128-byte line per wave-instruction, four instructions
// deep; and the search took three launches' worth of overheads (key fill, scan with one atomicMin per workgroup on
// ONE address, nothing fused).  K1c instead:
//   * every wave-instruction reads 1 KiB of CONTIGUOUS refs (lane i: 16 bytes at base + 16 i), U of them per tile and
//     the next tile's U already in flight while the current one is consumed (2U = 8 loads of 16 B per lane);
//   * a row of k = 4 L floats is then spread over L consecutive lanes (L = 1, 2, 4, 8).  V0's sum is a strict
//     t-ascending chain, so partial sums per lane would change the rounding: instead the chain is PASSED ALONG the L
//     lanes — L phases, in phase p every lane extends the sum it holds by its own four dimensions and hands the
//     result to its right neighbour (DPP row_shr:1, no LDS).  The lane that owns quarter p has, in phase p, exactly
//     V0's partial sum over dimensions 0 .. 4p - 1 (induction from the zero every lane starts with; what the other
//     lanes compute in that phase is never used), so after phase L - 1 the row's last lane holds V0's distance bit for
//     bit.  L-fold redundant VALU work — ~50 instructions per KiB at k = 16, a quarter of the HBM time per CU;
//   * k = 1, 2, 3: a row per lane with ONE k-dword load (global_load_dwordx3 at k = 3: 768 contiguous bytes per
//     wave-instruction);
//   * ONE launch: lane -> wave (packed-key shuffles) -> workgroup (LDS) -> the grid through returning 64-bit
//     atomic mins into EIGHT accumulator shards (workgroup b uses shard b & 7 — a sharding, not a placement
//     assumption: 512 workgroups finishing together would otherwise queue ~11 ns each on one address) and a
//     two-level arrival count (shard, then top); the last workgroup of all exchanges the shards back to
//     NNS_KEY_NONE (read + re-arm in one RMW), writes the keys and, optionally, the unpacked index / distance.
//     Every cross-workgroup access is an agent-scope atomic RMW, both sides: performed at the memory side, never
//     served from an XCD's L2 or a CU's L1.
struct StreamMerge {
    nns_key *acc;    // [8 shards][4 queries], NNS_KEY_NONE between launches
    int *cnt;        // [8] arrivals per shard + [8] = top-level arrivals; zero between launches
    int *idx_out;    // optional fused unpack
    float *dist_out;
};
constexpr int kStreamShards = 8, kStreamMaxQ = 4;
constexpr size_t kStreamWsKeys = kStreamShards * kStreamMaxQ + 8;   // accumulators + 16 ints

__device__ __forceinline__ float dpp_shr1(float v)
{
    // lane i <- lane i - 1 within its row of 16 lanes (row_shr:1); lanes without a source keep `v`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x111, 0xF, 0xF, false));
}

// the grid-level merge of K1c: `mine` = this workgroup's key of query qi (threads qi < mq of the workgroup call it)
__device__ __forceinline__ void stream_merge(const StreamMerge &mg, nns_key *keys, int mq, nns_key mine, int *s_flag)
{
    const int tid = threadIdx.x;
    const int shard = blockIdx.x & (kStreamShards - 1);
    if (gridDim.x == 1) {              // one workgroup (a short stream): its keys are the answer, no workspace involved
        if (tid < mq) {
            keys[tid] = mine;
            if (mg.idx_out) {
                mg.idx_out[tid] = (int)(uint32_t)(mine & 0xFFFFFFFFull);   // NNS_KEY_NONE -> 0, as V0
                if (mg.dist_out) mg.dist_out[tid] = __uint_as_float((uint32_t)(mine >> 32));
            }
        }
        return;
    }
    if (tid < mq) {
        const nns_key old = __hip_atomic_fetch_min(&mg.acc[shard * kStreamMaxQ + tid], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old));   // returning atomic: its value back = it has been performed at the memory side
    }
    __syncthreads();                   // every min of the workgroup is done before the arrival is counted
    if (tid == 0) {
        const int in_shard = ((int)gridDim.x - shard + kStreamShards - 1) / kStreamShards;   // workgroups with b & 7 == shard
        int last = 0;
        if (__hip_atomic_fetch_add(&mg.cnt[shard], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            const int nshards = (int)gridDim.x < kStreamShards ? (int)gridDim.x : kStreamShards;   // shards that have workgroups
            last = __hip_atomic_fetch_add(&mg.cnt[kStreamShards], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1;
        }
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;              // (workgroup-uniform)
    // read + re-arm in ONE memory-side RMW per (shard, query): thread 4 s + qi of the first wave takes shard s of
    // query qi, all 32 exchanges in flight together (one round trip, not eight), then a shuffle min over the shards
    static_assert(kStreamShards * kStreamMaxQ == 32 && kStreamMaxQ == 4, "thread 4 s + qi of the first wave");
    if (tid < 64) {
        nns_key v = NNS_KEY_NONE;
        if (tid < kStreamShards * kStreamMaxQ && (tid & (kStreamMaxQ - 1)) < mq)
            v = __hip_atomic_exchange(&mg.acc[tid], (nns_key)NNS_KEY_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = kStreamMaxQ; off < kStreamShards * kStreamMaxQ; off <<= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)v, off, 64), hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
            const nns_key o = ((nns_key)hi << 32) | lo;
            v = o < v ? o : v;
        }
        if (tid < mq) {
            keys[tid] = v;
            if (mg.idx_out) {
                mg.idx_out[tid] = (int)(uint32_t)(v & 0xFFFFFFFFull);   // NNS_KEY_NONE -> 0, as V0
                if (mg.dist_out) mg.dist_out[tid] = __uint_as_float((uint32_t)(v >> 32));
            }
        }
        if (tid >= 32 && tid <= 32 + kStreamShards)   // re-arm the counters (other lanes of the same wave)
            __hip_atomic_store(&mg.cnt[tid - 32], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void stream_arm_kernel(nns_key *acc, int *cnt)
{
    if (threadIdx.x < kStreamShards * kStreamMaxQ) acc[threadIdx.x] = NNS_KEY_NONE;
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
}

// lanes -> wave -> workgroup for QT queries; returns (threads qi < QT) the workgroup's key of query qi
template <int QT, int NW>
__device__ __forceinline__ nns_key stream_block_key(const float (&best)[QT], const int (&bidx)[QT], int64_t index_base,
                                                    nns_key (*wkeys)[QT])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const nns_key key = wave_min_key(make_key(best[u], index_base + bidx[u]));
        if (lane == 0) wkeys[wave][u] = key;
    }
    __syncthreads();
    nns_key mine = NNS_KEY_NONE;
    if (threadIdx.x < QT) {
        const int nw = blockDim.x >> 6;
        for (int w = 0; w < nw; ++w) mine = wkeys[w][threadIdx.x] < mine ? wkeys[w][threadIdx.x] : mine;
    }
    return mine;
}

// k = 4 L (L = 1, 2, 4, 8): L lanes per row, chain passed along them
template <int L, int QT, int U, int NW>
__global__ __launch_bounds__(64 * NW) void exact_stream_kernel(int n, int mq, const float *__restrict__ q,
                                                              const float4 *__restrict__ r4, int64_t index_base,
                                                              nns_key *__restrict__ keys, const StreamMerge mg)
{
    constexpr int K = 4 * L;
    constexpr int LOGL = L == 1 ? 0 : (L == 2 ? 1 : (L == 4 ? 2 : 3));
    constexpr int TILE = 64 * U;                                    // 16-byte pieces per wave and tile
    __shared__ nns_key wkeys[NW][QT];
    __shared__ int s_flag;
    const int lane = threadIdx.x & 63;
    const int part = lane & (L - 1);                                // which four dimensions of its row this lane holds
    const bool lastpart = part == L - 1;
    float qv[QT][4];
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[u][e] = u < mq ? q[(size_t)u * K + 4 * part + e] : 0.0f;
    float best[QT];
    int bidx[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        best[u] = __builtin_inff();
        bidx[u] = 0;
    }
    const int64_t total = (int64_t)n * L;                           // 16-byte pieces in all
    const int64_t full_tiles = total / TILE;
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    int64_t t = (int64_t)(threadIdx.x >> 6) * gridDim.x + blockIdx.x;   // waves of a workgroup: gridDim.x tiles apart

    auto consume = [&](const float4 (&b)[U], int64_t tile) __attribute__((always_inline)) {
        const int row0 = (int)((tile * TILE + lane) >> LOGL);       // (< n < 2^31)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u * (64 >> LOGL);
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                float sum = 0.0f;
#pragma unroll
                for (int p = 0; p < L; ++p) {
                    if (p) sum = dpp_shr1(sum);                     // the chain moves one lane to the right
                    sum = v0_step(sum, qv[qi][0], b[u].x);
                    sum = v0_step(sum, qv[qi][1], b[u].y);
                    sum = v0_step(sum, qv[qi][2], b[u].z);
                    sum = v0_step(sum, qv[qi][3], b[u].w);
                }
                const bool imp = lastpart && sum < best[qi];        // strict: the first (lowest) row of this lane wins; NaN / INF never
                best[qi] = imp ? sum : best[qi];
                bidx[qi] = imp ? row : bidx[qi];
            }
        }
    };
    {
        float4 cur[U], nxt[U];
        auto fetch = [&](float4 (&b)[U], int64_t tile) __attribute__((always_inline)) {
            const float4 *p = r4 + tile * TILE + lane;
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = p[u * 64];
        };
        if (t < full_tiles) fetch(cur, t);
        while (t < full_tiles) {
            const int64_t tn = t + nwaves;
            if (tn < full_tiles) fetch(nxt, tn);
            consume(cur, t);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            t = tn;
        }
        // the ragged last tile: the wave whose turn it is, every piece guarded (a piece past the end counts as NaN)
        if (t == full_tiles && total > full_tiles * TILE) {
            const float nanv = __builtin_nanf("");
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = full_tiles * TILE + u * 64 + lane;
                cur[u] = c < total ? r4[c] : make_float4(nanv, nanv, nanv, nanv);
            }
            consume(cur, full_tiles);
        }
    }
    const nns_key mine = stream_block_key<QT, NW>(best, bidx, index_base, wkeys);
    stream_merge(mg, keys, mq, mine, &s_flag);
}

// k = 1, 2, 3: a row per lane, one K-dword load each (contiguous across the wave)
template <int K>
struct __attribute__((packed, aligned(4))) RowK {
    float v[K];
};
template <int K, int QT, int U, int NW>
__global__ __launch_bounds__(64 * NW) void exact_stream_rows_kernel(int n, int mq, const float *__restrict__ q,
                                                                   const float *__restrict__ r, int64_t index_base,
                                                                   nns_key *__restrict__ keys, const StreamMerge mg)
{
    constexpr int TILE = 64 * U;                                    // rows per wave and tile
    __shared__ nns_key wkeys[NW][QT];
    __shared__ int s_flag;
    const int lane = threadIdx.x & 63;
    float qv[QT][K];
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int e = 0; e < K; ++e) qv[u][e] = u < mq ? q[(size_t)u * K + e] : 0.0f;
    float best[QT];
    int bidx[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        best[u] = __builtin_inff();
        bidx[u] = 0;
    }
    const int64_t full_tiles = n / TILE;
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    int64_t t = (int64_t)(threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    const RowK<K> *rows = reinterpret_cast<const RowK<K> *>(r);

    auto consume = [&](const RowK<K> (&b)[U], int64_t tile) __attribute__((always_inline)) {
        const int row0 = (int)(tile * TILE) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                float sum = 0.0f;
#pragma unroll
                for (int e = 0; e < K; ++e) sum = v0_step(sum, qv[qi][e], b[u].v[e]);
                const bool imp = sum < best[qi];
                best[qi] = imp ? sum : best[qi];
                bidx[qi] = imp ? row0 + u * 64 : bidx[qi];
            }
        }
    };
    {
        RowK<K> cur[U], nxt[U];
        auto fetch = [&](RowK<K> (&b)[U], int64_t tile) __attribute__((always_inline)) {
            const RowK<K> *p = rows + tile * TILE + lane;
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = p[u * 64];
        };
        if (t < full_tiles) fetch(cur, t);
        while (t < full_tiles) {
            const int64_t tn = t + nwaves;
            if (tn < full_tiles) fetch(nxt, tn);
            consume(cur, t);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            t = tn;
        }
        if (t == full_tiles && n > full_tiles * TILE) {
            RowK<K> nanrow;
#pragma unroll
            for (int e = 0; e < K; ++e) nanrow.v[e] = __builtin_nanf("");
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = full_tiles * TILE + u * 64 + lane;
                cur[u] = c < n ? rows[c] : nanrow;
            }
            consume(cur, full_tiles);
        }
    }
    const nns_key mine = stream_block_key<QT, NW>(best, bidx, index_base, wkeys);
    stream_merge(mg, keys, mq, mine, &s_flag);
}

#ifndef NNS_K1C_U
#define NNS_K1C_U 4
#endif
#ifndef NNS_K1C_NW
#define NNS_K1C_NW 8
#endif
#ifndef NNS_K1C_WGS
#define NNS_K1C_WGS 256    // workgroups at most (one of 8 waves per CU): few, long-lived waves — a wave's first-load
                           // latency and the merge's atomic round trips are per workgroup (1 x 1 M x 16, HIP events on
                           // one device: 2048 workgroups of 4 waves 33.8 us, 1024 24.9, 512 19.0; 256 of 8 waves 16.8)
#endif
static bool k1c_shape(int k, int m, const float *r)
{
    if (m < 1 || m > kStreamMaxQ) return false;
    if (k == 1 || k == 2 || k == 3) return true;
    return (k == 4 || k == 8 || k == 16 || k == 32) && (((uintptr_t)r & 15) == 0);
}

// workgroups of a K1c launch
static int k1c_workgroups(int k, int n)
{
    const int64_t pieces = k < 4 ? (int64_t)n : (int64_t)n * (k / 4);          // 16-byte pieces (k >= 4) or rows
    int64_t wgs = divup64(divup64(pieces, 64 * NNS_K1C_U), NNS_K1C_NW);         // one tile per wave at least
    if (wgs > NNS_K1C_WGS) wgs = NNS_K1C_WGS;
    // up to four tiles per wave (128 KiB of refs in all) stay in ONE workgroup: no cross-workgroup merge, nothing to arm
    if (wgs <= 4) wgs = 1;
    return (int)wgs;
}

template <int QT>
static int launch_k1c_q(int k, int m, int n, const float *q, const float *r, int64_t base, nns_key *keys, const StreamMerge &mg,
                        hipStream_t st)
{
    constexpr int U = NNS_K1C_U, NW = NNS_K1C_NW;
    const dim3 grid((unsigned)k1c_workgroups(k, n)), block(64 * NW);
    const float4 *r4 = reinterpret_cast<const float4 *>(r);
    switch (k) {
    case 1: hipLaunchKernelGGL((exact_stream_rows_kernel<1, QT, U, NW>), grid, block, 0, st, n, m, q, r, base, keys, mg); break;
    case 2: hipLaunchKernelGGL((exact_stream_rows_kernel<2, QT, U, NW>), grid, block, 0, st, n, m, q, r, base, keys, mg); break;
    case 3: hipLaunchKernelGGL((exact_stream_rows_kernel<3, QT, U, NW>), grid, block, 0, st, n, m, q, r, base, keys, mg); break;
    case 4: hipLaunchKernelGGL((exact_stream_kernel<1, QT, U, NW>), grid, block, 0, st, n, m, q, r4, base, keys, mg); break;
    case 8: hipLaunchKernelGGL((exact_stream_kernel<2, QT, U, NW>), grid, block, 0, st, n, m, q, r4, base, keys, mg); break;
    case 16: hipLaunchKernelGGL((exact_stream_kernel<4, QT, U, NW>), grid, block, 0, st, n, m, q, r4, base, keys, mg); break;
    default: hipLaunchKernelGGL((exact_stream_kernel<8, QT, U, NW>), grid, block, 0, st, n, m, q, r4, base, keys, mg); break;
    }
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

static int launch_k1c(int k, int m, int n, const float *q, const float *r, int64_t base, nns_key *keys, nns_key *ws,
                      bool ws_fresh, int *idx_out, float *dist_out, hipStream_t st)
{
    StreamMerge mg;
    mg.acc = ws;
    mg.cnt = reinterpret_cast<int *>(ws + kStreamShards * kStreamMaxQ);
    mg.idx_out = idx_out;
    mg.dist_out = dist_out;
    if (ws_fresh && k1c_workgroups(k, n) > 1) {   // accumulators and counters re-arm themselves; a fresh workspace is armed once
        hipLaunchKernelGGL(stream_arm_kernel, dim3(1), dim3(64), 0, st, mg.acc, mg.cnt);
        NNS_HIP(hipGetLastError());
    }
    if (m == 1) return launch_k1c_q<1>(k, m, n, q, r, base, keys, mg, st);
    if (m == 2) return launch_k1c_q<2>(k, m, n, q, r, base, keys, mg, st);
    return launch_k1c_q<4>(k, m, n, q, r, base, keys, mg, st);
}

template <int QT, typename T>
static int launch_k1b_t(int k, int n, const T *q, const T *r, const int *qlist,
                        const int *qcount, int mq, int groups, int64_t base, nns_key *keys,
                        hipStream_t st)
{
    const bool vec = (k % 4 == 0) && (((uintptr_t)r & (4 * sizeof(T) - 1)) == 0);
    const size_t lds = (size_t)QT * k * sizeof(float);
    int xblocks = divup(n, 256);
    // about 8 workgroups per CU in total; refs are strided over gridDim.x
    int target = divup(2048, groups);
    if (target < 1) target = 1;
    if (xblocks > target) xblocks = target;
    dim3 grid(xblocks, groups);
    const bool tiled = (k % (16 / (int)sizeof(T)) == 0) && (((uintptr_t)r & 15) == 0);
    if (tiled) {
        auto kern = exact_lane_ref_tiled_kernel<QT, T>;
        const size_t lds_t = lds + 4 * 64 * (8 * 16 + 16);
        if (lds_t > 48 * 1024)
            NNS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_t, st, k, n, q, r, qlist, qcount, mq, base, keys);
    } else if (vec) {
        auto kern = exact_lane_ref_kernel<QT, 4, T>;
        if (lds > 48 * 1024)
            NNS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k, n, q, r, qlist, qcount, mq, base, keys);
    } else {
        auto kern = exact_lane_ref_kernel<QT, 1, T>;
        if (lds > 48 * 1024)
            NNS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k, n, q, r, qlist, qcount, mq, base, keys);
    }
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// query-tile width: as wide as the LDS tile (<= 64 KiB) and the registers allow
static int pick_qt(int k, int nq)
{
    int qt = 32;
    while (qt > 1 && (size_t)qt * k * 4 > 64 * 1024) qt >>= 1;
    while (qt > 1 && qt / 2 >= nq) qt >>= 1;
    if (qt >= 32) return 32;
    if (qt >= 8) return 8;
    return qt >= 4 ? 4 : 1;
}

template <typename T>
static int launch_k1b(int k, int n, const T *q, const T *r, const int *qlist,
                      const int *qcount, int mq, int qt, int groups, int64_t base,
                      nns_key *keys, hipStream_t st)
{
    switch (qt) {
    case 32: return launch_k1b_t<32, T>(k, n, q, r, qlist, qcount, mq, groups, base, keys, st);
    case 8: return launch_k1b_t<8, T>(k, n, q, r, qlist, qcount, mq, groups, base, keys, st);
    case 4: return launch_k1b_t<4, T>(k, n, q, r, qlist, qcount, mq, groups, base, keys, st);
    default: return launch_k1b_t<1, T>(k, n, q, r, qlist, qcount, mq, groups, base, keys, st);
    }
}

static int check_k(int k)
{
    if ((size_t)k * 4 > 64 * 1024) {
        set_error("exact path: k = %d exceeds the LDS query tile (k <= 16384)", k);
        return NNS_ERR_UNSUPPORTED;
    }
    return NNS_OK;
}

static bool k1a_dim(int k) { return k == 1 || k == 2 || k == 3 || k == 4 || k == 8 || k == 16; }

size_t exact_workspace_keys(int k, int m, int n)
{
    if (m <= kStreamMaxQ && (k <= 4 || k == 8 || k == 16 || k == 32)) return kStreamWsKeys;   // K1c (if the refs are aligned)
    if (m < 64 || !k1a_dim(k)) return 0;
    switch (k) {
    case 1: return k1a_workspace_keys<1>(m, n);
    case 2: return k1a_workspace_keys<2>(m, n);
    case 3: return k1a_workspace_keys<3>(m, n);
    case 4: return k1a_workspace_keys<4>(m, n);
    case 8: return k1a_workspace_keys<8>(m, n);
    default: return k1a_workspace_keys<16>(m, n);
    }
}

// the geometry launch_exact_search would use (host only; nns_plan_exact): v[0] = kernel (0 K1a, 1 K1f, 2 K1b, 3 K1c),
// v[1] = query tiles (grid.x), v[2] = ref ranges (grid.y), v[3] = refs per range, v[4] = waves per workgroup,
// v[5] = queries per workgroup.  `aligned`: the refs are 16-byte aligned (K1c's condition); `have_ws`: the merge
// workspace could be allocated.
template <int K>
static void exact_plan_k1a(int m, int n, bool have_ws, int *v)
{
    K1aPlan p = k1a_plan<K>(m, n);
    v[0] = 0;
    v[5] = 64 * K1A_QPL;
    if (p.splits > 1 && !have_ws) {
        p.splits = 1;
        p.per = divup(n, K1aChunk<K>::value) * K1aChunk<K>::value;
    }
    if constexpr (K <= 3) {
        if (k1f_wanted<K>(m, n, p)) {   // (the same re-cut as launch_k1a)
            const int qtf = divup(m, 64 * K1F_QPL);
            int want = divup(NNS_K1F_WAVES, qtf * 8);
            const int maxs = divup(n, NNS_K1F_MIN_PER);
            if (want > maxs) want = maxs;
            if (want < 1) want = 1;
            const int per = divup(divup(n, want), K1F_CH) * K1F_CH;
            const int splits = divup(n, per);
            if (splits <= 1 || (have_ws && p.splits > 1)) {
                v[0] = 1;
                v[5] = 64 * K1F_QPL;
                p.qtiles = qtf;
                p.splits = splits;
                p.per = per;
                p.nw = 8;
            }
        }
    }
    v[1] = p.qtiles;
    v[2] = p.splits;
    v[3] = p.per;
    v[4] = p.nw;
}

int exact_plan(int k, int m, int n, bool aligned, bool have_ws, int *v)
{
    if (m >= 64 && k1a_dim(k)) {
        switch (k) {
        case 1: exact_plan_k1a<1>(m, n, have_ws, v); break;
        case 2: exact_plan_k1a<2>(m, n, have_ws, v); break;
        case 3: exact_plan_k1a<3>(m, n, have_ws, v); break;
        case 4: exact_plan_k1a<4>(m, n, have_ws, v); break;
        case 8: exact_plan_k1a<8>(m, n, have_ws, v); break;
        default: exact_plan_k1a<16>(m, n, have_ws, v); break;
        }
        return NNS_OK;
    }
    if (m <= kStreamMaxQ && have_ws && (k <= 3 || (aligned && (k == 4 || k == 8 || k == 16 || k == 32)))) {
        v[0] = 3;
        v[1] = 1;
        v[2] = k1c_workgroups(k, n);
        v[3] = divup(n, v[2]);
        v[4] = 8;
        v[5] = m;
        return NNS_OK;
    }
    NNS_TRY(check_k(k));
    const int qt = pick_qt(k, m);
    int groups = divup(m, qt);
    if (groups > 4096) groups = 4096;
    v[0] = 2;
    v[1] = groups;
    v[2] = 0;   // (K1b's grid over the refs is chosen at launch)
    v[3] = n;
    v[4] = 4;
    v[5] = qt;
    return NNS_OK;
}

int launch_exact_search(int k, int m, int n, const float *q, const float *r,
                        int64_t index_base, nns_key *keys, nns_key *ws, size_t ws_keys, bool ws_fresh,
                        int *idx_out, float *dist_out, hipStream_t st)
{
    // K1a needs enough queries to fill lanes; its query lives in K registers
    if (m >= 64) {
        switch (k) {
        case 1: return launch_k1a<1>(m, n, q, r, index_base, keys, ws, ws_keys, ws_fresh, idx_out, dist_out, st);
        case 2: return launch_k1a<2>(m, n, q, r, index_base, keys, ws, ws_keys, ws_fresh, idx_out, dist_out, st);
        case 3: return launch_k1a<3>(m, n, q, r, index_base, keys, ws, ws_keys, ws_fresh, idx_out, dist_out, st);
        case 4: return launch_k1a<4>(m, n, q, r, index_base, keys, ws, ws_keys, ws_fresh, idx_out, dist_out, st);
        case 8: return launch_k1a<8>(m, n, q, r, index_base, keys, ws, ws_keys, ws_fresh, idx_out, dist_out, st);
        case 16: return launch_k1a<16>(m, n, q, r, index_base, keys, ws, ws_keys, ws_fresh, idx_out, dist_out, st);
        default: break;
        }
    }
    // a handful of queries over short rows: the HBM-streaming form, one launch (needs its merge workspace)
    if (k1c_shape(k, m, r) && ws && ws_keys >= kStreamWsKeys)
        return launch_k1c(k, m, n, q, r, index_base, keys, ws, ws_fresh, idx_out, dist_out, st);
    NNS_TRY(check_k(k));
    NNS_TRY(launch_keys_fill(keys, m, NNS_KEY_NONE, st));
    const int qt = pick_qt(k, m);
    int groups = divup(m, qt);
    if (groups > 4096) groups = 4096;   // grid.y strides over the rest
    NNS_TRY(launch_k1b<float>(k, n, q, r, nullptr, nullptr, m, qt, groups, index_base, keys, st));
    if (idx_out) NNS_TRY(launch_keys_unpack(keys, m, idx_out, dist_out, st));
    return NNS_OK;
}

int launch_exact_search_bf16(int k, int m, int n, const uint16_t *q, const uint16_t *r,
                             int64_t index_base, nns_key *keys, hipStream_t st)
{
    NNS_TRY(check_k(k));
    NNS_TRY(launch_keys_fill(keys, m, NNS_KEY_NONE, st));
    const int qt = pick_qt(k, m);
    int groups = divup(m, qt);
    if (groups > 4096) groups = 4096;
    return launch_k1b<uint16_t>(k, n, q, r, nullptr, nullptr, m, qt, groups, index_base, keys, st);
}

int launch_exact_listed_bf16(int k, int n, const uint16_t *q, const uint16_t *r, const int *qlist,
                             const int *qcount, int max_listed, int64_t index_base, nns_key *keys,
                             hipStream_t st)
{
    if (max_listed <= 0) return NNS_OK;
    NNS_TRY(check_k(k));
    const int qt = pick_qt(k, max_listed >= 32 ? 32 : max_listed);
    int groups = divup(max_listed, qt);
    if (groups > 8) groups = 8;
    return launch_k1b<uint16_t>(k, n, q, r, qlist, qcount, 0, qt, groups, index_base, keys, st);
}

int launch_exact_listed(int k, int n, const float *q, const float *r, const int *qlist,
                        const int *qcount, int max_listed, int64_t index_base,
                        nns_key *keys, hipStream_t st)
{
    if (max_listed <= 0) return NNS_OK;
    NNS_TRY(check_k(k));
    // The number of listed queries lives on the device (no host sync): a fixed
    // grid of 8 query-group rows strides over however many 32-query groups there
    // are, so a handful of ambiguous queries and a pathological all-ambiguous
    // batch both keep ~2048 workgroups busy.
    const int qt = pick_qt(k, max_listed >= 32 ? 32 : max_listed);
    int groups = divup(max_listed, qt);
    if (groups > 8) groups = 8;
    return launch_k1b<float>(k, n, q, r, qlist, qcount, 0, qt, groups, index_base, keys, st);
}

}  // namespace nns
