// finalize.hip — K5: turn the filter's candidate lists into V0's answer.
//
// Role in the reference: the second-stage reductions — the LDS tree of get_min_kernel
// (core.cu:105-119), V7's host re-rank over per-block candidates (core.cu:675-696) and
// V8/V9's host merge over per-GPU candidates (core.cu:832-852; wrong for m > 1, SURVEY
// F4).  Here the candidates are re-ranked with V0's own arithmetic on the device and the
// result is a packed (distance, index) key, so splits / shards / GPUs merge with a plain
// integer min and the outcome is V0's argmin bit for bit.
//
// The proof.  For query i let s_j = |y'_j|^2 - 2 x'_i.y'_j be the filter score of ref j
// (fp32 MFMA chain, or bf16 MFMA with fp32 accumulate), a = min_j s_j, X^2 = |x'_i|^2,
// Y^2 = max_j |y'_j|^2 (K2's norms), u = 2^-24, K the tile depth, g = gamma_{K+2}.  Then for
// every j
//     | s_j + |x'|^2 - D_j |  <=  e3 + e2,        D_j = ||q_i - r_j||^2 in real arithmetic
//       e3 = g (Y^2 + 2XY) + 2u Y^2    K-step FMA chain seeded with the rounded norm
//                                       (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain;
//                                       tests check that on the hardware; the bf16 MFMA's
//                                       internal order is undocumented: 2u per add assumed
//                                       and checked against fp64 by the tests)
//       e2 = 2.5u (X + Y)^2            the single rounding of x' = fl(x-c), y' = fl(y-c)
//                                       (0 on the bf16 path: no centring)
// and V0's own fp32 value d0_j (core.cu:38-43: k+1 roundings per term) satisfies
// |d0_j - D_j| <= g D_j.  Hence with
//     tau(a) = 2 (e3 + e2) + 2g/(1-g) (max(a + X^2, 0) + e3 + e2)
// any ref with s_j > a + tau(a) has d0_j strictly above d0 of the filter's best ref:
// V0's argmin lies in C_i = { j : s_j <= a + tau(a) }.  The filter appends every such j
// to a list (its running threshold is never below a + tau(a), see filter_mfma.hip); this
// kernel evaluates V0's exact distance for the members of C_i (1-2 refs typically) and
// keeps the lexicographic (distance, index) minimum == V0's result, ties included.
// Lists that overflowed, NaN/INF/huge inputs (K2's max-|value| word) or an empty list
// send the query to the exact scan over all refs (K1b) instead.
//
// fp32 evaluation.  tau above is real arithmetic; the kernels compute, in fp32 (u = 2^-24, every operation
// rounded to nearest, nothing contracted),
//     tau_fl(a) = fl(c0 + fl(c1 * max(fl(a + x2), 0)))        K5:     T5(a) = fl(a + tau_fl(a))
//                                                              filter: Tf(t) = fl(t + fl(1.002 * tau_fl(t)))
// with c0 = 1.001 (2 + c1)(e3 + e2) + er, c1 = 1.001 * 2g/(1-g), x2 >= X^2 (tau_consts, then rounded UP to fp32).
//  (i)  tau_fl(a) >= tau_c(a) (1 - 3u), tau_c = the same expression in real arithmetic: three roundings, each
//       relative and downward by at most u.
//  (ii) T5(a) >= a + tau_fl(a) - u |a + tau_fl(a)| >= a + tau_c(a) (1 - 4u) - u |a|: the rounding of the threshold
//       sum is relative to the SCORE's magnitude, |a| <= max(X^2, Y^2 + 2XY) + e3 <= (X + Y)^2 + e3, i.e. up to
//       tau / (2 (K + 2)) — 2.8 % of tau at K = 16.  The factors 1.001 do not cover that; er = 2u ((X+Y)^2 + e3 + e2)
//       does: tau_c(a) - tau(a) >= 0.001 tau(a) + er, 0.001 tau >= 4u tau_c, er >= u |a|  ==>  T5(a) >= a + tau(a).
//       So K5 keeps every member of C_i.  (tests/test_tau_model.py evaluates T5 in fp32 over a grid of depths, norms
//       and scores and holds it against a + tau(a) in extended precision, and asserts slack >= u (|a| + tau).)
//  (iii) every fp32 operation above is monotone non-decreasing in a, so t >= a ==> Tf(t) >= Tf(a), and
//       fl(1.002 tau_fl) >= tau_fl gives Tf(a) >= T5(a): a lane's threshold — the minimum of Tf over the tile minima
//       it has seen, all >= the query's final minimum a — never drops below T5(a); whatever K5 selects was recorded.
#include "nns_internal.h"

namespace nns {

// |value| above this (or NaN/INF) voids the error analysis (squares overflow)
constexpr unsigned kHugeBits = 0x5BB1A2BCu;   // 1e17f

__device__ __forceinline__ float load_f(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float load_f(const uint16_t *p, size_t i)
{
    return __uint_as_float((unsigned)p[i] << 16);   // bf16 -> fp32 widening is exact
}
__device__ __forceinline__ float4 load_f4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load_f4(const uint16_t *p)
{
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    float4 o;
    o.x = __uint_as_float(v.x << 16);
    o.y = __uint_as_float(v.x & 0xFFFF0000u);
    o.z = __uint_as_float(v.y << 16);
    o.w = __uint_as_float(v.y & 0xFFFF0000u);
    return o;
}

// V0's exact distance (core.cu:38-43), t ascending; 4 dims per load when aligned
template <typename T>
__device__ __forceinline__ float v0_distance(const T *qi, const T *rj, int k, bool vec)
{
    float sum = 0.0f;
    if (vec) {
        for (int t = 0; t < k; t += 4) {
            const float4 qv = load_f4(qi + t), rv = load_f4(rj + t);
            sum = v0_step(sum, qv.x, rv.x);
            sum = v0_step(sum, qv.y, rv.y);
            sum = v0_step(sum, qv.z, rv.z);
            sum = v0_step(sum, qv.w, rv.w);
        }
    } else {
        for (int t = 0; t < k; ++t) sum = v0_step(sum, load_f(qi, t), load_f(rj, t));
    }
    return sum;
}

// One wave per list UNIT (the 64 lane-lists the filter's wave wrote for 32 queries x 2 lanes or 16
// queries x 4 lanes): lane L walks list lane L, so every load of the [entry][lane] layout is a
// fully coalesced 512-B row, the lpq lists of a query are walked in parallel, and the per-query
// minimum / best key is a 1- or 2-step xor shuffle across the lanes L ^ 32 (^ 16).
template <typename T>
__global__ __launch_bounds__(256) void finalize_kernel(
    int kt, int bf16, int lpq, int m_pad, int splits, int k, int m, int n, const T *__restrict__ q,
    const T *__restrict__ r, const CandEntry *__restrict__ lists, const int *__restrict__ counts,
    const float *__restrict__ qnorm, DevScalars *__restrict__ scal, int64_t index_base,
    nns_key *__restrict__ keys, int *__restrict__ amb_list, int *__restrict__ multi_list)
{
    const int ush = lpq == 4 ? 4 : 5, qmask = (1 << ush) - 1;   // queries per list unit: 16 or 32
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);       // wave-uniform
    const int units = m_pad >> ush;
    if (unit >= units) return;
    const int i = (unit << ush) + (lane & qmask);               // this lane's query
    const bool live = i < m;

    bool fallback = scal->q_maxabs_bits >= kHugeBits || scal->r_maxabs_bits >= kHugeBits;
    float a = __builtin_inff();
    int over = 0;
    if (!fallback && live) {
        for (int s = 0; s < splits; ++s) {
            const size_t lblk = (size_t)s * units + unit;
            const int cw = counts[lblk * 64 + lane];
            if (cw & kCandOverflow) over = 1;
            const int c = (cw & kCandCountMask) < kCandCap ? (cw & kCandCountMask) : kCandCap;
            const CandEntry *l = lists + lblk * (kCandCap * 64) + lane;
            for (int e = 0; e < c; ++e) a = fminf(a, l[e * 64].s);
        }
    }
    // combine the query's lpq lists: lanes that differ in the bits above the query index
    for (int off = 32; off >= (1 << ush); off >>= 1) {
        a = fminf(a, __shfl_xor(a, off, 64));
        over |= __shfl_xor(over, off, 64);
    }
    if (over || !(a < __builtin_inff())) fallback = true;

    nns_key best = NNS_KEY_NONE;
    int ncand = 0;   // candidates within tau of the filter's minimum: the refs V0's arithmetic decides among
    if (!fallback && live) {
        const TauConsts tc = tau_consts(kt, qnorm[i], __uint_as_float(scal->ymax2_bits), bf16);
        const float thr = a + tau_of(tc, a);
        const T *qi = q + (size_t)i * k;
        const bool vec = (k & 3) == 0 && (((uintptr_t)q | (uintptr_t)r) & (4 * sizeof(T) - 1)) == 0;
        for (int s = 0; s < splits; ++s) {
            const size_t lblk = (size_t)s * units + unit;
            const int cw = counts[lblk * 64 + lane] & kCandCountMask;
            const int c = cw < kCandCap ? cw : kCandCap;
            const CandEntry *l = lists + lblk * (kCandCap * 64) + lane;
            for (int e = 0; e < c; ++e) {
                const CandEntry ce = l[e * 64];
                if (ce.s <= thr && ce.j < n) {
                    const float sum = v0_distance(qi, r + (size_t)ce.j * k, k, vec);
                    const nns_key key = make_key(sum, index_base + ce.j);
                    best = key < best ? key : best;
                    ++ncand;
                }
            }
        }
    }
    for (int off = 32; off >= (1 << ush); off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)best, off, 64);
        const unsigned hi = __shfl_xor((unsigned)(best >> 32), off, 64);
        const nns_key o = ((nns_key)hi << 32) | lo;
        best = o < best ? o : best;
        ncand += __shfl_xor(ncand, off, 64);
    }
    // every candidate NaN/INF cannot happen with bounded inputs; be safe anyway
    if (best == NNS_KEY_NONE) fallback = true;
    if (live && lane <= qmask) {   // one lane per query writes
        keys[i] = fallback ? (nns_key)NNS_KEY_NONE : best;
        if (fallback) {
            const int pos = atomicAdd(&scal->amb_count, 1);
            amb_list[pos] = i;
        } else if (ncand > 1) {    // decided by the exact re-rank of several candidates (rare)
            const int pos = atomicAdd(&scal->multi_count, 1);
            multi_list[pos] = i;
        }
    }
}

// Same result, one WAVE per query: with many ref-range splits (few queries) a query owns
// 2 * splits candidate lists, and walking them from one thread serialises hundreds of
// dependent loads.  Lanes take lists l, l + 64, ...; the minimum score and the final key
// are reduced across the wave with packed-key / float shuffles.
//
// TILE RECORDS (FilterGeom.tile_rec, short ref streams): an entry is (minimum of the lane's scores of one ref tile,
// first ref of the lane's rows of that tile) — 16 rows j + 8 g + e (g, e < 4) of a 32 x 32 MFMA tile, 4 rows j + e of a
// 16 x 16 one.  The minimum over the entries is still the filter's minimum a of the query, and every ref with a score
// <= a + tau(a) sits in a tile whose minimum is <= a + tau(a): the members of C_i are among the rows of the entries
// within the threshold.  Each such entry is evaluated by 16 (4) lanes side by side, one row each, with V0's
// arithmetic; rows that are not in C_i take part in the exact minimum too, which cannot change V0's answer.
template <typename T>
__global__ __launch_bounds__(256) void finalize_wave_kernel(
    int kt, int bf16, int lpq, int tile_rec, int m_pad, int splits, int k, int m, int n, const T *__restrict__ q,
    const T *__restrict__ r, const CandEntry *__restrict__ lists, const int *__restrict__ counts,
    const float *__restrict__ qnorm, DevScalars *__restrict__ scal, int64_t index_base,
    nns_key *__restrict__ keys, int *__restrict__ amb_list, int *__restrict__ multi_list)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform query
    if (i >= m) return;
    const int nlists = lpq * splits;
    const int ush = lpq == 4 ? 4 : 5, qmask = (1 << ush) - 1, lsh = lpq == 4 ? 2 : 1;
    bool fallback = scal->q_maxabs_bits >= kHugeBits || scal->r_maxabs_bits >= kHugeBits;
    float a = __builtin_inff();
    int over = 0;
    for (int l = lane; l < nlists; l += 64) {
        const int s = l >> lsh, h = l & (lpq - 1);
        const size_t lblk = (size_t)s * (m_pad >> ush) + (i >> ush);
        const int ln = (h << ush) + (i & qmask);
        const int cw = counts[lblk * 64 + ln];
        if (cw & kCandOverflow) over = 1;
        const int c = (cw & kCandCountMask) < kCandCap ? (cw & kCandCountMask) : kCandCap;
        const CandEntry *lp = lists + lblk * (kCandCap * 64) + ln;
        for (int e = 0; e < c; ++e) a = fminf(a, lp[e * 64].s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a = fminf(a, __shfl_xor(a, off, 64));
        over |= __shfl_xor(over, off, 64);
    }
    if (over || !(a < __builtin_inff())) fallback = true;

    nns_key best = NNS_KEY_NONE;
    int ncand = 0;
    if (!fallback) {
        const TauConsts tc = tau_consts(kt, qnorm[i], __uint_as_float(scal->ymax2_bits), bf16);
        const float thr = a + tau_of(tc, a);
        const T *qi = q + (size_t)i * k;
        const bool vec = (k & 3) == 0 && (((uintptr_t)q | (uintptr_t)r) & (4 * sizeof(T) - 1)) == 0;
        int over3 = 0;
        if (tile_rec) {
            // the lanes walk their lists in lockstep; an entry within the threshold is broadcast and its rows are
            // evaluated side by side, one lane per row (a V0 chain is sequential in t: the parallelism is over rows)
            const int nr = lpq == 4 ? 4 : 16;
            const int joff = lpq == 4 ? lane : (lane & 3) + 8 * (lane >> 2);
            for (int l0 = 0; l0 < nlists; l0 += 64) {
                const int l = l0 + lane;
                int c = 0;
                const CandEntry *lp = lists;
                if (l < nlists) {
                    const int s = l >> lsh, h = l & (lpq - 1);
                    const size_t lblk = (size_t)s * (m_pad >> ush) + (i >> ush);
                    const int ln = (h << ush) + (i & qmask);
                    const int cw = counts[lblk * 64 + ln] & kCandCountMask;
                    c = cw < kCandCap ? cw : kCandCap;
                    lp = lists + lblk * (kCandCap * 64) + ln;
                }
                int cmax = c;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const int o = __shfl_xor(cmax, off, 64);
                    cmax = o > cmax ? o : cmax;
                }
                for (int e = 0; e < cmax; ++e) {
                    CandEntry ce;
                    ce.s = __builtin_inff();
                    ce.j = 0;
                    if (e < c) ce = lp[e * 64];
                    bool hit = e < c && ce.s <= thr;
                    // record form 2: entry 2 is the lane's THIRD-best tile minimum, with no tile attached — inside
                    // the threshold means more than two tiles of this lane's stream may hold V0's answer: exact scan
                    if (tile_rec == 2 && e == 2) {
                        if (hit) over3 = 1;
                        hit = false;
                    }
                    if (hit) ++ncand;   // (entries, i.e. tiles, within the threshold)
                    unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                    while (mask) {      // (wave-uniform)
                        const int src = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const int j = __shfl(ce.j, src, 64) + joff;
                        if (lane < nr && j < n) {
                            const float sum = v0_distance(qi, r + (size_t)j * k, k, vec);
                            const nns_key key = make_key(sum, index_base + j);
                            best = key < best ? key : best;
                        }
                    }
                }
            }
        } else
        for (int l = lane; l < nlists; l += 64) {
            const int s = l >> lsh, h = l & (lpq - 1);
            const size_t lblk = (size_t)s * (m_pad >> ush) + (i >> ush);
            const int ln = (h << ush) + (i & qmask);
            const int cw = counts[lblk * 64 + ln] & kCandCountMask;
            const int c = cw < kCandCap ? cw : kCandCap;
            const CandEntry *lp = lists + lblk * (kCandCap * 64) + ln;
            for (int e = 0; e < c; ++e) {
                const CandEntry ce = lp[e * 64];
                if (ce.s <= thr && ce.j < n) {
                    const float sum = v0_distance(qi, r + (size_t)ce.j * k, k, vec);
                    const nns_key key = make_key(sum, index_base + ce.j);
                    best = key < best ? key : best;
                    ++ncand;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, off, 64);
            const unsigned hi = __shfl_xor((unsigned)(best >> 32), off, 64);
            const nns_key o = ((nns_key)hi << 32) | lo;
            best = o < best ? o : best;
            ncand += __shfl_xor(ncand, off, 64);
            over3 |= __shfl_xor(over3, off, 64);
        }
        if (best == NNS_KEY_NONE || over3) fallback = true;
    }
    if (lane == 0) {
        keys[i] = fallback ? (nns_key)NNS_KEY_NONE : best;
        if (fallback) {
            const int pos = atomicAdd(&scal->amb_count, 1);
            amb_list[pos] = i;
        } else if (ncand > 1) {
            const int pos = atomicAdd(&scal->multi_count, 1);
            multi_list[pos] = i;
        }
    }
}

int launch_finalize(const FilterGeom &g, int k, int m, int n, const void *q, const void *r,
                    const CandEntry *lists, const int *counts, const float *qnorm, DevScalars *scal,
                    int64_t index_base, nns_key *keys, int *amb_list, int *multi_list, hipStream_t st)
{
    const int mode = g.mixed ? 2 : g.bf16;        // tau mode (nns_internal.h)
    const bool data_bf16 = g.bf16 && !g.mixed;    // element type of q / r
    if (g.splits >= 4) {   // few queries, many lists per query: one wave per query
        if (data_bf16)
            hipLaunchKernelGGL(finalize_wave_kernel<uint16_t>, dim3(divup(m, 4)), dim3(256), 0, st, g.kt, mode,
                               g.lpq, g.tile_rec, g.m_pad, g.splits, k, m, n, (const uint16_t *)q, (const uint16_t *)r, lists,
                               counts, qnorm, scal, index_base, keys, amb_list, multi_list);
        else
            hipLaunchKernelGGL(finalize_wave_kernel<float>, dim3(divup(m, 4)), dim3(256), 0, st, g.kt, mode, g.lpq, g.tile_rec,
                               g.m_pad, g.splits, k, m, n, (const float *)q, (const float *)r, lists, counts, qnorm, scal,
                               index_base, keys, amb_list, multi_list);
        NNS_HIP(hipGetLastError());
        return NNS_OK;
    }
    if (g.tile_rec) {   // (filter_plan couples tile records to splits >= 4)
        set_error("internal: tile records need the one-wave-per-query finalize");
        return NNS_ERR_INVALID;
    }
    if (data_bf16)
        hipLaunchKernelGGL(finalize_kernel<uint16_t>, dim3(divup(g.m_pad / (64 / g.lpq), 4)), dim3(256), 0, st, g.kt, mode, g.lpq, g.m_pad,
                           g.splits, k, m, n, (const uint16_t *)q, (const uint16_t *)r, lists, counts, qnorm,
                           scal, index_base, keys, amb_list, multi_list);
    else
        hipLaunchKernelGGL(finalize_kernel<float>, dim3(divup(g.m_pad / (64 / g.lpq), 4)), dim3(256), 0, st, g.kt, mode, g.lpq, g.m_pad,
                           g.splits, k, m, n, (const float *)q, (const float *)r, lists, counts, qnorm, scal,
                           index_base, keys, amb_list, multi_list);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

}  // namespace nns
