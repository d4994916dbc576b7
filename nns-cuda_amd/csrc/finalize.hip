// finalize.hip — K5: merge the filter's per-split top-2 partials, PROVE each
// winner or hand the query to the exact scan, and emit packed (distance, index)
// keys carrying V0's exact distance.
//
// Role in the reference: the second-stage reductions — the LDS tree of
// get_min_kernel (core.cu:105-119), V7's host re-rank over per-block candidates
// (core.cu:675-696) and V8/V9's host merge over per-GPU candidates
// (core.cu:832-852; wrong for m > 1).  Here every level reduces
// (distance, index) pairs and the final distance is recomputed with V0's own
// arithmetic, so keys from different splits / shards / GPUs merge with a plain
// integer min and the result is V0's argmin bit for bit.
//
// The proof.  Let a <= b be the smallest and second-smallest filter score
// s = |y'|^2 - 2 x'.y' of query i over the shard, X^2 = |x'_i|^2 and
// Y^2 = max_j |y'_j|^2 (centred norms from K2), u = 2^-24, K = tile depth,
// g = gamma_{K+2} = (K+2)u / (1 - (K+2)u).  For every ref j
//     | s_j + |x'|^2 - D_j |  <=  e3 + e2,      D_j = ||q_i - r_j||^2 (real arithmetic)
//       e3 = g (Y^2 + 2XY) + 2u Y^2     fp32 FMA chain of K steps seeded with the
//                                        rounded norm (v_mfma_f32_32x32x2_f32 is a
//                                        k-ordered fmaf chain; tests check that)
//       e2 = 2.5u (X + Y)^2             the single rounding of x' = fl(x - c), y' = fl(y - c)
// and V0's own fp32 value d0_j (core.cu:38-43: k+1 roundings per term) satisfies
// |d0_j - D_j| <= g D_j.  Hence if
//     b - a  >  tau = 2 (e3 + e2) + 2 g/(1-g) (a + X^2 + e3 + e2)
// then d0 of the filter's winner is strictly below d0 of every other ref: V0
// would return exactly this index (ties impossible).  Otherwise the query is
// "ambiguous" and is re-ranked by the exact scan over all refs (K1b), which IS
// V0's arithmetic.  Either way the index is V0's.  NaN/INF/huge inputs void the
// bound; K2's max-|value| word detects them and every query goes to the scan.
#include "nns_internal.h"

namespace nns {

// |value| above this (or NaN/INF) voids the error analysis (squares overflow)
constexpr unsigned kHugeBits = 0x5BB1A2BCu;   // 1e17f

__global__ __launch_bounds__(256) void finalize_kernel(
    int kt, int m_pad, int splits, int k, int m, int n, const float *__restrict__ q,
    const float *__restrict__ r, const Partial *__restrict__ partials,
    const float *__restrict__ qnorm, DevScalars *__restrict__ scal, int64_t index_base,
    nns_key *__restrict__ keys, int *__restrict__ amb_list)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;

    float a1 = __builtin_inff(), a2 = __builtin_inff();
    int idx = 0;
    for (int s = 0; s < splits; ++s) {
        const Partial p = partials[(size_t)s * m_pad + i];
        const float hi = fmaxf(a1, p.m1);
        if (p.m1 < a1) idx = p.idx;
        a1 = fminf(a1, p.m1);
        a2 = fminf(hi, fminf(a2, p.m2));
    }

    const bool bad_inputs = scal->q_maxabs_bits >= kHugeBits || scal->r_maxabs_bits >= kHugeBits;
    bool certain = false;
    if (!bad_inputs && a1 < __builtin_inff() && idx < n) {
        const double u = 5.9604644775390625e-08;   // 2^-24
        const double X2 = (double)qnorm[i] * (1.0 + 4.0 * u);
        const double Y2 = (double)__uint_as_float(scal->ymax2_bits) * (1.0 + 4.0 * u);
        const double X = sqrt(X2), Y = sqrt(Y2);
        const double gk = (kt + 2) * u / (1.0 - (kt + 2) * u);
        const double e3 = gk * (Y2 + 2.0 * X * Y) + 2.0 * u * Y2;
        const double e2 = 2.5 * u * (X + Y) * (X + Y);
        double ap = (double)a1 + X2;
        if (ap < 0.0) ap = 0.0;
        ap += e3 + e2;
        const double tau = (2.0 * (e3 + e2) + 2.0 * gk / (1.0 - gk) * ap) * 1.001 + 1e-30;
        certain = ((double)a2 - (double)a1) > tau;   // a2 = +INF (single ref) is certain
    }

    if (certain) {
        // V0's exact distance of the proven winner (core.cu:38-43)
        const float *qi = q + (size_t)i * k;
        const float *rj = r + (size_t)idx * k;
        float sum = 0.0f;
        if ((k & 3) == 0 && ((((uintptr_t)q) | ((uintptr_t)r)) & 15) == 0) {
            for (int t = 0; t < k; t += 4) {
                const float4 qv = *reinterpret_cast<const float4 *>(qi + t);
                const float4 rv = *reinterpret_cast<const float4 *>(rj + t);
                sum = v0_step(sum, qv.x, rv.x);
                sum = v0_step(sum, qv.y, rv.y);
                sum = v0_step(sum, qv.z, rv.z);
                sum = v0_step(sum, qv.w, rv.w);
            }
        } else {
            for (int t = 0; t < k; ++t) sum = v0_step(sum, qi[t], rj[t]);
        }
        keys[i] = make_key(sum, index_base + idx);
    } else {
        keys[i] = NNS_KEY_NONE;
        const int pos = atomicAdd(&scal->amb_count, 1);
        amb_list[pos] = i;
    }
}

int launch_finalize(const FilterGeom &g, int k, int m, int n, const float *q, const float *r,
                    const Partial *partials, const float *qnorm, DevScalars *scal,
                    int64_t index_base, nns_key *keys, int *amb_list, hipStream_t st)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(divup(m, 256)), dim3(256), 0, st, g.kt, g.m_pad,
                       g.splits, k, m, n, q, r, partials, qnorm, scal, index_base, keys, amb_list);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

}  // namespace nns
