// prep_kernels.hip — K2: the layout pre-pass in front of the MFMA filter.
//
// Role of the reference's mat_inv_kernel (core.cu:293-306 and its copies in
// v6..v9: AoS [n][k] -> SoA [k][n] so the fused kernel's loads coalesce) —
// re-designed for the MFMA operand feed of gfx950 instead of translated:
//
//   1. per-dimension mean c[t] of the reference shard (deterministic two-stage
//      reduction in fp64) — the clouds are centred before the -2*Q*R^T GEMM so
//      that |r'|^2 - 2 q'.r' loses as few bits as possible to cancellation
//      (SURVEY 7.3-1); correctness never depends on c, only the filter's
//      ambiguity margin tau does;
//   2. "tile image" of the centred (and, for refs, -2-scaled) points: per block of
//      32 points img[b][h][i][e] = v(point i, dim 8b+4h+e), the exact order in
//      which the 64 lanes of a wave consume float4 #b as operands of
//      v_mfma_f32_32x32x2_f32 k-steps 4b..4b+3.  One image block is 32*KT*4
//      contiguous bytes, so the filter's global->LDS DMA is a pure linear copy
//      and every ds_read_b128 of it is lane-linear (conflict-free);
//   3. squared norms of the centred points (fp64 accumulate, one fp32 rounding):
//      refs' norms seed the MFMA accumulators, queries' norms only enter tau;
//   4. max |value| and max norm (uint-ordered atomicMax on the fp32 bits), used to
//      bound the filter error and to detect NaN/INF/huge inputs, which are routed
//      to the exact kernels instead.
#include "nns_internal.h"

namespace nns {

constexpr int PREP_MEAN_BLOCKS = 1024;

// max-reduction into one device word: a plain read first (the word only grows), so that
// after the first few workgroups almost no atomic is issued — tens of thousands of
// workgroups hammering one address otherwise serialise the whole pre-pass
__device__ __forceinline__ void max_word(unsigned *word, unsigned v)
{
    if (v > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, v);
}

int prep_workspace_bytes(int kt, size_t *bytes)
{
    *bytes = (size_t)PREP_MEAN_BLOCKS * kt * sizeof(double);
    return NNS_OK;
}

// stage 1: per-block column sums.  Thread (rl, c4): c4 walks the dims four at a time (one
// 16-byte load per lane when k % 4 == 0), rl walks the rows, so a wave reads consecutive
// bytes.  Sums are fp64 in a fixed order -> the mean is deterministic.
template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(int k, int kt, int n, int rows_per_block,
                                                     const float *__restrict__ r,
                                                     double *__restrict__ partial,
                                                     unsigned *__restrict__ maxabs_bits)
{
    extern __shared__ double ssum[];   // [rl_count][kt]
    const int cgroups = kt / VEC;                       // column groups per row
    const int cw = cgroups < 256 ? cgroups : 256;
    const int rl_count = 256 / cw;
    const int g0 = threadIdx.x % cw;
    const int rl = threadIdx.x / cw;
    const int row0 = blockIdx.x * rows_per_block;
    int row1 = row0 + rows_per_block;
    if (row1 > n) row1 = n;
    unsigned mx = 0;
    for (int g = g0; g < cgroups; g += cw) {
        double acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.0;
        const int c = g * VEC;
        if (c < k && rl < rl_count)
            for (int j = row0 + rl; j < row1; j += rl_count) {
                float v[VEC];
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(r + (size_t)j * k + c);
                    v[0] = q.x;
                    v[VEC > 1 ? 1 : 0] = q.y;
                    v[VEC > 2 ? 2 : 0] = q.z;
                    v[VEC > 3 ? 3 : 0] = q.w;
                } else {
                    v[0] = r[(size_t)j * k + c];
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    acc[e] += (double)v[e];
                    const unsigned b = __float_as_uint(v[e]) & 0x7FFFFFFFu;
                    mx = b > mx ? b : mx;
                }
            }
        if (rl < rl_count)
#pragma unroll
            for (int e = 0; e < VEC; ++e) ssum[rl * kt + c + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < kt; c += 256) {
        double acc = 0.0;
        for (int i = 0; i < rl_count; ++i) acc += ssum[i * kt + c];   // fixed order
        partial[(size_t)blockIdx.x * kt + c] = acc;
    }
    // wave max then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) max_word(maxabs_bits, mx);
}

// stage 2: one workgroup per dimension, fixed-shape tree over the partials -> fp32 mean
__global__ __launch_bounds__(256) void mean_kernel(int kt, int nblocks, int n, const double *__restrict__ partial,
                                                   float *__restrict__ mean)
{
    __shared__ double red[256];
    const int c = blockIdx.x;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partial[(size_t)b * kt + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[c] = (float)(red[0] / (double)n);
}

int launch_prep_mean(int k, int kt, int n, const float *r, double *partial_ws, float *mean,
                     unsigned *maxabs_bits, hipStream_t st)
{
    int nblocks = PREP_MEAN_BLOCKS;
    int rows = divup(n, nblocks);
    if (rows < 64) rows = 64;
    nblocks = divup(n, rows);
    const bool vec = (k % 4 == 0) && (((uintptr_t)r & 15) == 0);
    const int cgroups = vec ? kt / 4 : kt;
    const int cw = cgroups < 256 ? cgroups : 256;
    const size_t lds = (size_t)(256 / cw) * kt * sizeof(double);
    if (vec)
        hipLaunchKernelGGL(colsum_kernel<4>, dim3(nblocks), dim3(256), lds, st, k, kt, n, rows, r, partial_ws,
                           maxabs_bits);
    else
        hipLaunchKernelGGL(colsum_kernel<1>, dim3(nblocks), dim3(256), lds, st, k, kt, n, rows, r, partial_ws,
                           maxabs_bits);
    NNS_HIP(hipGetLastError());
    hipLaunchKernelGGL(mean_kernel, dim3(kt), dim3(256), 0, st, kt, nblocks, n, partial_ws, mean);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// image kernel: one workgroup = one block of 32 points.
// BF = 1 (NNS_FILTER_BF16, fp32 points through the bf16 filter): the same centring and fp32 norms, but
// the image holds the centred, scaled values ROUNDED to bf16 (RNE) in the 16x16x32 operand order of
// image_bf16_kernel (order 1).  KT = 256 only.
template <int KT, int BF = 0>
__global__ __launch_bounds__(256) void image_kernel(int k, int npts, const float *__restrict__ pts,
                                                    const float *__restrict__ mean, float scale,
                                                    float pad_norm, float *__restrict__ img,
                                                    float *__restrict__ norms,
                                                    unsigned *__restrict__ max_norm_bits,
                                                    unsigned *__restrict__ maxabs_bits)
{
    constexpr int LD = KT + 4;                  // padded row (floats), keeps float4 alignment
    __shared__ __attribute__((aligned(16))) float tile[32 * LD];
    __shared__ double nrm[32][8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    const int p0 = blk * 32;

    // load + centre: the 32 rows are one contiguous span of 32*k floats
    unsigned mx = 0;
    if ((k & 3) == 0 && k <= KT && p0 + 32 <= npts && (((uintptr_t)pts) & 15) == 0) {
        // full block of rows: 16-byte loads, consecutive lanes -> consecutive bytes (the 32 rows
        // are one contiguous span of 32 * k floats); dims k .. KT-1 are zero padding
        const int k4 = k >> 2;
        const float4 *src = reinterpret_cast<const float4 *>(pts + (size_t)p0 * k);
        const float4 *mean4 = reinterpret_cast<const float4 *>(mean);
        if (k < KT)
            for (int e = tid; e < 32 * (KT / 4); e += 256) {
                const int i = e / (KT / 4), t4 = e - i * (KT / 4);
                if (t4 >= k4) *reinterpret_cast<float4 *>(&tile[i * LD + 4 * t4]) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        for (int e = tid; e < 32 * k4; e += 256) {
            const int i = e / k4, t4 = e - i * k4;
            const float4 v = src[e];
            const float4 mc = mean4[t4];
            const unsigned b0 = __float_as_uint(v.x) & 0x7FFFFFFFu, b1 = __float_as_uint(v.y) & 0x7FFFFFFFu;
            const unsigned b2 = __float_as_uint(v.z) & 0x7FFFFFFFu, b3 = __float_as_uint(v.w) & 0x7FFFFFFFu;
            const unsigned bm = max(max(b0, b1), max(b2, b3));
            mx = bm > mx ? bm : mx;
            float4 c;   // ONE rounding each: x' = fl(x - c)
            c.x = __fsub_rn(v.x, mc.x);
            c.y = __fsub_rn(v.y, mc.y);
            c.z = __fsub_rn(v.z, mc.z);
            c.w = __fsub_rn(v.w, mc.w);
            *reinterpret_cast<float4 *>(&tile[i * LD + 4 * t4]) = c;
        }
    } else {
        for (int e = tid; e < 32 * KT; e += 256) {
            const int i = e / KT, t = e - i * KT;
            float c = 0.0f;
            if (t < k && p0 + i < npts) {
                const float v = pts[(size_t)(p0 + i) * k + t];
                const unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;
                mx = b > mx ? b : mx;
                c = __fsub_rn(v, mean[t]);   // ONE rounding: x' = fl(x - c)
            }
            tile[i * LD + t] = c;
        }
    }
    __syncthreads();

    // squared norm of the centred row: 8 threads per row, fp64, fixed order
    {
        const int i = tid >> 3, part = tid & 7;
        double acc = 0.0;
        for (int t = part; t < KT; t += 8) {
            const double v = (double)tile[i * LD + t];
            acc += v * v;
        }
        nrm[i][part] = acc;
    }
    __syncthreads();
    if (tid < 32) {
        double acc = 0.0;
        for (int p = 0; p < 8; ++p) acc += nrm[tid][p];
        float nv = (float)acc;
        unsigned nb = 0;
        if (p0 + tid >= npts) nv = pad_norm;           // padding never wins (refs: +INF)
        else nb = __float_as_uint(nv);
        norms[p0 + tid] = nv;
        if (max_norm_bits) {                           // block max (lanes 0..31), one update
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                const unsigned o = __shfl_xor(nb, off, 64);
                nb = o > nb ? o : nb;
            }
            if (tid == 0) max_word(max_norm_bits, nb);
        }
    }

    if constexpr (BF != 0) {
        static_assert(BF == 0 || KT == 256 || KT == 128, "the bf16 operand image is 128 or 256 deep");
        // fragment f = NKS * tile + k-step; lane l: point 16 tile + (l & 15), dims 32 ks + 8 (l >> 4) .. + 7
        constexpr int NKS = KT / 32;
        uint4 *outb = reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(img) + (size_t)blk * 32 * KT);
        for (int f = tid; f < 2 * NKS * 64; f += 256) {
            const int s16 = f >> 6, lane = f & 63;
            const int i = 16 * (s16 / NKS) + (lane & 15), d0 = 32 * (s16 % NKS) + 8 * (lane >> 4);
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // scale (+1 / -2) is exact in fp32 and in bf16; ONE rounding to bf16, NaN stays NaN
                const __bf16 lo = (__bf16)(tile[i * LD + d0 + 2 * e] * scale);
                const __bf16 hi = (__bf16)(tile[i * LD + d0 + 2 * e + 1] * scale);
                w[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) |
                       ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
            }
            outb[f] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
    // write the image: float4 #f of the block = img[b][lane][0..3], lane = 32h + i
    float4 *out = reinterpret_cast<float4 *>(img + (size_t)blk * 32 * KT);
    for (int f = tid; f < (KT / 8) * 64; f += 256) {
        const int b = f >> 6, lane = f & 63;
        const int h = lane >> 5, i = lane & 31;
        const float4 v = *reinterpret_cast<const float4 *>(&tile[i * LD + 8 * b + 4 * h]);
        float4 o;
        o.x = v.x * scale;   // scale is +1 or -2: exact
        o.y = v.y * scale;
        o.z = v.z * scale;
        o.w = v.w * scale;
        out[f] = o;
    }
    }

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0 && maxabs_bits) max_word(maxabs_bits, mx);
}

// NNS_FILTER_BF16 at KT = 512 (fp32 points, 256 < k <= 512; operands in the 32x32x16 order 0 of
// image_bf16_kernel): the fp32 tile of image_kernel would not fit the 64 KiB of static LDS, so
// each thread centres, squares (fp64) and rounds a 64-dim run of one point on the fly and only
// the bf16 values are staged.  Same values as image_kernel<KT, 1> would produce.
__global__ __launch_bounds__(256) void image_mixed512_kernel(int order, int k, int npts, const float *__restrict__ pts,
                                                             const float *__restrict__ mean, float scale,
                                                             float pad_norm, uint16_t *__restrict__ img,
                                                             float *__restrict__ norms,
                                                             unsigned *__restrict__ max_norm_bits,
                                                             unsigned *__restrict__ maxabs_bits)
{
    constexpr int KT = 512, LD = KT + 8;
    __shared__ __attribute__((aligned(16))) uint16_t tile[32 * LD];
    __shared__ double nrm[32][8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    const int p0 = blk * 32;
    const int i = tid >> 3, part = tid & 7;   // 8 threads per point, 64 consecutive dims each
    const bool live = p0 + i < npts;
    const float *row = pts + (size_t)(p0 + i) * k;
    unsigned mx = 0;
    double acc = 0.0;
    for (int t = part * 64; t < part * 64 + 64; ++t) {
        float c = 0.0f;
        if (live && t < k) {
            const float v = row[t];
            const unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;
            mx = b > mx ? b : mx;
            c = __fsub_rn(v, mean[t]);   // ONE rounding: x' = fl(x - c)
        }
        acc += (double)c * (double)c;
        // scale (+1 / -2) is exact; ONE rounding to bf16 (RNE), NaN stays NaN
        tile[i * LD + t] = __builtin_bit_cast(unsigned short, (__bf16)(c * scale));
    }
    nrm[i][part] = acc;
    __syncthreads();
    if (tid < 32) {
        double a2 = 0.0;
        for (int p = 0; p < 8; ++p) a2 += nrm[tid][p];   // fixed order
        float nv = (float)a2;
        unsigned nb = 0;
        if (p0 + tid >= npts) nv = pad_norm;             // padding never wins (refs: +INF)
        else nb = __float_as_uint(nv);
        norms[p0 + tid] = nv;
        if (max_norm_bits) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                const unsigned o = __shfl_xor(nb, off, 64);
                nb = o > nb ? o : nb;
            }
            if (tid == 0) max_word(max_norm_bits, nb);
        }
    }
    // 16-byte fragments: f = s * 64 + lane.  order 0 (32x32x16 operands): lane = 32 h + i, dims 16 s + 8 h ..;
    // order 1 (16x16x32 operands): fragment (KT / 32) t + ks, lane: point 16 t + (l & 15), dims 32 ks + 8 (l >> 4) ..
    uint4 *out = reinterpret_cast<uint4 *>(img + (size_t)blk * 32 * KT);
    for (int f = tid; f < (KT / 16) * 64; f += 256) {
        const int s16 = f >> 6, lane = f & 63;
        int i, d0;
        if (order == 0) {
            i = lane & 31;
            d0 = 16 * s16 + 8 * (lane >> 5);
        } else {
            constexpr int NKS = KT / 32;
            i = 16 * (s16 / NKS) + (lane & 15);
            d0 = 32 * (s16 % NKS) + 8 * (lane >> 4);
        }
        out[f] = *reinterpret_cast<const uint4 *>(&tile[i * LD + d0]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0 && maxabs_bits) max_word(maxabs_bits, mx);
}

// KT = 1024 (512 < k <= 1024), operands of v_mfma_f32_32x32x16_bf16 (order 0), for BOTH point types: a
// 32-point block is 64 KiB of bf16 — more than one staging tile — so the block is produced in two halves of
// 512 dims through the same LDS tile; the squared norm accumulates in a register across the halves (fp64,
// fixed order).  T = float: centred by the mean, ONE rounding to bf16 (NNS_FILTER_BF16 / AUTO for fp32 points
// beyond the fp32 tiles); T = uint16_t: bf16 bit patterns, no centring (a centred bf16 value would need a
// second rounding).  Same values as image_mixed512_kernel / image_bf16_kernel would produce at their depths.
// (KT = 768: three parts of 256 dims; KT = 640 / 384: five / three parts of 128.)
template <int KT, typename T, int HD = 512>
__global__ __launch_bounds__(256) void image_deep_kernel(int k, int npts, const T *__restrict__ pts,
                                                         const float *__restrict__ mean, float scale, float pad_norm,
                                                         uint16_t *__restrict__ img, float *__restrict__ norms,
                                                         unsigned *__restrict__ max_norm_bits,
                                                         unsigned *__restrict__ maxabs_bits)
{
    static_assert(KT % HD == 0 && HD % 64 == 0, "parts of HD dims, 8 threads per point");
    constexpr int LD = HD + 8, TPD = HD / 8;   // dims per thread and part
    __shared__ __attribute__((aligned(16))) uint16_t tile[32 * LD];
    __shared__ double nrm[32][8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    const int p0 = blk * 32;
    const int i = tid >> 3, part = tid & 7;   // 8 threads per point, 64 consecutive dims each per half
    const bool live = p0 + i < npts;
    const T *row = pts + (size_t)(p0 + i) * k;
    unsigned mx = 0;
    double acc = 0.0;
    uint4 *out = reinterpret_cast<uint4 *>(img + (size_t)blk * 32 * KT);
    for (int half = 0; half < KT / HD; ++half) {
        for (int tt = part * TPD; tt < part * TPD + TPD; ++tt) {
            const int t = half * HD + tt;
            float c = 0.0f;
            if (live && t < k) {
                float v;
                if constexpr (sizeof(T) == 4) v = (float)row[t];
                else v = __uint_as_float((unsigned)row[t] << 16);
                const unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;
                mx = b > mx ? b : mx;
                if constexpr (sizeof(T) == 4) c = __fsub_rn(v, mean[t]);   // ONE rounding: x' = fl(x - c)
                else c = v;
            }
            acc += (double)c * (double)c;
            // scale (+1 / -2) is exact; fp32 points: ONE rounding to bf16 (RNE), NaN stays NaN; bf16 points: exact
            tile[i * LD + tt] = __builtin_bit_cast(unsigned short, (__bf16)(c * scale));
        }
        __syncthreads();
        // 16-byte fragments of the 32x32x16 operand order: fragment s (16 dims), lane = 32 h + i
        for (int f = tid; f < (HD / 16) * 64; f += 256) {
            const int s16 = f >> 6, lane = f & 63;
            out[(half * (HD / 16) + s16) * 64 + lane] =
                *reinterpret_cast<const uint4 *>(&tile[(lane & 31) * LD + 16 * s16 + 8 * (lane >> 5)]);
        }
        __syncthreads();
    }
    nrm[i][part] = acc;
    __syncthreads();
    if (tid < 32) {
        double a2 = 0.0;
        for (int p = 0; p < 8; ++p) a2 += nrm[tid][p];   // fixed order
        float nv = (float)a2;
        unsigned nb = 0;
        if (p0 + tid >= npts) nv = pad_norm;             // padding never wins (refs: +INF)
        else nb = __float_as_uint(nv);
        norms[p0 + tid] = nv;
        if (max_norm_bits) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                const unsigned o = __shfl_xor(nb, off, 64);
                nb = o > nb ? o : nb;
            }
            if (tid == 0) max_word(max_norm_bits, nb);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0 && maxabs_bits) max_word(maxabs_bits, mx);
}

int launch_prep_image(int k, int kt, int npts, int npts_pad, const float *pts, const float *mean,
                      float scale, float pad_norm, float *img, float *norms,
                      unsigned *max_norm_bits, unsigned *maxabs_bits, hipStream_t st, bool out_bf16)
{
    const int blocks = npts_pad / 32;
    if (out_bf16) {
        if (kt != 1024 && kt != 768 && kt != 640 && kt != 512 && kt != 384 && kt != 256 && kt != 128) {
            set_error("prep: the bf16 operand image is 128, 256, 384, 512, 640, 768 or 1024 deep (kt = %d)", kt);
            return NNS_ERR_UNSUPPORTED;
        }
        if (kt == 384)
            hipLaunchKernelGGL((image_deep_kernel<384, float, 128>), dim3(blocks), dim3(256), 0, st, k, npts, pts, mean, scale,
                               pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
        else if (kt == 640)
            hipLaunchKernelGGL((image_deep_kernel<640, float, 128>), dim3(blocks), dim3(256), 0, st, k, npts, pts, mean, scale,
                               pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
        else if (kt == 768)
            hipLaunchKernelGGL((image_deep_kernel<768, float, 256>), dim3(blocks), dim3(256), 0, st, k, npts, pts, mean, scale,
                               pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
        else if (kt == 1024)
            hipLaunchKernelGGL((image_deep_kernel<1024, float>), dim3(blocks), dim3(256), 0, st, k, npts, pts, mean, scale,
                               pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
        else if (kt == 512)
            hipLaunchKernelGGL(image_mixed512_kernel, dim3(blocks), dim3(256), 0, st, NNS_BF16_TILE16 ? 1 : 0, k, npts, pts, mean,
                               scale, pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
        else if (kt == 256)
            hipLaunchKernelGGL((image_kernel<256, 1>), dim3(blocks), dim3(256), 0, st, k, npts, pts, mean, scale,
                               pad_norm, img, norms, max_norm_bits, maxabs_bits);
        else
            hipLaunchKernelGGL((image_kernel<128, 1>), dim3(blocks), dim3(256), 0, st, k, npts, pts, mean, scale,
                               pad_norm, img, norms, max_norm_bits, maxabs_bits);
        NNS_HIP(hipGetLastError());
        return NNS_OK;
    }
    switch (kt) {
    case 16:
        hipLaunchKernelGGL(image_kernel<16>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
                           scale, pad_norm, img, norms, max_norm_bits, maxabs_bits);
        break;
    case 32:
        hipLaunchKernelGGL(image_kernel<32>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
                           scale, pad_norm, img, norms, max_norm_bits, maxabs_bits);
        break;
    case 64:
        hipLaunchKernelGGL(image_kernel<64>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
                           scale, pad_norm, img, norms, max_norm_bits, maxabs_bits);
        break;
    case 128:
        hipLaunchKernelGGL(image_kernel<128>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
                           scale, pad_norm, img, norms, max_norm_bits, maxabs_bits);
        break;
    case 256:
        hipLaunchKernelGGL(image_kernel<256>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
                           scale, pad_norm, img, norms, max_norm_bits, maxabs_bits);
        break;
    default:
        set_error("prep: unsupported tile K %d", kt);
        return NNS_ERR_UNSUPPORTED;
    }
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// ---- bf16 image (K4's operands) --------------------------------------------------------
// One workgroup = one block of 32 points, KT = 256.  The image is [16 steps][64 lanes][8 bf16]:
// lane (i = lane & 31, h = lane >> 5) of step s holds dims 16s + 8h .. 16s + 8h + 7 of point
// i — one v_mfma_f32_32x32x16_bf16 operand per ds_read_b128.  No centring (a centred bf16
// value would need a second rounding); the scale (-2 for refs) is exact in bf16.
//
// order 1 (v_mfma_f32_16x16x32_bf16 operands, the product): a fragment is one 16-point tile t x
// one k-step ks of 32 dims; lane l holds dims 32 ks + 8 (l >> 4) .. + 7 of point 16 t + (l & 15);
// the block stores fragment (KT / 32) t + ks (the filter walks a ref tile's k-steps in a row; a wave
// keeps all k-steps of its four query tiles in registers).  order 0: the 32x32x16 layout above.
template <int KT>
__global__ __launch_bounds__(256) void image_bf16_kernel(int order, int k, int npts, const uint16_t *__restrict__ pts,
                                                         float scale, float pad_norm,
                                                         uint16_t *__restrict__ img, float *__restrict__ norms,
                                                         unsigned *__restrict__ max_norm_bits,
                                                         unsigned *__restrict__ maxabs_bits)
{
    constexpr int LD = KT + 8, NKS = KT / 32;
    __shared__ __attribute__((aligned(16))) uint16_t tile[32 * LD];
    __shared__ double nrm[32][8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    const int p0 = blk * 32;
    unsigned mx = 0;
    if (k == KT && p0 + 32 <= npts && (((uintptr_t)pts) & 15) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(pts + (size_t)p0 * KT);   // 8 bf16 per load
        for (int e = tid; e < 32 * (KT / 8); e += 256) {
            const int i = e / (KT / 8), t8 = e - i * (KT / 8);
            const uint4 v = src[e];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned lo = (w[u] << 16) & 0x7FFFFFFFu, hi = w[u] & 0x7FFF0000u;
                mx = max(mx, max(lo, hi));
            }
            *reinterpret_cast<uint4 *>(&tile[i * LD + 8 * t8]) = v;
        }
    } else {
        for (int e = tid; e < 32 * KT; e += 256) {
            const int i = e / KT, t = e - i * KT;
            uint16_t v = 0;
            if (t < k && p0 + i < npts) {
                v = pts[(size_t)(p0 + i) * k + t];
                const unsigned b = ((unsigned)v << 16) & 0x7FFFFFFFu;
                mx = b > mx ? b : mx;
            }
            tile[i * LD + t] = v;
        }
    }
    __syncthreads();
    {
        const int i = tid >> 3, part = tid & 7;
        double acc = 0.0;
        for (int t = part; t < KT; t += 8) {
            const double v = (double)__uint_as_float((unsigned)tile[i * LD + t] << 16);
            acc += v * v;
        }
        nrm[i][part] = acc;
    }
    __syncthreads();
    if (tid < 32) {
        double acc = 0.0;
        for (int p = 0; p < 8; ++p) acc += nrm[tid][p];
        float nv = (float)acc;
        unsigned nb = 0;
        if (p0 + tid >= npts) nv = pad_norm;
        else nb = __float_as_uint(nv);
        norms[p0 + tid] = nv;
        if (max_norm_bits) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                const unsigned o = __shfl_xor(nb, off, 64);
                nb = o > nb ? o : nb;
            }
            if (tid == 0) max_word(max_norm_bits, nb);
        }
    }
    // 16-byte fragments: f = s * 64 + lane
    uint4 *out = reinterpret_cast<uint4 *>(img + (size_t)blk * 32 * KT);
    for (int f = tid; f < (KT / 16) * 64; f += 256) {
        const int s = f >> 6, lane = f & 63;
        int i, d0;   // point of the block and first dim of this lane's 8
        if (order == 0) {
            i = lane & 31;
            d0 = 16 * s + 8 * (lane >> 5);
        } else {
            const int t = s / NKS, ks = s % NKS;
            i = 16 * t + (lane & 15);
            d0 = 32 * ks + 8 * (lane >> 4);
        }
        const uint4 sv = *reinterpret_cast<const uint4 *>(&tile[i * LD + d0]);
        const unsigned in[4] = {sv.x, sv.y, sv.z, sv.w};
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // scale in fp32 and narrow back: exact for scale in {1, -2} (no overflow below 1e17)
            const float lo = __uint_as_float(in[e] << 16) * scale;
            const float hi = __uint_as_float(in[e] & 0xFFFF0000u) * scale;
            w[e] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
        }
        out[f] = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0 && maxabs_bits) max_word(maxabs_bits, mx);
}

int launch_prep_image_bf16(int order, int kt, int k, int npts, int npts_pad, const uint16_t *pts, float scale, float pad_norm,
                           void *img, float *norms, unsigned *max_norm_bits, unsigned *maxabs_bits,
                           hipStream_t st)
{
    if (kt == 256)
        hipLaunchKernelGGL(image_bf16_kernel<256>, dim3(npts_pad / 32), dim3(256), 0, st, order, k, npts, pts, scale,
                           pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else if (kt == 512)   // (either operand order)
        hipLaunchKernelGGL(image_bf16_kernel<512>, dim3(npts_pad / 32), dim3(256), 0, st, order, k, npts, pts, scale,
                           pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else if (kt == 1024 && order == 0)
        hipLaunchKernelGGL((image_deep_kernel<1024, uint16_t>), dim3(npts_pad / 32), dim3(256), 0, st, k, npts, pts,
                           (const float *)nullptr, scale, pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else if (kt == 384 && order == 0)
        hipLaunchKernelGGL((image_deep_kernel<384, uint16_t, 128>), dim3(npts_pad / 32), dim3(256), 0, st, k, npts, pts,
                           (const float *)nullptr, scale, pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else if (kt == 640 && order == 0)
        hipLaunchKernelGGL((image_deep_kernel<640, uint16_t, 128>), dim3(npts_pad / 32), dim3(256), 0, st, k, npts, pts,
                           (const float *)nullptr, scale, pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else if (kt == 768 && order == 0)
        hipLaunchKernelGGL((image_deep_kernel<768, uint16_t, 256>), dim3(npts_pad / 32), dim3(256), 0, st, k, npts, pts,
                           (const float *)nullptr, scale, pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else if (kt == 128 && order == 1)
        hipLaunchKernelGGL(image_bf16_kernel<128>, dim3(npts_pad / 32), dim3(256), 0, st, order, k, npts, pts, scale,
                           pad_norm, (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    else {
        set_error("prep: unsupported bf16 tile K %d (order %d)", kt, order);
        return NNS_ERR_UNSUPPORTED;
    }
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// ---- SoA -> AoS (NNS_REFS_SOA) --------------------------------------------------------------
// Tile = 32 dims x 64 points through LDS: reads are 64 consecutive points of one dimension row
// (256 B / 128 B runs), writes 32 consecutive dims of one point (128 B / 64 B runs).  HBM-bound:
// 2 * n * k * esz bytes.  (The reference's mat_inv_kernel, core.cu:293-306, goes the other way
// with one thread per element and strided writes.)
template <typename T>
__global__ __launch_bounds__(256) void soa_to_aos_kernel(int k, int n, const T *__restrict__ src, T *__restrict__ dst)
{
    __shared__ T tile[32][64 + 2];
    const int j0 = blockIdx.x * 64, t0 = blockIdx.y * 32;
    for (int e = threadIdx.x; e < 32 * 64; e += 256) {
        const int t = e >> 6, j = e & 63;
        if (t0 + t < k && j0 + j < n) tile[t][j] = src[(size_t)(t0 + t) * n + j0 + j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int j = e >> 5, t = e & 31;
        if (t0 + t < k && j0 + j < n) dst[(size_t)(j0 + j) * k + t0 + t] = tile[t][j];
    }
}

int launch_soa_to_aos(int k, int n, const void *src, void *dst, int esz, hipStream_t st)
{
    const dim3 grid(divup(n, 64), divup(k, 32));
    if (esz == 4)
        hipLaunchKernelGGL(soa_to_aos_kernel<float>, grid, dim3(256), 0, st, k, n, (const float *)src, (float *)dst);
    else
        hipLaunchKernelGGL(soa_to_aos_kernel<uint16_t>, grid, dim3(256), 0, st, k, n, (const uint16_t *)src,
                           (uint16_t *)dst);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

}  // namespace nns
