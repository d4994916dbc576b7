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

int prep_workspace_bytes(int kt, size_t *bytes)
{
    *bytes = (size_t)PREP_MEAN_BLOCKS * kt * sizeof(double);
    return NNS_OK;
}

// stage 1: per-block column sums.  Thread (rl, col): col = tid % cw walks the
// dims, rl = tid / cw walks the rows, so a wave reads consecutive floats.
__global__ __launch_bounds__(256) void colsum_kernel(int k, int kt, int n, int rows_per_block,
                                                     const float *__restrict__ r,
                                                     double *__restrict__ partial,
                                                     unsigned *__restrict__ maxabs_bits)
{
    extern __shared__ double ssum[];   // [rl_count][kt]
    const int cw = kt < 256 ? kt : 256;
    const int rl_count = 256 / cw;
    const int col0 = threadIdx.x % cw;
    const int rl = threadIdx.x / cw;
    const int row0 = blockIdx.x * rows_per_block;
    int row1 = row0 + rows_per_block;
    if (row1 > n) row1 = n;
    unsigned mx = 0;
    for (int c = col0; c < kt; c += cw) {
        double acc = 0.0;
        if (c < k && rl < rl_count)
            for (int j = row0 + rl; j < row1; j += rl_count) {
                const float v = r[(size_t)j * k + c];
                acc += (double)v;
                const unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;
                mx = b > mx ? b : mx;
            }
        if (rl < rl_count) ssum[rl * kt + c] = acc;
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
    if ((threadIdx.x & 63) == 0) atomicMax(maxabs_bits, mx);
}

// stage 2: fixed-order sum of the partials -> fp32 mean
__global__ void mean_kernel(int kt, int nblocks, int n, const double *__restrict__ partial,
                            float *__restrict__ mean)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= kt) return;
    double acc = 0.0;
    for (int b = 0; b < nblocks; ++b) acc += partial[(size_t)b * kt + c];
    mean[c] = (float)(acc / (double)n);
}

int launch_prep_mean(int k, int kt, int n, const float *r, double *partial_ws, float *mean,
                     unsigned *maxabs_bits, hipStream_t st)
{
    int nblocks = PREP_MEAN_BLOCKS;
    int rows = divup(n, nblocks);
    if (rows < 64) rows = 64;
    nblocks = divup(n, rows);
    const int cw = kt < 256 ? kt : 256;
    const size_t lds = (size_t)(256 / cw) * kt * sizeof(double);
    hipLaunchKernelGGL(colsum_kernel, dim3(nblocks), dim3(256), lds, st, k, kt, n, rows, r,
                       partial_ws, maxabs_bits);
    NNS_HIP(hipGetLastError());
    hipLaunchKernelGGL(mean_kernel, dim3(divup(kt, 64)), dim3(64), 0, st, kt, nblocks, n,
                       partial_ws, mean);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// image kernel: one workgroup = one block of 32 points.
template <int KT>
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
        if (p0 + tid >= npts) nv = pad_norm;           // padding never wins (refs: +INF)
        else if (max_norm_bits) atomicMax(max_norm_bits, __float_as_uint(nv));
        norms[p0 + tid] = nv;
    }

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

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0 && maxabs_bits) atomicMax(maxabs_bits, mx);
}

int launch_prep_image(int k, int kt, int npts, int npts_pad, const float *pts, const float *mean,
                      float scale, float pad_norm, float *img, float *norms,
                      unsigned *max_norm_bits, unsigned *maxabs_bits, hipStream_t st)
{
    const int blocks = npts_pad / 32;
    switch (kt) {
    case 64:
        hipLaunchKernelGGL(image_kernel<64>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
                           scale, pad_norm, img, norms, max_norm_bits, maxabs_bits);
        break;
    case 128:
        hipLaunchKernelGGL(image_kernel<128>, dim3(blocks), dim3(256), 0, st, k, npts, pts, mean,
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
__global__ __launch_bounds__(256) void image_bf16_kernel(int k, int npts, const uint16_t *__restrict__ pts,
                                                         float scale, float pad_norm,
                                                         uint16_t *__restrict__ img, float *__restrict__ norms,
                                                         unsigned *__restrict__ max_norm_bits,
                                                         unsigned *__restrict__ maxabs_bits)
{
    constexpr int KT = 256, LD = KT + 8;
    __shared__ __attribute__((aligned(16))) uint16_t tile[32 * LD];
    __shared__ double nrm[32][8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    const int p0 = blk * 32;
    unsigned mx = 0;
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
        if (p0 + tid >= npts) nv = pad_norm;
        else if (max_norm_bits) atomicMax(max_norm_bits, __float_as_uint(nv));
        norms[p0 + tid] = nv;
    }
    // 16-byte fragments: f = s * 64 + lane
    uint4 *out = reinterpret_cast<uint4 *>(img + (size_t)blk * 32 * KT);
    for (int f = tid; f < 16 * 64; f += 256) {
        const int s = f >> 6, lane = f & 63;
        const int h = lane >> 5, i = lane & 31;
        const uint16_t *src = &tile[i * LD + 16 * s + 8 * h];
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // scale in fp32 and narrow back: exact for scale in {1, -2} (no overflow below 1e17)
            const float lo = __uint_as_float((unsigned)src[2 * e] << 16) * scale;
            const float hi = __uint_as_float((unsigned)src[2 * e + 1] << 16) * scale;
            w[e] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
        }
        out[f] = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0 && maxabs_bits) atomicMax(maxabs_bits, mx);
}

int launch_prep_image_bf16(int k, int npts, int npts_pad, const uint16_t *pts, float scale, float pad_norm,
                           void *img, float *norms, unsigned *max_norm_bits, unsigned *maxabs_bits,
                           hipStream_t st)
{
    hipLaunchKernelGGL(image_bf16_kernel, dim3(npts_pad / 32), dim3(256), 0, st, k, npts, pts, scale, pad_norm,
                       (uint16_t *)img, norms, max_norm_bits, maxabs_bits);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

}  // namespace nns
