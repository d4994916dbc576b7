// Microbenchmark: approximate the filter's bare loop (2 waves/SIMD, QB = 2): per step one
// ds_read_b128 + 8 MFMAs on two accumulators; optional per-tile seed reads, epilogue-free.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: plain; 1: + seeds (4 ds_read_b128 into acc every 16 steps); 2: + walking LDS addresses (ring of 4 x 33 KB)
__global__ __launch_bounds__(512) void k(float *out, int iters, float a0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds = reinterpret_cast<float4 *>(smem);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8256; i += blockDim.x) lds[i] = make_float4(a0 + i, a0, a0 * 2, a0 * 3);
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = threadIdx.x * 0.001f;
    float b0[64], b1[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) { b0[j] = a0 + j + lane; b1[j] = a0 - j + lane; }
    float4 fr[4];
    for (int i = 0; i < iters; ++i) {
        const float4 *base = lds + (MODE == 2 ? (i & 3) * 2064 : 0);
        fr[0] = base[lane];
        fr[1] = base[64 + lane];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t + 2 < 32) fr[(t + 2) & 3] = base[((t + 2) & 31) * 64 + lane];
            if (MODE >= 1 && (t & 15) == 0) {
                const float4 *nrm = base + 2048 + (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 nv = nrm[2 * g];
                    acc0[4 * g] = nv.x; acc0[4 * g + 1] = nv.y; acc0[4 * g + 2] = nv.z; acc0[4 * g + 3] = nv.w;
                    acc1[4 * g] = nv.x; acc1[4 * g + 1] = nv.y; acc1[4 * g + 2] = nv.z; acc1[4 * g + 3] = nv.w;
                }
            }
            const float4 a = fr[t & 3];
            const int b = (t & 15) * 4;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0[b + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1[b + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0[b + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1[b + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0[b + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1[b + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0[b + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1[b + 3], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1 && (t & 15) == 15) asm volatile("" :: "v"(acc0), "v"(acc1));
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    int iters = 9000;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 135000);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 135000, 0, out, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double mf = 256.0 * 8 * (double)iters * 256;
        double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
        if (rep == 2) printf("%-44s %8.3f ms  %7.1f TF  (%.1f%% of 157.3)\n", name, ms, tf, tf / 157.3 * 100);
    }
    hipFree(out);
}

int main()
{
    run<0>("QB=2 loop, no seeds");
    run<1>("+ seeds every 16 steps");
    run<2>("+ walking ring addresses");
    return 0;
}
