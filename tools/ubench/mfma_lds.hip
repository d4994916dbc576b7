// Microbenchmark: MFMA chain fed by ds_read_b128 (one read per 4 MFMAs), 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: no LDS; 1: ds_read_b128 prefetch 2; 2: same + sched_barrier; 3: B operands from 64 regs
__global__ __launch_bounds__(512) void k(float *out, int iters, float a0)
{
    __shared__ __attribute__((aligned(16))) float4 lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(a0 + i, a0, a0 * 2, a0 * 3);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = threadIdx.x * 0.001f;
    float b[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) b[j] = a0 + j + lane;
    float4 fr[4];
    fr[0] = lds[lane];
    fr[1] = lds[64 + lane];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (MODE >= 1) fr[(t + 2) & 3] = lds[((t + 2) & 15) * 64 + lane];
            float4 a = MODE >= 1 ? fr[t & 3] : make_float4(a0, a0, a0, a0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * t + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * t + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * t + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * t + 3], acc, 0, 0, 0);
            if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double mf = 256.0 * 8 * (double)iters * 64;
        double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
        if (rep == 2) printf("%-40s %8.3f ms  %7.1f TF  (%.1f%% of 157.3)\n", name, ms, tf, tf / 157.3 * 100);
    }
    hipFree(out);
}

int main()
{
    run<0>("no LDS, B from 64 regs");
    run<1>("ds_read_b128 per 4 MFMA, prefetch 2");
    run<2>("same + sched_barrier per step");
    return 0;
}
