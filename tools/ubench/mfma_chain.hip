// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate for dependent chains.
//   variant CH = independent accumulator chains per wave, WPS = waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 0.001f + c;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
void run(int threads, const char *name)
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    int iters = 20000 / CH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, iters, 0.5f, 0.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double mf = 256.0 * (threads / 64) * (double)iters * 16 * CH;
        double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
        if (rep == 2) printf("%-34s %8.3f ms  %7.1f TF  (%.1f%% of 157.3)\n", name, ms, tf, tf / 157.3 * 100);
    }
    hipFree(out);
}

int main()
{
    run<1>(256, "1 wave/SIMD, 1 dependent chain");
    run<2>(256, "1 wave/SIMD, 2 chains");
    run<4>(256, "1 wave/SIMD, 4 chains");
    run<1>(512, "2 waves/SIMD, 1 chain each");
    run<2>(512, "2 waves/SIMD, 2 chains each");
    return 0;
}
