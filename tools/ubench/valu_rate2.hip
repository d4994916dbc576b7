// Microbenchmark: fp32 VALU issue rate on gfx950 vs waves per SIMD, independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters)
{
    float a[16], x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = threadIdx.x * 0.001f + i;
        x[i] = in[(threadIdx.x + i) & 255];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x[i]));
                if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x[i]));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x[i]));
                if (MODE == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                if (MODE == 4) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&a[i & 14]) : "v"(*(double *)&x[i & 14]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    float *out, *in;
    (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
    (void)hipMalloc(&in, 1024);
    (void)hipMemset(in, 0, 1024);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("%-22s", name);
    for (int w : {1, 2, 3, 4, 8}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, in, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double winstr_per_simd = (double)w * iters * 64;
        printf("  w=%d: %.2f cyc/instr", w, ms * 1e-3 * 2.4e9 / winstr_per_simd);
    }
    printf("\n");
}

int main()
{
    run<0>("v_add_f32 v,v,v");
    run<1>("v_mul_f32 v,v,v");
    run<2>("v_fma_f32");
    run<3>("v_sub_f32 (no dep)");
    run<4>("v_min3_f32");
    run<5>("v_pk_add_f32");
    return 0;
}
