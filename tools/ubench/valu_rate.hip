// Microbenchmark: fp32 VALU issue rate on gfx950 (ops per cycle per SIMD), 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *sc, int iters)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    float s0 = sc[0], s1 = sc[1];   // uniform -> SGPR
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __fmul_rn(a[i], a[(i + 1) & 7]);           // v_mul v,v,v
                if (MODE == 1) a[i] = __fsub_rn(s0, a[i]);                       // v_sub v,s,v
                if (MODE == 2) a[i] = __fmaf_rn(a[i], a[(i + 1) & 7], a[(i + 2) & 7]);   // v_fma
                if (MODE == 3) { float d = __fsub_rn(a[i], s0); a[i] = __fadd_rn(a[(i+1)&7], __fmul_rn(d, d)); } // sub,mul,add
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + s1;
}

typedef float f2 __attribute__((ext_vector_type(2)));

// packed fp32: one wave-instruction does two lanes' worth of ops
template <int MODE>
__global__ __launch_bounds__(256) void kp(float *out, const float *sc, int iters)
{
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
            }
        }
    }
    f2 s = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + sc[1];
}

template <int MODE>
void runp(const char *name)
{
    float *out, *sc;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipMalloc(&sc, 16);
    float h[2] = {1.0001f, 0.5f};
    hipMemcpy(sc, h, 8, hipMemcpyHostToDevice);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kp<MODE>, dim3(2048), dim3(256), 0, 0, out, sc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double winstr = 2048.0 * 4 * (double)iters * 64;
        double per_simd_cycle = winstr / 1024.0 / (ms * 1e-3 * 2.4e9);
        if (rep == 2) printf("%-28s %8.3f ms  -> %.2f cycles per packed wave-instr\n", name, ms, 1.0 / per_simd_cycle);
    }
}

template <int MODE>
void run(const char *name, int ops_per_inner)
{
    float *out, *sc;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipMalloc(&sc, 16);
    float h[2] = {1.0001f, 0.5f};
    hipMemcpy(sc, h, 8, hipMemcpyHostToDevice);
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, out, sc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double winstr = 2048.0 * 4 * (double)iters * 64 * ops_per_inner;   // wave-instructions
        double per_simd_cycle = winstr / 1024.0 / (ms * 1e-3 * 2.4e9);
        if (rep == 2) printf("%-28s %8.3f ms  %.3f wave-instr / cycle / SIMD (at 2.4 GHz)  -> %.2f cycles per wave-instr\n", name, ms, per_simd_cycle, 1.0 / per_simd_cycle);
    }
}

int main()
{
    run<0>("v_mul_f32 v,v,v", 1);
    run<1>("v_sub_f32 v,s,v", 1);
    run<2>("v_fma_f32 v,v,v,v", 1);
    run<3>("sub(s) + mul + add", 3);
    runp<0>("v_pk_mul_f32");
    runp<1>("v_pk_add_f32");
    runp<2>("v_pk_fma_f32");
    runp<3>("v_pk_add_f32 neg (sub)");
    return 0;
}
