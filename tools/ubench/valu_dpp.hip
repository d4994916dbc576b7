// Microbenchmark: does a DPP row broadcast ride along for free?  v_fmac_f32 against v_fmac_f32_dpp row_newbcast:n
// (gfx90a+: lane n of each 16-lane row feeds every lane of the row) and v_mov_b32_dpp, 8 waves per SIMD, independent chains.
// What it is for: K1f reads every ref by a broadcast ds_read_b128 (its walk runs at the LDS's pace); with the refs of a chunk
// lane-striped in FOUR registers (one lane-linear read per 16 refs) a ref could reach the lanes through the FMA's own DPP
// operand instead.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *src, int iters)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    float r0 = src[threadIdx.x & 63], r1 = src[64 + (threadIdx.x & 63)], x = src[128 + (threadIdx.x & 63)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(r0), "v"(x));
                if (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(r0), "v"(x));
                if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(r1));
                if (MODE == 3) {   // the chain K1f would run: mov_dpp (norm) + 3 fmac_dpp
                    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(r1));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(r0), "v"(x));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(r1), "v"(x));
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(r0), "v"(x));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int ops)
{
    float *out, *src;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipMalloc(&src, 256 * sizeof(float));
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0.001f * i;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, out, src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winstr = 2048.0 * 4 * (double)iters * 64 * ops;
        const double per = winstr / 1024.0 / (ms * 1e-3 * 2.4e9);
        if (rep == 2) printf("%-44s %8.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, ms, 1.0 / per);
    }
}

int main()
{
    run<0>("v_fmac_f32 v, v, v", 1);
    run<1>("v_fmac_f32_dpp row_newbcast", 1);
    run<2>("v_mov_b32_dpp row_newbcast", 1);
    run<3>("mov_dpp + 3 fmac_dpp (one score)", 4);
    return 0;
}
