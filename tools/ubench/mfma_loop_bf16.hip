// Microbenchmark: the bf16 filter's bare loop in isolation — per step one ds_read_b128 (1 KiB
// fragment) + four in-place v_mfma_f32_16x16x32_bf16 (asm, as in the product), walking a
// 4 x 32 KiB LDS ring, on RANDOM operands (the clock the chip holds depends on the data).
// Reports TFLOP/s, MFMA-pipe occupancy at the measured in-kernel clock, and the clock.
//   MODE 0: MFMAs only (operands in registers)      MODE 1: + one ds_read_b128 per step (prefetch 2)
//   WAVES 4 / 8 per workgroup = 1 / 2 per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma(const f32x4 &a, const f32x4 &b, f32x4 &c)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float *out, const f32x4 *src, int iters, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(i + blockIdx.x * 37) & 8191];
    f32x4 bq[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[q][s] = src[(threadIdx.x * 8 + q * 2048 + s * 64 + blockIdx.x) & 8191];
    __syncthreads();
    f32x4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fr[4];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        const f32x4 *base = lds + (i & 3) * 2048;
        if (MODE == 1) {
            fr[0] = base[lane];
            fr[1] = base[64 + lane];
        } else if (i == 0) {
            fr[0] = lds[lane]; fr[1] = lds[64 + lane]; fr[2] = lds[128 + lane]; fr[3] = lds[192 + lane];
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (MODE == 1 && t + 2 < 32) fr[(t + 2) & 3] = base[(t + 2) * 64 + lane];
            const int rt = (t >> 3) & 1, ks = t & 7;
#pragma unroll
            for (int q = 0; q < 4; ++q) mma(fr[t & 3], bq[q][ks], acc[rt][q]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 7\n\ts_nop 7");
    float s = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += acc[r][q][0] + acc[r][q][1] + acc[r][q][2] + acc[r][q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE, int WAVES>
void run(const char *name, const f32x4 *src)
{
    float *out;
    unsigned long long *st;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&st, 512 * sizeof(unsigned long long));
    const int iters = 20000 * 8 / WAVES / 4;   // ~ the same wall time for every variant
    (void)hipFuncSetAttribute((const void *)k<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 131072, 0, out, src, iters, st);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(512);
    (void)hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < 256; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz[128];
    const double mfmas = 256.0 * WAVES * (double)iters * 32 * 4;
    const double tf = mfmas * 16384.0 / (ms * 1e-3) / 1e12;
    const double busy = mfmas * 16.0 / 1024.0 / (ms * 1e-3 * clk * 1e9);
    printf("%-34s %7.2f ms  %6.0f TFLOP/s (%4.1f %% of 2.5 PF)  clock %.2f GHz  MFMA pipe %.1f %% busy\n", name, ms, tf,
           tf / 2500 * 100, clk, busy * 100);
    (void)hipFree(out);
    (void)hipFree(st);
}

static unsigned host_bf16()
{
    const float v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    unsigned u;
    ::memcpy(&u, &v, 4);
    return u >> 16;
}

// B operand from the accumulation-register half of the unified file ("a"): arch VGPRs stop at 256
__device__ __forceinline__ void mma_a(const f32x4 &a, const f32x4 &b, f32x4 &c)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}

// MODE 2: 128 queries per wave (8 query tiles, 256 operand registers, ONE wave per SIMD): one
// ds_read_b128 per EIGHT MFMAs — half the LDS bytes per flop of the product's shape
__global__ __launch_bounds__(256) void k8(float *out, const f32x4 *src, int iters, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(i + blockIdx.x * 37) & 8191];
    f32x4 bq[8][8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[q][s] = src[(threadIdx.x * 8 + q * 1024 + s * 64 + blockIdx.x) & 8191];
    __syncthreads();
    f32x4 acc[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fr[4];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        const f32x4 *base = lds + (i & 3) * 2048;
        fr[0] = base[lane];
        fr[1] = base[64 + lane];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t + 2 < 32) fr[(t + 2) & 3] = base[(t + 2) * 64 + lane];
            const int rt = (t >> 3) & 1, ks = t & 7;
#pragma unroll
            for (int q = 0; q < 8; ++q) mma_a(fr[t & 3], bq[q][ks], acc[rt][q]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 7\n\ts_nop 7");
    float s = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q) s += acc[r][q][0] + acc[r][q][1] + acc[r][q][2] + acc[r][q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

void run8(const f32x4 *src)
{
    float *out;
    unsigned long long *st;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&st, 512 * sizeof(unsigned long long));
    const int iters = 5000;
    (void)hipFuncSetAttribute((const void *)k8, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k8, dim3(256), dim3(256), 131072, 0, out, src, iters, st);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(512);
    (void)hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < 256; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz[128];
    const double mfmas = 256.0 * 4 * (double)iters * 32 * 8;
    const double tf = mfmas * 16384.0 / (ms * 1e-3) / 1e12;
    const double busy = mfmas * 16.0 / 1024.0 / (ms * 1e-3 * clk * 1e9);
    printf("%-34s %7.2f ms  %6.0f TFLOP/s (%4.1f %% of 2.5 PF)  clock %.2f GHz  MFMA pipe %.1f %% busy\n",
           "ds_read per 8 MFMAs, 1 wave/SIMD", ms, tf, tf / 2500 * 100, clk, busy * 100);
    (void)hipFree(out);
    (void)hipFree(st);
}

int main()
{
    std::vector<unsigned> hsrc(8192 * 4);
    srand(7);
    for (auto &w : hsrc) {   // two random bf16 values in [-1, 1) per word
        w = host_bf16() | (host_bf16() << 16);
    }
    f32x4 *src;
    (void)hipMalloc(&src, hsrc.size() * 4);
    (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    run<0, 4>("MFMA only, 1 wave/SIMD", src);
    run<0, 8>("MFMA only, 2 waves/SIMD", src);
    run<1, 4>("+ ds_read_b128/step, 1 wave/SIMD", src);
    run<1, 8>("+ ds_read_b128/step, 2 waves/SIMD", src);
    run8(src);
    return 0;
}
