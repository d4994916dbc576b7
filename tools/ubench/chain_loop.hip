// Microbenchmark: the skeleton of a K-split "chained" deep tile before building it.  Eight waves (two per SIMD), each
// with the operands of 32 queries x 512 dims resident (two 16-query tiles x 16 k-steps = 128 registers); per step one
// ds_read_b128 (1 KiB fragment: 16 refs x 32 dims) + two in-place v_mfma_f32_16x16x32_bf16.  Waves 0-3 walk the even
// 16 KiB chunks of a linear stream (low K half of a ref tile), waves 4-7 the odd ones (high K half) two intervals
// behind; the stream comes from global memory by LDS-DMA into a ring of 8 x 16 KiB, STEPS / 4 pieces per wave and
// interval, counted vmcnt + one s_barrier per interval of STEPS steps.
//   what it answers: the cost of a barrier every 16 steps (512 MFMA cycles per wave) against every 32, and of the DMA
//   issue at this density (one 1 KiB piece per 4 steps and wave), on random operands.
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
__device__ __forceinline__ void dma16(const void *g, unsigned lds_byte)
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_byte)
                 : "memory");
}

// other forms of the same 1 KiB fill
__device__ __forceinline__ void dma16_saddr(const void *sbase, unsigned voff, unsigned lds_byte)
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte)
                 : "memory");
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf(u32x4 rsrc, unsigned voff, unsigned lds_byte)
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_byte)
                 : "memory");
}
__device__ __forceinline__ void dma4x4(const char *g, unsigned lds_byte)   // four dword pieces: 4 x 256 B
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\t"
                 "global_load_lds_dword %1, off offset:256\n\t"
                 "global_load_lds_dword %1, off offset:512\n\t"
                 "global_load_lds_dword %1, off offset:768\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_byte)
                 : "memory");
}

// a wave's PPW consecutive pieces in ONE statement: M0 set once, instruction offsets 0 / 1024 / ... (burst forms)
template <int N>
__device__ __forceinline__ void dma16_burst_one_m0(const void *g, unsigned lds_byte)
{
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    if constexpr (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
    else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"((const char *)g + 4096), "s"(lds_byte + 4096) : "memory");
    }
}

// STEPS: fragment steps per barrier interval (16: one chunk per wave and interval; 32: two); DMA / BAR: on / off
template <int STEPS, int DMA, int BAR, int COMPUTE = 1, int STAGE = 0>
__global__ __launch_bounds__(512) void k(float *out, const char *stream, size_t stream_bytes, const f32x4 *src, int intervals,
                                        unsigned long long *stamps, size_t phase)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = wave >> 2;
    constexpr int CPI = STEPS / 16;             // chunks per wave and interval
    constexpr int PPW = 2 * STEPS / 8;          // DMA pieces per wave and interval
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    {   // the ring starts with random operands (the clock the chip holds depends on the data)
        f32x4 *lds = reinterpret_cast<f32x4 *>(smem);
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(i + blockIdx.x * 37) & 8191];
    }
    f32x4 bq[2][16];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 16; ++s) bq[q][s] = src[(threadIdx.x * 8 + q * 2048 + s * 64 + blockIdx.x) & 8191];
    __syncthreads();
    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 rsrc;
    {
        const unsigned long long sb = (unsigned long long)(uintptr_t)stream;
        rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)sb);
        rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32) & 0xFFFFu);
        rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)stream_bytes);
        rsrc[3] = 0x00020000u;
    }
    f32x4 fr[4];
    f32x4 stg[PPW];   // STAGE 1: the wave's pieces of the previous interval on their way through registers
#pragma unroll
    for (int p = 0; p < PPW; ++p) stg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int j = 0; j < intervals; ++j) {
        // this interval's chunks: low waves 2 CPI j + 2 c, high waves two intervals behind, odd chunks
        const int c0 = 2 * CPI * j + (hh ? 1 - 4 * CPI : 0);
        const f32x4 *base = reinterpret_cast<const f32x4 *>(smem + ((c0 & 7) * 16384)) + lane;
        if (COMPUTE) {
            fr[0] = base[0];
            fr[1] = base[64];
        }
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            if (COMPUTE == 1 && t + 2 < STEPS) {
                const int tt = t + 2;
                const f32x4 *b2 = reinterpret_cast<const f32x4 *>(smem + (((c0 + 2 * (tt / 16)) & 7) * 16384)) + lane;
                fr[tt & 3] = b2[(tt & 15) * 64];
            }
            if (DMA && (STAGE == 5 || STAGE == 6)) {
                if (t == 1) {   // the wave's PPW consecutive KiB of the interval's chunks, all at once
                    const size_t cg = (size_t)(2 * CPI * (j + 2) - 1) * 16384 + (size_t)(wave * PPW) * 1024;
                    const char *g = stream + ((cg + (size_t)(blockIdx.x >> 3) * phase) % stream_bytes) + lane * 16;
                    const unsigned dst = lds_base + (unsigned)((cg >> 10) & 127) * 1024;
                    if (STAGE == 5) dma16_burst_one_m0<PPW>(g, dst);
                    else
#pragma unroll
                        for (int p = 0; p < PPW; ++p) dma16(g + p * 1024, dst + p * 1024);
                }
            } else if (DMA && (t & 3) == 1) {
                // piece p of the two chunks (per chunk of this interval's pair) that land two intervals ahead
                const int p = t >> 2;                         // 0 .. PPW - 1
                const int piece = wave * PPW + p;             // 0 .. 2 STEPS - 1 (KiB of the interval's 2 CPI chunks)
                const size_t cg = (size_t)(2 * CPI * (j + 2) - 1) * 16384 + (size_t)piece * 1024;   // chunks 2j+3, 2j+4 at CPI 1
                const char *g = stream + ((cg + (size_t)(blockIdx.x >> 3) * phase) % stream_bytes) + lane * 16;
                if (STAGE == 0) {
                    dma16(g, lds_base + (unsigned)((cg >> 10) & 127) * 1024);
                } else if (STAGE == 2) {   // scalar base + 32-bit lane offset
                    const size_t go = (cg + (size_t)(blockIdx.x >> 3) * phase) % stream_bytes;
                    dma16_saddr(stream, (unsigned)go + lane * 16, lds_base + (unsigned)((cg >> 10) & 127) * 1024);
                } else if (STAGE == 3) {   // buffer resource + 32-bit lane offset
                    const size_t go = (cg + (size_t)(blockIdx.x >> 3) * phase) % stream_bytes;
                    dma16_buf(rsrc, (unsigned)go + lane * 16, lds_base + (unsigned)((cg >> 10) & 127) * 1024);
                } else if (STAGE == 4) {   // four dword pieces (each lane 4 B, 256 B per instruction)
                    dma4x4(stream + ((cg + (size_t)(blockIdx.x >> 3) * phase) % stream_bytes) + lane * 4, lds_base + (unsigned)((cg >> 10) & 127) * 1024);
                } else {
                    // register staging: last interval's piece p goes to LDS (the compiler waits for its load here), this
                    // interval's is loaded into the same registers
                    const size_t cgp = cg - (size_t)(2 * CPI) * 16384;
                    *reinterpret_cast<f32x4 *>(smem + ((cgp >> 10) & 127) * 1024 + lane * 16) = stg[p];
                    stg[p] = *reinterpret_cast<const f32x4 *>(g);
                }
            }
            if (!COMPUTE) continue;
            const int ks = t & 15, rt = (t >> 4) & 1;
            mma(fr[t & 3], bq[0][ks], acc[rt][0]);
            mma(fr[t & 3], bq[1][ks], acc[rt][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) {
            if (DMA && STAGE != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGE == 4 ? 4 * PPW : PPW) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) s += acc[r][q][0] + acc[r][q][1] + acc[r][q][2] + acc[r][q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int STEPS, int DMA, int BAR, int COMPUTE = 1, int STAGE = 0>
void run(const char *name, const f32x4 *src, const char *stream, size_t stream_bytes, size_t phase = 0)
{
    float *out;
    unsigned long long *st;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&st, 512 * sizeof(unsigned long long));
    const int intervals = 40000 * 16 / STEPS;
    (void)hipFuncSetAttribute((const void *)k<STEPS, DMA, BAR, COMPUTE, STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<STEPS, DMA, BAR, COMPUTE, STAGE>), dim3(256), dim3(512), 131072, 0, out, stream, stream_bytes, src, intervals, st, phase);
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
    const double mfmas = 256.0 * 8 * (double)intervals * STEPS * 2;
    const double tf = mfmas * 16384.0 / (ms * 1e-3) / 1e12;
    const double busy = mfmas * 16.0 / 1024.0 / (ms * 1e-3 * clk * 1e9);
    printf("%-46s %7.2f ms  %6.0f TFLOP/s (%4.1f %% of 2.5 PF)  clock %.2f GHz  MFMA pipe %.1f %% busy\n", name, ms, tf,
           tf / 2500 * 100, clk, busy * 100);
    fflush(stdout);
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

int main()
{
    std::vector<unsigned> hsrc(8192 * 4);
    srand(7);
    for (auto &w : hsrc) w = host_bf16() | (host_bf16() << 16);
    f32x4 *src;
    (void)hipMalloc(&src, hsrc.size() * 4);
    (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    // the streamed image: 1 GiB of random bf16 pairs (every workgroup streams the same bytes, as the filter's do)
    const size_t stream_bytes = (size_t)1 << 30;
    char *stream;
    (void)hipMalloc(&stream, stream_bytes + (1 << 20));   // (+ slack: the burst forms read a wave's 8 KiB linearly past a wrapped start)
    for (size_t off = 0; off < stream_bytes + (1 << 20); off += hsrc.size() * 4)
        (void)hipMemcpy(stream + off, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    run<16, 0, 0>("bare: 2 MFMAs per ds_read, 8 waves", src, stream, stream_bytes);
    run<16, 0, 1>("+ barrier per 16 steps", src, stream, stream_bytes);
    run<32, 0, 1>("+ barrier per 32 steps", src, stream, stream_bytes);
    run<16, 1, 1>("+ barrier per 16 steps + DMA (4 pieces / wave)", src, stream, stream_bytes);
    run<32, 1, 1>("+ barrier per 32 steps + DMA (8 pieces / wave)", src, stream, stream_bytes);
    run<16, 1, 0>("DMA, no barrier (results racy: timing only)", src, stream, stream_bytes);
    // the workgroups of an XCD (blockIdx.x >> 3 = 0 .. 31) walk the stream PHASE bytes apart instead of in lock-step on
    // the same lines: do the 32 CUs of an XCD queue on the same L2 channels?
    run<32, 1, 1>("32 steps + DMA, phase 4 KiB", src, stream, stream_bytes, 4096);
    run<32, 1, 1>("32 steps + DMA, phase 36 KiB", src, stream, stream_bytes, 36864);
    run<32, 1, 1>("32 steps + DMA, phase 68 KiB", src, stream, stream_bytes, 69632);
    run<32, 1, 1, 2, 5>("MFMAs w/o ds_reads + DMA burst, ONE m0 setup", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 2, 6>("MFMAs w/o ds_reads + DMA burst, m0 per piece", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 1, 5>("32 steps + DMA burst, ONE m0 setup", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 2, 2>("MFMAs w/o ds_reads + DMA saddr + voffset", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 2, 3>("MFMAs w/o ds_reads + DMA buffer_load offen lds", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 2, 4>("MFMAs w/o ds_reads + DMA 4 dword pieces", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 2, 1>("MFMAs w/o ds_reads + register-staged", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 2, 0>("32 steps + DMA, MFMAs WITHOUT their ds_reads", src, stream, stream_bytes, 4096);
    run<32, 0, 1, 2, 0>("32 steps, MFMAs without ds_reads, no DMA", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 0, 0>("LDS-DMA alone (no MFMA, no ds_read), 32 steps", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 0, 1>("register-staged alone, 32 steps", src, stream, stream_bytes, 4096);
    run<32, 1, 1, 1, 1>("32 steps + register-staged fill", src, stream, stream_bytes, 4096);
    run<16, 1, 1, 1, 1>("16 steps + register-staged fill", src, stream, stream_bytes, 4096);
    run<32, 1, 1>("32 steps + DMA, phase 1 MiB + 4 KiB", src, stream, stream_bytes, (1 << 20) + 4096);
    run<32, 1, 1>("32 steps + DMA, phase 32 MiB + 4 KiB", src, stream, stream_bytes, (32 << 20) + 4096);
    return 0;
}
