// Microbenchmark behind DESIGN.md's "KT = 4 MFMA filter for low-k problems" note (round 3): what would a
// v_mfma_f32_16x16x4_f32 filter — ONE MFMA per 16 x 16 (ref, query) tile for k <= 4 — retire per second on the
// reference's low-dimensional shapes, epilogue included?  Not product code: scores only, no candidate lists, no K5.
//   wave: NQ query tiles of 16 resident (B operand: 1 VGPR each); per ref tile of 16: A operand 1 dword per lane
//   (lane l: ref l & 15, dim l >> 4) + the tile's norms (float4 per lane: rows 4 (l >> 4) ..) from global memory
//   (L2-resident image), NQ MFMAs seeded with the norms, then per (ref tile, query tile):
//     MODE 0  nothing (bare MFMA + operand feed)
//     MODE 1  threshold test: min of the 4 scores, compare, wave-uniform rare branch that tightens a running minimum
//             and stores a record (the filter's form with tau = 0: the FEWEST slow paths any margin can give)
//     MODE 2  branch-free best / second-best / third-best tracking (record form 2)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_k4 mfma_k4.hip ; run: ./mfma_k4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NQ>
__global__ __launch_bounds__(512) void k4(const float *__restrict__ qimg, const float *__restrict__ rimg,
                                          const float *__restrict__ rnorm, int tiles_per_split, float *__restrict__ out,
                                          float2 *__restrict__ recs)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qt0 = (blockIdx.x * 8 + wave) * NQ;
    float b[NQ], thr[NQ], m2[NQ], m3[NQ];
    int t1[NQ], t2[NQ], cnt = 0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        b[i] = qimg[(size_t)(qt0 + i) * 64 + lane];
        thr[i] = m2[i] = m3[i] = __builtin_inff();
        t1[i] = t2[i] = 0;
    }
    const int tile0 = blockIdx.y * tiles_per_split;
    const float *ap = rimg + (size_t)tile0 * 64 + lane;
    const float4 *np = reinterpret_cast<const float4 *>(rnorm + (size_t)tile0 * 16 + 4 * (lane >> 4));
    float a_next = ap[0];
    float4 n_next = np[0];
    for (int t = 0; t < tiles_per_split; ++t) {
        const float a = a_next;
        const float4 nv = n_next;
        if (t + 1 < tiles_per_split) {
            a_next = ap[(size_t)(t + 1) * 64];
            n_next = np[(size_t)(t + 1) * 4];
        }
        const f32x4 seed = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[i], seed, 0, 0, 0);
            if (MODE == 0) {
                asm volatile("" ::"v"(s));
            } else {
                const float tm = fminf(fminf(fminf(s[0], s[1]), s[2]), s[3]);
                if (MODE == 1) {
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(tm <= thr[i]) != 0ull, 0)) {
                        if (tm <= thr[i]) {
                            recs[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x) * 16 + (cnt & 15)] = make_float2(tm, (float)(tile0 + t));
                            ++cnt;
                            thr[i] = tm;
                        }
                    }
                } else {
                    float o1 = thr[i], o2 = m2[i];
                    int p1 = t1[i], p2 = t2[i];
                    asm volatile("" : "+v"(o1), "+v"(o2), "+v"(p1), "+v"(p2));
                    const bool lt1 = tm < o1, lt2 = tm < o2;
                    m3[i] = fminf(m3[i], fmaxf(o2, tm));
                    m2[i] = fminf(o2, fmaxf(o1, tm));
                    thr[i] = fminf(o1, tm);
                    t2[i] = lt1 ? p1 : (lt2 ? tile0 + t : p2);
                    t1[i] = lt1 ? tile0 + t : p1;
                }
            }
        }
    }
    float acc = (float)cnt;
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc += thr[i] + m2[i] + m3[i] + (float)(t1[i] + t2[i]);
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x] = acc;
}

template <int MODE, int NQ>
static void run(const char *name, int m, int n)
{
    const int qtiles = m / 16, rtiles = n / 16;
    const int qgroups = qtiles / (8 * NQ);
    int splits = 256 / (qgroups > 0 ? qgroups : 1);
    if (splits < 1) splits = 1;
    while (rtiles % splits) --splits;
    const int tps = rtiles / splits;
    float *q, *r, *nrm, *out;
    float2 *recs;
    hipMalloc(&q, (size_t)qtiles * 64 * 4);
    hipMalloc(&r, (size_t)rtiles * 64 * 4);
    hipMalloc(&nrm, (size_t)n * 4 + 64);
    hipMalloc(&out, (size_t)qgroups * splits * 512 * 4);
    hipMalloc(&recs, (size_t)qgroups * splits * 512 * 16 * 8);
    float *h = (float *)malloc((size_t)rtiles * 64 * 4);
    srand(7);
    for (size_t i = 0; i < (size_t)rtiles * 64; ++i) h[i] = -2.0f * (rand() / (float)RAND_MAX - 0.5f);
    hipMemcpy(r, h, (size_t)rtiles * 64 * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)qtiles * 64; ++i) h[i] = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(q, h, (size_t)qtiles * 64 * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)n; ++i) h[i] = 0.25f * (rand() / (float)RAND_MAX);
    hipMemcpy(nrm, h, (size_t)n * 4, hipMemcpyHostToDevice);
    free(h);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k4<MODE, NQ>), dim3(qgroups, splits), dim3(512), 0, 0, q, r, nrm, tps, out, recs);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    const double pairs = (double)m * n;
    printf("%-34s m=%6d n=%8d NQ=%2d grid %3d x %3d: %8.1f us  %.3e pairs/s  %.2f pairs/clk/SIMD @2.4GHz  (MFMA bound 8.0)\n", name, m, n, NQ,
           qgroups, splits, best * 1e3, pairs / (best * 1e-3), pairs / (best * 1e-3) / (1024 * 2.4e9));
    hipFree(q); hipFree(r); hipFree(nrm); hipFree(out); hipFree(recs);
}

int main()
{
    const int shapes[][2] = {{1024, 1048576}, {4096, 65536}, {65536, 1048576}};
    for (auto &s : shapes) {
        run<0, 8>("bare MFMA + operand feed", s[0], s[1]);
        run<1, 8>("threshold test, tau = 0", s[0], s[1]);
        run<2, 8>("best / 2nd / 3rd, branch-free", s[0], s[1]);
        if (s[0] >= 4096) {
            run<0, 16>("bare MFMA + operand feed", s[0], s[1]);
            run<1, 16>("threshold test, tau = 0", s[0], s[1]);
            run<2, 16>("best / 2nd / 3rd, branch-free", s[0], s[1]);
        }
    }
    return 0;
}
