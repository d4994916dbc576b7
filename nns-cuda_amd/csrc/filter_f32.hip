// filter_f32.hip — K3: the fp32 MFMA filter.  This is the dominant kernel.
//
// What the reference does here: V1/V2 write the full m x n distance matrix
// (get_dis_kernel, core.cu:58-78) and reduce its rows (get_min_kernel
// core.cu:87-122 / thrust::min_element core.cu:197-198); V3-V9 fuse the two but
// give one 1024-thread block to each query and re-stream every reference point
// per query (core.cu:589-633), so nothing is reused.  At 65536 x 1048576 x 128
// the matrix would be 256 GiB.
//
// What this kernel does instead: the ||q - r||^2 expansion.  With x' = q - c,
// y' = r - c (centred by K2),
//     s(i, j) = |y'_j|^2 - 2 x'_i . y'_j            ( = ||q_i - r_j||^2 - |x'_i|^2 )
// has the same argmin over j.  -2 * Y' * X'^T is a dense GEMM: it runs on the
// matrix cores as v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD)
// with |y'_j|^2 preloaded as the accumulator's initial value, so the finished
// accumulator IS s(i, j) and the matrix never leaves the registers.  A fused
// running top-2 (v_med3 + v_min per element) keeps, per query, the smallest and
// second-smallest score and the index of the smallest.  K5 then proves each
// winner from the gap between the two (error bound tau) or hands the query to
// the exact scan — indices end up bit-identical to V0.
//
// Geometry (KT = 128):
//   * operands: MFMA A = refs (rows i of the 32x32 tile), B = queries (columns).
//     The C layout puts a query on a lane (col = lane & 31) and 16 refs in the
//     lane's 16 accumulator registers, so the running top-2 state is THREE
//     registers per 32 queries and needs no cross-lane traffic until the end.
//   * a wave owns F_QB = 2 blocks of 32 queries; their B operands for all of K
//     (2 x 64 VGPRs) are loaded once and stay resident.  A workgroup is 8 waves
//     (2 per SIMD, <= 256 VGPRs each) = 512 queries; all 8 waves consume the same
//     stream of ref blocks from LDS, so a ref block fetched once feeds 512
//     queries (V7 re-reads it per query).
//   * refs stream through a ring of F_D LDS slots of 64 refs (2 image blocks,
//     32 KiB + 256 B of norms), filled by LDS-DMA (global_load_lds_dwordx4:
//     K2's image is stored in LDS order, so it is a linear 1 KiB-per-instruction
//     copy) with counted vmcnt waits and ONE raw s_barrier per slot; A operands
//     are lane-linear ds_read_b128 (4 k-steps each, bank-conflict free).
//   * grid = (m_pad / 512) x splits: each workgroup sweeps one contiguous range
//     of ref slots; splits are chosen so that the grid covers the 256 CUs.
//     Partials (top-2 per query per split) are merged by K5.
//
// Roofline: MFMA-bound.  Per 32x32 tile and K = 128: 64 MFMAs x 64 cycles per
// SIMD; the epilogue is 32 VALU + 4 ds_read_b128 per tile per wave (< 1 VALU
// per MFMA) and issues in the MFMA shadow.  HBM sees each image about once per
// XCD (the 8-wave workgroups of an XCD march through the same slots and hit
// L2); algorithmic HBM traffic is the image + queries, ~0.6 GB at C3.
#include <stdlib.h>
#include "nns_internal.h"

namespace nns {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int F_SB = 2;   // 32-ref image blocks per ring slot

// KT: tile depth; F_QB: 32-query blocks per wave (B operands resident in VGPRs);
// F_NW: waves per workgroup; F_D: ring depth in slots.
template <int KT, int F_QB, int F_NW, int F_D>
struct FCfg {
    static constexpr int QW = F_NW * F_QB * 32;         // queries per workgroup
    static constexpr int KB = KT / 8;                   // float4 per lane per image block
    static constexpr int BLK_BYTES = 32 * KT * 4;
    static constexpr int SLOT_COORD = F_SB * BLK_BYTES;
    static constexpr int SLOT_NORM = F_SB * 32 * 4;     // 256 B: one 4-B-per-lane DMA
    static constexpr int SLOT_BYTES = SLOT_COORD + SLOT_NORM;
    static constexpr int PIECES = SLOT_COORD / 1024;    // 1 KiB DMA pieces per slot
    static constexpr int PPW = PIECES / F_NW;           // pieces per wave
    static constexpr int OPS = PPW + 1;                 // DMA instructions per wave per slot
    static constexpr int LDS_BYTES = F_D * SLOT_BYTES;
    static_assert(PIECES % F_NW == 0, "slot must split evenly over the waves");
    static_assert(F_SB * 32 == 64, "norm piece is one dword per lane");
};


// LDS-DMA (global_load_lds_*): the wave copies 64 x {16, 4} bytes from per-lane
// global addresses to LDS at M0 + lane * size, with no VGPR destination.  Issued
// from inline asm on purpose: through the builtin, hipcc (ROCm 7.2) treats every
// later ds_read as possibly aliasing the in-flight DMA and drains it with
// s_waitcnt vmcnt(0) right after the issue, which serialises the ring.  The asm
// form is invisible to that pass; completion is tracked by OUR counted vmcnt
// waits + the slot barrier.  M0 is compiler-reserved: saved and restored inside
// the same statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma16(const void *g, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_byte)
                 : "memory");
}
__device__ __forceinline__ void dma4(const void *g, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_byte)
                 : "memory");
}

// min(a, b) as v_med3_f32(a, b, -INF): hipcc puts a canonicalising v_max in front of
// fminf() on MFMA results (3 VALU per score instead of 2); med3 needs none.  Scores
// are finite or +INF here (K2 routes NaN inputs to the exact path).
__device__ __forceinline__ float vmin(float a, float b)
{
    return __builtin_amdgcn_fmed3f(a, b, -__builtin_inff());
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KT, int F_QB, int F_NW, int F_D>
__global__ __launch_bounds__(F_NW * 64) void filter_f32_kernel(
    const float *__restrict__ qimg, const float *__restrict__ rimg,
    const float *__restrict__ rnorm, Partial *__restrict__ partials, int total_slots,
    int slots_per_split, int m_pad)
{
    using C = FCfg<KT, F_QB, F_NW, F_D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    // ---- resident B operands: this wave's 2 x 32 queries, all of K ---------------
    float bq[F_QB][KT / 2];
    {
        const int qblk0 = (blockIdx.x * F_NW + wave) * F_QB;
#pragma unroll
        for (int qb = 0; qb < F_QB; ++qb) {
            const float4 *src =
                reinterpret_cast<const float4 *>(qimg + (size_t)(qblk0 + qb) * 32 * KT) + lane;
#pragma unroll
            for (int b = 0; b < C::KB; ++b) {
                const float4 v = src[b * 64];
                bq[qb][4 * b + 0] = v.x;
                bq[qb][4 * b + 1] = v.y;
                bq[qb][4 * b + 2] = v.z;
                bq[qb][4 * b + 3] = v.w;
            }
        }
        // Pin the loads here: hipcc must wait for them BEFORE the ring starts, not
        // with a vmcnt(0) at their first use inside the loop (it cannot see the asm
        // DMAs, and such a wait would drain them every iteration).
#pragma unroll
        for (int qb = 0; qb < F_QB; ++qb)
#pragma unroll
            for (int i = 0; i < KT / 2; ++i) asm volatile("" : "+v"(bq[qb][i]));
    }

    const int slot0 = blockIdx.y * slots_per_split;
    int ns = total_slots - slot0;
    if (ns > slots_per_split) ns = slots_per_split;

    const char *rimg_b = reinterpret_cast<const char *>(rimg);
    // LDS byte address of the ring (low 32 bits of the generic pointer = LDS offset)
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);

    // DMA of ring slot s (relative to slot0) into ring position p
    auto issue = [&](int s, int p) {
        const size_t gslot = (size_t)(slot0 + s);
        const unsigned dst = lds_base + p * C::SLOT_BYTES + wave * (C::PPW * 1024);
        const char *src = rimg_b + gslot * C::SLOT_COORD + wave * (C::PPW * 1024) + lane * 16;
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) dma16(src + i * 1024, dst + i * 1024);
        // the slot's 64 norms: every wave copies the same 256 B (same bytes to the
        // same LDS words), which keeps each wave's DMA count per slot identical
        dma4(rnorm + gslot * 64 + lane, lds_base + p * C::SLOT_BYTES + C::SLOT_COORD);
    };

    float m1[F_QB], m2[F_QB];
    int code[F_QB];
#pragma unroll
    for (int qb = 0; qb < F_QB; ++qb) {
        m1[qb] = __builtin_inff();
        m2[qb] = __builtin_inff();
        code[qb] = 0;
    }

    // prologue: F_D - 1 slots in flight
#pragma unroll
    for (int s = 0; s < F_D - 1; ++s)
        if (s < ns) issue(s, s);

    for (int s = 0; s < ns; ++s) {
        // my share of slot s has landed once at most (F_D-2) younger slots' DMAs remain
        if (s + (F_D - 2) < ns)
            wait_vmcnt<(F_D - 2) * C::OPS>();
        else
            wait_vmcnt<0>();
        // everyone's share has landed, and everyone is done reading slot s-1
        __builtin_amdgcn_s_barrier();
        if (s + F_D - 1 < ns) issue(s + F_D - 1, (s + F_D - 1) % F_D);

        const char *slot = smem + (s % F_D) * C::SLOT_BYTES;
#pragma unroll
        for (int blk = 0; blk < F_SB; ++blk) {
            // accumulators start at |y'_j|^2 of their row: rows (r&3) + 8(r>>2) + 4h
            f32x16 acc[F_QB];
            const float *nrm = reinterpret_cast<const float *>(slot + C::SLOT_COORD) + blk * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 nv = *reinterpret_cast<const float4 *>(nrm + 8 * g);
#pragma unroll
                for (int qb = 0; qb < F_QB; ++qb) {
                    acc[qb][4 * g + 0] = nv.x;
                    acc[qb][4 * g + 1] = nv.y;
                    acc[qb][4 * g + 2] = nv.z;
                    acc[qb][4 * g + 3] = nv.w;
                }
            }
            const float4 *ap = reinterpret_cast<const float4 *>(slot + blk * C::BLK_BYTES) + lane;
#pragma unroll
            for (int b = 0; b < C::KB; ++b) {
                const float4 a = ap[b * 64];   // lane-linear ds_read_b128: 4 k-steps of A
#pragma unroll
                for (int qb = 0; qb < F_QB; ++qb)
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[qb][4 * b + 0], acc[qb], 0, 0, 0);
#pragma unroll
                for (int qb = 0; qb < F_QB; ++qb)
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[qb][4 * b + 1], acc[qb], 0, 0, 0);
#pragma unroll
                for (int qb = 0; qb < F_QB; ++qb)
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[qb][4 * b + 2], acc[qb], 0, 0, 0);
#pragma unroll
                for (int qb = 0; qb < F_QB; ++qb)
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[qb][4 * b + 3], acc[qb], 0, 0, 0);
            }
            // fused running top-2: 2 VALU per score
            const int blkcode = ((slot0 + s) * F_SB + blk) << 4;
#pragma unroll
            for (int qb = 0; qb < F_QB; ++qb) {
                const float m1_old = m1[qb];
                float a1 = m1_old, a2 = m2[qb];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = acc[qb][r];
                    a2 = __builtin_amdgcn_fmed3f(a1, a2, x);
                    a1 = vmin(a1, x);
                }
                m1[qb] = a1;
                m2[qb] = a2;
                const bool imp = a1 < m1_old;
                if (__builtin_amdgcn_ballot_w64(imp) != 0ull) {   // rare after warm-up
                    int cd = code[qb];
#pragma unroll
                    for (int r = 15; r >= 0; --r)
                        cd = (imp && acc[qb][r] == a1) ? (blkcode | r) : cd;
                    code[qb] = cd;
                }
            }
        }
    }

    // ---- lanes l and l+32 hold the same query over disjoint ref rows: merge --------
#pragma unroll
    for (int qb = 0; qb < F_QB; ++qb) {
        const int cd = code[qb];
        const int r = cd & 15;
        // shard-local ref index of the lane's best
        int idx = (cd >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float a1 = m1[qb], a2 = m2[qb];
        const float o1 = __shfl_xor(a1, 32, 64);
        const float o2 = __shfl_xor(a2, 32, 64);
        const int oi = __shfl_xor(idx, 32, 64);
        // top-2 of the union {a1 <= a2} U {o1 <= o2}
        const float lo = fminf(a1, o1), hi = fmaxf(a1, o1);
        const float second = fminf(hi, fminf(a2, o2));
        if (o1 < a1) idx = oi;
        if (lane < 32) {
            const int qi = ((blockIdx.x * F_NW + wave) * F_QB + qb) * 32 + lane;
            Partial p;
            p.m1 = lo;
            p.m2 = second;
            p.idx = idx;
            p.pad = 0;
            partials[(size_t)blockIdx.y * m_pad + qi] = p;
        }
    }
}

// ---- self-test: one 32x32 tile through the same MFMA k-order as the filter --------
// out[i][j] = chain over s = 0..KT/2-1 of the two-step FMA of v_mfma_f32_32x32x2_f32
// seeded with c0[i]; a is [32][KT] (rows = A operand), b is [32][KT] (rows = B operand
// columns).  Lets the tests check the hardware against a host fmaf() chain — the
// assumption behind tau (finalize.hip).
__global__ __launch_bounds__(64) void mfma_selftest_kernel(int kt, const float *__restrict__ a,
                                                           const float *__restrict__ b,
                                                           const float *__restrict__ c0,
                                                           float *__restrict__ out)
{
    const int lane = threadIdx.x, h = lane >> 5, i = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = c0[(r & 3) + 8 * (r >> 2) + 4 * h];
    for (int s = 0; s < kt / 2; ++s) {
        const int kk = 8 * (s >> 2) + 4 * h + (s & 3);   // same k permutation as the image
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * kt + kk], b[i * kt + kk], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int launch_mfma_selftest(int kt, const float *a, const float *b, const float *c0, float *out,
                         hipStream_t st)
{
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, st, kt, a, b, c0, out);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// ---- configurations ---------------------------------------------------------------
// (QB, NW, D): 32-query blocks per wave, waves per workgroup, ring depth; wgs_per_cu
// is what LDS + VGPRs admit (used only to size the ref-range splits).
struct FilterVariant {
    int qb, nw, d, wgs_per_cu;
};
static const FilterVariant kVariants[] = {
    {1, 8, 2, 2},    // 0: 8 waves x 32 queries, double-buffered ring (66 KiB): 2 WG / CU
    {1, 16, 2, 1},   // 1: 16 waves x 32 queries: 4 waves / SIMD in one workgroup
    {2, 8, 4, 1},    // 2: 8 waves x 64 queries, 4-slot ring
    {4, 4, 4, 1},    // 3: 4 waves x 128 queries, one wave per SIMD (512 VGPRs)
    {2, 4, 2, 2},    // 4: 4 waves x 64 queries, 2 WG / CU: SIMD partners in different WGs
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr int kDefaultVariant = 0;

static int pick_variant()
{
    const char *e = getenv("NNS_FILTER_VARIANT");   // tuning knob; default is the measured best
    if (e && *e) {
        const int v = atoi(e);
        if (v >= 0 && v < kNumVariants) return v;
    }
    return kDefaultVariant;
}

int filter_plan(int k, int m, int n, FilterGeom *g)
{
    int kt = 0;
    if (k <= 64) kt = 64;
    else if (k <= 128) kt = 128;
    else {
        set_error("MFMA filter: k = %d > 128 not tiled yet", k);
        return NNS_ERR_UNSUPPORTED;
    }
    g->variant = pick_variant();
    const FilterVariant &v = kVariants[g->variant];
    const int qw = v.nw * v.qb * 32;
    g->kt = kt;
    g->m_pad = divup(m, qw) * qw;
    const int slot_pts = 32 * F_SB;
    g->n_pad = divup(n, slot_pts) * slot_pts;
    g->total_slots = g->n_pad / slot_pts;
    g->qgroups = g->m_pad / qw;
    // cover the 256 CUs with whole rounds of co-resident workgroups
    const int resident = 256 * v.wgs_per_cu;
    int splits = 1;
    if (g->qgroups < resident) splits = divup(resident, g->qgroups);
    if (splits > g->total_slots) splits = g->total_slots;
    if (splits > 65535) splits = 65535;
    g->slots_per_split = divup(g->total_slots, splits);
    g->splits = divup(g->total_slots, g->slots_per_split);
    return NNS_OK;
}

template <int KT, int QB, int NW, int D>
static int launch_filter_t(const FilterGeom &g, const float *qimg, const float *rimg,
                           const float *rnorm, Partial *partials, hipStream_t st)
{
    using C = FCfg<KT, QB, NW, D>;
    auto kern = filter_f32_kernel<KT, QB, NW, D>;
    // > 64 KiB of dynamic LDS needs the opt-in, once per device
    static bool attr_set[64] = {};
    int dev = 0;
    NNS_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        NNS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.qgroups, g.splits), dim3(NW * 64), C::LDS_BYTES, st, qimg, rimg,
                       rnorm, partials, g.total_slots, g.slots_per_split, g.m_pad);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

template <int KT>
static int launch_filter_kt(const FilterGeom &g, const float *qimg, const float *rimg,
                            const float *rnorm, Partial *partials, hipStream_t st)
{
    switch (g.variant) {
    case 0: return launch_filter_t<KT, 1, 8, 2>(g, qimg, rimg, rnorm, partials, st);
    case 1: return launch_filter_t<KT, 1, 16, 2>(g, qimg, rimg, rnorm, partials, st);
    case 2: return launch_filter_t<KT, 2, 8, 4>(g, qimg, rimg, rnorm, partials, st);
    case 3: return launch_filter_t<KT, 4, 4, 4>(g, qimg, rimg, rnorm, partials, st);
    case 4: return launch_filter_t<KT, 2, 4, 2>(g, qimg, rimg, rnorm, partials, st);
    default: break;
    }
    set_error("filter: bad variant %d", g.variant);
    return NNS_ERR_INVALID;
}

int launch_filter_f32(const FilterGeom &g, const float *qimg, const float *rimg,
                      const float *rnorm, Partial *partials, hipStream_t st)
{
    if (g.kt == 64) return launch_filter_kt<64>(g, qimg, rimg, rnorm, partials, st);
    return launch_filter_kt<128>(g, qimg, rimg, rnorm, partials, st);
}

}  // namespace nns
