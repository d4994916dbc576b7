// filter_mfma.hip — K3 (fp32) / K4 (bf16): the MFMA filter.  The dominant kernel.
//
// What the reference does here: V1/V2 write the full m x n distance matrix
// (get_dis_kernel, core.cu:58-78) and reduce its rows (get_min_kernel core.cu:87-122 /
// thrust::min_element core.cu:197-198); V3-V9 fuse the two but give one 1024-thread
// block to each query and re-stream every reference point per query
// (core.cu:589-633).  At 65536 x 1048576 x 128 the matrix would be 256 GiB.
//
// What this kernel does instead: the ||q - r||^2 expansion.  With x' = q - c, y' = r - c
// (centred by K2; c = 0 on the bf16 path),
//     s(i, j) = |y'_j|^2 - 2 x'_i . y'_j            ( = ||q_i - r_j||^2 - |x'_i|^2 )
// has the same argmin over j.  -2 * Y' * X'^T is a dense GEMM on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain; v_mfma_f32_16x16x32_bf16: bf16 inputs,
// fp32 accumulate) with |y'_j|^2 preloaded as the accumulator's initial value, so the
// finished accumulator IS s(i, j) and the matrix never leaves the registers.
//
// Fused epilogue = RECORD COLLECTION, not a plain argmin: a lane keeps the running
// minimum m1 of its query over the refs it has seen and a threshold
// thr = m1 + tau(m1), where tau bounds |s + |x'|^2 - V0's fp32 distance| from both sides
// (derivation in finalize.hip).  Per 32x32 tile it takes the minimum of its 16 scores
// (8 v_min3 + 1 v_cmp); only when that beats thr (a new record or a near-record: about
// ln(n) times per lane in total) does a wave-uniform slow path append (score, index)
// entries to the lane's private candidate list in HBM.  Every ref whose score is within
// tau of the query's final minimum is therefore in some list, and K5 re-ranks exactly
// those few candidates with V0's own arithmetic: indices come out bit-identical to V0
// although the filter itself is approximate.
//
// Geometry (fp32; what the bf16 instantiation does differently is described at OpBF16):
//   * MFMA A = refs (rows of the 32x32 tile), B = queries (columns): the C layout puts
//     a query on a lane (col = lane & 31) and 16 refs in the lane's 16 accumulator
//     registers, so the running state is per lane and needs no cross-lane traffic.
//   * a wave owns 64 queries (two blocks of 32; one block at the 256-deep tile); their B
//     operands for all of K (128 VGPRs) are loaded once and stay resident.  A workgroup is
//     8 waves = 512 queries; all 8 waves consume the same stream of ref blocks from LDS (a
//     block fetched once feeds 512 queries; the reference's V7 re-reads it per query).
//   * refs stream through a 4-slot LDS ring (slot = 64 refs = 32 KiB of K2's tile image
//     + 256 B of norms) filled by LDS-DMA (global_load_lds_dwordx4: the image is stored in
//     LDS order, so the copy is linear, 1 KiB per wave-instruction), waited for with
//     vmcnt and ONE raw s_barrier per slot; A fragments are lane-linear ds_read_b128
//     (bank-conflict free by construction of the image).
//   * the two waves that share a SIMD (w and w + 4) run HALF A BLOCK out of phase: waves
//     4..7 lag by half a 32-ref block (they finish the previous slot's last half block
//     after the barrier), so one partner's epilogue / accumulator re-seed / LDS latency
//     falls in the middle of the other's MFMA chain instead of both stalling together
//     at every block boundary (MI355X guide, "Two waves per SIMD", item 9).
//   * grid = (m_pad / 512) x splits; the split count minimises rounds x work per
//     workgroup (one resident workgroup per CU), see filter_plan.
//
// Roofline: MFMA-bound.  fp32: 64 MFMAs x 64 cycles per 32x32x128 tile per SIMD; bf16:
// 32 MFMAs x 16 cycles per 32x32x256 tile (four 16x16 tiles x 8 k-steps).  Algorithmic HBM
// traffic = the images once.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <vector>
#include "nns_internal.h"

namespace nns {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time unrolled loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// accumulators of a wave's query blocks as NAMED members (an array of ext-vectors passed by
// reference is not scalarised by hipcc and lands in scratch)
struct AccSet {
    f32x16 v0, v1;
    template <int I>
    __device__ __forceinline__ f32x16 &at()
    {
        if constexpr (I == 0) return v0;
        else return v1;
    }
    template <int I>
    __device__ __forceinline__ const f32x16 &at() const
    {
        if constexpr (I == 0) return v0;
        else return v1;
    }
};

// The same 32 registers as eight 16x16 tiles [ref tile rt][query tile qt] (OpBF16)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct AccSet16 {
    f32x4 t00, t01, t02, t03, t10, t11, t12, t13;
    template <int RT, int QT>
    __device__ __forceinline__ f32x4 &at()
    {
        static_assert(RT >= 0 && RT < 2 && QT >= 0 && QT < 4, "2 ref tiles x 4 query tiles");
        if constexpr (RT == 0) {
            if constexpr (QT == 0) return t00;
            else if constexpr (QT == 1) return t01;
            else if constexpr (QT == 2) return t02;
            else return t03;
        } else {
            if constexpr (QT == 0) return t10;
            else if constexpr (QT == 1) return t11;
            else if constexpr (QT == 2) return t12;
            else return t13;
        }
    }
    template <int RT, int QT>
    __device__ __forceinline__ const f32x4 &at() const
    {
        return const_cast<AccSet16 *>(this)->template at<RT, QT>();
    }
};

// sixteen 16x16 tiles [ref tile rt][query tile qt]: 128 queries per wave (the one-wave-per-SIMD form of OpBF16)
struct AccSet16W {
    f32x4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
    template <int RT, int QT>
    __device__ __forceinline__ f32x4 &at()
    {
        static_assert(RT >= 0 && RT < 2 && QT >= 0 && QT < 8, "2 ref tiles x 8 query tiles");
        if constexpr (RT == 0) {
            if constexpr (QT == 0) return a0;
            else if constexpr (QT == 1) return a1;
            else if constexpr (QT == 2) return a2;
            else if constexpr (QT == 3) return a3;
            else if constexpr (QT == 4) return a4;
            else if constexpr (QT == 5) return a5;
            else if constexpr (QT == 6) return a6;
            else return a7;
        } else {
            if constexpr (QT == 0) return b0;
            else if constexpr (QT == 1) return b1;
            else if constexpr (QT == 2) return b2;
            else if constexpr (QT == 3) return b3;
            else if constexpr (QT == 4) return b4;
            else if constexpr (QT == 5) return b5;
            else if constexpr (QT == 6) return b6;
            else return b7;
        }
    }
    template <int RT, int QT>
    __device__ __forceinline__ const f32x4 &at() const
    {
        return const_cast<AccSet16W *>(this)->template at<RT, QT>();
    }
};

// Timing diagnostics only (results are wrong): build with -DNNS_DIAG -DNNS_FILTER_ABLATE=<bits>
//   1 no ring sync (wait + barrier), 2 no epilogue, 16 no DMA issue.  The product build (no NNS_DIAG)
// has none of the diagnostic switches: no ablation, no NNS_FILTER_CLOCK / NNS_DIAG_FILTER_ONLY
// environment variables (tests/test_abi_cpu.py greps the shipped library for them).
#if !defined(NNS_DIAG)
#undef NNS_FILTER_ABLATE
#undef NNS_F_NOLATCHFENCE
#undef NNS_F_NOLAG
#endif
#ifndef NNS_FILTER_ABLATE
#define NNS_FILTER_ABLATE 0
#endif
constexpr int kAblate = NNS_FILTER_ABLATE;

constexpr int F_D = 4;                   // ring depth
#ifndef NNS_F_DMA_AHEAD
#define NNS_F_DMA_AHEAD 3
#endif
constexpr int F_DMA_AHEAD_MAX = NNS_F_DMA_AHEAD;   // (A/B builds: 2 = round 2's schedule for every operator)
constexpr int F_SLOT_COORD = 32768;      // image bytes of one ring slot (32 fragment steps x 1 KiB)
constexpr int F_SLOT_NORM = 2048;        // room for the slot's norms (up to 512 floats: the 16-deep tile)
constexpr int F_SLOT_BYTES = F_SLOT_COORD + F_SLOT_NORM;
constexpr int F_LDS_BYTES = F_D * F_SLOT_BYTES;

// ---- operand traits ---------------------------------------------------------------
// query blocks (of 32) per wave: their B operands stay resident in VGPRs (64 each).  More
// blocks = more MFMAs per LDS byte and per barrier interval, fewer waves' worth of registers.
#ifndef NNS_F_QB_F32
#define NNS_F_QB_F32 2
#endif
#ifndef NNS_F_QB_BF16
#define NNS_F_QB_BF16 2
#endif
// waves per workgroup: 8 = two per SIMD (<= 256 VGPRs each), 4 = one per SIMD (<= 512)
#ifndef NNS_F_NW_F32
#define NNS_F_NW_F32 8
#endif
#ifndef NNS_F_NW_BF16
#define NNS_F_NW_BF16 8
#endif

template <int SPB, int QB_>
struct OpF32T {  // fp32 operands: float4 #b = operands of MFMA k-steps 4b .. 4b+3 (8 dims per fragment)
    static constexpr int kSPB = SPB;          // fragment steps per 32-point image block: KT = 8 * SPB
    static constexpr bool kTile16 = false;    // 32x32 MFMA tiles: a lane owns one query per query block
    // a slot's ring DMA pieces back to back (see the interval) — where it measured faster: KT = 128 (+0.65 % on long streams:
    // C3) and KT = 32 (+0.8 %); KT = 16 / 64 / 256 within -0.4 .. +0.1 %: one piece per step as before
    static constexpr bool kDmaBurst = SPB == 16 || SPB == 4;
    // SIMD partners half a block out of phase (+1.3 % on C3); a 2-step block (KT = 16) has no half to lag by
    static constexpr bool kLag = SPB >= 4;
    // the lanes' tau constants (c0, x2 per state) stay in registers: an LDS round trip on the slow path
    // queues behind the whole workgroup's fragment reads (measured: ~900 cycles each under this load)
    static constexpr bool kTauInRegs = true;
    using Acc = AccSet;
    static constexpr int kQB = QB_;           // 4 * SPB resident operand registers per query block
    static constexpr int kNW = NNS_F_NW_F32;
#ifndef NNS_F_PF
#define NNS_F_PF 2
#endif
    static constexpr int kPrefetch = NNS_F_PF;   // fragments in flight ahead of the MFMAs (4 x 64 cycles each)
    __device__ static __forceinline__ f32x16 mma(const float4 &a, const float4 &b, f32x16 acc)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        return acc;
    }
};
using OpF32 = OpF32T<16, NNS_F_QB_F32>;      // KT = 128
using OpF32K32 = OpF32T<4, NNS_F_QB_F32>;    // KT = 32: the mid-range dimensionalities (k = 17 .. 32)
// KT = 16: the reference driver's own 16-D samples (main.cu:38-51) without zero padding to 32: 2 fragment
// steps per 32-ref block, 16 blocks = 512 refs per ring slot (norms: two dwordx4 DMA pieces)
using OpF32K16 = OpF32T<2, NNS_F_QB_F32>;
using OpF32K64 = OpF32T<8, NNS_F_QB_F32>;    // KT = 64: 32 < k <= 64 without padding to 128
// KT = 256 (128 < k <= 256): the resident operands of ONE query block already take 128 registers,
// a ring slot holds one 32-ref block (32 KiB), and the ring turns twice as often per MFMA
using OpF32K256 = OpF32T<32, 1>;

// bf16, KT = 256, 16 fragment steps per 32-ref block either way; two MFMA shapes:
//
// OpBF16 (the product): v_mfma_f32_16x16x32_bf16.  Same flops per cycle as the 32x32x16 form,
// but the chip holds a ~14 % higher clock on it under this loop's load (measured here on C5:
// in-kernel clock 2.05 vs 1.80 GHz, filter 70 vs 80 ms in the bare loop; MI355X guide, 'DVFS
// give-back' item 7).  A step = one 1 KiB fragment (16 refs x 32 dims) x the wave's FOUR
// 16-query tiles; step b = 8 * rt + ks: the 8 k-steps of ref tile 0, then those of ref tile 1.
// C layout: lane l holds query column l & 15 and ref rows 4 * (l >> 4) .. + 3 of each tile, so a
// lane carries four queries (one per query tile) x 8 refs per block, and a query's refs are
// spread over the four lanes l & 15 + 16 g — four lane-private candidate lists per query.
//
// The record collection of one ref tile runs INSIDE the MFMA stream of the other: a burst of
// ~20 VALU instructions at a block boundary keeps the SIMD's vector issue port for ~80 cycles
// with no MFMA issued (neither by this wave nor by its SIMD partner: lagging the partners did
// not hide it, measured), which cost 7-9 % of this kernel.  Two v_min per MFMA gap fit the 8
// issue cycles a 16x16x32 leaves free, so the reduction of tile rt's finished scores is spread
// over k-step 1 of tile 1 - rt, and the threshold test + (rare) slow path follow at k-step 2.
template <int SPB_, int NW_ = NNS_F_NW_BF16, bool ASM_ = true, int QB_ = 2>
struct OpBF16T {
    static constexpr int kSPB = SPB_;         // 16: KT = 256 (8 k-steps per 16-ref tile); 8: KT = 128 (4 k-steps); 32: KT = 512
    static constexpr bool kAsmMfma = ASM_;    // false: compiler builtins (the 512-deep form: operands beyond the 256 ArchVGPRs an asm "v" can name)
    static constexpr bool kTile16 = true;
    static constexpr bool kDmaBurst = false;
    static constexpr bool kLag = false;       // lock-step SIMD partners (lagging them: +1..4 % time on C5)
    static constexpr bool kTauInRegs = false; // 222-234 VGPRs: the four states' constants live in LDS (read on the slow path)
    using Acc = std::conditional_t<QB_ == 4, AccSet16W, AccSet16>;
    static constexpr int kQB = QB_;           // 2: 64 queries per wave = 4 query tiles; 4: 128 queries = 8 tiles (one wave per SIMD)
    static constexpr int kNW = NW_;
#ifndef NNS_F_PF_BF16
#define NNS_F_PF_BF16 2
#endif
    static constexpr int kPrefetch = NNS_F_PF_BF16;
    // Inline asm, accumulating IN PLACE: through the builtin hipcc picks the three-address form
    // and rotates the eight 4-register accumulators through extra tuples (52-72 registers live
    // instead of 32), which spills the resident query operands.  The compiler does not see MFMA
    // hazards of an asm statement; the kernel keeps them by construction: an accumulator is
    // re-used as srcC only 8 MFMAs later, and VALU reads of it (the epilogue) wait behind an
    // the epilogue's fences (mma16_fence_lo / _hi).
    __device__ static __forceinline__ void mma16(const float4 &a, const float4 &b, f32x4 &acc)
    {
        if constexpr (ASM_ && QB_ == 4)   // the wave's 256 query-operand registers are the AGPR half of its file
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                         : "+v"(acc)
                         : "v"(__builtin_bit_cast(f32x4, a)), "a"(__builtin_bit_cast(f32x4, b)));
        else if constexpr (ASM_)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                         : "+v"(acc)
                         : "v"(__builtin_bit_cast(f32x4, a)), "v"(__builtin_bit_cast(f32x4, b)));
        else
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    // First MFMA of a tile: srcC = the refs' norms (three-address form).  The accumulator is an
    // in/out operand although its old value is not read: that pins every tile to ONE register
    // tuple for the whole kernel.  As a fresh output the allocator may give it another tuple and
    // reconcile at the loop back-edge with v_mov copies of the accumulators right behind the
    // interval's last MFMAs — a read hazard (caught by tools/check_mfma_hazards.py).
    __device__ static __forceinline__ void mma16_seed(const float4 &a, const float4 &b, f32x4 &acc, const f32x4 &c)
    {
        if constexpr (ASM_ && QB_ == 4)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3"
                         : "+v"(acc)
                         : "v"(__builtin_bit_cast(f32x4, a)), "a"(__builtin_bit_cast(f32x4, b)), "v"(c));
        else if constexpr (ASM_)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3"
                         : "+v"(acc)
                         : "v"(__builtin_bit_cast(f32x4, a)), "v"(__builtin_bit_cast(f32x4, b)), "v"(c));
        else
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // VALU may read an accumulator 8 wait states after the MFMA that wrote it issued (what hipcc
    // inserts behind the builtin: s_nop 7).  To the compiler an asm MFMA's result is ready at
    // once.  In the loop the readers sit a whole step (>= 64 cycles) behind the writers and
    // __builtin_amdgcn_sched_barrier keeps them there; the kernel's tail reads right behind the
    // last MFMAs and needs the explicit wait, with the accumulators as in/out operands so that
    // the reads are ordered behind it.
    template <class A>
    __device__ static __forceinline__ void mma16_tail_fence(A &c)
    {
        if constexpr (ASM_ && QB_ == 4) {
            // (volatile statements keep their order: the pins, and every read of the tiles behind them, follow the wait)
            asm volatile("s_nop 7");
            asm volatile("" : "+v"(c.template at<1, 0>()), "+v"(c.template at<1, 1>()), "+v"(c.template at<1, 2>()), "+v"(c.template at<1, 3>()));
            asm volatile("" : "+v"(c.template at<1, 4>()), "+v"(c.template at<1, 5>()), "+v"(c.template at<1, 6>()), "+v"(c.template at<1, 7>()));
        } else if constexpr (ASM_ && QB_ == 1) asm volatile("s_nop 7" : "+v"(c.t10), "+v"(c.t11));   // (two query tiles per wave)
        else if constexpr (ASM_) asm volatile("s_nop 7" : "+v"(c.t10), "+v"(c.t11), "+v"(c.t12), "+v"(c.t13));
        // (builtin MFMAs: hipcc's hazard recognizer places the wait states)
    }
};
using OpBF16 = OpBF16T<16>;       // KT = 256
using OpBF16K128 = OpBF16T<8>;    // KT = 128: k <= 128 without padding to 256 (half the MFMAs)
// KT = 512 in the 16x16x32 form (round 2): the resident operands of the wave's four query tiles are 256
// registers, so four waves per workgroup (one per SIMD) = 256 queries; a 1 KiB LDS fragment still feeds FOUR
// MFMAs (the 32x32x16 form of OpBF16K512: one — LDS-bandwidth bound at 55 % of peak).  Compiler builtins
// instead of inline asm: an asm "v" operand must sit in the 256 architectural VGPRs.
#ifndef NNS_K512_NW4
// Round 3: EIGHT waves x 32 queries (two query tiles per wave, 128 operand registers): the same 256 queries per
// workgroup, but two waves per SIMD — a partner issues MFMAs while a wave sits in an LDS-DMA issue — and in-place asm
// MFMAs (the four-wave form's operands spill into AGPRs and hipcc copies one back per MFMA: 70 v_accvgpr_read per 64
// MFMAs).  A 1 KiB fragment feeds two MFMAs per wave: LDS fragment reads 32 of every 64 cycles.
using OpBF16K512T = OpBF16T<32, 8, true, 1>;
#else
using OpBF16K512T = OpBF16T<32, 4, false>;
#endif
// KT = 256 with 128 queries per wave on four waves (experiment, -DNNS_BF16_WIDE): half the LDS fragment reads per MFMA
using OpBF16Wide = OpBF16T<16, 4, true, 4>;


// OpBF16T32: v_mfma_f32_32x32x16_bf16 (the first version; kept for A/B builds with
// -DNNS_BF16_TILE32): 16 bytes = 8 bf16 = one operand, lane = query, 16 refs per lane.
template <int SPB_, int QB_, int NW_ = NNS_F_NW_BF16, bool LAG_ = (NW_ == 8)>
struct OpBF16T32T {
    static constexpr int kSPB = SPB_;         // fragment steps (16 dims each) per 32-ref block
    static constexpr bool kTile16 = false;
    static constexpr bool kDmaBurst = false;
    static constexpr bool kLag = LAG_;        // (one wave per SIMD has no partner to stagger against)
    static constexpr bool kTauInRegs = true;
    using Acc = AccSet;
    static constexpr int kQB = QB_;
    static constexpr int kNW = NW_;
    static constexpr int kPrefetch = NNS_F_PF_BF16;   // one MFMA (32 cycles) per fragment and query block
    __device__ static __forceinline__ f32x16 mma(const float4 &a, const float4 &b, f32x16 acc)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
using OpBF16T32 = OpBF16T32T<16, NNS_F_QB_BF16>;   // KT = 256
// KT = 512 (256 < k <= 512): the 32x32x16 form with ONE query block per wave (its resident operands are
// 128 registers) and one 32-ref block per ring slot — the bf16 twin of OpF32K256
using OpBF16K512 = OpBF16T32T<32, 1>;
// KT = 1024 (512 < k <= 1024): K-split accumulation.  The resident operands of one query block are 256
// registers, so a workgroup is FOUR waves (one per SIMD, up to 512 registers each) = 128 queries; a 32-ref
// block is 64 fragment steps = TWO ring slots, and its accumulators carry across the slot barrier: seeded at
// step 0 of the even slot, retired at step 31 of the odd one.  One MFMA per 1 KiB LDS fragment, like the
// 512-deep tile: LDS-bandwidth bound (~50 % of the bf16 MFMA peak), still ~50x the exact VALU scan.
using OpBF16K1024 = OpBF16T32T<64, 1, 4>;
// KT = 768 (512 < k <= 768; round 3): 48 fragment steps per block — two blocks over three ring slots.  The resident
// operands of one query block are 192 registers, so — unlike at 1024 — TWO waves per SIMD fit (<= 256 registers
// each): eight waves = 256 queries per workgroup, and a SIMD partner covers a wave's LDS-DMA issue stalls (the
// one-wave-per-SIMD tiles lose about half their MFMA slots to them: profiles/r03_deep_ablate.txt).  Lock-step
// partners (a half-block lag would straddle slots).
#ifndef NNS_F_NW_K768
#define NNS_F_NW_K768 8
#endif
using OpBF16K768 = OpBF16T32T<48, 1, NNS_F_NW_K768, false>;
// KT = 640 (512 < k <= 640): 40 fragment steps per block — FOUR blocks over FIVE ring slots; 160 operand registers, eight waves
using OpBF16K640 = OpBF16T32T<40, 1, 8, false>;
// KT = 384 (256 < k <= 384, round 3): the same form below 512 — 24 steps per block, four blocks over three slots, 96 operand
// registers; k = 300 ran on the 512-deep tile at 36 % of peak
using OpBF16K384 = OpBF16T32T<24, 1, 8, false>;
#if defined(NNS_BF16_WIDE)
using OpBF16Active = OpBF16Wide;
using OpBF16K512Active = OpBF16K512T;
#elif NNS_BF16_TILE16
using OpBF16Active = OpBF16;
using OpBF16K512Active = OpBF16K512T;
#else
using OpBF16Active = OpBF16T32;
using OpBF16K512Active = OpBF16K512;
#endif

// min over the lanes that carry the same query: l ^ 32 (32x32 tiles), and l ^ 16 too (16x16 tiles).  Row
// swaps (v_permlane32_swap / v_permlane16_swap, gfx950), not ds_bpermute: no LDS round trip on the slow
// path.  swap(t, t) returns {old t with its upper rows replaced by the partner's, and vice versa}: the min of
// the two halves of the pair is min(t[l], t[l ^ 32]) on every lane (checked on hardware:
// nns_selftest_lane_share).
template <bool T16>
__device__ __forceinline__ float share_min(float t)
{
    const unsigned tb = __float_as_uint(t);
    const auto s32 = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
    t = fminf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));          // min(t[l], t[l ^ 32])
    if constexpr (T16) {
        const unsigned tb2 = __float_as_uint(t);
        const auto s16 = __builtin_amdgcn_permlane16_swap(tb2, tb2, false, false);
        t = fminf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));      // min(t[l], t[l ^ 16])
    }
    return t;
}

struct FilterArgs {
    const float4 *qimg;     // [m_pad/32][16][64] 16-byte fragments
    const char *rimg;       // [n_pad/32][16 KiB]
    const float *rnorm;     // [n_pad]
    const float *qnorm;     // [m_pad]
    const DevScalars *scal;
    CandEntry *lists;       // [splits][m_pad/32][kCandCap][64 lanes]
    int *counts;            // [splits][m_pad/32][64 lanes]
    int total_slots, slots_per_split, m_pad, kt;
    int bf16;               // tau mode: 0 fp32 operands, 1 bf16 points, 2 fp32 points rounded to bf16 operands
    int share_thr;          // short streams: a query's lanes adopt the smallest of their thresholds
    int tile_rec;           // short streams: ONE record per (lane, ref tile) — (tile minimum, first ref of the lane's
                            // rows) — instead of one per score within the threshold; K5 re-ranks the lane's rows
#ifdef NNS_DIAG
    unsigned long long *stamps;   // diagnostic (NNS_FILTER_CLOCK): per-workgroup s_memtime / s_memrealtime
#endif
};

template <class OP>
__global__ __launch_bounds__(OP::kNW * 64) void filter_kernel(const FilterArgs a)
{
    constexpr int F_NW = OP::kNW;
    constexpr int SPB = OP::kSPB;                      // fragment steps per image block
    // Blocks and ring slots: a slot is always 32 fragment steps.  Up to 32 steps per block a slot holds BPS whole
    // blocks; deeper blocks straddle slots with a SUPER-PERIOD of SUP_SLOTS slots = SUP_BLKS blocks — 64 steps
    // (1024-deep): 2 slots = 1 block; 48 steps (768-deep): 3 slots = 2 blocks, the second one starting in the middle
    // of the second slot.  Accumulators carry across the slot barriers; every slot of a super-period brings the
    // norms of all its blocks.  In general lcm(SPB, 32) steps: 40 steps (640-deep) = 5 slots = 4 blocks.
    constexpr int SUP_GCD = SPB % 32 == 0 ? 32 : (SPB % 16 == 0 ? 16 : (SPB % 8 == 0 ? 8 : 1));
    // (24 steps, 384-deep: 3 slots = 4 blocks.)  "Deep" = the block does not divide the slot.
    constexpr bool DEEP = 32 % SPB != 0;
    constexpr int SUP_SLOTS = DEEP ? SPB / SUP_GCD : 1;
    constexpr int SUP_BLKS = DEEP ? 32 / SUP_GCD : 32 / SPB;
    constexpr int SPBLK = SUP_SLOTS;                   // (name kept: slots of a deep block's super-period)
    constexpr int BPS = DEEP ? 1 : 32 / SPB;           // image blocks per ring slot (shallow tiles)
    constexpr int BLK_BYTES = SPB * 1024;
    constexpr int SLOT_REFS = 32 * SUP_BLKS;           // norms DMAed with a slot
    constexpr int F_PPW = F_SLOT_COORD / 1024 / F_NW;   // 1 KiB DMA pieces per wave per slot
    static_assert((32 % SPB == 0 && SPB >= 2) || SPB == 64 || SPB == 48 || SPB == 40 || SPB == 24, "a slot is 32 fragment steps");
    static_assert(SUP_SLOTS * 32 == SUP_BLKS * SPB, "a super-period is whole slots and whole blocks");
    static_assert(SPBLK == 1 || (!OP::kLag && !OP::kTile16), "blocks straddling slots: lock-step 32x32 tiles only");
    static_assert(SLOT_REFS == 32 || SLOT_REFS == 64 || SLOT_REFS == 128 || SLOT_REFS == 256 || SLOT_REFS == 512,
                  "norm pieces: one dword per lane, or dwordx4 pieces of 256 norms");
    constexpr int F_NP = SLOT_REFS <= 256 ? 1 : SLOT_REFS / 256;   // norm DMA pieces per slot
    static_assert(SLOT_REFS * 4 <= F_SLOT_NORM, "norm room of a ring slot");
    // How far ahead of its interval a slot's DMA is issued.  Lagging SIMD partners still read slot s - 1 during
    // the first half of interval s, so with four ring slots they can fill slot s + 2 at most: ONE interval between
    // issue and deadline — plenty when an interval is 16 000 cycles (fp32 tiles), not when it is 1 000 - 4 000
    // (bf16 tiles: a 1 KiB piece takes ~2 600 cycles from issue to landed under load, and round 2's waves sat in
    // vmcnt(0) for half their cycles at the deep tiles).  Lock-step operators are done with slot s - 1 at the
    // barrier that opens interval s, so they fill slot s + 3 = s - 1 (mod 4) and wait with a COUNTED vmcnt that
    // leaves the youngest slot's pieces in flight across the barrier: TWO intervals of latency cover.
    constexpr int AHEAD = (OP::kLag || F_DMA_AHEAD_MAX < 3) ? 2 : 3;
#ifndef NNS_F_NORM_ONE
#define NNS_F_NORM_ONE 1
#endif
    // DMA pieces per wave and slot that the counted vmcnt waits may leave in flight: with the norms issued by one wave
    // per slot, a wave's youngest slot has F_PPW or F_PPW + F_NP pieces — counting F_PPW is safe for both
    // (not the 768-deep tile: 252 of its 256 registers are taken, the turn-taking test spills)
    constexpr bool NORM_ONE = NNS_F_NORM_ONE && SPB != 48;
    constexpr int F_PPS = NORM_ONE ? F_PPW : F_PPW + F_NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    constexpr int QB = OP::kQB;
    static_assert(QB <= 2 || OP::kTile16, "the 32x32 accumulator sets hold two query blocks");
    const int qblk0 = (blockIdx.x * F_NW + wave) * QB;
    // Lane STATES: the running minimum / threshold / candidate list a lane keeps per query it
    // carries.  32x32 tiles: one query per query block (state = block, 16 scores per ref block);
    // 16x16 tiles: one query per 16-query tile (state = tile, 8 scores per ref block).
    constexpr bool T16 = OP::kTile16;
    constexpr int NS = T16 ? 2 * QB : QB;
    constexpr int QPS = T16 ? 16 : 32;          // queries per state
    constexpr int NBQ = T16 ? SPB / 2 : SPB;    // resident operand fragments per state

    // ---- resident B operands: this wave's QB x 32 queries, all of K (128 VGPRs at QB = 2) ----
    float4 bq[NS][NBQ];
    TauConsts tc[NS];
    {
        const float4 *src = a.qimg + (size_t)qblk0 * (SPB * 64) + lane;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
#pragma unroll
            for (int b = 0; b < NBQ; ++b) bq[st][b] = src[(st * NBQ + b) * 64];
            // (record form 2 keeps no thresholds: skip the margin's double-precision arithmetic, ~1 us of a short stream)
            if (T16 || a.tile_rec != 2)
                tc[st] = tau_consts(a.kt, a.qnorm[qblk0 * 32 + st * QPS + (lane & (QPS - 1))],
                                    __uint_as_float(a.scal->ymax2_bits), a.bf16);
            else
                tc[st] = TauConsts{0.0f, 0.0f, 0.0f};
        }
    }
    // Pin the loads here: hipcc must wait for them BEFORE the ring starts, not with a
    // vmcnt(0) at their first use inside the loop (it cannot see the asm DMAs).
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
        for (int b = 0; b < NBQ; ++b)
            if constexpr (T16 && QB > 2)   // (AGPR-resident operands: see OpBF16T::mma16)
                asm volatile("" : "+a"(bq[st][b].x), "+a"(bq[st][b].y), "+a"(bq[st][b].z), "+a"(bq[st][b].w));
            else
                asm volatile("" : "+v"(bq[st][b].x), "+v"(bq[st][b].y), "+v"(bq[st][b].z), "+v"(bq[st][b].w));
        asm volatile("" : "+v"(tc[st].c0), "+v"(tc[st].c1), "+v"(tc[st].x2));
    }
    // The tau constants of the lane's states live in LDS behind the ring (2 KiB per wave, read only on
    // the slow path) instead of 3 registers per state; c1 depends on the tile depth alone and is
    // wave-uniform
    float *tcl = reinterpret_cast<float *>(smem + F_LDS_BYTES) + wave * (NS > 4 ? 1024 : 512) + lane;
    if constexpr (!OP::kTauInRegs) {
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            tcl[(2 * st) * 64] = tc[st].c0;
            tcl[(2 * st + 1) * 64] = tc[st].x2;
        }
    }
    const float c1u = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(tc[0].c1)));

#ifdef NNS_DIAG
    unsigned long long st_t0 = 0, st_r0 = 0;
    if (a.stamps) {   // diagnostic launches only: in-kernel clock = d(memtime) / d(memrealtime) * 100 MHz
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int slot0 = blockIdx.y * a.slots_per_split;
    int ns = a.total_slots - slot0;
    if (ns > a.slots_per_split) ns = a.slots_per_split;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);

    // DMA piece p (0 .. F_PPW-1: 1 KiB image pieces; F_PPW .. F_PPW+F_NP-1: the norms) of slot s
    // (relative to slot0) into ring position s % F_D
    auto issue_piece = [&](int s, int p) __attribute__((always_inline)) {
        const size_t gslot = (size_t)(slot0 + s);
        const unsigned dst = lds_base + (s & (F_D - 1)) * F_SLOT_BYTES;
        if (p < F_PPW) {
            const int piece = wave * F_PPW + p;
            dma16(a.rimg + gslot * F_SLOT_COORD + piece * 1024 + lane * 16, dst + piece * 1024);
        } else if (NORM_ONE && ((int)gslot & (F_NW - 1)) != wave) {
            // the slot's norms are ONE wave's business, the waves taking turns slot by slot (round 1 - 3: every wave copied
            // the same bytes to the same words, to keep the DMA counts per slot identical — eight pieces where one does,
            // and an LDS-DMA piece costs its SIMD ~130 cycles of MFMA issue whatever its size); the counted vmcnt waits
            // below therefore count IMAGE pieces only (a wave whose youngest slot carries norm pieces waits for them too)
        } else if constexpr (SLOT_REFS <= 64) {   // (128- and 256-ref slots: one dwordx4 piece of 256 norms)
            // (a 32-ref slot also copies the next slot's 32)
            dma4(a.rnorm + (gslot / SPBLK) * SLOT_REFS + lane, dst + F_SLOT_COORD);   // (a deep block: both its slots)
        } else {
            const int np = p - F_PPW;             // 256 norms per piece (a deep block: its super-period's, + over-read)
            dma16(a.rnorm + (gslot / SPBLK) * SLOT_REFS + np * 256 + lane * 4, dst + F_SLOT_COORD + np * 1024);
        }
    };
    auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < F_PPW + F_NP; ++p) issue_piece(s, p);
    };

    // ---- per-lane record state ---------------------------------------------------------
    float thr[NS];
    int cnt[NS];
    // candidate lists are stored [split][state unit][entry][lane] (unit = the 64 lanes' lists of
    // one query block / query tile) so that both the appends of a wave and K5's per-query reads
    // touch consecutive 8-byte words
    const size_t lblk0 = (size_t)blockIdx.y * (a.m_pad / QPS) + (size_t)qblk0 * (32 / QPS);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        thr[st] = __builtin_inff();
#if defined(NNS_DIAG) && defined(NNS_F_THR_NEGINF)   // timing experiment: the fast path alone (results are wrong)
        thr[st] = -__builtin_inff();
#endif
        cnt[st] = 0;
    }
    // The same lists as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: the appends of the
    // slow path then need no 64-bit VALU address arithmetic (global_store saddr + voffset)
    char *lbase;
    {
        const uintptr_t lb = (uintptr_t)(a.lists + lblk0 * (kCandCap * 64));
        // (readfirstlane returns a signed int: go through unsigned, or a low word with bit 31 set
        //  sign-extends over the high half)
        const unsigned lb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(lb >> 32));
        const unsigned lb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)lb);
        lbase = (char *)(((uintptr_t)lb_hi << 32) | (uintptr_t)lb_lo);
    }
    const unsigned loff = lane * (unsigned)sizeof(CandEntry);
    f32x4 nseed0 = {0.0f, 0.0f, 0.0f, 0.0f}, nseed1 = nseed0;   // 16x16 tiles: the two ref tiles' norms
    // accumulators start at |y'_j|^2 of their rows: rows (r&3) + 8(r>>2) + 4h
    auto seed = [&](typename OP::Acc &acc, const char *slot, int blk) __attribute__((always_inline)) {
        if constexpr (T16) {
            return;   // (16x16 tiles: seed16 below)
        } else {
        const float *nrm = reinterpret_cast<const float *>(slot + F_SLOT_COORD) + blk * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 nv = *reinterpret_cast<const float4 *>(nrm + 8 * g);
            // (compile-time qb: a runtime-looking index would send the accumulators to scratch)
            static_for<QB>([&](auto qc) __attribute__((always_inline)) {
                constexpr int qb = decltype(qc)::value;
                acc.template at<qb>()[4 * g + 0] = nv.x;
                acc.template at<qb>()[4 * g + 1] = nv.y;
                acc.template at<qb>()[4 * g + 2] = nv.z;
                acc.template at<qb>()[4 * g + 3] = nv.w;
            });
        }
        }
    };
    // Shallow lock-step tiles (KT = 16: a tile is two steps, and both SIMD partners reach their tile boundaries
    // together): the NEXT tile's norms are read a whole tile ahead into 16 registers, so the tile's first MFMA does not
    // wait for an LDS round trip every 8 MFMAs (at the deeper tiles the lagging partner's MFMA chain covers it).
#ifndef NNS_F_SEED_AHEAD
#define NNS_F_SEED_AHEAD 1
#endif
    constexpr bool kSeedAhead = NNS_F_SEED_AHEAD && !T16 && !OP::kLag && SPB <= 4;
    float4 nsd[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) nsd[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto seed_fetch = [&](const char *slot, int blk) __attribute__((always_inline)) {
        if constexpr (kSeedAhead) {
            const float *nrm = reinterpret_cast<const float *>(slot + F_SLOT_COORD) + blk * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) nsd[g] = *reinterpret_cast<const float4 *>(nrm + 8 * g);
        }
    };
    auto seed_apply = [&](typename OP::Acc &acc) __attribute__((always_inline)) {
        if constexpr (kSeedAhead) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                static_for<QB>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int qb = decltype(qc)::value;
                    acc.template at<qb>()[4 * g + 0] = nsd[g].x;
                    acc.template at<qb>()[4 * g + 1] = nsd[g].y;
                    acc.template at<qb>()[4 * g + 2] = nsd[g].z;
                    acc.template at<qb>()[4 * g + 3] = nsd[g].w;
                });
            }
        }
    };
    // 16x16 tiles: the norms of ref tile rt of block blk (rows 16 rt + 4 (lane >> 4) + i), loaded
    // two steps ahead of the tile's first MFMAs, which take them as srcC (no register copies)
    auto seed16 = [&](const char *slot, int blk, auto rt_c) __attribute__((always_inline)) {
        constexpr int rt = decltype(rt_c)::value;
        const float4 nv = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(slot + F_SLOT_COORD) +
                                                            blk * 32 + 16 * rt + 4 * (lane >> 4));
        if constexpr (rt == 0) nseed0 = f32x4{nv.x, nv.y, nv.z, nv.w};
        else nseed1 = f32x4{nv.x, nv.y, nv.z, nv.w};
    };
    // one finished score x of ref j (slow path): append to the state's candidate ring, tighten
    // Slow path, step 1: tighten the lane's threshold with the minimum tmin of the scores about to be
    // examined.  Doing this FIRST is what keeps the slow path short: every ref within tau of the
    // FINAL minimum is below every threshold the lane ever holds (thresholds only shrink), so
    // appending against the already-tightened threshold still collects all of them, while a run of
    // descending scores no longer appends (and stores) each one.
    auto tighten = [&](auto st_c, float tmin) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;
        // x -> x + 1.002 tau(x) is monotone, so the threshold of the running minimum is the minimum
        // of the thresholds: no separate running-minimum register.  (1.002: a hair wider than K5's
        // own tau, so that the lists are supersets of what K5 needs.)
        float c0v, x2v;
        if constexpr (OP::kTauInRegs) {
            c0v = tc[st].c0;
            x2v = tc[st].x2;
        } else {
            c0v = tcl[(2 * st) * 64];
            x2v = tcl[(2 * st + 1) * 64];
        }
        const float d = tmin + x2v;
        const float tn = tmin + (c0v + c1u * (d > 0.0f ? d : 0.0f)) * 1.002f;
        float t = tn < thr[st] ? tn : thr[st];
        // A query lives on 2 (32x32 tiles: lanes l, l ^ 32) or 4 (16x16: l ^ 16, l ^ 32) lanes, each seeing
        // a different part of every ref block.  Any of their thresholds is valid for all of them (each
        // is >= final minimum + tau), so they may adopt the smallest: a lane then stops taking the slow
        // path for refs a sibling lane has already beaten.  Pays on SHORT ref streams (where nearly
        // every tile is slow), costs 0.2 % on C3 / C5: a launch-time choice by stream length
        // (launch_filter).  Row swaps (v_permlane32_swap / 16_swap, gfx950), not ds_bpermute: no LDS
        // round trip.  (Wave-uniform slow path: every lane is active here.)
        if (a.share_thr) t = share_min<T16>(t);
        // (a finite threshold: "x <= thr" then also excludes the +INF scores of padding refs)
        thr[st] = fminf(t, 3.4028234663852886e38f);
    };
    // Slow path, step 2: one finished score x of ref j: append to the state's candidate ring (thr is finite
    // here — tighten clamps it — so "x <= thr" also excludes the +INF scores of padding refs).  `roomy`
    // (wave-uniform, computed once per slow tile): no lane of the wave is within a tile's worth of entries of
    // its ring's capacity, so the ring-wrap check — a load, hence a vmcnt wait that also drains the DMA ring —
    // is skipped by a scalar branch; it only runs on monotone inputs (every ref a new record).
    auto record = [&](auto st_c, float x, int j, bool roomy) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;
        if (x <= thr[st]) {
            const unsigned off = loff + (unsigned)(st * (kCandCap * 64 * (int)sizeof(CandEntry))) +
                                 (unsigned)(cnt[st] & (kCandCap - 1)) * (unsigned)(64 * sizeof(CandEntry));
            CandEntry *const dst = reinterpret_cast<CandEntry *>(lbase + off);
            if (__builtin_expect(!roomy, 0)) {
                // ring wrap: the slot's old entry may only be dropped if it is above the current
                // threshold (then it can never be within tau of the final minimum)
                if ((cnt[st] & kCandCountMask) >= kCandCap && dst->s <= thr[st]) cnt[st] |= kCandOverflow;
            }
            CandEntry e;
            e.s = x;
            e.j = j;
#if defined(NNS_DIAG) && defined(NNS_F_NOSTORE)   // timing experiment: the slow path without its global store
            asm volatile("" ::"v"(e.s), "v"(e.j), "v"(dst));
#else
            *dst = e;
#endif
            ++cnt[st];
        }
    };
    // minimum of the finished scores of one ref block for lane state st
    auto tile_min = [&](const typename OP::Acc &acc, auto st_c) __attribute__((always_inline)) -> float {
        constexpr int st = decltype(st_c)::value;
        if constexpr (T16) {
            const f32x4 &lo = acc.template at<0, st>();   // refs 4 g + i
            const f32x4 &hi = acc.template at<1, st>();   // refs 16 + 4 g + i
            const float t0 = fminf(fminf(lo[0], lo[1]), lo[2]);
            const float t1 = fminf(fminf(lo[3], hi[0]), hi[1]);
            return fminf(fminf(t0, t1), fminf(hi[2], hi[3]));
        } else {
            const f32x16 &t = acc.template at<st>();
            const float t0 = fminf(fminf(t[0], t[1]), t[2]);
            const float t1 = fminf(fminf(t[3], t[4]), t[5]);
            const float t2 = fminf(fminf(t[6], t[7]), t[8]);
            const float t3 = fminf(fminf(t[9], t[10]), t[11]);
            const float t4 = fminf(fminf(t[12], t[13]), t[14]);
            return fminf(fminf(fminf(t0, t1), t[15]), fminf(fminf(t2, t3), t4));
        }
    };
    // slow path of one state: every score of the block against the lane's threshold
#ifdef NNS_DIAG
    unsigned diag_slow = 0, diag_tiles = 0;   // slow-path executions / retired (tile, state) pairs of this wave
    unsigned long long diag_cyc = 0;   // s_memtime ticks spent inside record_all
#endif
    auto record_all = [&](const typename OP::Acc &acc, int blk_global, auto st_c) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;
#ifdef NNS_DIAG
        ++diag_slow;
        const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
#endif
        // (the tile minimum is recomputed here, on the cold path, rather than kept live across the branch)
        const float tmv = tile_min(acc, st_c);
        tighten(st_c, tmv);
        // is any lane of the wave about to wrap its 64-entry ring (monotone inputs)?
        const bool roomy = __builtin_amdgcn_ballot_w64((cnt[st] & kCandCountMask) + 16 > kCandCap) == 0ull;
        if (a.tile_rec) {
            // TILE RECORDS (short ref streams).  A wave carries 32 queries per state and a query improves on its
            // t-th tile with probability ~1/t, so for the first ~32-64 tiles of a stream nearly EVERY tile takes
            // this path in some lane, and a stream of 256 tiles (1024 queries x 1 M refs: 128 splits) takes it on
            // a third of them — at the 16-deep tile, where a tile is 8 MFMAs, that was 40 % of the kernel.  Here
            // the path is short whatever the scores look like: the lane records the TILE — its minimum and the first
            // of the lane's 16 rows — and K5 evaluates V0's distance for those rows.  Completeness is unchanged:
            // a ref within tau of the final minimum makes its tile's minimum <= every threshold the lane ever holds.
            record(st_c, tmv, blk_global * 32 + 4 * h, roomy);
        } else if constexpr (T16) {
            const f32x4 &lo = acc.template at<0, st>();
            const f32x4 &hi = acc.template at<1, st>();
            const int jbase = blk_global * 32 + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) record(st_c, lo[r], jbase + r, roomy);
#pragma unroll
            for (int r = 0; r < 4; ++r) record(st_c, hi[r], jbase + 16 + r, roomy);
        } else {
            // Typically ONE lane has ONE score to append, and every instruction of this path delays the whole
            // workgroup at the next slot barrier: test the scores four at a time first (wave-uniform), descend
            // only into a group that has one.
            const f32x16 &t = acc.template at<st>();
            const int jbase = blk_global * 32 + 4 * h;
            // (the groups are tile_min's own triples + the sixteenth score: hipcc shares the group minima
            //  between the two, and with any other grouping it splits the hot path's v_min3 to do so —
            //  10 VALU instructions per retired tile instead of 8)
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                const float gm = fminf(fminf(t[3 * g], t[3 * g + 1]), t[3 * g + 2]);
                if (__builtin_amdgcn_ballot_w64(gm <= thr[st]) != 0ull) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const int r = 3 * g + e;
                        record(st_c, t[r], jbase + (r & 3) + 8 * (r >> 2), roomy);
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(t[15] <= thr[st]) != 0ull) record(st_c, t[15], jbase + 3 + 8 * 3, roomy);
        }
#ifdef NNS_DIAG
        diag_cyc += __builtin_amdgcn_s_memtime() - diag_t0;
#endif
    };
    // ---- record form 2 (fp32 32 x 32 tiles on short ref streams): the lane's TWO BEST TILES, branch-free ---------
    // A wave carries 32 queries per state and a query improves on its t-th tile with probability ~1/t, so on a stream
    // of a few hundred tiles a third of the tiles took the threshold test's slow path in SOME lane (335 195 of
    // 1 048 576 tile retirements at 1024 x 1 M x 16, ~800 cycles each with one record per tile, ~1400 with one per
    // score: profiles/r03_streams.txt) — at the 16-deep tile, where a tile is 8 MFMAs, a quarter of the kernel.
    // Short streams do not need lists at all: a lane keeps the minima of its best and second-best tile, where they
    // came from, and the minimum of the third-best — 3 compares + 7 selects behind the tile's 8 v_min3, in the
    // shadow of the SIMD partner's MFMAs, no branch, no store, no threshold.  At the end of the stream the lane writes
    // three entries.  K5 takes a = the minimum over the entries and evaluates V0's distance for the rows of every
    // recorded tile whose minimum is <= a + tau(a); if a lane's THIRD minimum is within the threshold too (three
    // tiles of one lane's stream within tau: duplicates, or refs packed closer than the filter resolves) the query
    // goes to the exact scan.  Completeness: tiles of a lane's stream with a minimum <= a + tau(a) are a prefix of
    // its tiles sorted by minimum — of length <= 2 unless the third is inside too.
    float m1[NS], m2[NS], m3[NS];
    int t1[NS], t2[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        m1[st] = m2[st] = m3[st] = __builtin_inff();
        t1[st] = t2[st] = 0;
    }
    auto epilogue_top2 = [&](const typename OP::Acc &acc, int blk_global) __attribute__((always_inline)) {
        if constexpr (!T16) {
            static_for<NS>([&](auto st_c) __attribute__((always_inline)) {
                constexpr int st = decltype(st_c)::value;
                const float tm = tile_min(acc, st_c);
                // sorted insert of tm into m1 <= m2 <= m3 by v_med3 / v_min; the tile numbers follow with selects.
                // (The old values go through empty asm statements: a select between two elements of these arrays is
                //  otherwise rewritten by hipcc into a load through a selected POINTER, which keeps the arrays in scratch.)
                float o1 = m1[st], o2 = m2[st];
                int p1 = t1[st], p2 = t2[st];
                asm volatile("" : "+v"(o1), "+v"(o2), "+v"(p1), "+v"(p2));
                const bool lt1 = tm < o1, lt2 = tm < o2;   // strict: equal minima fill the next rank
                // (o1 <= o2 <= m3: the new second is the median of {tm, o1, o2}, the new third that of {tm, o2, m3}; scores
                //  are finite or +INF, never NaN: K2 routes NaN inputs to the exact kernels)
                m3[st] = __builtin_amdgcn_fmed3f(tm, o2, m3[st]);
                m2[st] = __builtin_amdgcn_fmed3f(tm, o1, o2);
                m1[st] = fminf(o1, tm);
                t2[st] = lt1 ? p1 : (lt2 ? blk_global : p2);
                t1[st] = lt1 ? blk_global : p1;
            });
        }
    };
    // record collection at the end of a ref block
    auto epilogue = [&](const typename OP::Acc &acc, int blk_global) __attribute__((always_inline)) {
        if constexpr ((kAblate & 2) != 0) {   // diagnostic: keep the accumulators alive, collect nothing
            static_for<NS>([&](auto st_c) __attribute__((always_inline)) {
                constexpr int st = decltype(st_c)::value;
                if constexpr (T16) {
                    const f32x4 lo = acc.template at<0, st>(), hi = acc.template at<1, st>();
                    asm volatile("" ::"v"(lo), "v"(hi));
                } else {
                    const f32x16 t = acc.template at<st>();
                    asm volatile("" ::"v"(t));
                }
            });
            return;
        }
        if constexpr (T16) {
            // (16x16 tiles retire their ref tiles inside t16_step)
        } else {
            static_for<NS>([&](auto st_c) __attribute__((always_inline)) {
                constexpr int st = decltype(st_c)::value;
                const float tm = tile_min(acc, st_c);
#ifdef NNS_DIAG
                ++diag_tiles;
#endif
#ifdef NNS_F_NOEXPECT
                if (__builtin_amdgcn_ballot_w64(tm <= thr[st]) != 0ull)   // rare: ~ln(n) tiles per lane
#else
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(tm <= thr[st]) != 0ull, 0))   // rare: ~ln(n) tiles per lane
#endif
                    record_all(acc, blk_global, st_c);
            });
        }
    };
    auto mma_all = [&](typename OP::Acc &acc, const float4 &frag, auto b_c) __attribute__((always_inline)) {
        constexpr int b = decltype(b_c)::value;
        if constexpr (T16) {
            // (16x16 tiles go through t16_step)
        } else {
            static_for<QB>([&](auto qc) __attribute__((always_inline)) {
                constexpr int qb = decltype(qc)::value;
                acc.template at<qb>() = OP::mma(frag, bq[qb][b], acc.template at<qb>());
            });
        }
    };

    // ---- 16x16 tiles: one step, with the other ref tile's record collection folded in --------
    constexpr int NT16 = T16 ? NS : 4;
    float tmh[NT16];   // minima of the retiring ref tile, one per state
    unsigned long long hm[NT16];
#pragma unroll
    for (int i = 0; i < NT16; ++i) {
        tmh[i] = 0.0f;
        hm[i] = 0ull;
    }
    // threshold test of the retiring ref tile ot of block oblk: ONE wave-uniform branch for the
    // common case (no lane has a record in any of its four states), then per state inside
    auto t16_test = [&](typename OP::Acc &acc, int oblk, auto ot_c) __attribute__((always_inline)) {
        if constexpr (T16) {
            constexpr int ot = decltype(ot_c)::value;
            unsigned long long any = 0ull;
#pragma unroll
            for (int i = 0; i < NT16; ++i) any |= hm[i];
            if (__builtin_expect(any != 0ull, 0)) {   // cold: laid out off the MFMA stream
                const int jbase = oblk * 32 + 16 * ot + 4 * (lane >> 4);
                static_for<NT16>([&](auto st_c) __attribute__((always_inline)) {
                    constexpr int st = decltype(st_c)::value;
                    if (hm[st] != 0ull) {
                        const f32x4 &o = acc.template at<ot, st>();
                        tighten(st_c, tmh[st]);
                        const bool roomy = __builtin_amdgcn_ballot_w64((cnt[st] & kCandCountMask) + 4 > kCandCap) == 0ull;
                        if (a.tile_rec) {   // one record for the lane's four rows of this 16-ref tile (see record_all)
                            record(st_c, tmh[st], jbase, roomy);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) record(st_c, o[r], jbase + r, roomy);
                        }
                    }
                });
            }
        }
    };
    // the four states' lane masks "this lane has a score within its threshold" (v_cmp straight to
    // SGPR pairs, computed a step before the branch that reads them)
    auto t16_masks = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < NT16; ++st) hm[st] = __builtin_amdgcn_ballot_w64(tmh[st] <= thr[st]);
    };
    // step b = 8 rt + ks of the block blk_global: the 4 MFMAs of ref tile rt's k-step ks; at
    // ks = 1 the other tile's four accumulators (finished at its k-step 7, >= 5 MFMAs ago) are
    // reduced, two v_min behind each MFMA; at ks = 2 they are tested.  The other tile of rt = 0
    // is tile 1 of the PREVIOUS block (kernel start: +INF accumulators, nothing to record).
    auto t16_step = [&](typename OP::Acc &acc, const float4 &frag, int blk_global, auto b_c) __attribute__((always_inline)) {
        if constexpr (T16) {
            constexpr int b = decltype(b_c)::value;
            constexpr int NKS = SPB / 2;              // k-steps of 32 dims per 16-ref tile
            // (with two query tiles per wave a step is two MFMAs: retire one k-step later, so that the reads of the other
            //  tile's accumulators still sit >= 5 MFMAs behind the MFMAs that finished them)
            constexpr int RET = NT16 >= 4 ? 1 : 2;
            static_assert(NKS >= RET + 2, "the other tile is retired at k-steps RET and RET + 1");
            constexpr int rt = b / NKS, ks = b % NKS, ot = 1 - rt;
            static_for<NT16>([&](auto qc) __attribute__((always_inline)) {
                constexpr int qt = decltype(qc)::value;
                if constexpr (ks == 0) OP::mma16_seed(frag, bq[qt][0], acc.template at<rt, qt>(), rt == 0 ? nseed0 : nseed1);
                else OP::mma16(frag, bq[qt][ks], acc.template at<rt, qt>());
                if constexpr (ks == RET) {
                    // Name the retiring tile's accumulator as ONE in/out tuple before its elements are read:
                    // without it hipcc carries elements 1 and 3 of each tile across the loop back-edge in
                    // separate registers (8 v_mov at the latch, 8 more to put them back in front of the tile's
                    // seed MFMA, whose in/out constraint "reads" them): 16 of the 74 VALU instructions of an
                    // interval, in a loop where every VALU issue cycle is an MFMA issue cycle lost.
                    if constexpr (OP::kAsmMfma) asm volatile("" : "+v"(acc.template at<ot, qt>()));
                    const f32x4 o = acc.template at<ot, qt>();
                    if constexpr ((kAblate & 2) != 0) asm volatile("" ::"v"(o));
                    else tmh[qt] = fminf(fminf(fminf(o[0], o[1]), o[2]), o[3]);
                    __builtin_amdgcn_sched_barrier(0);   // keep the pair right behind its MFMA
                }
            });
            if constexpr (ks == RET && (kAblate & 2) == 0) t16_masks();
            if constexpr (ks == RET + 1 && (kAblate & 2) == 0) t16_test(acc, rt == 0 ? blk_global - 1 : blk_global, std::integral_constant<int, ot>{});
        }
    };

    // ---- the software pipeline of one barrier interval -----------------------------------
    // An interval is the 32 fragment steps of one ring slot (BPS image blocks x SPB fragments;
    // one step = one ds_read_b128 + its MFMAs for the wave's query blocks).  Waves 0..3
    // (LAG = 0) run the slot's blocks in order; waves 4..7 (LAG = 1) run HALF A BLOCK behind
    // (they start each interval with the second half of the previous slot's last block), so
    // one SIMD partner's tile boundary (epilogue, accumulator re-seed from the norms) falls in
    // the middle of the other's MFMA chain.  Fragments are prefetched PF steps ahead through
    // a register ring that is carried ACROSS the barrier: the LDS ring is 4 slots deep and the
    // barrier of interval s confirms slot s + 1, so the first fragments of interval s + 1 are
    // already in flight when its barrier releases and the MFMA chain restarts at once.
    constexpr int PF = OP::kPrefetch;
    constexpr int RING = PF < 4 ? 4 : 8;
    constexpr int LAGOFF = SPB / 2;
    static_assert(PF < RING && 32 % RING == 0 && (!OP::kLag || PF <= LAGOFF), "prefetch ring");
    typename OP::Acc acc;
    if constexpr (T16) {
        const f32x4 inf4 = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
        static_for<NT16>([&](auto qc) __attribute__((always_inline)) {
            acc.template at<0, decltype(qc)::value>() = inf4;
            acc.template at<1, decltype(qc)::value>() = inf4;
        });
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.v0[r] = acc.v1[r] = __builtin_inff();
    }
    float4 fr[RING];

    auto frag_ptr = [&](const char *slot, int blk, int f) __attribute__((always_inline)) {
        return (reinterpret_cast<const float4 *>(slot + (DEEP ? 0 : blk * BLK_BYTES)) + lane) + f * 64;
    };

    using I0c = std::integral_constant<int, 0>;
    using I1c = std::integral_constant<int, 1>;
    // s: slot index relative to slot0; cur/prev/nxt: ring images of slots s, s-1, s+1
    auto interval = [&](auto lag_c, auto dph_c, auto half_c, auto rec_c, int s, const char *cur, const char *prev, const char *nxt) __attribute__((always_inline)) {
        constexpr int LAG = decltype(lag_c)::value;
        constexpr int REC2 = decltype(rec_c)::value;   // 1: record form 2 (the lane's two best tiles), 32 x 32 tiles only
        constexpr int DPH = decltype(dph_c)::value;   // DMA phase: the SIMD partners issue at different steps
        constexpr int HALF = decltype(half_c)::value; // blocks spanning two slots: which of them this interval is
        const bool first = s == 0;
        // first block of this slot (shallow) / of this slot's super-period (deep: slot0 and ns are whole super-periods)
        const int blk0_global = SPBLK == 1 ? (slot0 + s) * BPS : (slot0 + s) / SUP_SLOTS * SUP_BLKS;
        // compile-time schedule: step t works on position u = t - LAG * LAGOFF of the slot's
        // fragment stream (u < 0: tail of the previous slot; u >= 32: head of the next one)
        auto load = [&](auto tc_) __attribute__((always_inline)) {
            constexpr int t = decltype(tc_)::value;
            constexpr int u = t - LAG * LAGOFF;
            // (fragment u of a slot sits at u KiB whatever the block depth: block * SPB + step = u)
            if constexpr (u < 0) fr[t % RING] = *frag_ptr(prev, 0, 32 + u);
            else if constexpr (u < 32) fr[t % RING] = *frag_ptr(cur, 0, u);
            else fr[t % RING] = *frag_ptr(nxt, 0, u - 32);   // (last interval: stale slot, unused)
        };
        static_for<32>([&](auto tc_) __attribute__((always_inline)) {
            constexpr int t = decltype(tc_)::value;
            constexpr int u = t - LAG * LAGOFF;
            // block (of the slot / of the super-period) and operand / k position inside it
            constexpr int blk = SPBLK == 1 ? (u < 0 ? BPS - 1 : u / SPB) : (32 * HALF + u) / SPB;
            constexpr int b = SPBLK == 1 ? (u < 0 ? SPB + u : u % SPB) : (32 * HALF + u) % SPB;
            load(std::integral_constant<int, t + PF>{});
            // one DMA piece per step, two slots ahead, in the MFMA shadow, at different steps
            // for the two SIMD partners (an LDS-DMA issue stalls the issuing wave ~100 cycles;
            // past the end of the shard it reads the image's padding)
            if constexpr ((kAblate & 16) == 0) {
#ifdef NNS_F_DMA_SAMEPHASE
                constexpr int d0 = 2;
#else
                constexpr int d0 = DPH == 0 ? 2 : 16;
#endif
#ifdef NNS_F_DMA_SP
                constexpr int sp = NNS_F_DMA_SP;
#else
                // steps between pieces: two where they fit — except K4's 256-deep 16x16x32 form, whose pieces go out on consecutive
                // steps (same-device A/B on C5: 78.2 -> 77.5 ms, +0.9 %; the 512- / 384-deep tiles lose 1 % that way, three steps
                // apart loses everywhere: profiles/r03_ab_dma_burst.txt)
                constexpr int sp = (T16 && SPB == 16 && QB == 2) ? 1 : ((F_PPW + F_NP) * 2 <= 14 ? 2 : 1);
#endif
                static_assert(d0 + sp * (F_PPW + F_NP) <= 32, "DMA pieces must fit the interval");
                // fp32 operators issue a slot's pieces BACK TO BACK at one step (round 3, second session): a lone LDS-DMA piece
                // costs its SIMD ~130 cycles of MFMA issue, a burst of them far less per piece (tools/ubench/chain_loop.hip:
                // 36.8 -> 25.2 ms); C3 118.3 -> 117.6 ms (OP::kDmaBurst: by tile depth).  The bf16 operators keep one piece per step: bursts cost THEM 1 - 6 %
                // (profiles/r03_ab_dma_burst.txt) — their intervals are 8 - 16x shorter and a burst stalls the lock-step partner too.
#ifdef NNS_F_DMA_BURST
                constexpr bool BURST = NNS_F_DMA_BURST != 0;
#else
                constexpr bool BURST = OP::kDmaBurst;
#endif
                if constexpr (BURST) {
                    if constexpr (t == d0) issue(s + AHEAD);
                } else if constexpr (t >= d0 && t < d0 + sp * (F_PPW + F_NP) && (t - d0) % sp == 0) {
                    issue_piece(s + AHEAD, (t - d0) / sp);
                }
            }
            if constexpr (T16) {
                // norms two steps ahead: tile 1 of this block; tile 0 of the next block (next ring
                // slot after the slot's last block: confirmed by this interval's barrier)
                if constexpr (b == SPB / 2 - 2) seed16(cur, blk, I1c{});
                if constexpr (b == SPB - 2) seed16(blk + 1 < BPS ? cur : nxt, (blk + 1) % BPS, I0c{});
            } else if constexpr (b == 0) {   // a tile is seeded right where it starts
                if constexpr (kSeedAhead) {
                    seed_apply(acc);   // (read a tile ago; the next tile's: block blk + 1 of this slot, or the next slot's first —
                    seed_fetch(blk + 1 < BPS ? cur : nxt, (blk + 1) % BPS);   // landed and confirmed by this interval's barrier)
                } else {
                    seed(acc, cur, blk);
                }
            }
            // (LAG 1, very first interval: its first LAGOFF steps chew on a not-yet-written ring
            //  slot; that accumulator is discarded below and re-seeded at the next tile)
            if constexpr (T16) {
                t16_step(acc, fr[t % RING], blk0_global + blk, std::integral_constant<int, b>{});
            } else {
                mma_all(acc, fr[t % RING], std::integral_constant<int, b>{});
                if constexpr (b == SPB - 1) {                // the tile's epilogue right at its end
                    if constexpr (REC2) {   // (a first interval's lagging half block chewed on garbage: skipped like below)
                        if constexpr (u < 0) {
                            if (!first) epilogue_top2(acc, blk0_global - 1);
                        } else {
                            epilogue_top2(acc, blk0_global + blk);
                        }
                    } else if constexpr (u < 0) {
                        if (!first) epilogue(acc, blk0_global - 1);
                    } else {
                        epilogue(acc, blk0_global + blk);
                    }
                }
            }
#ifndef NNS_F_NOSCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
        });
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const bool half = wave >= F_NW / 2;   // wave-uniform: the second wave of each SIMD
#ifdef NNS_F_NOLAG
    const bool lag = false;   // diagnostic: SIMD partners in lock-step
#else
    const bool lag = OP::kLag && half;
#endif
    auto ring = [&](int s) __attribute__((always_inline)) { return smem + ((s + F_D) & (F_D - 1)) * F_SLOT_BYTES; };
    static_assert((F_D & (F_D - 1)) == 0, "ring depth must be a power of two");

    // prologue: slots 0 .. AHEAD - 1 in flight; confirm slot 0; start interval 0's first fragments
    issue(0);
    issue(1);
    if constexpr (AHEAD == 3) issue(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * F_PPS) : "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (T16) seed16(ring(0), 0, I0c{});   // the first block's tile-0 norms
    seed_fetch(ring(0), 0);                          // (shallow lock-step tiles: the first tile's norms)
    // (LAG 1 reads ring slot -1 here: garbage in, discarded — see the interval)
    static_for<PF>([&](auto t) __attribute__((always_inline)) {
        constexpr int tt = decltype(t)::value;
        if constexpr (OP::kLag) {
            if (lag) fr[tt % RING] = *frag_ptr(ring(-1), BPS - 1, SPB - LAGOFF + tt);
            else fr[tt % RING] = *frag_ptr(ring(0), 0, tt);
        } else {
            fr[tt % RING] = *frag_ptr(ring(0), 0, tt);   // (slot-local fragment tt)
        }
    });
    // The slot loop, one copy per interval variant (the dispatch is wave-uniform and loop-invariant): with
    // the three variants as branches inside ONE loop, hipcc parks the accumulators in different register
    // tuples at the merge and at the loop header and reconciles them with ~48 v_mov at every back-edge.
    // Every wave executes exactly ns barriers whichever copy it runs.
    auto slot_loop = [&](auto lag_c, auto dph_c, auto rec_c) __attribute__((always_inline)) {
        auto sync_slot = [&]() __attribute__((always_inline)) {
            if constexpr ((kAblate & 1) == 0) {
                // my share of slot s+1 has landed.  AHEAD 2: issued an interval ago, the only DMA in flight.  AHEAD 3:
                // issued two intervals ago; the F_PPS pieces of slot s+2 issued since stay in flight (vmcnt counts
                // in issue order; a candidate store of the slow path issued in between only makes the wait longer)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 2) * F_PPS) : "memory");
                // everyone's share of slot s+1 has landed; everyone is done with slot s-2
                __builtin_amdgcn_s_barrier();
            }
        };
        if constexpr (SPBLK == 2) {
            // a block = two slots (ns is even: filter_plan): even slot = k-steps 0..31, odd slot = 32..63
            for (int s = 0; s < ns; s += 2) {
                sync_slot();
                interval(lag_c, dph_c, I0{}, rec_c, s, ring(s), ring(s - 1), ring(s + 1));
                sync_slot();
                interval(lag_c, dph_c, I1{}, rec_c, s + 1, ring(s + 1), ring(s), ring(s + 2));
            }
        } else if constexpr (SPBLK > 2) {
            // a super-period of SPBLK slots (ns is a multiple of it: filter_plan) — 48-step blocks: A 0..31 | A 32..47,
            // B 0..15 | B 16..47; 40-step blocks: four blocks over five slots
            for (int s = 0; s < ns; s += SPBLK) {
                static_for<SPBLK>([&](auto ph_c) __attribute__((always_inline)) {
                    constexpr int ph = decltype(ph_c)::value;
                    sync_slot();
                    interval(lag_c, dph_c, ph_c, rec_c, s + ph, ring(s + ph), ring(s + ph - 1), ring(s + ph + 1));
                });
            }
        } else {
            for (int s = 0; s < ns; ++s) {
                sync_slot();
                interval(lag_c, dph_c, I0{}, rec_c, s, ring(s), ring(s - 1), ring(s + 1));
                if constexpr (!OP::kLag) {
                    // lock-step partners (staggering only their DMA issue steps, or packing / spreading the
                    // pieces differently, measured +-0.5 % on C5).  The interval ends on the MFMAs that finish
                    // ref tile 1, which is retired in the NEXT interval: hipcc is free to split / copy those
                    // accumulators at the loop back-edge (it did: v_mov of single elements right behind the
                    // MFMAs = stale reads).  Whatever it does with them now happens behind the 8 wait states
                    // their readers need.
#ifndef NNS_F_NOLATCHFENCE   // (timing experiments only: without it the results are wrong)
                    if constexpr (T16) OP::mma16_tail_fence(acc);
#endif
                }
            }
        }
    };
    // (the record form, like the interval variant, is wave-uniform and loop-invariant: one copy of the loop each)
    const bool top2 = !T16 && a.tile_rec == 2;
    if constexpr (OP::kLag) {
        if (top2) {
            if (!half) slot_loop(I0{}, I0{}, I1{});
            else if (lag) slot_loop(I1{}, I1{}, I1{});
            else slot_loop(I0{}, I1{}, I1{});
        } else {
            if (!half) slot_loop(I0{}, I0{}, I0{});
            else if (lag) slot_loop(I1{}, I1{}, I0{});
            else slot_loop(I0{}, I1{}, I0{});
        }
    } else if constexpr (!T16) {
        if (top2) slot_loop(I0{}, I0{}, I1{});
        else slot_loop(I0{}, I0{}, I0{});
    } else {
        slot_loop(I0{}, I0{}, I0{});
    }
    if constexpr (T16) {   // ref tile 1 of the last block is still to be retired
        OP::mma16_tail_fence(acc);
        if constexpr ((kAblate & 2) == 0) {
            static_for<NT16>([&](auto qc) __attribute__((always_inline)) {
                const f32x4 &o = acc.template at<1, decltype(qc)::value>();
                tmh[decltype(qc)::value] = fminf(fminf(fminf(o[0], o[1]), o[2]), o[3]);
            });
            t16_masks();
            t16_test(acc, (slot0 + ns) * BPS - 1, std::integral_constant<int, 1>{});
        }
    }
    if constexpr (OP::kLag) if (lag) {   // the lagging half block of the last slot; its first PF fragments are in the ring
        const char *lastp = ring(ns - 1);
        static_for<LAGOFF>([&](auto tc_) __attribute__((always_inline)) {
            constexpr int t = decltype(tc_)::value;
            if constexpr (t + PF < LAGOFF) fr[(t + PF) % RING] = *frag_ptr(lastp, BPS - 1, SPB - LAGOFF + t + PF);
            mma_all(acc, fr[t % RING], std::integral_constant<int, SPB - LAGOFF + t>{});
        });
        if (top2) epilogue_top2(acc, (slot0 + ns) * BPS - 1);
        else epilogue(acc, (slot0 + ns) * BPS - 1);
    }
    if constexpr (!T16) if (top2) {
        // record form 2: three entries per lane state — (best tile's minimum, first ref of the lane's rows of that tile),
        // the same for the second best, and the third-best minimum alone (K5: within the threshold = exact scan)
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            CandEntry *const dst = reinterpret_cast<CandEntry *>(lbase + loff + (unsigned)(st * (kCandCap * 64 * (int)sizeof(CandEntry))));
            CandEntry e;
            e.s = m1[st];
            e.j = t1[st] * 32 + 4 * h;
            dst[0] = e;
            e.s = m2[st];
            e.j = t2[st] * 32 + 4 * h;
            dst[64] = e;
            e.s = m3[st];
            e.j = 0;
            dst[128] = e;
            cnt[st] = 3;
        }
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) a.counts[(lblk0 + st) * 64 + lane] = cnt[st];
#ifdef NNS_DIAG
    if (a.stamps && lane == 0) {   // totals behind the per-workgroup stamps
        unsigned long long *tot = a.stamps + 4 * (size_t)gridDim.x * gridDim.y;
        atomicAdd(tot, (unsigned long long)diag_slow);
        atomicAdd(tot + 1, (unsigned long long)diag_tiles);
        atomicAdd(tot + 2, diag_cyc);
    }
    if (a.stamps && threadIdx.x == 0) {
        unsigned long long *o = a.stamps + 4 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = st_t0;
        o[1] = st_r0;
        o[2] = __builtin_amdgcn_s_memtime();
        o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---- self-test: one 32x32 tile through the same MFMA k-order as the filter --------
// out[i][j] = accumulate over the image's k order of a[i][.] * b[j][.] seeded with c0[i];
// a, b are [32][kt] fp32 (bf16 mode: values must be bf16-representable).  Lets the tests
// check the hardware against host arithmetic — the error model behind tau.
__global__ __launch_bounds__(64) void mfma_selftest_kernel(int kt, int bf16, const float *__restrict__ a,
                                                           const float *__restrict__ b,
                                                           const float *__restrict__ c0,
                                                           float *__restrict__ out)
{
    const int lane = threadIdx.x, h = lane >> 5, i = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = c0[(r & 3) + 8 * (r >> 2) + 4 * h];
    if (!bf16) {
        for (int s = 0; s < kt / 2; ++s) {
            const int kk = 8 * (s >> 2) + 4 * h + (s & 3);   // same k permutation as the image
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * kt + kk], b[i * kt + kk], acc, 0, 0, 0);
        }
    } else if (bf16 == 1) {
        for (int s = 0; s < kt / 16; ++s) {
            bf16x8 av, bv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                av[e] = (__bf16)a[i * kt + 16 * s + 8 * h + e];
                bv[e] = (__bf16)b[i * kt + 16 * s + 8 * h + e];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
    } else {
        // bf16 == 2: the same 32x32 product as four 16x16 tiles of v_mfma_f32_16x16x32_bf16, with
        // the operand / result lane mapping the filter's OpBF16 and K2's order 1 image assume
        const int c = lane & 15, g = lane >> 4;
        for (int rt = 0; rt < 2; ++rt)
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = c0[16 * rt + 4 * g + e];
                for (int ks = 0; ks < kt / 32; ++ks) {
                    bf16x8 av, bv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        av[e] = (__bf16)a[(16 * rt + c) * kt + 32 * ks + 8 * g + e];
                        bv[e] = (__bf16)b[(16 * qt + c) * kt + 32 * ks + 8 * g + e];
                    }
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, t, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(16 * rt + 4 * g + e) * 32 + 16 * qt + c] = t[e];
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

// self-test of share_min: out[l] = the value lane l would adopt from in[0..63]
__global__ __launch_bounds__(64) void lane_share_selftest_kernel(int t16, const float *__restrict__ in, float *__restrict__ out)
{
    const float t = in[threadIdx.x];
    out[threadIdx.x] = t16 ? share_min<true>(t) : share_min<false>(t);
}

int launch_lane_share_selftest(int t16, const float *in, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(lane_share_selftest_kernel, dim3(1), dim3(64), 0, st, t16, in, out);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

int launch_mfma_selftest(int kt, int bf16, const float *a, const float *b, const float *c0, float *out,
                         hipStream_t st)
{
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, st, kt, bf16, a, b, c0, out);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

// ---- planning + launch ---------------------------------------------------------------
// streams of at most this many 32-ref tiles per workgroup run with shared lane thresholds
constexpr int64_t kShareThrMaxTiles = 2048;
#ifndef NNS_F_TILEREC_MAX
#define NNS_F_TILEREC_MAX 2048
#endif
constexpr int64_t kTileRecMaxTiles = NNS_F_TILEREC_MAX;

int filter_plan(int k, int m, int n, bool bf16, FilterGeom *g, bool mixed, bool per_ref)
{
    if (mixed) bf16 = true;   // fp32 points, bf16 operands: the bf16 filter's geometry
    int kt = 0;
    if (bf16) {
        if (k <= 128) kt = 128;        // OpBF16K128: 4 k-steps per 16-ref tile, 4 blocks per slot
        else if (k <= 256) kt = 256;
        else if (k <= 384) kt = 384;   // OpBF16K384: four 24-step blocks over three ring slots
        else if (k <= 512) kt = 512;   // OpBF16K512
        else if (k <= 640) kt = 640;   // OpBF16K640: four 40-step blocks over five ring slots
        else if (k <= 768) kt = 768;   // OpBF16K768: two 48-step blocks over three ring slots
        else if (k <= 1024) kt = 1024; // OpBF16K1024: K-split accumulation over two ring slots per block
    } else {
        if (k <= 16) kt = 16;          // OpF32K16: 2 fragment steps per block, 16 blocks per slot
        else if (k <= 32) kt = 32;     // OpF32K32: 4 fragment steps per block, 8 blocks per slot
        else if (k <= 64) kt = 64;     // OpF32K64: 8 steps per block, 4 blocks per slot
        else if (k <= 128) kt = 128;
        else if (k <= 256) kt = 256;   // OpF32K256: 32 fragment steps per block, 1 block per slot
    }
    if (!kt) {
        set_error("MFMA filter: k = %d exceeds the deepest tile (fp32 operands: 256, bf16 operands: 1024)", k);
        return NNS_ERR_UNSUPPORTED;
    }
    g->bf16 = bf16 ? 1 : 0;
    g->mixed = mixed ? 1 : 0;
    g->kt = kt;
    g->lpq = (bf16 && kt <= 512 && kt != 384 && OpBF16Active::kTile16) ? 4 : 2;
    // queries per workgroup
    const int qw = 32 * (bf16 ? (kt == 1024  ? OpBF16K1024::kQB * OpBF16K1024::kNW
                              : kt == 768 ? OpBF16K768::kQB * OpBF16K768::kNW
                              : kt == 640 ? OpBF16K640::kQB * OpBF16K640::kNW
                              : kt == 384 ? OpBF16K384::kQB * OpBF16K384::kNW
                              : kt == 512 ? OpBF16K512Active::kQB * OpBF16K512Active::kNW
                                          : OpBF16Active::kQB * OpBF16Active::kNW)
                              : (kt == 256 ? OpF32K256::kQB * OpF32K256::kNW : OpF32::kQB * OpF32::kNW));
    g->m_pad = divup(m, qw) * qw;
    // refs per ring slot (32 fragment steps of 8 fp32 / 16 bf16 dims)
    const int steps_per_block = bf16 ? kt / 16 : kt / 8;
    // deep blocks straddle slots: super-periods of `slots_per_block` slots = `pad_pts` refs (1024-deep: 2 slots = one
    // block of 32; 768-deep: 3 slots = two blocks = 64 refs, i.e. 21.33 refs per slot — slot_pts, which only sizes
    // paddings from here on, is rounded up)
    // (640-deep: 5 slots = four blocks = 128 refs; 384-deep: 3 slots = four blocks = 128 refs)
    const bool deep = 32 % steps_per_block != 0;
    const int sup_gcd = steps_per_block % 32 == 0 ? 32 : (steps_per_block % 16 == 0 ? 16 : 8);
    const int slots_per_block = deep ? steps_per_block / sup_gcd : 1;                         // slots of a super-period
    const int pad_pts = deep ? 32 * (32 / sup_gcd) : 32 * (32 / steps_per_block);            // refs of a super-period: whole blocks
    const int slot_pts = deep ? (pad_pts + slots_per_block - 1) / slots_per_block : pad_pts;
    g->n_pad = divup(n, pad_pts) * pad_pts;
    g->total_slots = g->n_pad / pad_pts * slots_per_block;
    g->qgroups = g->m_pad / qw;
    // One 8-wave workgroup is resident per CU (132 KiB of LDS), so the grid runs in rounds of
    // 256 workgroups and a round that is mostly empty costs as much as a full one.  Choose the
    // number of ref-range splits that minimises rounds x (work per workgroup), i.e.
    // ceil(qgroups * s / 256) / s, with a small per-split charge (prologue, lists, merge) and a
    // cap on the candidate-list memory (512 B per lane-list: 1 or 2 KiB per query per split).
    int splits = 1;
    {
        const int64_t list_cap = (int64_t)2 << 30;
        double best_cost = 1e30;
        for (int sp = 1; sp <= 64; ++sp) {
            if (sp > g->total_slots) break;
            if (sp > 1 && (int64_t)sp * g->m_pad * g->lpq * 512 > list_cap) break;
            const int rounds = divup(g->qgroups * sp, 256);
            const double cost = (double)rounds / sp * (1.0 + 0.004 * sp);
            if (cost < best_cost - 1e-12) {
                best_cost = cost;
                splits = sp;
            }
        }
        // very few queries: more splits than the scan above tries, to cover all CUs
        if (g->qgroups * splits < 256) splits = divup(256, g->qgroups);
    }
    if (splits > g->total_slots) splits = g->total_slots;
    if (splits > 65535) splits = 65535;
    g->slots_per_split = divup(divup(g->total_slots, splits), slots_per_block) * slots_per_block;   // whole blocks
    g->splits = divup(g->total_slots, g->slots_per_split);
    g->slot_pts = slot_pts;
    // short streams (every tile of a stream of T tiles is slow until ~64 tiles in): share thresholds among a query's
    // lanes, and record tiles instead of single scores; long streams (C3: 16384 tiles, C5: 8192) keep private
    // thresholds and per-score records.  Tile records need K5's one-wave-per-query form (splits >= 4: a tile's rows
    // are evaluated by 16 lanes side by side).
    const int64_t stream_tiles = (int64_t)g->slots_per_split / slots_per_block * (pad_pts / 32);
    // (fp32 points through bf16 operands: the rounding term makes tau ~2^-6 |x'||y'|, thousands of refs lie within it
    //  of the running minimum, and private thresholds fill the 64-entry rings — 5 504 of 65 536 queries overflowed into
    //  the exact scan at k = 1024 — so their lanes always share, whatever the stream length)
    g->share_thr = (stream_tiles <= kShareThrMaxTiles || mixed) ? 1 : 0;
    // record forms: 0 per score (long streams), 1 per (lane, ref tile) behind the threshold test (short streams, 16 x 16
    // bf16 tiles: their epilogue has no vector slots to spare), 2 the lane's two best tiles, branch-free (short streams,
    // 32 x 32 tiles: every fp32 depth and the 768- / 1024-deep bf16-operand tiles)
    g->tile_rec = (stream_tiles <= kTileRecMaxTiles && g->splits >= 4 && !per_ref) ? (g->lpq == 4 ? 1 : 2) : 0;
#ifdef NNS_F_NOTOP2
    if (g->tile_rec == 2) g->tile_rec = 1;
#endif
#ifdef NNS_F_NOSHARE   // (A/B builds)
    g->share_thr = 0;
#endif
#ifdef NNS_F_NOTILEREC
    g->tile_rec = 0;
#endif
    return NNS_OK;
}

template <class OP>
static int launch_filter_t(const FilterGeom &g, const FilterArgs &args, hipStream_t st)
{
    auto kern = filter_kernel<OP>;
    // + 2 KiB per wave for the lanes' tau constants
    constexpr int lds_bytes = F_LDS_BYTES + OP::kNW * ((OP::kTile16 && OP::kQB > 2) ? 4096 : 2048);
    // > 64 KiB of dynamic LDS needs the opt-in, once per device
    static std::atomic<bool> attr_set[64];   // (two threads racing here both set it: harmless)
    int dev = 0;
    NNS_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        NNS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(g.qgroups, g.splits), dim3(OP::kNW * 64), lds_bytes, st, args);
    NNS_HIP(hipGetLastError());
    return NNS_OK;
}

int launch_filter(const FilterGeom &g, const void *qimg, const void *rimg, const float *rnorm,
                  const float *qnorm, const DevScalars *scal, CandEntry *lists, int *counts,
                  hipStream_t st)
{
    FilterArgs a;
    a.qimg = reinterpret_cast<const float4 *>(qimg);
    a.rimg = reinterpret_cast<const char *>(rimg);
    a.rnorm = rnorm;
    a.qnorm = qnorm;
    a.scal = scal;
    a.lists = lists;
    a.counts = counts;
    a.total_slots = g.total_slots;
    a.slots_per_split = g.slots_per_split;
    a.m_pad = g.m_pad;
    a.kt = g.kt;
    a.bf16 = g.mixed ? 2 : g.bf16;
    a.share_thr = g.share_thr;   // (filter_plan)
    a.tile_rec = g.tile_rec;
#ifdef NNS_DIAG
    a.stamps = nullptr;
    const char *clk = getenv("NNS_FILTER_CLOCK");
    const size_t nwg = (size_t)g.qgroups * g.splits;
    if (clk && atoi(clk)) {
        NNS_HIP(hipMalloc(&a.stamps, (nwg * 4 + 8) * sizeof(unsigned long long)));
        NNS_HIP(hipMemsetAsync(a.stamps, 0, (nwg * 4 + 8) * sizeof(unsigned long long), st));
    }
#endif
    const int rc = g.bf16 ? (g.kt == 128    ? launch_filter_t<OpBF16K128>(g, a, st)
                             : g.kt == 512  ? launch_filter_t<OpBF16K512Active>(g, a, st)
                             : g.kt == 1024 ? launch_filter_t<OpBF16K1024>(g, a, st)
                             : g.kt == 768  ? launch_filter_t<OpBF16K768>(g, a, st)
                             : g.kt == 640  ? launch_filter_t<OpBF16K640>(g, a, st)
                             : g.kt == 384  ? launch_filter_t<OpBF16K384>(g, a, st)
                                           : launch_filter_t<OpBF16Active>(g, a, st))
                          : (g.kt == 16    ? launch_filter_t<OpF32K16>(g, a, st)
                             : g.kt == 32  ? launch_filter_t<OpF32K32>(g, a, st)
                             : g.kt == 64  ? launch_filter_t<OpF32K64>(g, a, st)
                             : g.kt == 256 ? launch_filter_t<OpF32K256>(g, a, st)
                                           : launch_filter_t<OpF32>(g, a, st));
#ifdef NNS_DIAG
    if (a.stamps) {   // diagnostic: synchronous read-out, median clock over workgroups
        std::vector<unsigned long long> h(nwg * 4 + 8);
        NNS_HIP(hipStreamSynchronize(st));
        NNS_HIP(hipMemcpy(h.data(), a.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> ghz, us;
        unsigned long long first = ~0ull, last = 0;
        for (size_t w = 0; w < nwg; ++w)
            if (h[4 * w + 3] > h[4 * w + 1]) {
                ghz.push_back((double)(h[4 * w + 2] - h[4 * w]) / (double)(h[4 * w + 3] - h[4 * w + 1]) * 0.1);
                us.push_back((double)(h[4 * w + 3] - h[4 * w + 1]) * 0.01);   // 100 MHz ticks
                first = std::min(first, h[4 * w + 1]);
                last = std::max(last, h[4 * w + 3]);
            }
        std::sort(ghz.begin(), ghz.end());
        std::sort(us.begin(), us.end());
        if (!ghz.empty())
            fprintf(stderr, "[nns] filter in-kernel clock: median %.3f GHz (min %.3f, max %.3f) over %zu workgroups; "
                            "workgroup main loop %.1f us median (%.1f .. %.1f), first start to last end %.1f us\n",
                    ghz[ghz.size() / 2], ghz.front(), ghz.back(), ghz.size(), us[us.size() / 2], us.front(), us.back(),
                    (double)(last - first) * 0.01);
        fprintf(stderr, "[nns] filter slow path: %llu of %llu (tile, state) retirements (fp32 tiles only), %.0f memtime ticks each, share=%d\n",
                h[nwg * 4], h[nwg * 4 + 1], h[nwg * 4] ? (double)h[nwg * 4 + 2] / (double)h[nwg * 4] : 0.0, a.share_thr);
        (void)hipFree(a.stamps);
    }
#endif
    return rc;
}

}  // namespace nns
