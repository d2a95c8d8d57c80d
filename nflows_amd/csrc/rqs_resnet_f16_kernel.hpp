// The kernel of rqs_resnet_f16.hip (K8h) as a header: the template is instantiated in several translation units
// (rqs_resnet_f16.hip: 8 and 10 bins, contexts; rqs_resnet_f16_bins_{a,b,c}.hip: the other bin counts and the other
// block activations, round 4) so that the 200 instances compile side by side.  Design notes: rqs_resnet_f16.hip.
#pragma once

#include "k8h_common.hpp"

namespace nfa {
namespace k8h {

// ---- VALU work woven between the MFMAs of a tile: `step<SLOT>()` runs behind MFMA number SLOT ----
struct NoWeave {
    template <int SLOT>
    __device__ __forceinline__ void step() {}
};

// Conversion of a finished accumulator tile (ReLU'd when RELU, times `scale`, a power of two) into the
// f16 pieces of k-steps 2t and 2t + 1 of the next GEMM, one pair of values per slice, behind every other
// MFMA of the first sixteen.  (Members are references to fixed registers-to-be: one object per tile,
// nothing re-pointed at run time, so that the arrays behind them stay in registers.)
template <int RELU>   // (an activation code: kActNone / kActRelu / ...)
struct ConvWeave {
    const f32x16& src;            // finished tile
    uvec4 &h0, &l0, &h1, &l1;     // pieces of k-steps 2t, 2t + 1
    float scale;
    float& peak;                  // max |value| seen (before the scale)

    template <int J>
    __device__ __forceinline__ void pair() {
        unsigned hi, lo;
        convert_pair<RELU>(src[2 * J], src[2 * J + 1], scale, peak, hi, lo);
        if constexpr (J < 4) {
            h0[J] = hi;
            l0[J] = lo;
        } else {
            h1[J - 4] = hi;
            l1[J - 4] = lo;
        }
    }
    template <int SLOT>
    __device__ __forceinline__ void step() {
        if constexpr (SLOT % 2 == 1 && SLOT < 16) pair<SLOT / 2>();
    }
    __device__ __forceinline__ void all() {   // un-woven (the last tile of a GEMM)
        pair<0>(); pair<1>(); pair<2>(); pair<3>(); pair<4>(); pair<5>(); pair<6>(); pair<7>();
    }
};

// The same conversion cut for the 24 MFMAs of TWO k-steps of a k-major GEMM (12 each): pair J of the
// tile takes slots 3J (ReLU, peak), 3J + 1 (high pieces), 3J + 2 (low pieces): 2-3 VALU instructions
// behind every MFMA.
template <int RELU>   // (an activation code: kActNone / kActRelu / ...)
struct ConvSlices {
    const f32x16& src;
    uvec4 &h0, &l0, &h1, &l1;
    float scale;
    float& peak;
    float v0, v1;
    unsigned hi;

    template <int SLOT>
    __device__ __forceinline__ void step() {
        constexpr int J = SLOT / 3, PH = SLOT % 3;
        if constexpr (PH == 0) {
            if constexpr (RELU == kActRelu) {
                asm("v_max_f32 %0, %3, 0\n\t"
                    "v_max_f32 %1, %4, 0\n\t"
                    "v_max3_f32 %2, %2, %0, %1"
                    : "=&v"(v0), "=&v"(v1), "+v"(peak)
                    : "v"(src[2 * J]), "v"(src[2 * J + 1]));
            } else if constexpr (activation_is_homogeneous(RELU)) {
                v0 = activate<RELU>(src[2 * J]);       // (kActNone: the value itself)
                v1 = activate<RELU>(src[2 * J + 1]);
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(peak) : "v"(v0), "v"(v1));
            } else {   // ELU, tanh: of the value at its own scale (the pieces are then taken with a factor of one)
                v0 = activate<RELU>(src[2 * J] * scale);
                v1 = activate<RELU>(src[2 * J + 1] * scale);
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(peak) : "v"(v0), "v"(v1));
            }
        } else if constexpr (PH == 1) {
            unsigned h;
            const float sc = activation_is_homogeneous(RELU) ? scale : 1.0f;
            asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
                "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
                : "=&v"(h)
                : "v"(v0), "v"(v1), "v"(sc));
            hi = h;
        } else {
            unsigned lo;
            const float sc = activation_is_homogeneous(RELU) ? scale : 1.0f;
            asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
                "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                : "=&v"(lo)
                : "v"(v0), "v"(v1), "v"(sc), "v"(hi));
            if constexpr (J < 4) {
                h0[J] = hi;
                l0[J] = lo;
            } else {
                h1[J - 4] = hi;
                l1[J - 4] = lo;
            }
        }
    }
};

// Preparation of the accumulator of the NEXT tile of a skip-connection GEMM: acc = acc * ratio + bias
// (bias from the layer's parameter block in LDS), four values per slice.
struct InitWeave {
    f32x16& nxt;
    const float* bias;   // its 16 biases (this lane-half's)
    float ratio;
    template <int Q4>
    __device__ __forceinline__ void init4() {
        const vec4f b = reinterpret_cast<const vec4f*>(bias)[Q4];
        nxt[Q4 * 4 + 0] = __builtin_fmaf(nxt[Q4 * 4 + 0], ratio, b.x);
        nxt[Q4 * 4 + 1] = __builtin_fmaf(nxt[Q4 * 4 + 1], ratio, b.y);
        nxt[Q4 * 4 + 2] = __builtin_fmaf(nxt[Q4 * 4 + 2], ratio, b.z);
        nxt[Q4 * 4 + 3] = __builtin_fmaf(nxt[Q4 * 4 + 3], ratio, b.w);
    }
    template <int SLOT>
    __device__ __forceinline__ void step() {
        if constexpr (SLOT % 4 == 2 && SLOT < 18) init4<SLOT / 4>();
    }
    __device__ __forceinline__ void all() { init4<0>(); init4<1>(); init4<2>(); init4<3>(); }
};

template <class A, class B>
struct BothWeaves {
    A& a;
    B& b;
    template <int SLOT>
    __device__ __forceinline__ void step() {
        a.template step<SLOT>();
        b.template step<SLOT>();
    }
};

// ---- the final layer's spline evaluation, three units per group of three tiles ----
enum { kUnitNumA = 1, kUnitFinishA = 2, kUnitFinishB = 3 };

template <int UNIT, class Steps>
constexpr int spline_unit_slices() {
    return UNIT == kUnitNumA ? 2 * Steps::kNumSlices : Steps::kNumSlices + Steps::kFinishSlices;
}

// Slice I of a unit.  U0: width / height numerators of A alternate (two independent chains);
// U1: finish A with the width numerators of B on every third position; U2: height numerators of
// B, then finish B (its single walk needs both numerator sets).
template <int UNIT, int I, class Steps>
__device__ __forceinline__ void spline_unit_slice(Steps& fa, Steps& fb, const RqsDev& sp) {
    constexpr int N = Steps::kNumSlices;
    if constexpr (UNIT == kUnitNumA) {
        if constexpr ((I & 1) == 0) fa.template num_w<(I >> 1)>();
        else fa.template num_h<(I >> 1)>();
    } else if constexpr (UNIT == kUnitFinishA) {
        static_assert(Steps::kFinishSlices >= 2 * N, "one numerator slice behind every two finish slices");
        if constexpr (I % 3 == 2 && I / 3 < N) fb.template num_w<I / 3>();
        else fa.template finish<I - ((I + 1) / 3 < N ? (I + 1) / 3 : N)>(sp);
    } else {
        if constexpr (I < N) fb.template num_h<I>();
        else fb.template finish<I - N>(sp);
    }
}

template <int UNIT, int I, int END, class Steps>
__device__ __forceinline__ void spline_unit_range(Steps& fa, Steps& fb, const RqsDev& sp) {
    if constexpr (I < END) {
        spline_unit_slice<UNIT, I>(fa, fb, sp);
        spline_unit_range<UNIT, I + 1, END>(fa, fb, sp);
    }
}

template <int UNIT, class Steps>
struct SplineWeave {
    Steps &fa, &fb;
    const RqsDev& sp;
    template <int SLOT>
    __device__ __forceinline__ void step() {
        constexpr int N = spline_unit_slices<UNIT, Steps>();
        spline_unit_range<UNIT, (SLOT * N) / kSlots, ((SLOT + 1) * N) / kSlots>(fa, fb, sp);
    }
};

// ---- 10 bins (the reference's default): one feature per lane-half and group of two tiles.  The width
//      numerators run behind the second tile's MFMAs, everything else behind the next group's first tile.
enum { kUnitNumW10 = 4, kUnitRest10 = 5 };

template <int UNIT, int I, int END, class Steps>
__device__ __forceinline__ void spline10_range(Steps& f, const RqsDev& sp) {
    if constexpr (I < END) {
        constexpr int N = Steps::kNumSlices;
        if constexpr (UNIT == kUnitNumW10) f.template num_w<I>();
        else if constexpr (I < N) f.template num_h<I>();
        else f.template finish<I - N>(sp);
        spline10_range<UNIT, I + 1, END>(f, sp);
    }
}

template <int UNIT, class Steps>
struct SplineWeave10 {
    Steps& f;
    const RqsDev& sp;
    static constexpr int kCount = UNIT == kUnitNumW10 ? Steps::kNumSlices : Steps::kNumSlices + Steps::kFinishSlices;
    template <int SLOT>
    __device__ __forceinline__ void step() {
        spline10_range<UNIT, (SLOT * kCount) / kSlots, ((SLOT + 1) * kCount) / kSlots>(f, sp);
    }
};

// ---- any other bin count from 2 to 16 (round 4): one feature per lane-half and group of T = ceil((3 K - 1) / 16) tiles
//      (the lane-half's 16 T accumulator values are the feature's 3 K - 1 logits -- K widths, K heights, K - 1 derivatives
//      -- then padding).  ONE accumulator tile: a finished tile's sixteen values are copied into the evaluation's arrays
//      (`take_chunk`) and the accumulator takes the next tile's biases.  What runs behind a tile's MFMAs only needs
//      logits of EARLIER tiles: the width numerators behind tile 1 (widths: K <= 16 values, all in tile 0), the height
//      numerators behind tile 2 (T = 3) and everything that is left behind tile 0 of the NEXT group.
enum { kSeqW = 1, kSeqH = 2, kSeqFinish = 4 };

template <int MASK, class Steps>
constexpr int spline_seq_count() {
    return ((MASK & kSeqW) ? Steps::kNumSlices : 0) + ((MASK & kSeqH) ? Steps::kNumSlices : 0) +
           ((MASK & kSeqFinish) ? Steps::kFinishSlices : 0);
}

// slices [I, END) of the sequence MASK names: numerators first (width / height alternating when both are in it:
// two independent chains), then the rest of the evaluation
template <int MASK, int I, int END, class Steps>
__device__ __forceinline__ void spline_seq_range(Steps& f, const RqsDev& sp) {
    if constexpr (I < END) {
        constexpr int N = Steps::kNumSlices;
        constexpr bool W = (MASK & kSeqW) != 0, H = (MASK & kSeqH) != 0;
        constexpr int NUM = (W ? N : 0) + (H ? N : 0);
        if constexpr (I < NUM) {
            if constexpr (W && H) {
                if constexpr ((I & 1) == 0) f.template num_w<(I >> 1)>();
                else f.template num_h<(I >> 1)>();
            } else if constexpr (W) {
                f.template num_w<I>();
            } else {
                f.template num_h<I>();
            }
        } else {
            f.template finish<I - NUM>(sp);
        }
        spline_seq_range<MASK, I + 1, END>(f, sp);
    }
}

template <int MASK, class Steps>
struct SplineWeaveSeq {
    Steps& f;
    const RqsDev& sp;
    static constexpr int kCount = spline_seq_count<MASK, Steps>();
    template <int SLOT>
    __device__ __forceinline__ void step() {
        spline_seq_range<MASK, (SLOT * kCount) / kSlots, ((SLOT + 1) * kCount) / kSlots>(f, sp);
    }
};

// values 16 C .. 16 C + 15 of the lane-half's logits, from the tile that has just been finished
template <int C, int KB, class Steps>
__device__ __forceinline__ void take_chunk(Steps& f, const f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = 16 * C + q;
        if (j < KB) f.ew[j < KB ? j : 0] = acc[q];
        else if (j < 2 * KB) f.eh[j < 2 * KB ? j - KB : 0] = acc[q];
        else if (j < 3 * KB - 1) f.sd[j < 3 * KB - 1 ? j - 2 * KB : 0] = acc[q];
    }
}

#ifndef NFA_K8H_ORDER
#define NFA_K8H_ORDER 0   // 1: the srcB-grouped order of the three products of a cell (round-4 experiment: no gain, profiles/r4/k8h_mfma_order.txt)
#endif

// ---- one 32-row output tile: 8 k-steps x 3 products, two stages, a weave slice behind every MFMA
template <int KS, class W, class SM>
__device__ __forceinline__ void tile_kstep(f32x16& acc, uvec4 bhw, uvec4 blw, Frags& fr, unsigned cur, unsigned nxt, W& w, SM& sm) {
    const f16x8 bh = __builtin_bit_cast(f16x8, bhw), bl = __builtin_bit_cast(f16x8, blw);
    if constexpr (KS == kPairs - 1) stream_ensure_next(sm);   // (the next read goes to the next stage)
    const Frags nf = next_frags<KS>(cur, nxt);   // the next k-step's fragments, three MFMAs ahead of their use
    await_frags(fr);
    const f16x8 ah = __builtin_bit_cast(f16x8, fr.h), al = __builtin_bit_cast(f16x8, fr.l);
#ifndef NFA_ABL_CONST_FRAGS
    fr = nf;
#else
    (void)nf;
#endif
    // Order of the three products (round 4): the matrix pipe's energy depends on how often its SECOND operand (srcB: the
    // activation pieces here) CHANGES between consecutive instructions -- tools/mfma_toggle_probe.hip under the power cap:
    // srcB new on every MFMA 1 219 TFLOP/s, on every 4th 1 616, never 1 662; a new srcA (the weights) costs nothing --,
    // and K8h runs at that cap.  The two products on bh are therefore adjacent (bh, bh, bl: two changes per k-step
    // instead of three).
#if NFA_K8H_ORDER == 0   // (round 3: smallest terms first)
    acc = NFA_K8H_MFMA(al, bh, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    NFA_K8H_WEAVE(w.template step<KS * 3 + 0>());
    __builtin_amdgcn_sched_barrier(0);
    acc = NFA_K8H_MFMA(ah, bl, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    NFA_K8H_WEAVE(w.template step<KS * 3 + 1>());
    __builtin_amdgcn_sched_barrier(0);
    acc = NFA_K8H_MFMA(ah, bh, acc, 0, 0, 0);
#else
    acc = NFA_K8H_MFMA(al, bh, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    NFA_K8H_WEAVE(w.template step<KS * 3 + 0>());
    __builtin_amdgcn_sched_barrier(0);
    acc = NFA_K8H_MFMA(ah, bh, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    NFA_K8H_WEAVE(w.template step<KS * 3 + 1>());
    __builtin_amdgcn_sched_barrier(0);
    acc = NFA_K8H_MFMA(ah, bl, acc, 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    NFA_K8H_WEAVE(w.template step<KS * 3 + 2>());
    __builtin_amdgcn_sched_barrier(0);
}

template <class W, class SM>
__device__ __forceinline__ void tile_gemm(f32x16& acc, const uvec4 (&ph)[8], const uvec4 (&pl)[8], SM& sm, Frags& fr,
                                          int lane, W&& w) {
    unsigned cur, nxt;
    stage_begin(sm, cur, nxt, lane);
    tile_kstep<0>(acc, ph[0], pl[0], fr, cur, nxt, w, sm);
    tile_kstep<1>(acc, ph[1], pl[1], fr, cur, nxt, w, sm);
    tile_kstep<2>(acc, ph[2], pl[2], fr, cur, nxt, w, sm);
    tile_kstep<3>(acc, ph[3], pl[3], fr, cur, nxt, w, sm);
    tile_kstep<4>(acc, ph[4], pl[4], fr, cur, nxt, w, sm);
    tile_kstep<5>(acc, ph[5], pl[5], fr, cur, nxt, w, sm);
    tile_kstep<6>(acc, ph[6], pl[6], fr, cur, nxt, w, sm);
    tile_kstep<7>(acc, ph[7], pl[7], fr, cur, nxt, w, sm);
    stream_advance(sm);
}

// two k-steps of a k-major GEMM = one stage (pair g = tile g of the first k-step, 4 + g of the second) with a
// weave slice behind every MFMA (slots 0 .. 23)
template <class W, class SM>
__device__ __forceinline__ void kstep_pair_woven(f32x16 (&acc)[4], uvec4 bh0, uvec4 bl0, uvec4 bh1, uvec4 bl1, SM& sm,
                                                 Frags& fr, int lane, W&& w) {
#define NFA_K8H_CELL(T, G, SLOT, BH, BL)                                                         \
    {                                                                                            \
        if (G == kPairs - 1) stream_ensure_next(sm);                                             \
        const Frags nf = next_frags<G>(cur, nxt);                                                \
        await_frags(fr);                                                                         \
        const f16x8 ah = __builtin_bit_cast(f16x8, fr.h), al = __builtin_bit_cast(f16x8, fr.l);  \
        NFA_K8H_KEEP_FRAGS(fr, nf)                                                               \
        /* srcB order (see tile_kstep): even cells bh, bh, bl -- odd cells bl, bh, bh: the four cells of a k-step  */ \
        /* share their pieces, so srcB changes four times per twelve MFMAs instead of eight                          */ \
        acc[T] = (NFA_K8H_ORDER == 0 || !(G & 1)) ? NFA_K8H_MFMA(al, BH, acc[T], 0, 0, 0)                             \
                                                  : NFA_K8H_MFMA(ah, BL, acc[T], 0, 0, 0);                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        NFA_K8H_WEAVE(w.template step<SLOT + 0>());                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        acc[T] = NFA_K8H_ORDER == 0 ? NFA_K8H_MFMA(ah, BL, acc[T], 0, 0, 0) : NFA_K8H_MFMA(ah, BH, acc[T], 0, 0, 0);    \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        NFA_K8H_WEAVE(w.template step<SLOT + 1>());                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        acc[T] = NFA_K8H_ORDER == 0 ? NFA_K8H_MFMA(ah, BH, acc[T], 0, 0, 0)                                             \
                 : (!(G & 1) ? NFA_K8H_MFMA(ah, BL, acc[T], 0, 0, 0) : NFA_K8H_MFMA(al, BH, acc[T], 0, 0, 0));          \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        NFA_K8H_WEAVE(w.template step<SLOT + 2>());                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
    unsigned cur, nxt;
    stage_begin(sm, cur, nxt, lane);
    {
        const f16x8 bh = __builtin_bit_cast(f16x8, bh0), bl = __builtin_bit_cast(f16x8, bl0);
        NFA_K8H_CELL(0, 0, 0, bh, bl)
        NFA_K8H_CELL(1, 1, 3, bh, bl)
        NFA_K8H_CELL(2, 2, 6, bh, bl)
        NFA_K8H_CELL(3, 3, 9, bh, bl)
    }
    {
        const f16x8 bh = __builtin_bit_cast(f16x8, bh1), bl = __builtin_bit_cast(f16x8, bl1);
        NFA_K8H_CELL(0, 4, 12, bh, bl)
        NFA_K8H_CELL(1, 5, 15, bh, bl)
        NFA_K8H_CELL(2, 6, 18, bh, bl)
        NFA_K8H_CELL(3, 7, 21, bh, bl)
    }
    stream_advance(sm);
#undef NFA_K8H_CELL
}

// k-major 128 -> 128 GEMM whose input pieces are made on the way from the accumulator tiles `src` of the
// previous GEMM (ReLU, x `scale`): tile 0 is converted up front, tile t + 1 behind the MFMAs of k-steps
// 2t, 2t + 1 -- which only read the pieces of tile t.  `worst`: running max of |value x scale| over the
// row block's conversions (the f16-range check).
template <int ACT = kActRelu, class SM>
__device__ __forceinline__ void gemm_kmajor_converting(f32x16 (&acc)[4], uvec4 (&ph)[8], uvec4 (&pl)[8],
                                                       const f32x16 (&src)[4], float scale, float& worst, SM& sm,
                                                       Frags& fr, int lane) {
    float peak = 0.0f;
    ConvWeave<ACT>{src[0], ph[0], pl[0], ph[1], pl[1], scale, peak}.all();
    kstep_pair_woven(acc, ph[0], pl[0], ph[1], pl[1], sm, fr, lane,
                     ConvSlices<ACT>{src[1], ph[2], pl[2], ph[3], pl[3], scale, peak});
    kstep_pair_woven(acc, ph[2], pl[2], ph[3], pl[3], sm, fr, lane,
                     ConvSlices<ACT>{src[2], ph[4], pl[4], ph[5], pl[5], scale, peak});
    kstep_pair_woven(acc, ph[4], pl[4], ph[5], pl[5], sm, fr, lane,
                     ConvSlices<ACT>{src[3], ph[6], pl[6], ph[7], pl[7], scale, peak});
    kstep_pair_woven(acc, ph[6], pl[6], ph[7], pl[7], sm, fr, lane, NoWeave{});
    // (ELU / tanh: `peak` was taken behind the scale)
    worst = __builtin_fmaxf(worst, activation_is_homogeneous(ACT) ? peak * scale : peak);
}

// the initial layer: NKS k-steps (2 or 4) on the pieces of the identity features
template <int NKS, class SM>
__device__ __forceinline__ void gemm_kmajor(f32x16 (&acc)[4], const uvec4 (&ph)[8], const uvec4 (&pl)[8], SM& sm,
                                            Frags& fr, int lane) {
    kstep_pair_woven(acc, ph[0], pl[0], ph[1], pl[1], sm, fr, lane, NoWeave{});
    if constexpr (NKS == 4) kstep_pair_woven(acc, ph[2], pl[2], ph[3], pl[3], sm, fr, lane, NoWeave{});
}

// k-major 128 -> 128 GEMM on finished pieces (the gated block's second Linear)
template <class SM>
__device__ __forceinline__ void gemm_kmajor_full(f32x16 (&acc)[4], const uvec4 (&ph)[8], const uvec4 (&pl)[8], SM& sm,
                                                 Frags& fr, int lane) {
    kstep_pair_woven(acc, ph[0], pl[0], ph[1], pl[1], sm, fr, lane, NoWeave{});
    kstep_pair_woven(acc, ph[2], pl[2], ph[3], pl[3], sm, fr, lane, NoWeave{});
    kstep_pair_woven(acc, ph[4], pl[4], ph[5], pl[5], sm, fr, lane, NoWeave{});
    kstep_pair_woven(acc, ph[6], pl[6], ph[7], pl[7], sm, fr, lane, NoWeave{});
}

// the gate of a block with a context: hacc = hacc * ratio + v * sigmoid(g * inv_t), tile by tile.  Sigmoid on
// v_exp_f32 / v_rcp_f32 with one residual correction of the reciprocal; the exponent is capped so that 1 + 2^t
// stays finite (sigmoid < 2^-126 there); NaN propagates.
__device__ __forceinline__ void gate_tile(f32x16& hacc, const f32x16& v, const f32x16& g, float ratio, float inv_t) {
    const float c = -1.44269502162933349609375f * inv_t;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float t = g[q] * c;
        t = t > 126.0f ? 126.0f : t;   // (a comparison, not fminf: NaN stays NaN)
        const float e2 = __builtin_amdgcn_exp2f(t);
        const float dn = 1.0f + e2;
        const float r0 = __builtin_amdgcn_rcpf(dn);
        const float sg = __builtin_fmaf(__builtin_fmaf(-dn, r0, 1.0f), r0, r0);
        hacc[q] = __builtin_fmaf(hacc[q], ratio, v[q] * sg);
    }
}

__device__ __forceinline__ void load_bias_tile(f32x16& acc, const float* bias_tile_half) {
    const vec4f* bp = reinterpret_cast<const vec4f*>(bias_tile_half);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const vec4f b = bp[q4];
        acc[q4 * 4 + 0] = b.x;
        acc[q4 * 4 + 1] = b.y;
        acc[q4 * 4 + 2] = b.z;
        acc[q4 * 4 + 3] = b.w;
    }
}

__device__ __forceinline__ bool not_finite(float v) { return !(__builtin_fabsf(v) < INFINITY); }

// which part of the evaluation runs behind tile TI (>= 1) of a group, and what is left for the next group's first tile.
// Up to 16 bins all widths sit in tile 0: their numerators behind tile 1, the heights' behind tile 2 (T = 3) or with the
// rest (T <= 2).  More than 16 bins (round 4: 20, 24, 32 -- T = 4 .. 6): a numerator set runs behind the tile that
// follows the one completing it, tw = (K - 1) / 16 for the widths, th = (2 K - 1) / 16 for the heights.
constexpr int any_tile_mask(int KB, int TI) {
    if (KB <= 16) return TI == 1 ? kSeqW : TI == 2 ? kSeqH : 0;
    return (((KB - 1) / 16 == TI - 1) ? kSeqW : 0) | (((2 * KB - 1) / 16 == TI - 1) ? kSeqH : 0);
}
constexpr int any_rest_mask(int KB) {
    const int T = (3 * KB - 1 + 15) / 16;
    if (KB <= 16) return T == 1 ? (kSeqW | kSeqH | kSeqFinish) : T == 2 ? (kSeqH | kSeqFinish) : kSeqFinish;
    return (((KB - 1) / 16 == T - 1) ? kSeqW : 0) | (((2 * KB - 1) / 16 == T - 1) ? kSeqH : 0) | kSeqFinish;
}

// tiles TI .. T - 1 of a group: biases, the tile's 24 MFMAs with their share of the evaluation, the tile's values
template <int TI, int T, int KB, class Steps, class SM>
__device__ __forceinline__ void any_group_tiles(f32x16& acc, Steps& f, const float* gb, const uvec4 (&ph)[8],
                                                const uvec4 (&pl)[8], SM& sm, Frags& fr, int lane, const RqsDev& sp) {
    if constexpr (TI < T) {
        constexpr int kMask = any_tile_mask(KB, TI);
        load_bias_tile(acc, gb + TI * 32);
        if constexpr (kMask != 0) tile_gemm(acc, ph, pl, sm, fr, lane, SplineWeaveSeq<kMask, Steps>{f, sp});
        else tile_gemm(acc, ph, pl, sm, fr, lane, NoWeave{});
        take_chunk<TI, KB>(f, acc);
        any_group_tiles<TI + 1, T, KB, Steps>(acc, f, gb, ph, pl, sm, fr, lane, sp);
    }
}

// CTX: conditioners with a context.  Its ce <= 32 columns are the initial layer's LAST two k-steps (the host
// packs the weight as [identity features, zero-padded to 32 | context, zero-padded to 32]: INIT_KS = 4,
// d_i <= 32), kept as f16 pieces in registers for the whole run; every block ends with the gate
// h + (W_1 relu(u) + b_1) * sigmoid(W_c context + b_c): the second Linear then has accumulators of its own
// (its input pieces are finished first: u's registers are needed), the gate's Linear is one more stage
// (two k-steps, k-major) and the residual stream takes the product in.
// DBG (rqs_resnet_f16_dbg.hip): the same kernel with the last layer's chosen bins stored to a.dbg_bins (FusedSteps' kbin).
template <bool INVERSE, int INIT_KS, int NW, int KB = 8, bool CTX = false, int RING = kRing, int ACT = kActRelu, bool DBG = false>
__global__ void __launch_bounds__(NW * kWave, 2) rqs_resnet_f16_kernel(const Args a) {
    static_assert(!DBG || KB == 8, "the diagnostic instances: 8 bins");
    static_assert(!CTX || INIT_KS == 4, "context: two identity k-steps + two context k-steps");
    static_assert(ACT == kActRelu || (ACT >= kActLeakyRelu && ACT <= kActTanh), "the blocks' activation");
    constexpr int kThreads = NW * kWave;
    // dynamic LDS: the weight ring, per wave a [D][33] row tile, two parameter blocks (current / next layer)
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    __shared__ int s_final[128];
    __shared__ int s_bad[NW];
    __shared__ unsigned s_sync[8];   // elastic stream: per-slot counters (WeightStream)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, dt = a.dt;
    int my_status = 0;
    if (tid < 128) {
        const int v = a.final_tab[tid];
        if (tid < D && (v < 0 || v >= D)) my_status |= NFA_STATUS_BAD_INDEX;
        s_final[tid] = v < 0 ? 0 : (v >= D ? D - 1 : v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no ordinary load in flight once the stream starts)

    using Stream = WeightStream<NW, RING>;
    Stream sm;
    sm.w = a.w;
    sm.ring = reinterpret_cast<vec4f*>(lds_dyn);
    sm.fetch = 0;
    sm.num_stages = a.num_stages * a.num_layers;
    sm.tid = tid;
    sm.sync = lds_address(s_sync);
    sm.gen = NW;
    sm.peek = 0;
    // stages 0 .. 2 -> slots 0 .. 2 (three stages in flight in both forms of the ring)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        sm.slot = ring_next<Stream>(j, Stream::ELASTIC ? 2 : 1);   // (stream_request targets slot - 2 / slot - 1)
        stream_request(sm);
    }
    sm.slot = 0;
    // (elastic: stages 0 and 1 are complete after the barrier below and never get ticks: start at NW)
    if (tid < 8) s_sync[tid] = tid < 2 ? NW : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frags fr;   // the weight fragments the next MFMAs need (carried across stages, layers and row blocks)
    fr.h = sm.ring[lane];
    fr.l = sm.ring[64 + lane];

    // a layer's parameter words live in one of two blocks of `pblock` floats (the words actually used)
    const int pblock = (a.param_words + 3) & ~3;
    float* s_row = lds_dyn + RING * kStageVec4 * 4 + wave * D * kRowPad;
    float* s_param = lds_dyn + RING * kStageVec4 * 4 + NW * D * kRowPad;   // [2][pblock]
    const int groups = dt >> 2;
    const int64_t num_quads = a.batch / (32 * NW);   // row blocks of this workgroup size
    int pb = 0;  // which parameter block the current layer uses

    const bool tracing = a.trace != nullptr && __builtin_amdgcn_readfirstlane(wave) == 0;
    int ti = 1;
    if (tracing) {
        if (lane == 0)
            a.trace[(size_t)blockIdx.x * 64] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID, XCC_ID
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int64_t quad = blockIdx.x; quad < num_quads; quad += gridDim.x) {
        const int64_t row0 = quad * (32 * NW) + (wave << 5);
        // (lane-derived values are made opaque per iteration: hoisted out of this loop they would
        // stay live through the whole kernel and push the register allocation into scratch)
        int lane_here = lane, di = a.di;
        asm volatile("" : "+v"(lane_here), "+s"(di));
        const int half = lane_here >> 5, r = lane_here & 31;
        // ---- the wave's 32 rows: one coalesced read; slot j of the tile = input column j.  (The
        //      stream is drained at the end of every row block: these loads are alone in flight.)
        {
            const vec4f* xv = reinterpret_cast<const vec4f*>(a.x + row0 * D);
            const int nvec = D * 8;
            for (int e0 = lane; e0 < nvec; e0 += kWave * 4) {
                vec4f v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kWave;
                    v[u] = xv[e < nvec ? e : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * kWave;
                    if (e < nvec) {
                        const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                        s_row[(c0 + 0) * kRowPad + rr] = v[u].x;
                        s_row[(c0 + 1) * kRowPad + rr] = v[u].y;
                        s_row[(c0 + 2) * kRowPad + rr] = v[u].z;
                        s_row[(c0 + 3) * kRowPad + rr] = v[u].w;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        [[maybe_unused]] uvec4 cph[2], cpl[2];   // CTX: this lane's context values as f16 pieces, k = ks*16 + half*8 + j
        if constexpr (CTX) {
            const float* crow = a.ctx + (row0 + r) * a.ce;
            float cv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = (i >> 3) * 16 + half * 8 + (i & 7);
                cv[i] = crow[c < a.ce ? c : 0];
            }
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int c = (i >> 3) * 16 + half * 8 + (i & 7);
                unsigned hi, lo;
                split2(c < a.ce ? cv[i] : 0.0f, c + 1 < a.ce ? cv[i + 1] : 0.0f, hi, lo);
                cph[i >> 3][(i & 7) >> 1] = hi;
                cpl[i >> 3][(i & 7) >> 1] = lo;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        float lad_acc = 0.0f;
        float worst = 0.0f;   // max |activation x scale| handed to an f16 conversion in this row block
        int quad_status = 0;
        for (int layer = 0; layer < a.num_layers; ++layer) {
            // (the two waves of a SIMD alternate the higher issue priority layer by layer)
            if ((layer + (NW == 8 ? (wave >> 2) : (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0))) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            NFA_HSTAMP()
            // ---- the layer's parameter stage(s): ring -> parameter block `pb` (table entries clamped
            //      and checked on the way).  Nobody reads block pb any more: its previous user was the
            //      layer before the last, a whole layer of stage barriers ago.
            float* prm = s_param + pb * pblock;
            for (int p = 0; p < a.param_stages; ++p) {
                unsigned cur, nxt;
                stage_begin(sm, cur, nxt, lane);
                const vec4f* src = sm.ring + sm.slot * kStageVec4;
                vec4f* dst = reinterpret_cast<vec4f*>(prm) + p * kParamVec4;
                const int used = (pblock >> 2) - p * kParamVec4;   // vec4s of this stage that carry words
                for (int i = tid; i < (used < kParamVec4 ? used : kParamVec4); i += kThreads) {
                    vec4f v = src[i];
                    if (p == 0 && i < kTabWords / 4) {
                        // (whole-vector bit casts: a bit cast of a single vector ELEMENT reads element 0)
                        uvec4 u = __builtin_bit_cast(uvec4, v);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int idx = i * 4 + c;
                            const int e = (int)u[c];
                            const bool used_entry = idx < kTabTr ? idx < a.di : idx - kTabTr < dt;
                            if (used_entry && (e < 0 || e >= D)) my_status |= NFA_STATUS_BAD_INDEX;
                            u[c] = (unsigned)(e < 0 ? 0 : (e >= D ? D - 1 : e));
                        }
                        v = __builtin_bit_cast(vec4f, u);
                    }
                    dst[i] = v;
                }
                stream_ensure_next(sm);
#ifdef NFA_ABL_CONST_FRAGS
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(fr.h), "=v"(fr.l) : "v"(nxt));   // (early-clobber: see next_frags)
#else
                fr = next_frags<kPairs - 1>(cur, nxt);   // pair 0 of the stage behind this one (a weight stage after the last p)
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr.h), "+v"(fr.l));
                stream_advance(sm, true);   // (every wave reads the parameter block all threads have just written)
            }
            const int* tab = reinterpret_cast<const int*>(prm);
            const float* gemm = prm + kTabWords;   // header + biases of the next GEMM
            pb ^= 1;

            uvec4 ph[8], pl[8];   // the current activations (128 k per sample) as f16 pieces (8 per register quad)
            f32x16 hacc[4];       // the residual stream h in fp32 (x the scale of the GEMM that wrote it)

            // ---- identity features (scale 1): k = ks*16 + half*8 + j
#pragma unroll
            for (int ks = 0; ks < (CTX ? INIT_KS - 2 : INIT_KS); ++ks) {
                uvec4 hw, lw;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const int i0 = ks * 16 + half * 8 + j2 * 2;
                    float v0 = s_row[tab[kTabId + i0] * kRowPad + r], v1 = s_row[tab[kTabId + i0 + 1] * kRowPad + r];
                    v0 = i0 < di ? v0 : 0.0f;
                    v1 = i0 + 1 < di ? v1 : 0.0f;
                    unsigned hi, lo;
                    split2(v0, v1, hi, lo);
                    hw[j2] = hi;
                    lw[j2] = lo;
                }
                ph[ks] = hw;
                pl[ks] = lw;
            }
            if constexpr (CTX) {   // input of the initial layer = [identity features | context] (resnet.py:93-94)
                ph[INIT_KS - 2] = cph[0];
                pl[INIT_KS - 2] = cpl[0];
                ph[INIT_KS - 1] = cph[1];
                pl[INIT_KS - 1] = cpl[1];
            }

            // ---- initial layer (k-major: one stage of four tile pairs per k-step)
            {
                const float* bias = gemm + kHdr + half * 16;
#pragma unroll
                for (int t = 0; t < 4; ++t) load_bias_tile(hacc[t], bias + t * 32);
                gemm_kmajor<INIT_KS>(hacc, ph, pl, sm, fr, lane);
            }
            // (the pieces of a GEMM's result are made by the GEMM that consumes them, behind its MFMAs)
            float conv_scale = gemm[0];
            gemm += kHdr + 128;
            NFA_HSTAMP()

            // ---- residual blocks: h += W_1 relu(W_0 relu(h) + b_0) + b_1
            for (int blk = 0; blk < a.num_blocks; ++blk) {
                uvec4 qh[8], ql[8];   // pieces of relu(u)
                f32x16 u[4];
                {
                    // first Linear on the pieces of relu(h)
                    const float* bias = gemm + kHdr + half * 16;
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                    gemm_kmajor_converting<ACT>(u, ph, pl, hacc, conv_scale, worst, sm, fr, lane);
                    conv_scale = gemm[0];
                }
                gemm += kHdr + 128;
                NFA_HSTAMP()
                if constexpr (CTX) {
                    // temps = W_1 relu(u) + b_1 in accumulators of its own (u's registers: its pieces first) ...
                    {
                        float peak = 0.0f;
                        ConvWeave<ACT>{u[0], qh[0], ql[0], qh[1], ql[1], conv_scale, peak}.all();
                        ConvWeave<ACT>{u[1], qh[2], ql[2], qh[3], ql[3], conv_scale, peak}.all();
                        ConvWeave<ACT>{u[2], qh[4], ql[4], qh[5], ql[5], conv_scale, peak}.all();
                        ConvWeave<ACT>{u[3], qh[6], ql[6], qh[7], ql[7], conv_scale, peak}.all();
                        // (ELU / tanh: `peak` was taken behind the scale, see gemm_kmajor_converting)
                        worst = __builtin_fmaxf(worst, activation_is_homogeneous(ACT) ? peak * conv_scale : peak);
                    }
                    const float* bias = gemm + kHdr + half * 16;
                    const float ratio = gemm[1];
                    const float next_scale = gemm[0];
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(u[t], bias + t * 32);
                    gemm_kmajor_full(u, qh, ql, sm, fr, lane);
                    gemm += kHdr + 128;
                    NFA_HSTAMP()
                    // ... the gate's Linear on the context pieces (one stage), and h = h * ratio + temps * sigmoid(gate)
                    // (resnet.py:46-52: F.glu of the concatenation)
                    f32x16 g[4];
                    const float* gbias = gemm + kHdr + half * 16;
                    const float inv_t = gemm[0];
#pragma unroll
                    for (int t = 0; t < 4; ++t) load_bias_tile(g[t], gbias + t * 32);
                    kstep_pair_woven(g, cph[0], cpl[0], cph[1], cpl[1], sm, fr, lane, NoWeave{});
#pragma unroll
                    for (int t = 0; t < 4; ++t) gate_tile(hacc[t], u[t], g[t], ratio, inv_t);
                    conv_scale = next_scale;
                } else {
                    // second Linear accumulates into the residual stream itself: hacc = hacc * ratio + bias
                    // (the skip connection), then + W_1 relu(u)
                    const float* bias = gemm + kHdr + half * 16;
                    const float ratio = gemm[1];
                    InitWeave{hacc[0], bias + 0 * 32, ratio}.all();
                    InitWeave{hacc[1], bias + 1 * 32, ratio}.all();
                    InitWeave{hacc[2], bias + 2 * 32, ratio}.all();
                    InitWeave{hacc[3], bias + 3 * 32, ratio}.all();
                    gemm_kmajor_converting<ACT>(hacc, qh, ql, u, conv_scale, worst, sm, fr, lane);
                    conv_scale = gemm[0];
                }
                gemm += kHdr + 128;
                NFA_HSTAMP()
            }
            // pieces of h itself for the final layer (no ReLU in front of it: resnet.py:99-100)
            {
                float peak = 0.0f;
                ConvWeave<false>{hacc[0], ph[0], pl[0], ph[1], pl[1], conv_scale, peak}.all();
                ConvWeave<false>{hacc[1], ph[2], pl[2], ph[3], pl[3], conv_scale, peak}.all();
                ConvWeave<false>{hacc[2], ph[4], pl[4], ph[5], pl[5], conv_scale, peak}.all();
                ConvWeave<false>{hacc[3], ph[6], pl[6], ph[7], pl[7], conv_scale, peak}.all();
                worst = __builtin_fmaxf(worst, peak * conv_scale);
            }

            // ---- final layer with the spline evaluation woven into the MFMAs: the three tiles of a group
            //      hold the logits of this lane's two features A, B (A = T0 + T1[0:8], B = T1[8:16] + T2)
            if constexpr (KB == 10) {
                // 29 logits per feature padded to 32 rows = the 16 + 16 accumulator values a lane-half gets
                // from the two tiles of a group: [10 widths, 6 heights | 4 heights, 9 derivatives, 3 pads]
                using Steps = FusedSteps<INVERSE, 10>;
                Steps f;
                const float kappa = gemm[0];
                f.kappa = kappa;
                f.kl2e = 1.44269502162933349609375f * kappa;
                f.tail_s = a.sp.tail_logit * gemm[1];
                const float* fbias = gemm + kHdr + half * 16;
                const int groups10 = dt >> 1;
                f32x16 acc0, acc1;
                float hrest[6];
                float* slot = s_row + tab[kTabTr + half] * kRowPad + r;
                SplineWeave10<kUnitNumW10, Steps> wn{f, a.sp};
                SplineWeave10<kUnitRest10, Steps> wr{f, a.sp};
                load_bias_tile(acc0, fbias);
                tile_gemm(acc0, ph, pl, sm, fr, lane, NoWeave{});
                for (int g = 0; g < groups10; ++g) {
                    f.x = *slot;
#pragma unroll
                    for (int j = 0; j < 10; ++j) f.ew[j] = acc0[j];
#pragma unroll
                    for (int j = 0; j < 6; ++j) hrest[j] = acc0[10 + j];
                    load_bias_tile(acc1, fbias + (g * 2 + 1) * 32);
                    tile_gemm(acc1, ph, pl, sm, fr, lane, wn);
#pragma unroll
                    for (int j = 0; j < 6; ++j) f.eh[j] = hrest[j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f.eh[6 + j] = acc1[j];
#pragma unroll
                    for (int j = 0; j < 9; ++j) f.sd[j] = acc1[4 + j];
                    if (g + 1 < groups10) {
                        float* next_slot = s_row + tab[kTabTr + (g + 1) * 2 + half] * kRowPad + r;
                        load_bias_tile(acc0, fbias + (g * 2 + 2) * 32);
                        tile_gemm(acc0, ph, pl, sm, fr, lane, wr);
                        *slot = f.y;
                        slot = next_slot;
                    } else {
                        spline10_range<kUnitRest10, 0, SplineWeave10<kUnitRest10, Steps>::kCount>(f, a.sp);
                        *slot = f.y;
                    }
                    lad_acc += f.lad;
                    quad_status |= f.status;
                }
            } else if constexpr (KB != 8) {
                // other bin counts: T tiles per group of two features (one per lane-half), see SplineWeaveSeq
                constexpr int T = (3 * KB - 1 + 15) / 16;
                static_assert(T >= 1 && T <= 6, "2 .. 32 bins");
                using Steps = FusedSteps<INVERSE, KB>;
                // what is left for the next group's first tile: everything (T = 1), heights + rest (T = 2), the rest (T >= 3)
                constexpr int kRest = any_rest_mask(KB);
                Steps f;
                const float kappa = gemm[0];
                f.kappa = kappa;
                f.kl2e = 1.44269502162933349609375f * kappa;
                f.tail_s = a.sp.tail_logit * gemm[1];
                const float* fbias = gemm + kHdr + half * 16;
                const int groups_any = dt >> 1;
                f32x16 acc;
                float* slot = s_row + tab[kTabTr + half] * kRowPad + r;
                load_bias_tile(acc, fbias);
                tile_gemm(acc, ph, pl, sm, fr, lane, NoWeave{});
                for (int g = 0; g < groups_any; ++g) {
                    const float* gb = fbias + g * T * 32;
                    f.x = *slot;
                    take_chunk<0, KB>(f, acc);
                    any_group_tiles<1, T, KB, Steps>(acc, f, gb, ph, pl, sm, fr, lane, a.sp);
                    if (g + 1 < groups_any) {
                        float* next_slot = s_row + tab[kTabTr + (g + 1) * 2 + half] * kRowPad + r;
                        load_bias_tile(acc, gb + T * 32);
                        tile_gemm(acc, ph, pl, sm, fr, lane, SplineWeaveSeq<kRest, Steps>{f, a.sp});
                        *slot = f.y;
                        slot = next_slot;
                    } else {
                        spline_seq_range<kRest, 0, spline_seq_count<kRest, Steps>()>(f, a.sp);
                        *slot = f.y;
                    }
                    lad_acc += f.lad;
                    quad_status |= f.status;
                }
            } else {
                using Steps = FusedSteps<INVERSE, 8, DBG>;
                Steps fa, fb;
                const float kappa = gemm[0];
                fa.kappa = fb.kappa = kappa;
                fa.kl2e = fb.kl2e = 1.44269502162933349609375f * kappa;
                fa.tail_s = fb.tail_s = a.sp.tail_logit * gemm[1];  // gemm[1] = 1 / kappa
                float* slot_b = nullptr;
                const float* fbias = gemm + kHdr + half * 16;
                f32x16 acc[3];
                auto commit = [&](Steps& f, float* slot, [[maybe_unused]] int feature) {
                    *slot = f.y;
                    lad_acc += f.lad;
                    quad_status |= f.status;
                    if constexpr (DBG) {
                        if (layer == a.num_layers - 1) a.dbg_bins[(row0 + r) * dt + feature] = f.kbin;
                    }
                };
                // DBG: the tile's sixteen logits of this lane (accumulators x kappa) of the run's last layer
                [[maybe_unused]] auto store_logits = [&](const f32x16& t, int tile) {
                    if constexpr (DBG) {
                        if (a.dbg_logits != nullptr && layer == a.num_layers - 1) {
                            float* dst = a.dbg_logits + (size_t)(row0 + r) * (dt * 24) + tile * 32 + half * 16;
#pragma unroll
                            for (int q_ = 0; q_ < 16; ++q_) dst[q_] = t[q_] * kappa;
                            // (stores share vmcnt with the LDS-DMA requests and complete out of order with them: the
                            //  stream's counted waits must not see them)
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                };
                SplineWeave<kUnitNumA, Steps> w0{fa, fb, a.sp};
                SplineWeave<kUnitFinishA, Steps> w1{fa, fb, a.sp};
                SplineWeave<kUnitFinishB, Steps> w2{fa, fb, a.sp};
                for (int g = 0; g < groups; ++g) {
                    float* slot0 = s_row + tab[kTabTr + g * 4 + half * 2] * kRowPad + r;
                    float* slot1 = s_row + tab[kTabTr + g * 4 + half * 2 + 1] * kRowPad + r;
                    load_bias_tile(acc[0], fbias + (g * 3 + 0) * 32);
                    if (g > 0) {
                        tile_gemm(acc[0], ph, pl, sm, fr, lane, w2);
                        commit(fb, slot_b, (g - 1) * 4 + half * 2 + 1);
                    } else {
                        tile_gemm(acc[0], ph, pl, sm, fr, lane, NoWeave{});
                    }
                    store_logits(acc[0], g * 3 + 0);
                    fa.x = *slot0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                    }
                    load_bias_tile(acc[1], fbias + (g * 3 + 1) * 32);
                    tile_gemm(acc[1], ph, pl, sm, fr, lane, w0);
                    store_logits(acc[1], g * 3 + 1);
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                    }
                    load_bias_tile(acc[2], fbias + (g * 3 + 2) * 32);
                    tile_gemm(acc[2], ph, pl, sm, fr, lane, w1);
                    store_logits(acc[2], g * 3 + 2);
                    commit(fa, slot0, g * 4 + half * 2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fb.eh[j] = acc[2][j];
                        if (j < 7) fb.sd[j] = acc[2][8 + j];
                    }
                    slot_b = slot1;
                }
                spline_unit_range<kUnitFinishB, 0, spline_unit_slices<kUnitFinishB, Steps>()>(fa, fb, a.sp);
                commit(fb, slot_b, (groups - 1) * 4 + half * 2 + 1);
            }
            NFA_HSTAMP()
            // this wave's spline results must be visible to its own gathers of the next layer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- results: position p of a row comes from slot final[p].  A block with any non-finite
        //      value (f16 range exceeded somewhere, or non-finite inputs) is not written at all:
        //      the exact kernel redoes it from the inputs.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the stream: ordinary stores / loads follow
        lad_acc += __shfl_xor(lad_acc, 32, kWave);
        // sum_j z_j^2 of every row: the standard-normal epilogue needs it, and it is non-finite exactly
        // when one of the row's values is (or a square overflows: such a block is redone like the others)
        const float sumsq = tile_row_sumsq(s_row, a.Ds, half, r);
        const bool bad = not_finite(lad_acc) || not_finite(sumsq) || !(worst < kF16Overflow);
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) s_bad[wave] = wave_bad ? 1 : 0;
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) any_bad |= s_bad[w_];
        const bool quad_bad = any_bad != 0;
        if (!quad_bad) {
            if (!a.skip_out) {
                const int nvec = D * 8;
                vec4f* ov = reinterpret_cast<vec4f*>(a.out + row0 * D);
                for (int e = lane; e < nvec; e += kWave) {
                    const int rr = (e * 4) / D, c0 = e * 4 - rr * D;
                    vec4f v;
                    v.x = s_row[s_final[c0 + 0] * kRowPad + rr];
                    v.y = s_row[s_final[c0 + 1] * kRowPad + rr];
                    v.z = s_row[s_final[c0 + 2] * kRowPad + rr];
                    v.w = s_row[s_final[c0 + 3] * kRowPad + rr];
                    ov[e] = v;
                }
            }
            if (half == 0) {
                float* dst = a.lad + row0 + r;
                float v = a.accumulate ? *dst + lad_acc : lad_acc;
                if (a.normal) v = (-0.5f * sumsq - a.log_z) + v;   // normal.py:31-33, flows/base.py:49
                *dst = v;
            }
            my_status |= quad_status;
        }
        if (tid < NW / 4) a.redo[quad * (NW / 4) + tid] = quad_bad ? 1 : 0;   // one flag per 128 rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // s_bad is rewritten by the next row block
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace k8h
}  // namespace nfa

namespace nfa {
namespace k8h {
typedef void (*KernelFn)(const Args);
// the instances of the other translation units: nullptr when the unit does not hold the combination
KernelFn bins_kernel_a(int K, bool inverse, int init_ks, int waves);     // 2 .. 7, 9 bins
KernelFn bins_kernel_b(int K, bool inverse, int init_ks, int waves);     // 11 .. 16 bins
KernelFn bins_kernel_c(int K, bool inverse, int init_ks, int waves);     // 20, 24, 32 bins
KernelFn activation_kernel(int activation, int K, bool inverse, int init_ks, int waves);   // NFA_ACTIVATION_* > 0, 8 / 10 bins
// conditioners with a context (init_ks = 4) beyond the two tuned bin counts with ReLU (round 5): ReLU with 2 .. 7, 9,
// 11 .. 16, 20, 24, 32 bins, the other activations with 8 / 10 bins
KernelFn context_kernel_a(int K, int activation, bool inverse, int waves);   // ReLU, 2 .. 7, 9, 11, 12 bins
KernelFn context_kernel_b(int K, int activation, bool inverse, int waves);   // ReLU, 13 .. 16, 20, 24, 32 bins; leaky ReLU / ELU / tanh, 8 / 10 bins
KernelFn debug_kernel(bool inverse, int init_ks, int waves);             // 8 bins, ReLU, no context: the DBG instances
}  // namespace k8h
}  // namespace nfa
