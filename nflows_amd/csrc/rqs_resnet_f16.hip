// K8h: the whole-layer kernel (ResidualNet conditioner, nn/nets/resnet.py:55-100, + everything K1
// replaces, coupling.py:73-130, :549-582, for a run of layers in one launch) with its GEMMs on the
// f16 matrix pipe from TWO pieces per fp32 operand.
//
//   x = hi + lo,  hi = RN16(x),  lo = RN16(x - hi)        |x - hi - lo| <= 2^-24 |x|
//   x * w ~= hi_x hi_w + hi_x lo_w + lo_x hi_w            dropped: lo_x lo_w <= 2^-24 |x w|
//
// Three v_mfma_f32_32x32x16_f16 per k-step and tile instead of the six bf16 products of the
// three-piece scheme (rqs_resnet.hip): half the matrix-pipe time, 4 bytes per weight instead of 6,
// a piece conversion of ~8 instead of 11+ VALU instructions per pair.  Measured on the GPU against
// float64 (tools/f16x2_probe.hip, K = 128): max 5.9e-7 / rms 5.2e-8 -- the figures of a sequential
// fp32 fma chain (5.2e-7 / 5.2e-8), better than the six-product bf16 scheme (6.6e-7 / 5.8e-8).
//
// What f16 needs that bf16 did not:
//   * range of the LOW pieces: a low piece below 2^-14 is subnormal (gfx950's f16 MFMA honours
//     subnormals, checked by the probe) and keeps only absolute precision 2^-25.  Weights are
//     therefore pre-scaled per GEMM by a power of two T (host: max |w T| in [2^13, 2^14)); the scale
//     comes back out when accumulators are converted to the next layer's pieces (a product with a
//     power of two: exact) or, for the final layer, inside the spline evaluation.  Activations stay
//     at scale S (1 by default): an activation below 0.25 / S carries an absolute error <= 2^-25 / S,
//     ~3e-8 absolute on a layer output (probe) -- invisible next to the fp32 rounding of the sum.
//   * range of the HIGH pieces: |activation| * S > 65504 overflows.  Every row block checks its
//     results: a block with any non-finite output writes nothing and raises its entries of `redo`;
//     the caller then runs the exact kernel (nfa_rqs_flow_resnet_redo_f32, three bf16 pieces: full
//     fp32 range) on the flagged blocks.  Overflow always poisons: an f16 infinity enters the
//     products, its low piece is x - inf = -inf, and inf - inf = NaN reaches every logit that depends
//     on it.  Rows with NaN / inf INPUTS take the same route (the reference's propagation rules).
//
// Structure (32 samples per wave, rows in an LDS tile by slot, GEMMs transposed and chained through
// the register file as in rqs_resnet.hip), and what is different here:
//   * ONE stream of 16 KB stages per layer feeds everything through LDS-DMA: first the layer's
//     PARAMETER stage(s) -- column tables, per-GEMM headers {out_scale, skip_scale}, all biases --
//     then the weights.  Inside the layer loop a wave issues no global load other than its LDS-DMA
//     requests: a `s_waitcnt vmcnt(n)` of the compiler for an ordinary load counts on in-order
//     return, which LDS-DMA requests sharing the counter do not give it.  (The stream is drained
//     before the ordinary loads / stores at the two ends of a row block.)
//   * The ring is deep (four 16 KB slots, three stages in flight) and shared by EIGHT waves (one workgroup
//     per CU) where the batch allows: a request lands 1-2 us after it was issued, so bytes per
//     second = bytes in flight / latency; eight waves per stream also halve the bytes per CU.
//   * Weight fragments are read from LDS one MFMA group ahead by asm reads with a counted lgkmcnt
//     (hipcc waits with lgkmcnt(0), i.e. also for the reads just issued for the next group).
//   * The final layer is tile-major (a 32-row output tile = 24 MFMAs = two stages) with the spline
//     evaluation woven between its MFMAs, one slice behind each MFMA (~5 independent VALU
//     instructions behind an MFMA are free, tools/weave_probe.hip); the hidden Linears are k-major
//     (four accumulators, every input piece read once).
//   * The residual stream h lives in fp32 accumulator registers across a block (hacc): the block's
//     second Linear accumulates straight into it (skip connection = one fma per value when the
//     accumulator is prepared), ReLU is applied when a tile is converted into pieces, not per k-step.
//   * No packed fp32 arithmetic (see split2).
//
// Restrictions: 2 .. 16 bins (8 and 10: tuned final-layer loops and conditioners with a context), linear tails, hidden width 128 (narrower: zero-padded by the host), ReLU
// blocks, d_i <= 64, d_t % 4 == 0, d_t <= 64, D % 4 == 0, D <= 128, batch % 128 == 0 (other feature counts
// and batches: padded by the host); with a context: up to 32 context features beside d_i <= 32.



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

// CTX: conditioners with a context.  Its ce <= 32 columns are the initial layer's LAST two k-steps (the host
// packs the weight as [identity features, zero-padded to 32 | context, zero-padded to 32]: INIT_KS = 4,
// d_i <= 32), kept as f16 pieces in registers for the whole run; every block ends with the gate
// h + (W_1 relu(u) + b_1) * sigmoid(W_c context + b_c): the second Linear then has accumulators of its own
// (its input pieces are finished first: u's registers are needed), the gate's Linear is one more stage
// (two k-steps, k-major) and the residual stream takes the product in.
template <bool INVERSE, int INIT_KS, int NW, int KB = 8, bool CTX = false, int RING = kRing, int ACT = kActRelu>
__global__ void __launch_bounds__(NW * kWave, 2) rqs_resnet_f16_kernel(const Args a) {
    static_assert(!CTX || INIT_KS == 4, "context: two identity k-steps + two context k-steps");
    static_assert(ACT == kActRelu || (!CTX && ACT >= kActLeakyRelu && ACT <= kActTanh), "other activations: no context");
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
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(fr.h), "=v"(fr.l) : "v"(nxt));
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
                        ConvWeave<true>{u[0], qh[0], ql[0], qh[1], ql[1], conv_scale, peak}.all();
                        ConvWeave<true>{u[1], qh[2], ql[2], qh[3], ql[3], conv_scale, peak}.all();
                        ConvWeave<true>{u[2], qh[4], ql[4], qh[5], ql[5], conv_scale, peak}.all();
                        ConvWeave<true>{u[3], qh[6], ql[6], qh[7], ql[7], conv_scale, peak}.all();
                        worst = __builtin_fmaxf(worst, peak * conv_scale);
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
                static_assert(T >= 1 && T <= 3, "2 .. 16 bins");
                using Steps = FusedSteps<INVERSE, KB>;
                // what is left for the next group's first tile: everything (T = 1), heights + rest (T = 2), the rest (T = 3)
                constexpr int kRest = T == 1 ? (kSeqW | kSeqH | kSeqFinish) : T == 2 ? (kSeqH | kSeqFinish) : kSeqFinish;
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
                    if constexpr (T >= 2) {
                        load_bias_tile(acc, gb + 32);
                        tile_gemm(acc, ph, pl, sm, fr, lane, SplineWeaveSeq<kSeqW, Steps>{f, a.sp});
                        take_chunk<1, KB>(f, acc);
                    }
                    if constexpr (T >= 3) {
                        load_bias_tile(acc, gb + 64);
                        tile_gemm(acc, ph, pl, sm, fr, lane, SplineWeaveSeq<kSeqH, Steps>{f, a.sp});
                        take_chunk<2, KB>(f, acc);
                    }
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
                using Steps = FusedSteps8<INVERSE>;
                Steps fa, fb;
                const float kappa = gemm[0];
                fa.kappa = fb.kappa = kappa;
                fa.kl2e = fb.kl2e = 1.44269502162933349609375f * kappa;
                fa.tail_s = fb.tail_s = a.sp.tail_logit * gemm[1];  // gemm[1] = 1 / kappa
                float* slot_b = nullptr;
                const float* fbias = gemm + kHdr + half * 16;
                f32x16 acc[3];
                auto commit = [&](Steps& f, float* slot) {
                    *slot = f.y;
                    lad_acc += f.lad;
                    quad_status |= f.status;
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
                        commit(fb, slot_b);
                    } else {
                        tile_gemm(acc[0], ph, pl, sm, fr, lane, NoWeave{});
                    }
                    fa.x = *slot0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fa.ew[j] = acc[0][j];
                        fa.eh[j] = acc[0][8 + j];
                    }
                    load_bias_tile(acc[1], fbias + (g * 3 + 1) * 32);
                    tile_gemm(acc[1], ph, pl, sm, fr, lane, w0);
                    fb.x = *slot1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < 7) fa.sd[j] = acc[1][j];
                        fb.ew[j] = acc[1][8 + j];
                    }
                    load_bias_tile(acc[2], fbias + (g * 3 + 2) * 32);
                    tile_gemm(acc[2], ph, pl, sm, fr, lane, w1);
                    commit(fa, slot0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fb.eh[j] = acc[2][j];
                        if (j < 7) fb.sd[j] = acc[2][8 + j];
                    }
                    slot_b = slot1;
                }
                spline_unit_range<kUnitFinishB, 0, spline_unit_slices<kUnitFinishB, Steps>()>(fa, fb, a.sp);
                commit(fb, slot_b);
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

using namespace nfa;

static int launch_f16(const float* inputs, const float* context, int32_t context_features, const void* stream_packed,
                      int32_t param_stages, const int32_t* final_positions, int32_t num_layers, float* outputs,
                      float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch, int32_t features,
                      int32_t num_transform, int32_t num_identity, int32_t hidden_features, int32_t num_blocks,
                      const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (flags & ~(NFA_FLAG_INVERSE | NFA_FLAG_ACCUMULATE_LOGABSDET | NFA_FLAG_STANDARD_NORMAL_LOG_PROB |
                  NFA_FLAG_SKIP_OUTPUTS | NFA_FLAG_PAD_COLUMNS_MASK | NFA_FLAG_ACTIVATION_MASK))
        return NFA_ERR_INVALID_ARGUMENT;
    const int activation = (flags & NFA_FLAG_ACTIVATION_MASK) >> NFA_FLAG_ACTIVATION_SHIFT;
    if (activation > NFA_ACTIVATION_TANH) return NFA_ERR_INVALID_ARGUMENT;
    flags &= ~NFA_FLAG_ACTIVATION_MASK;
    if (!density_flags_valid(flags)) return NFA_ERR_INVALID_ARGUMENT;
    if (batch < 0 || features < 1 || num_transform < 1 || num_identity < 1 ||
        num_transform > features || num_identity > features || num_blocks < 0 || num_layers < 1 || param_stages < 1)
        return NFA_ERR_INVALID_ARGUMENT;
    k8h::Args a;
    int rc = make_dev_spec(spec, &a.sp);
    if (rc != NFA_OK) return rc;
    if (a.sp.beta != 1.0f) return NFA_ERR_UNSUPPORTED;
    // bin counts: 8 and 10 have their own final-layer loops; 2 .. 16 otherwise (no context there)
    const bool any_bins = a.sp.K != 8 && a.sp.K != 10;
    // activations other than ReLU: the two tuned bin counts, no context
    if (activation != NFA_ACTIVATION_RELU && (any_bins || context_features > 0)) return NFA_ERR_UNSUPPORTED;
    if (a.sp.K < 2 || a.sp.K > 16 || (any_bins && context_features > 0) || !a.sp.linear || hidden_features != 128 || (num_transform & 3) != 0 || num_transform > 64 ||
        num_identity > 64 || features > 128 || (features & 3) != 0 || (batch & 127) != 0 || num_blocks > 64 ||
        num_layers > 4096)
        return NFA_ERR_UNSUPPORTED;
    const bool with_ctx = context_features > 0;
    if (context_features < 0) return NFA_ERR_INVALID_ARGUMENT;
    // with a context: two identity k-steps + two context k-steps in the initial layer
    if (with_ctx && (context_features > 32 || num_identity > 32)) return NFA_ERR_UNSUPPORTED;
    // rows of the final layer per transformed feature: 23 logits padded to 24 (8 bins: two features share three
    // tiles), otherwise 3 K - 1 padded to whole 16-row lane-half shares
    const int rows_per_feature = a.sp.K == 8 ? 24 : 16 * ((3 * a.sp.K - 1 + 15) / 16);
    const int param_words = k8h::kTabWords + (k8h::kHdr + 128) * (1 + (with_ctx ? 3 : 2) * num_blocks) + k8h::kHdr +
                            num_transform * rows_per_feature;
    if (param_stages * 2048 < param_words || param_stages > 4) return NFA_ERR_INVALID_ARGUMENT;
    if (batch == 0) return NFA_OK;
    if (!inputs || !stream_packed || !final_positions || !logabsdet || !redo_blocks ||
        (!outputs && !(flags & NFA_FLAG_SKIP_OUTPUTS)) || (with_ctx && !context))
        return NFA_ERR_INVALID_ARGUMENT;
    a.ctx = with_ctx ? context : nullptr;
    a.ce = context_features;
    a.normal = (flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ? 1 : 0;
    a.skip_out = (flags & NFA_FLAG_SKIP_OUTPUTS) ? 1 : 0;
    a.Ds = density_columns(flags, features);
    if (a.Ds < 1) return NFA_ERR_INVALID_ARGUMENT;
    a.log_z = standard_normal_log_z(a.Ds);
    a.x = inputs;
    a.w = reinterpret_cast<const vec4f*>(stream_packed);
    a.final_tab = final_positions;
    a.out = outputs;
    a.lad = logabsdet;
    a.redo = redo_blocks;
    a.status = status;
    a.batch = batch;
    a.D = features;
    a.dt = num_transform;
    a.di = num_identity;
    a.num_blocks = num_blocks;
    a.num_layers = num_layers;
    a.param_stages = param_stages;
    a.param_words = param_words;
    const int init_ks = (with_ctx || num_identity > 32) ? 4 : 2;
    a.num_stages = param_stages + init_ks / 2 + (with_ctx ? 9 : 8) * num_blocks + num_transform * rows_per_feature / 32;
    a.accumulate = (flags & NFA_FLAG_ACCUMULATE_LOGABSDET) ? 1 : 0;
    a.trace = g_k7_trace;
    // workgroups of eight waves (256 rows, one per CU, one weight stream per CU) when the batch gives
    // every CU one; otherwise four waves (128 rows)
    const int cus = device_cu_count();
    static const int force_nw = getenv("NFA_K8H_WAVES") ? atoi(getenv("NFA_K8H_WAVES")) : 0;
    static const int force_ring = getenv("NFA_K8H_RING") ? atoi(getenv("NFA_K8H_RING")) : 0;   // 5: elastic stream (experiment, slower)
    int nw = ((batch & 255) == 0 && (batch >> 8) >= cus) ? 8 : 4;
    if (force_nw == 4 || (force_nw == 8 && (batch & 255) == 0)) nw = force_nw;
    const size_t lds_static = 1024;   // s_final, s_bad, s_sync (rounded up)
    const size_t lds_cap = 160 * 1024 - lds_static;
    auto lds_for = [&](int n, int ring) {
        return (size_t)ring * k8h::kStageVec4 * 16 + (size_t)n * features * k8h::kRowPad * sizeof(float) +
               (size_t)2 * ((param_words + 3) & ~3) * sizeof(float);
    };
    if (lds_for(nw, k8h::kRing) > lds_cap) nw = 4;
    if (lds_for(nw, k8h::kRing) > lds_cap) return NFA_ERR_UNSUPPORTED;
    // the elastic stream (five slots, counters instead of the per-stage barrier) where it fits: eight-wave
    // workgroups of the 8-bin kernel without a context (the bench's shape: 161 984 bytes at D = 64)
#ifdef NFA_K8H_ELASTIC
    const bool elastic = nw == 8 && !with_ctx && a.sp.K == 8 && force_ring == 5 && lds_for(8, k8h::kRingElastic) <= lds_cap;
#else
    const bool elastic = false;
    (void)force_ring;
#endif
    const size_t lds_launch = lds_for(nw, elastic ? k8h::kRingElastic : k8h::kRing);
    int64_t blocks = batch / (32 * nw);
    const int64_t per_cu = (nw == 4 && lds_launch + 2048 <= 80 * 1024) ? 2 : 1;
    const int64_t cap = (int64_t)cus * per_cu;
    if (blocks > cap) blocks = cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    profile_next_launch(&e0, &e1);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(nw * kWave);
    const bool inv = (flags & NFA_FLAG_INVERSE) != 0;
    void (*kern)(const k8h::Args) = nullptr;
    int which = with_ctx ? 16 + (inv ? 1 : 0) + (nw == 8 ? 2 : 0) + (a.sp.K == 10 ? 4 : 0)
                         : (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0) + (a.sp.K == 10 ? 8 : 0);
    if (elastic) which = 24 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0);
    if (any_bins) which = 32 + (a.sp.K - 2) * 8 + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0);
#define NFA_K8H_ANY(KB_)                                                                                             \
    case KB_:                                                                                                        \
        kern = nw == 8 ? (init_ks == 4 ? (inv ? k8h::rqs_resnet_f16_kernel<true, 4, 8, KB_> : k8h::rqs_resnet_f16_kernel<false, 4, 8, KB_>)   \
                                       : (inv ? k8h::rqs_resnet_f16_kernel<true, 2, 8, KB_> : k8h::rqs_resnet_f16_kernel<false, 2, 8, KB_>))  \
                       : (init_ks == 4 ? (inv ? k8h::rqs_resnet_f16_kernel<true, 4, 4, KB_> : k8h::rqs_resnet_f16_kernel<false, 4, 4, KB_>)   \
                                       : (inv ? k8h::rqs_resnet_f16_kernel<true, 2, 4, KB_> : k8h::rqs_resnet_f16_kernel<false, 2, 4, KB_>)); \
        break;
#define NFA_K8H_ACT(ACT_, KB_)                                                                                       \
    kern = nw == 8 ? (init_ks == 4 ? (inv ? k8h::rqs_resnet_f16_kernel<true, 4, 8, KB_, false, k8h::kRing, ACT_>           \
                                          : k8h::rqs_resnet_f16_kernel<false, 4, 8, KB_, false, k8h::kRing, ACT_>)          \
                                   : (inv ? k8h::rqs_resnet_f16_kernel<true, 2, 8, KB_, false, k8h::kRing, ACT_>           \
                                          : k8h::rqs_resnet_f16_kernel<false, 2, 8, KB_, false, k8h::kRing, ACT_>))         \
                   : (init_ks == 4 ? (inv ? k8h::rqs_resnet_f16_kernel<true, 4, 4, KB_, false, k8h::kRing, ACT_>           \
                                          : k8h::rqs_resnet_f16_kernel<false, 4, 4, KB_, false, k8h::kRing, ACT_>)          \
                                   : (inv ? k8h::rqs_resnet_f16_kernel<true, 2, 4, KB_, false, k8h::kRing, ACT_>           \
                                          : k8h::rqs_resnet_f16_kernel<false, 2, 4, KB_, false, k8h::kRing, ACT_>));
    if (activation != NFA_ACTIVATION_RELU) {
        which = 32 + 15 * 8 + (activation - 1) * 16 + (a.sp.K == 10 ? 8 : 0) + (inv ? 1 : 0) + (init_ks == 4 ? 2 : 0) + (nw == 8 ? 4 : 0);
        if (a.sp.K == 8) {
            if (activation == NFA_ACTIVATION_LEAKY_RELU) { NFA_K8H_ACT(kActLeakyRelu, 8) }
            else if (activation == NFA_ACTIVATION_ELU) { NFA_K8H_ACT(kActElu, 8) }
            else { NFA_K8H_ACT(kActTanh, 8) }
        } else {
            if (activation == NFA_ACTIVATION_LEAKY_RELU) { NFA_K8H_ACT(kActLeakyRelu, 10) }
            else if (activation == NFA_ACTIVATION_ELU) { NFA_K8H_ACT(kActElu, 10) }
            else { NFA_K8H_ACT(kActTanh, 10) }
        }
    } else
#undef NFA_K8H_ACT
    if (any_bins) switch (a.sp.K) {
        NFA_K8H_ANY(2) NFA_K8H_ANY(3) NFA_K8H_ANY(4) NFA_K8H_ANY(5) NFA_K8H_ANY(6) NFA_K8H_ANY(7) NFA_K8H_ANY(9)
        NFA_K8H_ANY(11) NFA_K8H_ANY(12) NFA_K8H_ANY(13) NFA_K8H_ANY(14) NFA_K8H_ANY(15) NFA_K8H_ANY(16)
    }
#undef NFA_K8H_ANY
    else switch (which) {
#ifdef NFA_K8H_ELASTIC   // (experiment builds only: measured 3 % slower than the rigid stream, profiles/r3/k8h_elastic_stream.txt)
        case 24: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8, 8, false, k8h::kRingElastic>; break;
        case 25: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8, 8, false, k8h::kRingElastic>; break;
        case 26: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 8, false, k8h::kRingElastic>; break;
        case 27: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 8, false, k8h::kRingElastic>; break;
#endif
        case 16: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4, 8, true>; break;
        case 17: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4, 8, true>; break;
        case 18: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 8, true>; break;
        case 19: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 8, true>; break;
        case 20: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4, 10, true>; break;
        case 21: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4, 10, true>; break;
        case 22: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 10, true>; break;
        case 23: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 10, true>; break;
        case 0: kern = k8h::rqs_resnet_f16_kernel<false, 2, 4>; break;
        case 1: kern = k8h::rqs_resnet_f16_kernel<true, 2, 4>; break;
        case 2: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4>; break;
        case 3: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4>; break;
        case 4: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8>; break;
        case 5: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8>; break;
        case 6: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8>; break;
        case 7: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8>; break;
        case 8: kern = k8h::rqs_resnet_f16_kernel<false, 2, 4, 10>; break;
        case 9: kern = k8h::rqs_resnet_f16_kernel<true, 2, 4, 10>; break;
        case 10: kern = k8h::rqs_resnet_f16_kernel<false, 4, 4, 10>; break;
        case 11: kern = k8h::rqs_resnet_f16_kernel<true, 4, 4, 10>; break;
        case 12: kern = k8h::rqs_resnet_f16_kernel<false, 2, 8, 10>; break;
        case 13: kern = k8h::rqs_resnet_f16_kernel<true, 2, 8, 10>; break;
        case 14: kern = k8h::rqs_resnet_f16_kernel<false, 4, 8, 10>; break;
        default: kern = k8h::rqs_resnet_f16_kernel<true, 4, 8, 10>; break;
    }
    static const char* const act_names[] = {"relu", "leaky_relu", "elu", "tanh"};
    note_layer_kernel("k8h::rqs_resnet_f16_kernel<inverse=%d, init_ks=%d, waves=%d, K=%d, ctx=%d, ring=%d, act=%s>", inv ? 1 : 0,
                      init_ks, nw, a.sp.K, with_ctx ? 1 : 0, elastic ? k8h::kRingElastic : k8h::kRing, act_names[activation]);
    if (lds_launch > 64 * 1024) {
        static unsigned long long raised[32 + 15 * 8 + 3 * 16] = {};   // device masks (raise_dynamic_lds)
        {
            const int rc_lds = raise_dynamic_lds((const void*)kern, &raised[which], (int)lds_cap);
            if (rc_lds != NFA_OK) return rc_lds;
        }
    }
    if (e0) hipExtLaunchKernelGGL(kern, grid, block, lds_launch, st, e0, e1, 0, a);
    else hipLaunchKernelGGL(kern, grid, block, lds_launch, st, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rqs_flow_resnet_f16x2_f32(const float* inputs, const void* stream_packed, int32_t param_stages,
                                             const int32_t* final_positions, int32_t num_layers, float* outputs,
                                             float* logabsdet, int32_t* redo_blocks, int32_t* status, int64_t batch,
                                             int32_t features, int32_t num_transform, int32_t num_identity,
                                             int32_t hidden_features, int32_t num_blocks,
                                             const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    return launch_f16(inputs, nullptr, 0, stream_packed, param_stages, final_positions, num_layers, outputs, logabsdet,
                      redo_blocks, status, batch, features, num_transform, num_identity, hidden_features, num_blocks,
                      spec, flags, stream);
}

extern "C" int nfa_rqs_flow_resnet_context_f16x2_f32(const float* inputs, const float* context,
                                                     int32_t context_features, const void* stream_packed,
                                                     int32_t param_stages, const int32_t* final_positions,
                                                     int32_t num_layers, float* outputs, float* logabsdet,
                                                     int32_t* redo_blocks, int32_t* status, int64_t batch,
                                                     int32_t features, int32_t num_transform, int32_t num_identity,
                                                     int32_t hidden_features, int32_t num_blocks,
                                                     const nfa_rqs_spec* spec, int32_t flags, void* stream) {
    if (context_features < 1) return NFA_ERR_INVALID_ARGUMENT;
    return launch_f16(inputs, context, context_features, stream_packed, param_stages, final_positions, num_layers,
                      outputs, logabsdet, redo_blocks, status, batch, features, num_transform, num_identity,
                      hidden_features, num_blocks, spec, flags, stream);
}
