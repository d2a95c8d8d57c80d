// The split-f16 GEMM machinery of K8x (rqs_resnet_f16x3.hip): every fp32 operand as THREE f16 pieces
//
//   x s = hi + lo + r' 2^-8,   hi = RN16(x s),  lo = RN16(x s - hi),  r' = RN16((x s - hi - lo) 2^8)      (s: a power of two)
//
// 11 + 11 + 11 significand bits with a sign each: the three pieces hold ANY fp32 value exactly while the last one stays
// above f16's smallest subnormal (2^-24), i.e. -- the last piece being kept at 2^8 times its value -- for |x s| >= 2^-9;
// below that the absolute error is <= 2^-33 (see rqs_resnet_f16x3.hip for the scales).  The product of two such operands,
//
//   x w = hi_x hi_w + (hi_x lo_w + lo_x hi_w) + (hi_x r_w + r_x hi_w) + [lo_x lo_w + ...]
//          1            2^-11                    2^-22                   dropped: <= 2^-22 x 2^-2 |x w|  (lo lo; rms 2^-24.6)
//
// runs as THREE v_mfma_f32_32x32x16_f16 per k-step and tile (hi hi, hi lo, lo hi) plus ONE v_mfma_scale_f32_32x32x64_f8f6f4
// per TWO k-steps and tile for the two 2^-22-level products: a last piece is a single bit (or zero) and its partner only
// has to be right to a few bits, so both go to the bf8 (e5m2: f16's exponent range, three significand bits) form of the
// instruction, which moves four times the k-range per instruction at half the issue rate per k (64.1 cycles for K = 64
// against 32.1 for K = 16: profiles/r6/mx_probe.txt) -- the hi factor truncated to bf8 costs <= 2^-3 of a 2^-22-level
// term, below the dropped lo lo product; the instruction's block scale (e8m0: 2^-8 on the A side) takes the last pieces'
// 2^8 out again.  Per k-step and tile: 3 + 1/2 x 2 = 4 f16-MFMA times where the five-product form of this kernel's first
// build took 5 and the three-piece bf16 scheme (bf16x3_gemm.hpp) takes 6.
// A bf8 operand of a lane (32 bytes, two k-steps ks0, ks1; the lane's eight k values each):
//   A (weights)      [bf8(hi_w) ks0 | bf8(r'_w) ks0 | bf8(hi_w) ks1 | bf8(r'_w) ks1]     packed by the host
//   B (activations)  [bf8(r'_x) ks0 | bf8(hi_x) ks0 | bf8(r'_x) ks1 | bf8(hi_x) ks1]     the high bytes of the f16 pieces
// (which k the hardware gives byte e of lane-half h is immaterial: A and B use the same map.)
// The LDS-DMA ring, its counted waits and its barriers are bf16x3_gemm.hpp's; a 12 KB stage holds twelve 1 KB fragments
// ([64 lanes] x 16 B): k-major stages cover TWO k-steps of TWO output tiles, [2 tiles][H0, L0, H1, L1, X lo, X hi]; a
// stage of the tile-major final layer covers four k-steps of its tile, [H0, L0, H1, L1][H2, L2, H3, L3][X01 lo, X01 hi, X23 lo,
// X23 hi] -- fragments 0 .. 3 of EVERY stage are the f16 fragments its first MFMAs need (read ahead: stage_kmajor).
#pragma once

#include "bf16x3_gemm.hpp"

namespace nfa {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

namespace k8x {

constexpr int kScaleA = 119, kScaleB = 127;   // e8m0 block scales of the bf8 instruction: 2^-8 (the last pieces' 2^8), 1

#ifdef NFA_ABL_NO_MFMA   // (measurement builds)
#define NFA_K8X_F16(a, b, c) (c)
#define NFA_K8X_BF8(a, b, c) (c)
#else
#define NFA_K8X_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define NFA_K8X_BF8(a, b, c) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, kScaleA, 0, kScaleB)
#endif

// the pieces of a k-step: 8 values per lane and piece = one register quad each (r: the last piece x 2^8)
struct Pieces {
    uvec4 h, l, r;
};

// The conversions are `asm volatile`: a plain asm statement is free to move, and hipcc's scheduler moved these across the
// stream's (volatile) waits and barriers to right behind the MFMAs whose accumulators they read -- instructions inside an
// asm statement are invisible to the hazard recogniser, which therefore pads nothing between an MFMA and an asm block
// that reads its result: the first build had every GEMM's last products missing from some values (logits 2^-12 off,
// bit-reproducible; found by tests/test_gpu_logits.py, round 6).  Volatile keeps them in program order behind the GEMM's
// last barrier, and tile_to_pieces puts the matrix pipe's write-back distance (11 wait states behind an 8-pass MFMA)
// in front of the first block that reads a tile.  NFA_K8X_NO_ASM (measurement builds): the same arithmetic in C++.
#define NFA_K8X_ASM asm volatile

// two fp32 values x `scale` (a power of two) -> packed f16 pairs of the three pieces.
//   hi = RN16(v s)                         v_fma_mixlo / mixhi_f16: fp32 fma, result rounded to f16
//   t  = v s - hi                          v_fma_mix_f32 with the f16 `hi` as negated addend: exact (<= 13 bits)
//   lo = RN16(t)
//   d  = t - lo                            exact in fp32
//   r' = RN16(256 d)                       one bit (or zero) where lo is a normal f16; all of d's bits while 256 d >= 2^-24
// |v s| >= 65520 gives hi = inf, t = -inf, lo = -inf, d = NaN: the overflow poisons every sum it enters, the row block is
// flagged and redone by the exact kernel.  (One asm block: hipcc puts an `s_nop 0` between two adjacent asm statements;
// early-clobber everywhere: every output is written before the last input is read.)
__device__ __forceinline__ void split3_scaled(float v0, float v1, float scale, unsigned& hi, unsigned& lo, unsigned& rr) {
#if defined(NFA_K8X_NO_ASM)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float x0 = v0 * scale, x1 = v1 * scale;
    const h2 h = {(_Float16)x0, (_Float16)x1};
    const float t0 = x0 - (float)h[0], t1 = x1 - (float)h[1];
    const h2 l = {(_Float16)t0, (_Float16)t1};
    const h2 r = {(_Float16)((t0 - (float)l[0]) * 256.0f), (_Float16)((t1 - (float)l[1]) * 256.0f)};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
    rr = __builtin_bit_cast(unsigned, r);
#else
    unsigned h, l, r;
    float t0, t1;
    NFA_K8X_ASM(
        "v_fma_mixlo_f16 %0, %5, %7, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %6, %7, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mix_f32 %3, %5, %7, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %6, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %3, 1.0, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %4, 1.0, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mix_f32 %3, %3, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %4, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %2, %3, %8, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %4, %8, 0 op_sel_hi:[0,0,0]"
        : "=&v"(h), "=&v"(l), "=&v"(r), "=&v"(t0), "=&v"(t1)
        : "v"(v0), "v"(v1), "v"(scale), "s"(256.0f));
    hi = h;
    lo = l;
    rr = r;
#endif
}

// the fp32 value of a piece triple (exact: hi + lo has at most 23 bits, + r' 2^-8 at most 24) times `mul`, plus what the
// accumulators hold: the skip connection, acc = bias + T x h
__device__ __forceinline__ void pieces_fma2(unsigned h, unsigned l, unsigned r, float mul, float& acc0, float& acc1) {
    float t0, t1;
#if defined(NFA_K8X_NO_ASM)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 hh = __builtin_bit_cast(h2, h), ll = __builtin_bit_cast(h2, l), rr = __builtin_bit_cast(h2, r);
    t0 = ((float)hh[0] + (float)ll[0]) + (float)rr[0] * 0.00390625f;
    t1 = ((float)hh[1] + (float)ll[1]) + (float)rr[1] * 0.00390625f;
#else
    NFA_K8X_ASM(
        "v_fma_mix_f32 %0, %2, 1.0, %3 op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %1, %2, 1.0, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t0), "=&v"(t1)
        : "v"(h), "v"(l), "v"(r), "s"(0.00390625f));
#endif
    acc0 = __builtin_fmaf(t0, mul, acc0);
    acc1 = __builtin_fmaf(t1, mul, acc1);
}

// ReLU applied to a value given as f16 pieces: all three are cleared where the leading piece is negative and not a
// NaN (f16 bit patterns 0x8000 .. 0xFC00 = int16 <= -1024), so that NaNs -- overflow poison included -- keep
// propagating like torch.relu's; -inf (a large negative pre-activation) becomes the zero it should.
__device__ __forceinline__ void relu_pieces(Pieces& p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned keep;
        // (volatile: a pure statement may be hoisted or merged by the compiler, and ReLU'd copies kept alive beside the
        //  originals would not fit the register file)
        asm volatile("v_pk_min_i16 %0, %1, 0\n\t"
            "v_pk_add_i16 %0, %0, %2\n\t"
            "v_pk_ashrrev_i16 %0, %3, %0\n\t"
            "v_not_b32 %0, %0"
            : "=&v"(keep)
            : "v"(p.h[i]), "s"(0x03FF03FFu), "s"(0x000F000Fu));  // (packed inline constants fill one half only)
        p.h[i] &= keep;
        p.l[i] &= keep;
        p.r[i] &= keep;
    }
}

// the bf8 B operand of two k-steps: the high bytes of the r' and hi pieces (an f16's high byte IS its bf8 truncation: same
// sign, same five exponent bits, the two leading fraction bits; infinities and NaNs stay what they are)
__device__ __forceinline__ i32x8 bf8_operand(const Pieces& p0, const Pieces& p1) {
    constexpr unsigned kHigh = 0x07050301u;   // bytes 1, 3 of the second argument, then bytes 1, 3 of the first
    i32x8 b;
    b[0] = (int)__builtin_amdgcn_perm(p0.r[1], p0.r[0], kHigh);
    b[1] = (int)__builtin_amdgcn_perm(p0.r[3], p0.r[2], kHigh);
    b[2] = (int)__builtin_amdgcn_perm(p0.h[1], p0.h[0], kHigh);
    b[3] = (int)__builtin_amdgcn_perm(p0.h[3], p0.h[2], kHigh);
    b[4] = (int)__builtin_amdgcn_perm(p1.r[1], p1.r[0], kHigh);
    b[5] = (int)__builtin_amdgcn_perm(p1.r[3], p1.r[2], kHigh);
    b[6] = (int)__builtin_amdgcn_perm(p1.h[1], p1.h[0], kHigh);
    b[7] = (int)__builtin_amdgcn_perm(p1.h[3], p1.h[2], kHigh);
    return b;
}

// accumulator tile (x `scale`), registers 8 hk .. 8 hk + 7  ->  pieces of k-step 2 t + hk.  RELU: `v < 0 ? 0 : v`
// (compare + select: NaN stays NaN, v_max_f32 would return the zero)
template <bool RELU>
__device__ __forceinline__ void tile_to_pieces(const f32x16& a, float scale, Pieces& p0, Pieces& p1) {
    asm volatile("s_nop 7\n\ts_nop 3" : : "v"(a));   // (see NFA_K8X_ASM: the tile's last MFMA has written back)
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) {
        float v0 = a[q2 * 2], v1 = a[q2 * 2 + 1];
        if (RELU) {
            v0 = v0 < 0.0f ? 0.0f : v0;
            v1 = v1 < 0.0f ? 0.0f : v1;
        }
        unsigned h, l, r;
        split3_scaled(v0, v1, scale, h, l, r);
        if (q2 < 4) {
            p0.h[q2] = h;
            p0.l[q2] = l;
            p0.r[q2] = r;
        } else {
            p1.h[q2 - 4] = h;
            p1.l[q2 - 4] = l;
            p1.r[q2 - 4] = r;
        }
    }
}

// value of the pieces of one k-step x `mul`, added to 8 accumulator registers (the skip connection)
__device__ __forceinline__ void add_pieces(f32x16& a, int q0, const Pieces& p, float mul) {
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
        float a0 = a[q0 + 2 * j2], a1 = a[q0 + 2 * j2 + 1];   // (no references to vector elements)
        pieces_fma2(p.h[j2], p.l[j2], p.r[j2], mul, a0, a1);
        a[q0 + 2 * j2] = a0;
        a[q0 + 2 * j2 + 1] = a1;
    }
}

__device__ __forceinline__ i32x8 join_x(vec4f lo, vec4f hi) {
    const uvec4 a = __builtin_bit_cast(uvec4, lo), b = __builtin_bit_cast(uvec4, hi);
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

// the f16 fragments (H ks0, L ks0, H ks1, L ks1) of the FIRST tile of a stage: fragments 0 .. 3 of every stage layout
struct Lead {
    vec4f h0, l0, h1, l1;
};
__device__ __forceinline__ Lead read_lead(const vec4f* stage) {
    return Lead{stage[0 * 64], stage[1 * 64], stage[2 * 64], stage[3 * 64]};
}

#define NFA_K8X_FENCE() __builtin_amdgcn_sched_barrier(0)

// One k-major stage: two k-steps (pieces b0, b1, bf8 operand bx) of two output tiles, fragments [H0, L0, H1, L1, X lo, X hi]
// per tile.  Software pipeline across the stage barrier (round 6): the first tile's f16 fragments arrive in `lead` -- read
// right behind the PREVIOUS stage's barrier --, and this stage's barrier stands in front of the second tile's last four
// MFMAs, which only need registers: the next stage's lead fragments are requested behind the barrier and land while those
// MFMAs run.  (The three-slot ring cannot be read ahead of the barrier that completes a stage; before this the wave paid one
// LDS latency at every stage start unless the SIMD's other wave happened to cover it.)  LAST: no successor to read ahead
// (the GEMM's last stage: the conversions that follow would have to keep the fragments alive).
// The three f16 products of a k-step share their srcB as far as they can (the matrix pipe's energy depends on how often
// srcB changes between consecutive instructions, a new srcA costs nothing: fused_common.hpp, round 4); within a stage the
// order of the additions is immaterial to the result's error.
template <bool LAST>
__device__ __forceinline__ void stage_kmajor(f32x16& acc0, f32x16& acc1, const Pieces& b0, const Pieces& b1, const i32x8& bx,
                                             WeightStream& sm, int lane, Lead& lead) {
    stream_request(sm);
    const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
    const f16x8 bh0 = __builtin_bit_cast(f16x8, b0.h), bl0 = __builtin_bit_cast(f16x8, b0.l);
    const f16x8 bh1 = __builtin_bit_cast(f16x8, b1.h), bl1 = __builtin_bit_cast(f16x8, b1.l);
    // everything else of the stage is requested now: tile 0's X, tile 1's six fragments
    const vec4f x0l = cur[4 * 64], x0h = cur[5 * 64];
    const vec4f h0 = cur[6 * 64], l0 = cur[7 * 64], h1 = cur[8 * 64], l1 = cur[9 * 64], x1l = cur[10 * 64], x1h = cur[11 * 64];
    NFA_K8X_FENCE();
    {
        const f16x8 ah0 = __builtin_bit_cast(f16x8, lead.h0), al0 = __builtin_bit_cast(f16x8, lead.l0);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, lead.h1), al1 = __builtin_bit_cast(f16x8, lead.l1);
        acc0 = NFA_K8X_F16(ah0, bl0, acc0);
        acc0 = NFA_K8X_F16(al0, bh0, acc0);
        acc0 = NFA_K8X_F16(ah0, bh0, acc0);
        acc0 = NFA_K8X_F16(ah1, bl1, acc0);
        acc0 = NFA_K8X_F16(al1, bh1, acc0);
        acc0 = NFA_K8X_F16(ah1, bh1, acc0);
        acc0 = NFA_K8X_BF8(join_x(x0l, x0h), bx, acc0);
    }
    const f16x8 ah0 = __builtin_bit_cast(f16x8, h0), al0 = __builtin_bit_cast(f16x8, l0);
    const f16x8 ah1 = __builtin_bit_cast(f16x8, h1), al1 = __builtin_bit_cast(f16x8, l1);
    const i32x8 ax = join_x(x1l, x1h);
    acc1 = NFA_K8X_F16(ah0, bl0, acc1);
    acc1 = NFA_K8X_F16(al0, bh0, acc1);
    acc1 = NFA_K8X_F16(ah0, bh0, acc1);
    NFA_K8X_FENCE();
    stream_advance(sm);          // every read of this stage has landed in registers; the next stage is complete
    if constexpr (!LAST) lead = read_lead(sm.ring + sm.slot * kStageVec4 + lane);
    NFA_K8X_FENCE();
    acc1 = NFA_K8X_F16(ah1, bl1, acc1);
    acc1 = NFA_K8X_F16(al1, bh1, acc1);
    acc1 = NFA_K8X_F16(ah1, bh1, acc1);
    acc1 = NFA_K8X_BF8(ax, bx, acc1);
}

// k-major GEMM (all four output tiles accumulate together): out^T[128 x 32 samples] += W[128 x 16 NKS] x act^T; two
// stages per pair of k-steps (tiles 0, 1 and tiles 2, 3).  RELU: applied to the input pieces on the fly (the pieces
// themselves stay: they are the residual stream)
template <bool RELU, int NKS>
__device__ __forceinline__ void gemm_kmajor(f32x16 (&acc)[4], const Pieces (&p)[8], WeightStream& sm, int lane) {
    static_assert(NKS % 2 == 0, "pairs of k-steps");
    Lead lead = read_lead(sm.ring + sm.slot * kStageVec4 + lane);
#pragma unroll
    for (int pr = 0; pr < NKS / 2; ++pr) {
        Pieces b0 = p[2 * pr], b1 = p[2 * pr + 1];
        if (RELU) {
            relu_pieces(b0);
            relu_pieces(b1);
        }
        const i32x8 bx = bf8_operand(b0, b1);
        stage_kmajor<false>(acc[0], acc[1], b0, b1, bx, sm, lane, lead);
        if (pr + 1 < NKS / 2) stage_kmajor<false>(acc[2], acc[3], b0, b1, bx, sm, lane, lead);
        else stage_kmajor<true>(acc[2], acc[3], b0, b1, bx, sm, lane, lead);
    }
}

}  // namespace k8x
}  // namespace nfa
