// The split-f16 GEMM machinery of K8x (rqs_resnet_f16x3.hip): every fp32 operand as THREE f16 pieces
//
//   x s = hi + lo + r,   hi = RN16(x s),  lo = RN16(x s - hi),  r = RN16(x s - hi - lo)      (s: a power of two)
//
// 11 + 11 + 11 significand bits with a sign each: the three pieces hold ANY fp32 value exactly while the last one
// stays above f16's smallest subnormal (2^-24), i.e. for |x s| >= 2^-2; below that the absolute error is <= 2^-25
// (see rqs_resnet_f16x3.hip for the scales).  Five v_mfma_f32_32x32x16_f16 per k-step and tile:
//
//   x w = hi_x hi_w + (hi_x lo_w + lo_x hi_w) + (hi_x r_w + r_x hi_w) + [lo_x lo_w + ...]
//          1            2^-11                    2^-22                   dropped: <= 2^-22 x 2^-2 |x w|  (lo lo; rms 2^-24.6)
//
// against the six bf16 products of the three-piece bf16 scheme (bf16x3_gemm.hpp): the same stage format ([4 tiles]
// [3 pieces][64 lanes] x 16 bytes k-major, [3 pieces][4 k-steps][64 lanes] tile-major: the LDS-DMA ring, its counted
// waits and its barriers are bf16x3_gemm.hpp's), 5/6 of the matrix-pipe time, and a piece conversion on
// v_fma_mix*_f16 (seven instructions per pair of values where the bf16 split takes eleven).
#pragma once

#include "bf16x3_gemm.hpp"

namespace nfa {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace k8x {

// Order of the five products: grouped by their SECOND operand (the activation pieces: r, lo, hi, hi, hi).  The matrix
// pipe's energy depends on how often srcB changes between consecutive instructions (fused_common.hpp, round 4), a new
// srcA costs nothing; within a k-step the order of the additions is immaterial to the result's error (the accumulator
// already holds the sum of the earlier k-steps).
#ifdef NFA_ABL_NO_MFMA5   // (measurement builds)
#define NFA_MFMA5(acc, ah, al, ar, bh, bl, br) asm volatile("" :: "v"(ah), "v"(al), "v"(ar), "v"(bh), "v"(bl), "v"(br))
#else
#define NFA_MFMA5(acc, ah, al, ar, bh, bl, br)                                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, br, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0)
#endif

// the pieces of a k-step: 8 values per lane and piece = one register quad each
struct Pieces {
    uvec4 h, l, r;
};

// two fp32 values x `scale` (a power of two) -> packed f16 pairs of the three pieces.
//   hi = RN16(v s)                         v_fma_mixlo / mixhi_f16: fp32 fma, result rounded to f16
//   t  = v s - hi                          v_fma_mix_f32 with the f16 `hi` as negated addend: exact (<= 13 bits)
//   lo = RN16(t)
//   r  = RN16(t - lo)                      exact in fp32; one bit (or zero) in f16 while it is >= 2^-24
// |v s| >= 65520 gives hi = inf, t = -inf, lo = -inf, r = NaN: the overflow poisons every sum it enters, the row
// block is flagged and redone by the exact kernel.  (One asm block: hipcc puts an `s_nop 0` between two adjacent asm
// statements; early-clobber everywhere: every output is written before the last input is read.)
// The conversions are `asm volatile`: a plain asm statement is free to move, and hipcc's scheduler moved these across the
// stream's (volatile) waits and barriers to right behind the MFMAs whose accumulators they read -- instructions inside an
// asm statement are invisible to the hazard recogniser, which therefore pads nothing between an MFMA and an asm block
// that reads its result: the first build had every GEMM's last products missing from some values (logits 2^-12 off,
// bit-reproducible; found by tests/test_gpu_logits.py, round 6).  Volatile keeps them in program order behind the GEMM's
// last barrier, and tile_to_pieces puts the matrix pipe's write-back distance (11 wait states behind an 8-pass MFMA)
// in front of the first block that reads a tile.  NFA_K8X_NO_ASM (measurement builds): the same arithmetic in C++.
#define NFA_K8X_PRE ""
#define NFA_K8X_ASM asm volatile
__device__ __forceinline__ void split3_scaled(float v0, float v1, float scale, unsigned& hi, unsigned& lo, unsigned& rr) {
#if defined(NFA_K8X_NO_ASM)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float x0 = v0 * scale, x1 = v1 * scale;
    const h2 h = {(_Float16)x0, (_Float16)x1};
    const float t0 = x0 - (float)h[0], t1 = x1 - (float)h[1];
    const h2 l = {(_Float16)t0, (_Float16)t1};
    const h2 r = {(_Float16)(t0 - (float)l[0]), (_Float16)(t1 - (float)l[1])};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
    rr = __builtin_bit_cast(unsigned, r);
#else
    unsigned h, l, r;
    float t0, t1;
    NFA_K8X_ASM(NFA_K8X_PRE
        "v_fma_mixlo_f16 %0, %5, %7, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %6, %7, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mix_f32 %3, %5, %7, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %6, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %3, 1.0, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %4, 1.0, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %3, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %4, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l), "=&v"(r), "=&v"(t0), "=&v"(t1)
        : "v"(v0), "v"(v1), "v"(scale));
    hi = h;
    lo = l;
    rr = r;
#endif
}

// the fp32 value of a piece triple (exact: hi + lo has at most 23 bits, + r at most 24) times `mul`, plus `add`:
// the skip connection, acc = bias + T x h
__device__ __forceinline__ void pieces_fma2(unsigned h, unsigned l, unsigned r, float mul, float& acc0, float& acc1) {
    float t0, t1;
#if defined(NFA_K8X_NO_ASM)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 hh = __builtin_bit_cast(h2, h), ll = __builtin_bit_cast(h2, l), rr = __builtin_bit_cast(h2, r);
    t0 = ((float)hh[0] + (float)ll[0]) + (float)rr[0];
    t1 = ((float)hh[1] + (float)ll[1]) + (float)rr[1];
#else
    NFA_K8X_ASM(NFA_K8X_PRE
        "v_fma_mix_f32 %0, %2, 1.0, %3 op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %1, %2, 1.0, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %0, %4, 1.0, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %4, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t0), "=&v"(t1)
        : "v"(h), "v"(l), "v"(r));
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

#define NFA_K8X_FRAGS(cur, i_h, i_l, i_r)                                   \
    const f16x8 ah = __builtin_bit_cast(f16x8, (cur)[(i_h) * 64]);          \
    const f16x8 al = __builtin_bit_cast(f16x8, (cur)[(i_l) * 64]);          \
    const f16x8 ar = __builtin_bit_cast(f16x8, (cur)[(i_r) * 64])

// k-major GEMM (all four output tiles accumulate together): out^T[128 x 32 samples] += W[128 x 16 NKS] x act^T; one
// stage ([4 tiles][3 pieces][64 lanes] x 16 bytes) per k-step.  RELU: applied to the input pieces on the fly (the
// pieces themselves stay: they are the residual stream)
template <bool RELU, int NKS>
__device__ __forceinline__ void gemm_kmajor(f32x16 (&acc)[4], const Pieces (&p)[8], WeightStream& sm, int lane) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        stream_request(sm);
        const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
        Pieces b = p[ks];
        if (RELU) relu_pieces(b);
        const f16x8 bh = __builtin_bit_cast(f16x8, b.h), bl = __builtin_bit_cast(f16x8, b.l),
                    br = __builtin_bit_cast(f16x8, b.r);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            NFA_K8X_FRAGS(cur, t * 3 + 0, t * 3 + 1, t * 3 + 2);
            NFA_MFMA5(acc[t], ah, al, ar, bh, bl, br);
        }
        stream_advance(sm);
    }
}

// one 32-row output tile of a 128-wide layer: acc += W_tile[32 x 128] x act^T; two stages of [3 pieces][4 k-steps]
// [64 lanes] x 16 bytes
__device__ __forceinline__ void gemm_tile(f32x16& acc, const Pieces (&p)[8], WeightStream& sm, int lane) {
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        stream_request(sm);
        const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int ks = hs * 4 + k4;
            const f16x8 bh = __builtin_bit_cast(f16x8, p[ks].h), bl = __builtin_bit_cast(f16x8, p[ks].l),
                        br = __builtin_bit_cast(f16x8, p[ks].r);
            NFA_K8X_FRAGS(cur, 0 * 4 + k4, 1 * 4 + k4, 2 * 4 + k4);
            NFA_MFMA5(acc, ah, al, ar, bh, bl, br);
        }
        stream_advance(sm);
    }
}

}  // namespace k8x
}  // namespace nfa
