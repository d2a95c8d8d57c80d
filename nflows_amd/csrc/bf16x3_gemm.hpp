// The split-bf16 GEMM machinery shared by the whole-layer kernels on the bf16 matrix pipe
// (rqs_resnet.hip: ResidualNet conditioner + spline layer; affine_mlp.hip: MLP conditioner + affine
// layer): the LDS-DMA weight ring, the six-product MFMA step, k-major and tile-major GEMMs on
// activations kept as three bf16 pieces in registers, accumulator <-> pieces conversions.
#pragma once

#include "fused_common.hpp"

namespace nfa {

constexpr int kStageVec4 = 768;    // 12 KB: [4 tiles][3 pieces][64 lanes] or [3 pieces][4 k-steps][64 lanes] x 16 B
constexpr int kRing = 3;
constexpr int kRowPad = 33;        // row tile: [output position][33]: conflict-free both ways
constexpr int kTabId = 0, kTabTr = 64, kTabLayer = 128;   // per-layer table: identity slots, transformed slots

// Weight stream through the LDS ring.  Stage s lives in slot s % 3; while stage s is consumed,
// stage s+1 has landed (or is landing) and stage s+2 is being requested.
struct WeightStream {
    const vec4f* w;
    vec4f* ring;
    int slot;        // ring slot of the stage being consumed
    int fetch;       // stage index (in the layer's list) to request next
    int num_stages;
    int tid;
};

// Addresses are kept in the shape "wave-uniform base + per-lane 32-bit offset": the stage's base
// (global) and the slot's base (LDS, goes to M0) are scalar arithmetic, the three per-lane byte
// offsets are loop-invariant registers -- no vector address arithmetic per request (it was ~5
// VALU instructions per request, 15 % of the kernel's VALU instructions).
__device__ __forceinline__ void stream_request(WeightStream& sm) {
    const int dst_slot = sm.slot >= 1 ? sm.slot - 1 : kRing - 1;  // (slot + 2) % 3
    const char* stage = reinterpret_cast<const char*>(sm.w) + (size_t)sm.fetch * (kStageVec4 * 16);
    const int wave = __builtin_amdgcn_readfirstlane(sm.tid >> 6);
    char* slot = reinterpret_cast<char*>(sm.ring) + dst_slot * (kStageVec4 * 16) + wave * (kWave * 16);
    const unsigned lane_off = (unsigned)sm.tid * 16u;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((stage + i * kBlock * 16) + lane_off),
            (__attribute__((address_space(3))) void*)(slot + i * kBlock * 16), 16, 0, 0);
    sm.fetch = (sm.fetch + 1 == sm.num_stages) ? 0 : sm.fetch + 1;
}

// end of a stage: the next stage's three LDS-DMA requests of this wave have landed (the three
// younger ones of the stage after it may still be in flight), every wave is done reading
__device__ __forceinline__ void stream_advance(WeightStream& sm) {
    asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    sm.slot = (sm.slot + 1 == kRing) ? 0 : sm.slot + 1;
}

// (NFA_MFMA6, the six products of a k-step, and their order: fused_common.hpp)

typedef unsigned uvec4 __attribute__((ext_vector_type(4)));

// ReLU applied to a value given as bf16 pieces: all three are cleared where the leading piece is
// negative and not a NaN (bf16 bit patterns 0x8000..0xFF80 = int16 <= -128), so that NaNs keep
// propagating like torch.relu's.  Three packed-int16 instructions make the mask of two values.
__device__ __forceinline__ void relu_pieces(bf16x8& h, bf16x8& m, bf16x8& l) {
    uvec4 hw = __builtin_bit_cast(uvec4, h), mw = __builtin_bit_cast(uvec4, m), lw = __builtin_bit_cast(uvec4, l);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned keep;
        // (volatile: a pure statement may be hoisted or merged by the compiler, and ReLU'd copies
        // kept alive beside the originals would not fit the register file)
        asm volatile("v_pk_min_i16 %0, %1, 0\n\t"
            "v_pk_add_i16 %0, %0, %2\n\t"
            "v_pk_ashrrev_i16 %0, %3, %0\n\t"
            "v_not_b32 %0, %0"
            : "=&v"(keep)
            : "v"(hw[i]), "s"(0x007F007Fu), "s"(0x000F000Fu));  // (packed inline constants fill one half only)
        hw[i] &= keep;
        mw[i] &= keep;
        lw[i] &= keep;
    }
    h = __builtin_bit_cast(bf16x8, hw);
    m = __builtin_bit_cast(bf16x8, mw);
    l = __builtin_bit_cast(bf16x8, lw);
}

// k-major GEMM (all four output tiles accumulate together, the input pieces of a k-step are dead
// after it): out^T[128 x 32 samples] += W[128 x 16*NKS] x act^T; one stage ([4 tiles][3 pieces]
// [64 lanes] x 16 bytes) per k-step
// any activation of a value given as bf16 pieces: the fp32 value back (two exact additions of three pieces that
// do not overlap), the activation, three new pieces -- ~15 VALU instructions per value where ReLU's sign mask takes
// two; only the exact kernel's first Linear of a block pays it, and only for the other activations
template <int ACT>
__device__ __forceinline__ void activate_pieces(bf16x8& h, bf16x8& m, bf16x8& l) {
    bf16x2 hh[4], mm[4], ll[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
        const float v0 = activate<ACT>(((float)h[2 * j2] + (float)m[2 * j2]) + (float)l[2 * j2]);
        const float v1 = activate<ACT>(((float)h[2 * j2 + 1] + (float)m[2 * j2 + 1]) + (float)l[2 * j2 + 1]);
        split3(vec2f{v0, v1}, hh[j2], mm[j2], ll[j2]);
    }
    h = join4(hh[0], hh[1], hh[2], hh[3]);
    m = join4(mm[0], mm[1], mm[2], mm[3]);
    l = join4(ll[0], ll[1], ll[2], ll[3]);
}

// ACT: kActNone (false) / kActRelu (true) / the other activations (fused_common.hpp), applied to the input pieces
template <int ACT, int NKS>
__device__ __forceinline__ void gemm_kmajor(f32x16 (&acc)[4], const bf16x8 (&ph)[8], const bf16x8 (&pm)[8],
                                            const bf16x8 (&pl)[8], WeightStream& sm, int lane) {
#ifdef NFA_BF16X3_SPLIT_ACC   // (K11: see NFA_MFMA6_SPLIT)
    f32x16 small[4] = {{0}, {0}, {0}, {0}};
#endif
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        stream_request(sm);
        const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
        bf16x8 bh = ph[ks], bm = pm[ks], bl = pl[ks];
        if constexpr (ACT == kActRelu) relu_pieces(bh, bm, bl);  // (the input pieces themselves stay: skip connection)
        else if constexpr (ACT != kActNone) activate_pieces<ACT>(bh, bm, bl);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(t * 3 + 0) * 64]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(t * 3 + 1) * 64]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(t * 3 + 2) * 64]);
#ifdef NFA_BF16X3_SPLIT_ACC
            NFA_MFMA6_SPLIT(acc[t], small[t], ah, am, al, bh, bm, bl);
#else
            NFA_MFMA6(acc[t], ah, am, al, bh, bm, bl);
#endif
        }
        stream_advance(sm);
    }
#ifdef NFA_BF16X3_SPLIT_ACC
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += small[t];
#endif
}

// one 32-row output tile of a 128-wide layer: acc += W_tile[32 x 128] x act^T, act given as pieces
// (ReLU applied to them on the fly if RELU); two stages of [3 pieces][4 k-steps][64 lanes] x 16 bytes
template <bool RELU>
__device__ __forceinline__ void gemm_tile(f32x16& acc, const bf16x8 (&ph)[8], const bf16x8 (&pm)[8],
                                          const bf16x8 (&pl)[8], WeightStream& sm, int lane) {
#ifdef NFA_BF16X3_SPLIT_ACC
    f32x16 small = {0};
#endif
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        stream_request(sm);
        const vec4f* cur = sm.ring + sm.slot * kStageVec4 + lane;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int ks = hs * 4 + k4;
            bf16x8 bh = ph[ks], bm = pm[ks], bl = pl[ks];
            if (RELU) relu_pieces(bh, bm, bl);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, cur[(0 * 4 + k4) * 64]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, cur[(1 * 4 + k4) * 64]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, cur[(2 * 4 + k4) * 64]);
#ifdef NFA_BF16X3_SPLIT_ACC
            NFA_MFMA6_SPLIT(acc, small, ah, am, al, bh, bm, bl);
#else
            NFA_MFMA6(acc, ah, am, al, bh, bm, bl);
#endif
        }
        stream_advance(sm);
    }
#ifdef NFA_BF16X3_SPLIT_ACC
    acc += small;
#endif
}

// accumulator tile t, registers 8*hk .. 8*hk+7  ->  pieces of k-step 2t + hk
template <int ACT>
__device__ __forceinline__ void tile_to_pieces(const f32x16& a, bf16x8& h0, bf16x8& m0, bf16x8& l0,
                                               bf16x8& h1, bf16x8& m1, bf16x8& l1) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = activate<ACT>(a[q]);  // (ReLU: NaN stays NaN)
    bf16x2 hh[8], mm[8], ll[8];
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) split3(vec2f{v[q2 * 2], v[q2 * 2 + 1]}, hh[q2], mm[q2], ll[q2]);
    h0 = join4(hh[0], hh[1], hh[2], hh[3]);
    m0 = join4(mm[0], mm[1], mm[2], mm[3]);
    l0 = join4(ll[0], ll[1], ll[2], ll[3]);
    h1 = join4(hh[4], hh[5], hh[6], hh[7]);
    m1 = join4(mm[4], mm[5], mm[6], mm[7]);
    l1 = join4(ll[4], ll[5], ll[6], ll[7]);
}

// value of the pieces of one k-step, added to 8 accumulator registers (the skip connection)
__device__ __forceinline__ void add_pieces(f32x16& a, int q0, const bf16x8& h, const bf16x8& m, const bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[q0 + j] += ((float)h[j] + (float)m[j]) + (float)l[j];
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

}  // namespace nfa
