// Pieces shared by the kernels that compute conditioner GEMMs on the matrix cores inside the
// spline coupling kernel (rqs_fused_linear.hip: K7 / K7b, rqs_resnet.hip: K8).
#pragma once

#include "rqs_math.hpp"

// Order of the six products.  A v_mfma_f32_32x32x16 whose SECOND operand (srcB: the activation pieces here) differs from
// the previous instruction's takes ~49 cycles instead of 32 when the two are issued back to back (round 4,
// tools/mfma_toggle_probe.hip: srcB new on every MFMA 1 192 TFLOP/s at 983 W -- not the power cap --, new on every
// 4th 1 677 at the cap; a new srcA costs nothing).  NFA_BF16X3_ORDER 1 groups the products by their srcB piece
// (bh, bh, bh, bm, bm, bl: three changes per k-step instead of six); 0 is the round-1 order (small terms first).
#ifndef NFA_BF16X3_ORDER
#define NFA_BF16X3_ORDER 0
#endif
#ifdef NFA_ABL_NO_MFMA6   // (measurement builds)
#define NFA_MFMA6(acc, ah, am, al, bh, bm, bl) asm volatile("" :: "v"(ah), "v"(am), "v"(al), "v"(bh), "v"(bm), "v"(bl))
#elif NFA_BF16X3_ORDER == 0
#define NFA_MFMA6(acc, ah, am, al, bh, bm, bl)                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0)
#else
#define NFA_MFMA6(acc, ah, am, al, bh, bm, bl)                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0)
#endif


// NFA_MFMA6_SPLIT (round 6; K11): the leading product accumulates alone, the five small ones in `small` -- the roundings of
// the small terms then happen at THEIR magnitude (2^-8 of the sum) and the main accumulator is rounded 8 times per 128-term dot
// product instead of 48: measured error against float64 0.39 x the single accumulator's (rms 2.28e-8 against 5.83e-8 and the
// sequential fp32 fma chain's 5.26e-8, profiles/r6/gemm_numerics_probe.txt); more products change nothing (9 products: 5.83e-8).
#define NFA_MFMA6_SPLIT(acc, small, ah, am, al, bh, bm, bl)                             \
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, small, 0, 0, 0);            \
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, small, 0, 0, 0);            \
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, small, 0, 0, 0);            \
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, small, 0, 0, 0);            \
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, small, 0, 0, 0);            \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0)

namespace nfa {

// Activation of the conditioner's residual blocks (nn/nets/resnet.py:27 `activation=F.relu`; round 4: the other
// functions users pass there).  Codes 0 / 1 are what a `bool RELU` template argument converts to.
//   leaky_relu: F.leaky_relu's default slope 0.01 -- x > 0 ? x : 0.01 x = max(x, 0.01 x), bit for bit torch's
//   elu:        F.elu's default alpha 1 -- x > 0 ? x : exp(x) - 1 (torch's CPU kernel subtracts, no expm1)
//   tanh:       (e^{2x} - 1) / (e^{2x} + 1) on v_exp_f32 / v_rcp_f32 with one residual correction: absolute error <= 2e-7
//               (the activation is the B operand of the next GEMM at magnitude <= 1: an absolute bound is what counts)
// NaN propagates through all of them.
enum : int { kActNone = 0, kActRelu = 1, kActLeakyRelu = 2, kActElu = 3, kActTanh = 4 };

// ReLU and leaky ReLU commute with a positive factor (K8h keeps its accumulators at a power-of-two scale and takes it
// out while it converts: act(v) x scale); ELU and tanh do not: they are applied to v x scale
constexpr bool activation_is_homogeneous(int act) { return act <= kActLeakyRelu; }

template <int ACT>
__device__ __forceinline__ float activate(float v) {
    if constexpr (ACT == kActRelu) {
        return v < 0.0f ? 0.0f : v;   // (NaN stays NaN)
    } else if constexpr (ACT == kActLeakyRelu) {
        return __builtin_fmaxf(v, 0.01f * v);
    } else if constexpr (ACT == kActElu) {
        const float e = __builtin_amdgcn_exp2f(v * 1.44269502162933349609375f) - 1.0f;
        return v > 0.0f ? v : e;
    } else if constexpr (ACT == kActTanh) {
        float t = v * 2.8853900432586669921875f;   // 2 log2(e)
        t = t > 126.0f ? 126.0f : t;                // (1 + 2^t stays finite; tanh is 1 to fp32 there; NaN stays NaN)
        const float e = __builtin_amdgcn_exp2f(t);
        const float dn = e + 1.0f;
        const float r0 = __builtin_amdgcn_rcpf(dn);
        const float r = __builtin_fmaf(__builtin_fmaf(-dn, r0, 1.0f), r0, r0);
        return (e - 1.0f) * r;   // (e - 1 is exact near e = 1, where 1 - 2 r would cancel)
    } else {
        return v;
    }
}


// debug aid (tools/k7_trace.py, tools/k8_trace.py): device buffer of 512 uint64 that lane 0 of
// wave 0 of workgroups 0 and 256 fills with cycle-counter stamps at phase boundaries; null = off
extern unsigned long long* g_k7_trace;
#define NFA_STAMP() if (tr && ti < 250) tr[ti++] = __builtin_readcyclecounter();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vec4f __attribute__((ext_vector_type(4)));

// Lane-private view of one feature's 24 (23 + pad) logits inside the three accumulators of a group:
// a lane's 48 accumulator registers are, in order, the logits of its two features (the host packs
// the weight rows so; see pack order in include/nflows_amd.h).
#define NFA_K7_FEATURE_A(p, a0, a1)                                                   \
    float p[24] = {a0[0], a0[1], a0[2],  a0[3],  a0[4],  a0[5],  a0[6],  a0[7],       \
                   a0[8], a0[9], a0[10], a0[11], a0[12], a0[13], a0[14], a0[15],      \
                   a1[0], a1[1], a1[2],  a1[3],  a1[4],  a1[5],  a1[6],  a1[7]}
#define NFA_K7_FEATURE_B(p, a1, a2)                                                   \
    float p[24] = {a1[8], a1[9], a1[10], a1[11], a1[12], a1[13], a1[14], a1[15],      \
                   a2[0], a2[1], a2[2],  a2[3],  a2[4],  a2[5],  a2[6],  a2[7],       \
                   a2[8], a2[9], a2[10], a2[11], a2[12], a2[13], a2[14], a2[15]}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float vec2f __attribute__((ext_vector_type(2)));

constexpr int kWTileVec4 = 3 * 8 * 64;  // one weight tile: [piece][k-step][lane] x 16 bytes

// (No packed fp32 arithmetic beside MFMAs: see rqs_resnet_f16.hip.  The residuals are computed per
// element and pinned against re-vectorisation; the MFMA files are built with -fno-slp-vectorize.)
__device__ __forceinline__ void split3(vec2f v, bf16x2& hi, bf16x2& mid, bf16x2& lo) {
    hi = __builtin_convertvector(v, bf16x2);
    const vec2f h = __builtin_convertvector(hi, vec2f);
    float a0 = v[0] - h[0], a1 = v[1] - h[1];
    asm volatile("" : "+v"(a0));
    asm volatile("" : "+v"(a1));
    mid = __builtin_convertvector(vec2f{a0, a1}, bf16x2);
    const vec2f m = __builtin_convertvector(mid, vec2f);
    float b0 = a0 - m[0], b1 = a1 - m[1];
    asm volatile("" : "+v"(b0));
    asm volatile("" : "+v"(b1));
    lo = __builtin_convertvector(vec2f{b0, b1}, bf16x2);
}

__device__ __forceinline__ bf16x8 join4(bf16x2 a, bf16x2 b, bf16x2 c, bf16x2 d) {
    return bf16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// Column bookkeeping of one coupling layer with its neighbouring permutations folded in, built
// once per workgroup in LDS.  Layer column c is read from input column src[c] and written to
// output position dst[c]; dinv is the inverse of dst.
struct LayerTables {
    int dinv[128];   // layer column stored at output position p
    int slot[128];   // index of a transformed column in transform_idx
    int src[128], dst[128];
    int tsrc[64];    // input column of transformed feature f
    int isrc[64];    // input column of identity feature i (K8 only)
    unsigned char ist[128];  // 1: column is transformed
};

// Returns status bits (NFA_STATUS_BAD_INDEX).  Ends with a workgroup barrier.
__device__ __forceinline__ int build_layer_tables(LayerTables& t, const int64_t* perm, const int64_t* scatter,
                                                  const int64_t* tidx, const int64_t* iidx, int D, int dt,
                                                  int di, int tid, int nthreads) {
    int st = 0;
    for (int c = tid; c < D; c += nthreads) {
        int src = c, dst = c;
        if (perm) {
            const int64_t p = perm[c];
            if (p < 0 || p >= D) st |= NFA_STATUS_BAD_INDEX;
            src = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        if (scatter) {
            const int64_t p = scatter[c];
            if (p < 0 || p >= D) st |= NFA_STATUS_BAD_INDEX;
            dst = (int)(p < 0 ? 0 : (p >= D ? D - 1 : p));
        }
        t.src[c] = src;
        t.dst[c] = dst;
        t.ist[c] = 0;
        t.slot[c] = 0;
    }
    __syncthreads();
    for (int c = tid; c < D; c += nthreads) t.dinv[t.dst[c]] = c;
    if (tid < dt) {
        const int64_t v = tidx[tid];
        if (v < 0 || v >= D) st |= NFA_STATUS_BAD_INDEX;
        const int col = (int)(v < 0 ? 0 : (v >= D ? D - 1 : v));
        t.tsrc[tid] = t.src[col];
        t.ist[col] = 1;
        t.slot[col] = tid;
    }
    if (iidx && tid < di) {
        const int64_t v = iidx[tid];
        if (v < 0 || v >= D) st |= NFA_STATUS_BAD_INDEX;
        t.isrc[tid] = t.src[(int)(v < 0 ? 0 : (v >= D ? D - 1 : v))];
    }
    __syncthreads();
    return st;
}

// One wave writes its 32 output rows contiguously: position p holds layer column c = dinv[p];
// transformed columns come from the wave's LDS y tile, the others are copied bit-exactly from the
// inputs (gathered through the fused permutation).  Eight gathers are in flight per store batch.
__device__ __forceinline__ void assemble_rows(const LayerTables& t, const float* s_y, int ystride,
                                              const float* x, float* out, int64_t row0, int D,
                                              FastDiv div_D, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int e0 = lane; e0 < 32 * D; e0 += kWave * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * kWave;
            v[u] = 0.0f;
            if (e < 32 * D) {
                const int rr = (int)fastdiv((uint32_t)e, div_D);
                const int c = t.dinv[e - rr * D];
                v[u] = t.ist[c] ? s_y[rr * ystride + t.slot[c]] : x[(row0 + rr) * D + t.src[c]];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * kWave;
            if (e < 32 * D) out[row0 * D + e] = v[u];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- standard-normal epilogue of the whole-layer kernels (NFA_FLAG_STANDARD_NORMAL_LOG_PROB) ----
// Sum of squares of the D values of row r of a wave's [D][pad] row tile: the lane pair (r, r + 32)
// shares the row, lane-half `half` takes slots [half D/2, (half + 1) D/2) (D % 4 == 0), the halves
// are added across the pair.  Every slot holds one output value whatever column it ends up in, so
// this is sum_j z_j^2 (in slot order, fp32; normal.py:31 sums in column order: same rounding class).
__device__ __forceinline__ float tile_row_sumsq(const float* s_row, int D, int half, int r, int pad = 33) {
    // lane-half 0 sums slots [0, n), lane-half 1 slots [n, D), n = ceil(D / 2); two accumulators each (for D % 4 == 0
    // the same order as ever)
    const int n = (D + 1) >> 1;
    const int lo = half ? n : 0, hi = half ? D : n;
    const float* p = s_row + r;
    float a0 = 0.0f, a1 = 0.0f;
    for (int j = lo; j < hi; j += 2) {
        const float v0 = p[j * pad], v1 = j + 1 < hi ? p[(j + 1) * pad] : 0.0f;
        a0 = __builtin_fmaf(v0, v0, a0);
        a1 = __builtin_fmaf(v1, v1, a1);
    }
    float s = a0 + a1;
    s += __shfl_xor(s, 32, kWave);
    return s;
}

static inline float standard_normal_log_z(int features) {
    return (float)(0.5 * (double)features * 1.8378770664093453);   // log(2 pi)
}

// NFA_FLAG_STANDARD_NORMAL_LOG_PROB belongs to the forward pass; NFA_FLAG_SKIP_OUTPUTS needs it
static inline bool density_flags_valid(int32_t flags) {
    if ((flags & NFA_FLAG_SKIP_OUTPUTS) && !(flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB)) return false;
    if ((flags & NFA_FLAG_PAD_COLUMNS_MASK) && !(flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB)) return false;
    if ((flags & NFA_FLAG_STANDARD_NORMAL_LOG_PROB) && (flags & NFA_FLAG_INVERSE)) return false;
    return true;
}

// columns the standard-normal epilogue sums over: all but the trailing NFA_FLAG_PAD_COLUMNS
static inline int density_columns(int32_t flags, int features) {
    return features - ((flags & NFA_FLAG_PAD_COLUMNS_MASK) >> NFA_FLAG_PAD_COLUMNS_SHIFT);
}

}  // namespace nfa
