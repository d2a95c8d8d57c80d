// Per-spline arithmetic of the rational-quadratic kernels (shared by the forward kernels in
// rqs.hip and the backward kernels in rqs_bwd.hip).  gfx950 only.
#pragma once

#include "common.hpp"

#include <math.h>

namespace nfa {

struct RqsDev {
    int K;          // bins
    int P;          // params per spline: 2K + nd
    int nd;         // derivative logits per spline: K-1 (linear tails) / K+1, or more (extras unused)
    int linear;     // 1: linear tails
    float left, right, bottom, top;
    float span_w, span_h;          // (float)(right-left), (float)(top-bottom)
    float right_eps, top_eps;      // last knot + 1e-6 (searchsorted)
    float min_w, min_h, min_d;
    float om_w, om_h;              // (float)(1 - min*K)
    float beta, tail_logit, divisor, rdivisor;  // rdivisor = RN(1/divisor)
};

// Storage of K values per lane: registers when K is a compile-time constant, the lane's own LDS
// words (updated in place) otherwise.
template <int KT>
struct Slots {
    float v[KT];
    __device__ __forceinline__ void bind(float*) {}
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <>
struct Slots<0> {
    float* s;
    __device__ __forceinline__ void bind(float* p) { s = p; }
    __device__ __forceinline__ float get(int i) const { return s[i]; }
    __device__ __forceinline__ void set(int i, float x) { s[i] = x; }
};

// ---- arithmetic building blocks -------------------------------------------------------------
// a / b given r = RN(1/b): one product, its exact residual (fma) and one correction (fma).
// This is the final step of the IEEE division algorithm; with a correctly rounded r it returns
// the correctly rounded quotient (no scaling needed here: |a|, |b| are far from the fp32 limits).
__device__ __forceinline__ float div_with_rcp(float a, float b, float r) {
    const float q = a * r;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(e, r, q);
}

// RN(1/b) for b in the normal range: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_refined(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
}

// a / b without the denormal / overflow scaling of the generic IEEE expansion (operands here are
// bin widths, heights and derivatives: well inside the normal range).
__device__ __forceinline__ float div_normal(float a, float b) {
    return div_with_rcp(a, b, rcp_refined(b));
}

// log(u) for normal u: v_log_f32 (log2, 1 ulp) times ln2 carried in two floats
__device__ __forceinline__ float log_normal(float u) {
    const float kLn2Hi = 0.693145751953125f;          // ln2, low 11 mantissa bits cleared
    const float kLn2Lo = 1.42860682030941723212e-06f;  // ln2 - kLn2Hi
    const float r = __builtin_amdgcn_logf(u);
    return __builtin_fmaf(r, kLn2Lo, r * kLn2Hi);
}

// log1p(t), t >= 0: log(u) with u = RN(1 + t) plus the first-order term for the rounding of u.
// (ocml's log1pf is ~120 VALU instructions; this is 9.)
__device__ __forceinline__ float log1p_nonneg(float t) {
    const float u = 1.0f + t;
    const float c = t - (u - 1.0f);  // exact: what the addition dropped
    return __builtin_fmaf(c, __builtin_amdgcn_rcpf(u), log_normal(u));
}

// exp(x) for x <= ~88 without range handling: 2^(x*log2e) with the product carried in two
// floats; v_exp_f32 (1 ulp) on the high part, first-order correction for the low part.
// Results below 2^-126 flush to zero (they are added to a softmax denominator >= 1).
__device__ __forceinline__ float exp_noclamp(float x) {
    const float kLog2e = 1.44269502162933349609375f;       // RN(log2(e))
    const float kLog2eLo = 1.925963033500011e-08f;          // log2(e) - kLog2e
    const float kLn2 = 0.693147182464599609375f;
    const float hi = x * kLog2e;
    float lo = __builtin_fmaf(x, kLog2e, -hi);
    lo = __builtin_fmaf(x, kLog2eLo, lo);
    const float e0 = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(e0, lo * kLn2, e0);
}

// softmax numerators exp(u_i - max) in place, returns the fp32 denominator.
// (aten's softmax sums the numerators in fp32 in a vector-lane order that depends on the host
// ISA; a balanced tree is the closest ISA-independent choice.)
template <int KT>
__device__ __forceinline__ float softmax_numerators(Slots<KT>& e, const float* logits, int K,
                                                    float divisor, float rdivisor) {
#pragma clang fp contract(off)
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < (KT > 0 ? KT : K); ++i) {
        float u = logits[i];
        if (divisor != 0.0f) u = div_with_rcp(u, divisor, rdivisor);
        e.set(i, u);
        m = fmaxf(m, u);
    }
    if (KT == 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            t[i] = exp_noclamp(e.get(i) - m);
            e.set(i, t[i]);
        }
        return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    double s = 0.0;  // runtime K: a long sequential fp32 sum would drift; double is exact enough
    for (int i = 0; i < (KT > 0 ? KT : K); ++i) {
        const float ex = exp_noclamp(e.get(i) - m);
        e.set(i, ex);
        s += (double)ex;
    }
    return (float)s;
}

// Walks the K bins, rebuilding knot_i / knot_{i+1} on the fly.
//   SEARCH : k <- last i with x >= knot_i (== count-1 for monotone knots), picks that bin's knots
//   !SEARCH: picks the knots of the given bin k
// The prefix sums are accumulated in double and rounded to fp32 per prefix, which is what
// aten's CPU cumsum does for float tensors (the double sums are exact for these magnitudes).
template <int KT, bool SEARCH>
__device__ __forceinline__ void walk_bins(const Slots<KT>& e, int K, float denom, float minbin,
                                          float om, float span, float lo, float hi, float x, int& k,
                                          float& knot_lo, float& knot_hi) {
#pragma clang fp contract(off)
    const float rden = rcp_refined(denom);
    double acc = 0.0;
    float prev = lo;
#pragma unroll
    for (int i = 0; i < (KT > 0 ? KT : K); ++i) {
        const float p = div_with_rcp(e.get(i), denom, rden);
        const float w = minbin + om * p;
        acc += (double)w;
        const float c = (float)acc;
        const float next = (i == (KT > 0 ? KT : K) - 1) ? hi : span * c + lo;
        const bool take = SEARCH ? (x >= prev) : (i == k);
        if (take) {
            if (SEARCH) k = i;
            knot_lo = prev;
            knot_hi = next;
        }
        prev = next;
    }
}

__device__ __forceinline__ float softplus_beta(float x, float beta) {
#pragma clang fp contract(off)
    const float xb = x * beta;
    const float sp = log1p_nonneg(exp_noclamp(xb));
    return xb > 20.0f ? x : (beta == 1.0f ? sp : sp / beta);
}

// The map inside one bin: knots (cw0, cw1) -> (ch0, ch1), end derivatives d0, d1.
// forward: rational_quadratic.py:162-181; inverse (quadratic root): :132-160.
template <bool INVERSE>
__device__ __forceinline__ int rqs_bin_eval(float x, float cw0, float cw1, float ch0, float ch1, float d0,
                                            float d1, float& y, float& lad) {
#pragma clang fp contract(off)
    const float in_w = cw1 - cw0;
    const float in_h = ch1 - ch0;
    const float r_w = rcp_refined(in_w);
    const float delta = div_with_rcp(in_h, in_w, r_w);
    const float s = (d0 + d1) - 2.0f * delta;
    int status = 0;

    if (INVERSE) {
        const float yc = x - ch0;
        const float a = yc * s + in_h * (delta - d0);
        const float b = in_h * d0 - yc * s;
        const float c = (-delta) * yc;
        const float disc = b * b - (4.0f * a) * c;
        if (!(disc >= 0.0f)) status = NFA_STATUS_NEG_DISCRIMINANT;
        const float root = div_normal(2.0f * c, (-b) - sqrtf(disc));
        y = root * in_w + cw0;
        const float t1mt = root * (1.0f - root);
        const float den = delta + s * t1mt;
        const float omr = 1.0f - root;
        const float dnum = (delta * delta) * ((d1 * (root * root) + (2.0f * delta) * t1mt) + d0 * (omr * omr));
        lad = -(log_normal(dnum) - 2.0f * log_normal(den));
    } else {
        const float theta = div_with_rcp(x - cw0, in_w, r_w);
        const float t1mt = theta * (1.0f - theta);
        const float num = in_h * (delta * (theta * theta) + d0 * t1mt);
        const float den = delta + s * t1mt;
        y = ch0 + div_normal(num, den);
        const float omt = 1.0f - theta;
        const float dnum = (delta * delta) * ((d1 * (theta * theta) + (2.0f * delta) * t1mt) + d0 * (omt * omt));
        lad = log_normal(dnum) - 2.0f * log_normal(den);
    }
    return status;
}

// One spline evaluation.  `sl` = this lane's P logits in LDS (may be clobbered when KT == 0).
//   LINEAR = true: linear tails, box [-B, B]^2 (B = sp.right); only sp.right / span_w / right_eps
//   and the per-side minimums are read, which keeps the kernel's scalar-register footprint down.
//   REGS = true: `sl` points at a per-lane register array (KT > 0, LINEAR): the two derivative
//   logits are picked with a select chain instead of a dynamic index (which would force the
//   array into scratch memory).
//   `bin` (optional, a per-lane address): receives the bin the search chose -- what
//   torchutils.searchsorted (utils/torchutils.py:134-136) returns for this element: 0 .. K-1, K when the
//   input reaches the nudged last knot (the reference's gather then fails; here OUTSIDE_DOMAIN) -- or -1
//   for an element the reference never searches (linear tails / outside the domain / NaN).  Stored the
//   moment it is known; with the default nullptr the stores fold away.
template <int KT, bool INVERSE, bool LINEAR, bool REGS = false>
__device__ __forceinline__ int rqs_eval(float x, float* sl, const RqsDev& sp, float& y, float& lad,
                                        int* bin = nullptr) {
#pragma clang fp contract(off)
    const int K = KT > 0 ? KT : sp.K;
    const float left = LINEAR ? -sp.right : sp.left;
    const float right = sp.right;
    const float bottom = LINEAR ? -sp.right : sp.bottom;
    const float top = LINEAR ? sp.right : sp.top;
    const float span_w = sp.span_w;
    const float span_h = LINEAR ? sp.span_w : sp.span_h;
    const float right_eps = sp.right_eps;
    const float top_eps = LINEAR ? sp.right_eps : sp.top_eps;
    if (LINEAR) {
        if (!(x >= left && x <= right)) {  // NaN falls outside too
            y = x;
            lad = 0.0f;
            if (bin) *bin = -1;
            return 0;
        }
    } else if (x < left || x > right) {
        y = x;
        lad = 0.0f;
        if (bin) *bin = -1;
        return NFA_STATUS_OUTSIDE_DOMAIN;
    }

    Slots<KT> ew, eh;
    ew.bind(sl);
    eh.bind(sl + K);
    const float den_w = softmax_numerators<KT>(ew, sl, K, sp.divisor, sp.rdivisor);
    const float den_h = softmax_numerators<KT>(eh, sl + K, K, sp.divisor, sp.rdivisor);

    int k = -1;
    float cw0 = 0.f, cw1 = 0.f, ch0 = 0.f, ch1 = 0.f;
    if (INVERSE) {
        walk_bins<KT, true>(eh, K, den_h, sp.min_h, sp.om_h, span_h, bottom, top, x, k, ch0, ch1);
        if (bin) *bin = (k >= 0 && x >= top_eps) ? K : k;
        if (k < 0 || x >= top_eps) {
            y = x;
            lad = 0.0f;
            return NFA_STATUS_OUTSIDE_DOMAIN;
        }
        walk_bins<KT, false>(ew, K, den_w, sp.min_w, sp.om_w, span_w, left, right, x, k, cw0, cw1);
    } else {
        walk_bins<KT, true>(ew, K, den_w, sp.min_w, sp.om_w, span_w, left, right, x, k, cw0, cw1);
        if (bin) *bin = (k >= 0 && x >= right_eps) ? K : k;
        if (k < 0 || x >= right_eps) {
            y = x;
            lad = 0.0f;
            return NFA_STATUS_OUTSIDE_DOMAIN;
        }
        walk_bins<KT, false>(eh, K, den_h, sp.min_h, sp.om_h, span_h, bottom, top, x, k, ch0, ch1);
    }

    const float* sd = sl + 2 * K;
    float u0, u1;
    if (LINEAR && REGS) {
        u0 = sp.tail_logit;
        u1 = sp.tail_logit;
#pragma unroll
        for (int q = 0; q < KT - 1; ++q) {
            u0 = (k == q + 1) ? sd[q] : u0;
            u1 = (k == q) ? sd[q] : u1;
        }
    } else if (LINEAR) {  // logits padded with the tail constant on both sides
        u0 = (k == 0) ? sp.tail_logit : sd[k - 1];
        u1 = (k >= sp.nd) ? sp.tail_logit : sd[k];  // padded index k+1 past the given logits
    } else if (REGS) {  // K + 1 logits in registers (the whole-layer kernel K8 with tails=None): selected, not indexed
        u0 = sd[0];
        u1 = sd[1];
#pragma unroll
        for (int q = 1; q < KT; ++q) {
            u0 = (k == q) ? sd[q] : u0;
            u1 = (k == q) ? sd[q + 1] : u1;
        }
    } else {
        u0 = sd[k];
        u1 = sd[k + 1];
    }
    const float d0 = sp.min_d + softplus_beta(u0, sp.beta);
    const float d1 = sp.min_d + softplus_beta(u1, sp.beta);

    return rqs_bin_eval<INVERSE>(x, cw0, cw1, ch0, ch1, d0, d1, y, lad);
}

// Branch-free form of rqs_eval<8, INVERSE, LINEAR = true, REGS = true>: same arithmetic, same
// results, but the tail / not-found cases are selected at the end instead of returning early, so
// that two independent evaluations placed back to back form one basic block and the scheduler
// can interleave their dependent chains (K7: a lane evaluates two features per group).
// softmax numerators of 8 logits that the producer already scaled by log2(e) (the scale folded
// into the weights of the GEMM that makes them): 2^(u_i - max), one v_exp_f32 each
__device__ __forceinline__ float softmax_numerators_log2(Slots<8>& e, const float* logits) {
#pragma clang fp contract(off)
    float m = logits[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, logits[i]);
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        t[i] = __builtin_amdgcn_exp2f(logits[i] - m);
        e.set(i, t[i]);
    }
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

// PRESCALED: 0 = width / height logits as the conditioner produced them (divided by sp.divisor
// here), 1 = already divided, 2 = already divided and multiplied by log2(e)
//
// The evaluation comes in two halves: the softmax numerators of both logit sets, then the walks
// over the bins, the derivatives and the map inside the bin (FlatSteps below cuts the same
// operations into slices for K8's woven final layer).
struct FlatEvalState {
    Slots<8> ew, eh;
    float den_w, den_h;
};

template <int PRESCALED>
__device__ __forceinline__ void rqs_flat8_numerators(const float* sl, const RqsDev& sp, FlatEvalState& st) {
#pragma clang fp contract(off)
    constexpr int KT = 8;
    const float div = PRESCALED ? 0.0f : sp.divisor;
    st.den_w = PRESCALED == 2 ? softmax_numerators_log2(st.ew, sl)
                              : softmax_numerators<KT>(st.ew, sl, KT, div, sp.rdivisor);
    st.den_h = PRESCALED == 2 ? softmax_numerators_log2(st.eh, sl + KT)
                              : softmax_numerators<KT>(st.eh, sl + KT, KT, div, sp.rdivisor);
}

template <bool INVERSE>
__device__ __forceinline__ int rqs_flat8_finish(float x, const float* sl, const RqsDev& sp, const FlatEvalState& st,
                                                float& y, float& lad) {
#pragma clang fp contract(off)
    constexpr int KT = 8;
    const float right = sp.right, left = -sp.right;
    const bool inside = (x >= left && x <= right);  // NaN is outside
    int k = -1;
    float cw0 = 0.f, cw1 = 0.f, ch0 = 0.f, ch1 = 0.f;
    if (INVERSE) {
        walk_bins<KT, true>(st.eh, KT, st.den_h, sp.min_h, sp.om_h, sp.span_w, left, right, x, k, ch0, ch1);
        walk_bins<KT, false>(st.ew, KT, st.den_w, sp.min_w, sp.om_w, sp.span_w, left, right, x, k, cw0, cw1);
    } else {
        walk_bins<KT, true>(st.ew, KT, st.den_w, sp.min_w, sp.om_w, sp.span_w, left, right, x, k, cw0, cw1);
        walk_bins<KT, false>(st.eh, KT, st.den_h, sp.min_h, sp.om_h, sp.span_w, left, right, x, k, ch0, ch1);
    }
    const bool found = (k >= 0) && !(x >= sp.right_eps);
    const float* sd = sl + 2 * KT;
    float u0 = sp.tail_logit, u1 = sp.tail_logit;
#pragma unroll
    for (int q = 0; q < KT - 1; ++q) {
        u0 = (k == q + 1) ? sd[q] : u0;
        u1 = (k == q) ? sd[q] : u1;
    }
    const float d0 = sp.min_d + softplus_beta(u0, sp.beta);
    const float d1 = sp.min_d + softplus_beta(u1, sp.beta);
    float ye, le;
    const int status = rqs_bin_eval<INVERSE>(x, cw0, cw1, ch0, ch1, d0, d1, ye, le);
    const bool valid = inside && found;
    y = valid ? ye : x;
    lad = valid ? le : 0.0f;
    return inside ? (found ? status : NFA_STATUS_OUTSIDE_DOMAIN) : 0;
}

template <bool INVERSE, int PRESCALED = 0>
__device__ __forceinline__ int rqs_eval_flat8(float x, const float* sl, const RqsDev& sp, float& y, float& lad) {
    FlatEvalState st;
    rqs_flat8_numerators<PRESCALED>(sl, sp, st);
    return rqs_flat8_finish<INVERSE>(x, sl, sp, st, y, lad);
}


// rqs_eval_flat8 cut into small slices: the same operations in the same order (bit-identical
// results), exposed a few instructions at a time so that a GEMM loop can issue one MFMA between two
// slices.  On gfx950 an MFMA holds the issue port for ~16 of its 32 cycles; VALU work that sits in
// the instruction stream right behind it fills the rest (tools/region_probe.hip).
//   num_w<S>, num_h<S>  S in [0, kNumSlices):    softmax numerators of the width / height logits
//   finish<S>           S in [0, kFinishSlices): the two walks over the bins, the derivatives, the
//                                                map inside the bin; the last slice selects y / lad
//
// FAST = true trades the reference's exact rounding sequence for fewer instructions where the
// logits carry more noise than the rounding anyway (K8's logits come out of an in-kernel GEMM
// whose summation order differs from the reference's: ~1e-6 relative): the exponent of a softmax
// numerator is one rounded product instead of a two-float one, the bin size is
// fma(e, om / den, min) instead of min + om * RN(e / den), a knot is one fma.  Prefix sums stay in
// double.  Not bit-identical to the plain evaluation; same error class (tests/test_gpu_flows.py).
//
// SCALED = true (K8h, rqs_resnet_f16.hip): the logits handed over are kappa^-1 times the true ones
// (a power of two: the f16 weight pieces of the GEMM that makes them are pre-scaled so that their
// low pieces stay in the normal f16 range); the factor rides on multiplications that exist anyway
// (`kl2e` = log2(e) * kappa in the softmax exponents) or costs one product (derivative logits).
template <bool INVERSE, int PRESCALED, bool FAST = false, int KT = 8, bool SCALED = false>
struct FlatSteps {
    static_assert(PRESCALED == 1 || PRESCALED == 2, "logits already divided by sqrt(hidden)");
    static_assert(KT == 8 || (KT == 10 && FAST && PRESCALED == 1), "10 bins: the shorter sequence only");
    static_assert(!SCALED || (FAST && PRESCALED == 1), "scaled logits: the shorter sequence only");
    float kl2e, kappa, tail_s;  // SCALED: log2(e) * kappa, kappa, tail_logit / kappa
    static constexpr int kNumSlices = 2 * KT + 4;
    static constexpr int kWalk = 3 * KT;                       // KT bins x 3 slices
    static constexpr int kFinishSlices = 1 + kWalk + 1 + kWalk + 6 + 5 + 1;  // = 62 for 8 bins
    static constexpr int kFirstWalkSlices = 1 + kWalk;  // these read only the first walk's numerators
    static constexpr bool kInverse = INVERSE;
    float ew[KT], eh[KT];  // logits, then softmax numerators
    float sd[KT - 1];      // derivative logits
    float x;
    float den_w, den_h, rden, prev;
    double acc;
    int k;
    float cw0, cw1, ch0, ch1, u0, u1, d0, d1;
    float y, lad;
    int status;
    float t0, t1, t2, t3, t4, t5;  // values that cross slice boundaries

    float m_w, m_h, lo_w, lo_h;  // per logit set, so that the two numerator passes can alternate

    template <int S>
    __device__ __forceinline__ void numerators(float (&e)[KT], float& den, float& m, float& tl) {
#pragma clang fp contract(off)
        if constexpr (S == 0) {          // max of the first four
            if (PRESCALED == 2) m = fmaxf(fmaxf(fmaxf(e[0], e[1]), e[2]), e[3]);
            else m = fmaxf(fmaxf(fmaxf(fmaxf(-INFINITY, e[0]), e[1]), e[2]), e[3]);
        } else if constexpr (S == 1) {
            m = fmaxf(fmaxf(fmaxf(fmaxf(m, e[4]), e[5]), e[6]), e[7]);
            if constexpr (KT == 10) m = fmaxf(fmaxf(m, e[8]), e[9]);
        } else if constexpr (S < 2 + 2 * KT) {   // one logit in two slices: exponent in two floats | 2^hi (1 + lo ln2)
            constexpr int I = (S - 2) >> 1;
            if constexpr (PRESCALED == 2) {
                if constexpr (((S - 2) & 1) == 0) e[I] = __builtin_amdgcn_exp2f(e[I] - m);
            } else if constexpr (FAST) {
                if constexpr (((S - 2) & 1) == 0)
                    e[I] = __builtin_amdgcn_exp2f((e[I] - m) * (SCALED ? kl2e : 1.44269502162933349609375f));
            } else if constexpr (((S - 2) & 1) == 0) {
                const float kLog2e = 1.44269502162933349609375f, kLog2eLo = 1.925963033500011e-08f;
                const float v = e[I] - m;
                const float hi = v * kLog2e;
                float lo = __builtin_fmaf(v, kLog2e, -hi);
                lo = __builtin_fmaf(v, kLog2eLo, lo);
                e[I] = hi;
                tl = lo;
            } else {
                const float kLn2 = 0.693147182464599609375f;
                const float e0 = __builtin_amdgcn_exp2f(e[I]);
                e[I] = __builtin_fmaf(e0, tl * kLn2, e0);
            }
        } else if constexpr (S == 2 + 2 * KT) {
            tl = (e[0] + e[1]) + (e[2] + e[3]);
        } else {
            den = tl + ((e[4] + e[5]) + (e[6] + e[7]));
            if constexpr (KT == 10) den += e[8] + e[9];
        }
    }
    template <int S>
    __device__ __forceinline__ void num_w() { numerators<S>(ew, den_w, m_w, lo_w); }
    template <int S>
    __device__ __forceinline__ void num_h() { numerators<S>(eh, den_h, m_h, lo_h); }

    // one bin of a walk (walk_bins) in three slices; SEARCH picks the bin x falls into, otherwise
    // bin k is picked
    template <bool SEARCH, int I, int PART>
    __device__ __forceinline__ void bin(const float (&e)[KT], float den, float minbin, float om, const RqsDev& sp,
                                        float& knot_lo, float& knot_hi) {
#pragma clang fp contract(off)
        if constexpr (PART == 0) {
            if constexpr (FAST) {
                t1 = __builtin_fmaf(e[I], rden, minbin);  // rden holds om / den here
            } else {
                const float p = div_with_rcp(e[I], den, rden);
                t1 = minbin + om * p;
            }
        } else if constexpr (PART == 1) {
            acc += (double)t1;
            const float c = (float)acc;
            if constexpr (FAST) t2 = (I == KT - 1) ? sp.right : __builtin_fmaf(sp.span_w, c, -sp.right);
            else t2 = (I == KT - 1) ? sp.right : sp.span_w * c + (-sp.right);
        } else {
            const bool take = SEARCH ? (x >= prev) : (I == k);
            if (take) {
                if (SEARCH) k = I;
                knot_lo = prev;
                knot_hi = t2;
            }
            prev = t2;
            if (!SEARCH) {  // the bin's two derivative logits (select chain on the same compare, no indexing)
                if (I >= 1) u0 = take ? sd[I >= 1 ? I - 1 : 0] : u0;
                if (I < KT - 1) u1 = take ? sd[I < KT - 1 ? I : 0] : u1;
            }
        }
    }

    // min_d + softplus_beta(u, beta) in three slices (exp | log1p | select)
    template <int PART>
    __device__ __forceinline__ void derivative(float u, float& d, const RqsDev& sp) {
#pragma clang fp contract(off)
        // (callers guarantee beta == 1: the coupling layers never enable the identity initialisation,
        // coupling.py:572-582; softplus_beta's general form carries an IEEE division per call)
        if constexpr (PART == 0) {
            t3 = SCALED ? u * kappa : u;
            t4 = exp_noclamp(t3);
        } else if constexpr (PART == 1) {
            t4 = log1p_nonneg(t4);
        } else {
            d = sp.min_d + (t3 > 20.0f ? t3 : t4);
        }
    }

    template <int S>
    __device__ __forceinline__ void finish(const RqsDev& sp) {
#pragma clang fp contract(off)
        constexpr int W1 = 1, MID = W1 + kWalk, W2 = MID + 1, D0 = W2 + kWalk, BE = D0 + 6, LAST = BE + 5;
        static_assert(LAST + 1 == kFinishSlices, "slice map");
        if constexpr (S == 0) {
            k = -1;
            cw0 = cw1 = ch0 = ch1 = 0.0f;
            rden = rcp_refined(INVERSE ? den_h : den_w);
            if constexpr (FAST) rden *= INVERSE ? sp.om_h : sp.om_w;
            acc = 0.0;
            prev = -sp.right;
        } else if constexpr (S < MID) {
            constexpr int I = (S - W1) / 3, PART = (S - W1) % 3;
            if (INVERSE) bin<true, I, PART>(eh, den_h, sp.min_h, sp.om_h, sp, ch0, ch1);
            else bin<true, I, PART>(ew, den_w, sp.min_w, sp.om_w, sp, cw0, cw1);
        } else if constexpr (S == MID) {
            rden = rcp_refined(INVERSE ? den_w : den_h);
            if constexpr (FAST) rden *= INVERSE ? sp.om_w : sp.om_h;
            acc = 0.0;
            prev = -sp.right;
            u0 = SCALED ? tail_s : sp.tail_logit;
            u1 = u0;
        } else if constexpr (S < D0) {
            constexpr int I = (S - W2) / 3, PART = (S - W2) % 3;
            if (INVERSE) bin<false, I, PART>(ew, den_w, sp.min_w, sp.om_w, sp, cw0, cw1);
            else bin<false, I, PART>(eh, den_h, sp.min_h, sp.om_h, sp, ch0, ch1);
        } else if constexpr (S < D0 + 3) {
            derivative<S - D0>(u0, d0, sp);
        } else if constexpr (S < BE) {
            derivative<S - D0 - 3>(u1, d1, sp);
        } else if constexpr (S < LAST) {
            bin_eval<S - BE>();
        } else {
            const bool inside = (x >= -sp.right && x <= sp.right);  // NaN is outside
            const bool found = (k >= 0) && !(x >= sp.right_eps);
            const bool valid = inside && found;
            y = valid ? y : x;
            lad = valid ? lad : 0.0f;
            status = inside ? (found ? status : NFA_STATUS_OUTSIDE_DOMAIN) : 0;
        }
    }

    // rqs_bin_eval in five slices
    float in_w, in_h, r_w, delta, s_, th, t1mt, den;
    template <int PART>
    __device__ __forceinline__ void bin_eval() {
#pragma clang fp contract(off)
        if constexpr (PART == 0) {
            in_w = cw1 - cw0;
            in_h = ch1 - ch0;
            r_w = rcp_refined(in_w);
            delta = div_with_rcp(in_h, in_w, r_w);
            s_ = (d0 + d1) - 2.0f * delta;
            status = 0;
        } else if constexpr (INVERSE) {
            if constexpr (PART == 1) {
                const float yc = x - ch0;
                const float a = yc * s_ + in_h * (delta - d0);
                const float b = in_h * d0 - yc * s_;
                const float c = (-delta) * yc;
                t0 = b * b - (4.0f * a) * c;   // discriminant
                t1 = 2.0f * c;
                t2 = -b;
            } else if constexpr (PART == 2) {
                if (!(t0 >= 0.0f)) status = NFA_STATUS_NEG_DISCRIMINANT;
                th = div_normal(t1, t2 - sqrtf(t0));  // root
                y = th * in_w + cw0;
            } else if constexpr (PART == 3) {
                t1mt = th * (1.0f - th);
                den = delta + s_ * t1mt;
                const float omr = 1.0f - th;
                t5 = (delta * delta) * ((d1 * (th * th) + (2.0f * delta) * t1mt) + d0 * (omr * omr));
            } else {
                lad = -(log_normal(t5) - 2.0f * log_normal(den));
            }
        } else {
            if constexpr (PART == 1) {
                th = div_with_rcp(x - cw0, in_w, r_w);
                t1mt = th * (1.0f - th);
                t0 = in_h * (delta * (th * th) + d0 * t1mt);  // numerator
            } else if constexpr (PART == 2) {
                den = delta + s_ * t1mt;
                y = ch0 + div_normal(t0, den);
            } else if constexpr (PART == 3) {
                const float omt = 1.0f - th;
                t5 = (delta * delta) * ((d1 * (th * th) + (2.0f * delta) * t1mt) + d0 * (omt * omt));
            } else {
                lad = log_normal(t5) - 2.0f * log_normal(den);
            }
        }
    }
};

// every slice of a FlatSteps evaluation in order (callers that do not interleave it with anything)
template <int PHASE, int I, class Steps>
__device__ __forceinline__ void flat_steps_run(Steps& f, const RqsDev& sp) {
    constexpr int N = PHASE == 2 ? Steps::kFinishSlices : Steps::kNumSlices;
    if constexpr (I < N) {
        if constexpr (PHASE == 0) f.template num_w<I>();
        else if constexpr (PHASE == 1) f.template num_h<I>();
        else f.template finish<I>(sp);
        flat_steps_run<PHASE, I + 1>(f, sp);
    }
}
template <class Steps>
__device__ __forceinline__ void flat_steps_all(Steps& f, const RqsDev& sp) {
    flat_steps_run<0, 0>(f, sp);
    flat_steps_run<1, 0>(f, sp);
    flat_steps_run<2, 0>(f, sp);
}

// host side: nfa_rqs_spec (doubles, as the reference's Python floats) -> fp32 device constants,
// rounded exactly where aten rounds them
inline int make_dev_spec(const nfa_rqs_spec* s, RqsDev* d) {
    if (!s) return NFA_ERR_INVALID_ARGUMENT;
    if (s->num_bins < 1 || s->num_bins > 4096) return NFA_ERR_INVALID_ARGUMENT;
    if (s->tails != NFA_TAILS_NONE && s->tails != NFA_TAILS_LINEAR) return NFA_ERR_INVALID_ARGUMENT;
    if (s->min_bin_width * s->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (s->min_bin_height * s->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    d->K = s->num_bins;
    d->linear = s->tails == NFA_TAILS_LINEAR;
    d->nd = d->linear ? d->K - 1 : d->K + 1;
    d->P = 2 * d->K + d->nd;
    d->left = (float)s->left;
    d->right = (float)s->right;
    d->bottom = (float)s->bottom;
    d->top = (float)s->top;
    d->span_w = (float)(s->right - s->left);
    d->span_h = (float)(s->top - s->bottom);
    d->right_eps = d->right + 1e-6f;
    d->top_eps = d->top + 1e-6f;
    d->min_w = (float)s->min_bin_width;
    d->min_h = (float)s->min_bin_height;
    d->min_d = (float)s->min_derivative;
    d->om_w = (float)(1.0 - s->min_bin_width * s->num_bins);
    d->om_h = (float)(1.0 - s->min_bin_height * s->num_bins);
    d->beta = (float)s->softplus_beta;
    d->tail_logit = (float)s->tail_logit;
    d->divisor = (float)s->wh_divisor;
    d->rdivisor = d->divisor != 0.0f ? 1.0f / d->divisor : 0.0f;
    return NFA_OK;
}


}  // namespace nfa
