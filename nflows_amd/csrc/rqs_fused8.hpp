// Spline evaluation of K8h's woven final layer (csrc/rqs_resnet_f16.hip): 2 .. 32 bins (8 and 10: the tuned forms), linear tails,
// logits handed over at scale 1/kappa straight from the MFMA accumulators.
//
// Same function as rational_quadratic.py:66-181 (+ :13-63 for the tails), arranged for the lowest
// VALU instruction count -- the layer kernel is bound by VALU issue, not by the matrix pipe:
//   * softmax numerators as 2^(fma(e, log2e*kappa, -max*log2e*kappa)): two instructions per logit;
//     the rounding of the shared term is common to all eight numerators and cancels in the
//     normalisation;
//   * ONE walk over the bins instead of two: knot_{i+1} = knot_i + fma(numerator_i, 2B(1-K min)/den,
//     2B min) for widths and heights side by side (fp32 running sums; the reference rounds each
//     normalised bin to fp32, sums in double and rounds every knot -- the same error class), and on
//     the compare x >= knot_i of the searched axis the candidates of BOTH axes and the two
//     derivative logits are selected with the same condition.  No bin index, no second walk;
//   * first / last knot exactly -B / +B as the reference forces them, bin sizes as knot
//     differences;
//   * softplus and the logarithms on v_exp_f32 / v_log_f32 (1 ulp) with first-order corrections
//     where the argument is near 1, divisions as reciprocal + one residual correction.
// The evaluation is cut into slices of 2-8 instructions (num_w / num_h / finish) so that the GEMM
// loop can issue one MFMA between two slices.  Error class: the reference's own fp32 path
// (tests/test_gpu_headline_parity.py holds the kernel to 2x the reference's error against float64).
#pragma once

#include "rqs_math.hpp"

namespace nfa {

// DBG = true (the diagnostic instances of rqs_resnet_f16_dbg.hip only): `kbin` follows the walk -- the last bin whose
// lower knot the input reached, i.e. the value torchutils.searchsorted (utils/torchutils.py:134-136) returns on THIS
// evaluation's knots; -1 for an input outside the box.  The same compares on the same values as the selects use.
template <bool INVERSE, int KT = 8, bool DBG = false>
struct FusedSteps {
    static_assert(KT >= 2 && KT <= 32, "2 .. 32 bins (8 and 10 keep their own maximum / sum chains)");
    static constexpr int kNumSlices = KT + 3;                   // max, one exponential per logit, sum x 2
    static constexpr int kWalkSlices = 3 + (KT - 1);            // setup x 2, bin 0, bins 1..KT-1
    static constexpr int kBinSlices = INVERSE ? 9 : 7;
    static constexpr int kFinishSlices = kWalkSlices + 6 + kBinSlices + 1;
    static constexpr int kFirstWalkSlices = 0;                  // (no part of finish runs on one numerator set alone)
    static constexpr bool kInverse = INVERSE;

    float ew[KT], eh[KT];   // logits (scaled), then softmax numerators
    float sd[KT - 1];       // derivative logits (scaled)
    float x;
    float kl2e, kappa, tail_s;   // log2(e) * kappa, kappa, tail_logit / kappa (uniform)
    float m_w, m_h, den_w, den_h, tw_, th_;
    float aw, ah, kw, kh, kwn, khn;
    unsigned long long take;     // lanes whose x is at or above the lower knot of the bin the next walk slice visits
    float cw0, cw1, ch0, ch1, u0, u1, d0, d1;
    float y, lad;
    int status;
    float t3, t4;
    float in_w, in_h, r_w, r_den, delta, s_, th, t1mt, den, t0, t1, t2, t5;
    int kbin;   // DBG only

    // Softmax numerators 2^((logit - max) x log2(e) x kappa): max in one slice (two v_max3_f32 + one
    // v_max_f32 for 8 bins), then fma + v_exp_f32 per logit; the rounding of the shared term is common to
    // all numerators and cancels in the normalisation.  (Round 3 tried the numerators without the maximum:
    // 10 instructions fewer per evaluation, no measurable gain, and logit sets beyond +-87 -- softmax is
    // shift-invariant, a trained network may sit anywhere -- would have gone to the exact kernel.)
    // e[A] + ... + e[B - 1], neighbours paired first (two independent chains for the adder)
    template <int A, int B>
    __device__ __forceinline__ static float half_sum(const float (&e)[KT]) {
        float s0 = e[A], s1 = B - A > 1 ? e[A + 1] : 0.0f;
#pragma unroll
        for (int i = A + 2; i < B; i += 2) {
            s0 += e[i];
            if (i + 1 < B) s1 += e[i + 1];
        }
        return B - A > 1 ? s0 + s1 : s0;
    }
    template <int S>
    __device__ __forceinline__ void numerators(float (&e)[KT], float& den_, float& m, float& t) {
        if constexpr (S == 0) {
            if constexpr (KT == 8 || KT == 10) {
                m = __builtin_fmaxf(__builtin_fmaxf(e[0], e[1]), e[2]);      // (v_max3_f32)
                m = __builtin_fmaxf(__builtin_fmaxf(m, e[3]), e[4]);
                m = __builtin_fmaxf(__builtin_fmaxf(m, e[5]), e[6]);
                if constexpr (KT == 10) m = __builtin_fmaxf(__builtin_fmaxf(m, e[7]), e[8]);
                m = __builtin_fmaxf(m, e[KT - 1]) * kl2e;                     // max * log2e * kappa
            } else {   // other bin counts (round 4): the same chain of v_max3_f32 over however many logits there are
                m = e[0];
#pragma unroll
                for (int i = 1; i + 1 < KT; i += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, e[i]), e[i + 1]);
                if constexpr (KT % 2 == 0) m = __builtin_fmaxf(m, e[KT - 1]);
                m *= kl2e;
            }
        } else if constexpr (S < 1 + KT) {
            e[S - 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[S - 1], kl2e, -m));
        } else if constexpr (KT == 8 || KT == 10) {
            if constexpr (S == 1 + KT) {
                t = (e[0] + e[1]) + (e[2] + e[3]);
            } else {
                den_ = t + ((e[4] + e[5]) + (e[6] + e[7]));
                if constexpr (KT == 10) den_ += e[8] + e[9];
            }
        } else {   // the two halves of the sum in two slices, pairs first
            constexpr int H = KT / 2;
            if constexpr (S == 1 + KT) t = half_sum<0, H>(e);
            else den_ = t + half_sum<H, KT>(e);
        }
    }
    template <int S>
    __device__ __forceinline__ void num_w() { numerators<S>(ew, den_w, m_w, tw_); }
    template <int S>
    __device__ __forceinline__ void num_h() { numerators<S>(eh, den_h, m_h, th_); }

    // min_d + softplus(u * kappa) in three slices, as max(t, 0) + log1p(exp(-|t|)): no overflow for any t (a
    // derivative logit of 200 is a legitimate slope of 200), full relative accuracy for very negative t.
    template <int PART>
    __device__ __forceinline__ void derivative(float u, float& d, const RqsDev& sp) {
        if constexpr (PART == 0) {
            t3 = u * kappa;
            t4 = __builtin_amdgcn_exp2f(-__builtin_fabsf(u * kl2e));
        } else if constexpr (PART == 1) {
            const float u1p = 1.0f + t4;
            const float c = t4 - (u1p - 1.0f);   // what the addition dropped
            const float lg = __builtin_fmaf(__builtin_amdgcn_logf(u1p), 0.693147182464599609375f, sp.min_d);
            t4 = __builtin_fmaf(c, __builtin_amdgcn_rcpf(u1p), lg);
        } else {
            d = __builtin_fmaxf(t3, 0.0f) + t4;
        }
    }

    // One bin of the walk.  The compare for the NEXT bin and the six selects of THIS bin are one asm block
    // with the compare in front: a v_cndmask needs two wait states behind the VALU instruction that wrote
    // its mask (hipcc pads compare + select with s_nop 1), here the mask was written a whole slice earlier.
    __device__ __forceinline__ void walk_select(float next_lower, float cand_u0, float cand_u1) {
        unsigned long long next;
        asm("v_cmp_ge_f32 %6, %7, %8\n\t"
            "v_cndmask_b32 %0, %0, %9, %15\n\t"
            "v_cndmask_b32 %1, %1, %10, %15\n\t"
            "v_cndmask_b32 %2, %2, %11, %15\n\t"
            "v_cndmask_b32 %3, %3, %12, %15\n\t"
            "v_cndmask_b32 %4, %4, %13, %15\n\t"
            "v_cndmask_b32 %5, %5, %14, %15"
            : "+v"(cw0), "+v"(cw1), "+v"(ch0), "+v"(ch1), "+v"(u0), "+v"(u1), "=&s"(next)
            : "v"(x), "v"(next_lower), "v"(kw), "v"(kwn), "v"(kh), "v"(khn), "v"(cand_u0), "v"(cand_u1), "s"(take));
        take = next;
    }

    template <int S>
    __device__ __forceinline__ void finish(const RqsDev& sp) {
        constexpr int W = kWalkSlices, D0 = W, BE = D0 + 6, LAST = BE + kBinSlices;
        static_assert(LAST + 1 == kFinishSlices, "slice map");
        const float B = sp.right;
        if constexpr (S == 0) {
            const float r0 = __builtin_amdgcn_rcpf(den_w);
            const float r = __builtin_fmaf(__builtin_fmaf(-den_w, r0, 1.0f), r0, r0);
            aw = r * (sp.span_w * sp.om_w);
        } else if constexpr (S == 1) {
            const float r0 = __builtin_amdgcn_rcpf(den_h);
            const float r = __builtin_fmaf(__builtin_fmaf(-den_h, r0, 1.0f), r0, r0);
            ah = r * (sp.span_w * sp.om_h);
        } else if constexpr (S == 2) {   // bin 0: always a candidate; knots of bin 1 and its compare
            const float k1w = -B + __builtin_fmaf(ew[0], aw, sp.span_w * sp.min_w);
            const float k1h = -B + __builtin_fmaf(eh[0], ah, sp.span_w * sp.min_h);
            cw0 = -B;
            ch0 = -B;
            cw1 = k1w;
            ch1 = k1h;
            u0 = tail_s;
            u1 = sd[0];
            kw = k1w;
            kh = k1h;
            kwn = k1w + __builtin_fmaf(ew[1], aw, sp.span_w * sp.min_w);
            khn = k1h + __builtin_fmaf(eh[1], ah, sp.span_w * sp.min_h);
            asm("v_cmp_ge_f32 %0, %1, %2" : "=s"(take) : "v"(x), "v"(INVERSE ? k1h : k1w));
            if constexpr (DBG) kbin = 0;
        } else if constexpr (S < W) {
            constexpr int I = S - 2;   // bins 1..KT-1: (kw, kh) lower, (kwn, khn) upper knots, `take` = x >= lower
            if constexpr (DBG) kbin = (x >= (INVERSE ? kh : kw)) ? I : kbin;
            float kw2 = B, kh2 = B;    // upper knots of bin I + 1
            if constexpr (I + 1 < KT - 1) {
                kw2 = kwn + __builtin_fmaf(ew[I + 1], aw, sp.span_w * sp.min_w);
                kh2 = khn + __builtin_fmaf(eh[I + 1], ah, sp.span_w * sp.min_h);
            }
            walk_select(INVERSE ? khn : kwn, sd[I - 1], I < KT - 1 ? sd[I < KT - 1 ? I : 0] : tail_s);
            kw = kwn;
            kh = khn;
            kwn = kw2;
            khn = kh2;
        } else if constexpr (S < D0 + 3) {
            derivative<S - D0>(u0, d0, sp);
        } else if constexpr (S < BE) {
            derivative<S - D0 - 3>(u1, d1, sp);
        } else if constexpr (S < LAST) {
            bin_eval<S - BE>();
        } else {
            const bool inside = (x >= -B && x <= B);  // NaN is outside
            y = inside ? y : x;
            lad = inside ? lad : 0.0f;
            status = inside ? status : 0;
            if constexpr (DBG) kbin = inside ? kbin : -1;
        }
    }

    // the map inside the bin (rational_quadratic.py:132-181) in seven slices
    template <int PART>
    __device__ __forceinline__ void bin_eval() {
        if constexpr (PART == 0) {
            in_w = cw1 - cw0;
            in_h = ch1 - ch0;
            const float r0 = __builtin_amdgcn_rcpf(in_w);
            r_w = __builtin_fmaf(__builtin_fmaf(-in_w, r0, 1.0f), r0, r0);
            status = 0;
        } else if constexpr (PART == 1) {
            const float q = in_h * r_w;
            delta = __builtin_fmaf(__builtin_fmaf(-q, in_w, in_h), r_w, q);
            s_ = __builtin_fmaf(-2.0f, delta, d0 + d1);
        } else if constexpr (INVERSE) {
            if constexpr (PART == 2) {
                const float yc = x - ch0;
                const float ys = yc * s_;
                const float a = __builtin_fmaf(in_h, delta - d0, ys);
                const float b = __builtin_fmaf(in_h, d0, -ys);
                const float c = -delta * yc;
                t0 = __builtin_fmaf(b, b, -4.0f * a * c);   // discriminant
                t1 = 2.0f * c;
                t2 = -b;
            } else if constexpr (PART == 3) {
                if (!(t0 >= 0.0f)) status = NFA_STATUS_NEG_DISCRIMINANT;
                const float dn = t2 - __builtin_sqrtf(t0);
                const float r0 = __builtin_amdgcn_rcpf(dn);
                const float r = __builtin_fmaf(__builtin_fmaf(-dn, r0, 1.0f), r0, r0);
                const float q = t1 * r;
                th = __builtin_fmaf(__builtin_fmaf(-q, dn, t1), r, q);   // root of the quadratic (:144)
            } else if constexpr (PART == 4) {
                // One Newton step on the FORWARD map g(theta) of this bin, evaluated the way the forward
                // pass evaluates it: the quadratic formula loses digits where b^2 ~ 4ac or the
                // denominator is small, and whatever it loses reappears as forward/inverse inconsistency
                // (amplified by every following layer).  g'(theta) = in_w delta^2 (...) / den^2; delta^2 (...) is
                // the quantity the log-determinant needs anyway.
                const float omr = 1.0f - th;
                t1mt = th * omr;
                den = __builtin_fmaf(s_, t1mt, delta);
                t0 = in_h * __builtin_fmaf(delta, th * th, d0 * t1mt);   // numerator of g - ch0
                t1 = omr * omr;
            } else if constexpr (PART == 5) {
                float c = d0 * t1;
                c = __builtin_fmaf(delta + delta, t1mt, c);
                c = __builtin_fmaf(d1, th * th, c);
                t5 = (delta * delta) * c;                                 // den^2 g' / in_w
                const float r0 = __builtin_amdgcn_rcpf(den);
                const float r = __builtin_fmaf(__builtin_fmaf(-den, r0, 1.0f), r0, r0);
                const float q = t0 * r;
                const float g = ch0 + __builtin_fmaf(__builtin_fmaf(-q, den, t0), r, q);
                t2 = x - g;                                               // residual in y
            } else if constexpr (PART == 6) {
                // d g / d theta = in_w * (d y / d x) = in_w delta^2 (...) / den^2.  (Until the end of round 3 this line read
                // in_h * t5 -- the slope with respect to theta taken for in_h instead of in_w times the derivative --,
                // i.e. every step came out scaled by 1 / delta: right for the near-identity splines of an untrained
                // flow (delta ~ 1), up to 80 x the reference's error on steep ones.  Found by running this file on
                // the CPU against the reference's vectors: tests/test_rqs_f32_host.py.)
#ifdef NFA_MUTATION_NEWTON_SLOPE   // (tools/build_variant.sh only: the round-3 defect restored, which tests/test_gpu_steep.py must FAIL)
                const float slope = in_h * t5;
#else
                const float slope = in_w * t5;                            // g' den^2
#endif
                const float r0 = __builtin_amdgcn_rcpf(slope);
                const float r = __builtin_fmaf(__builtin_fmaf(-slope, r0, 1.0f), r0, r0);
                const float step = (t2 * den) * (den * r);
                // (a bin visited through a NaN / degenerate path keeps the closed-form root)
                th = (__builtin_fabsf(step) <= 0.25f) ? th + step : th;
                y = __builtin_fmaf(th, in_w, cw0);
            } else if constexpr (PART == 7) {
                const float omr = 1.0f - th;
                t1mt = th * omr;
                den = __builtin_fmaf(s_, t1mt, delta);
                float c = d0 * (omr * omr);
                c = __builtin_fmaf(delta + delta, t1mt, c);
                c = __builtin_fmaf(d1, th * th, c);
                t5 = (delta * delta) * c;
            } else {
                const float l1 = __builtin_amdgcn_logf(t5), l2 = __builtin_amdgcn_logf(den);
                lad = -(__builtin_fmaf(-2.0f, l2, l1) * 0.693147182464599609375f);
            }
        } else {
            if constexpr (PART == 2) {
                const float xc = x - cw0;
                const float q = xc * r_w;
                th = __builtin_fmaf(__builtin_fmaf(-q, in_w, xc), r_w, q);
                const float omt = 1.0f - th;
                t1mt = th * omt;
                t1 = omt * omt;
            } else if constexpr (PART == 3) {
                t2 = th * th;
                t0 = in_h * __builtin_fmaf(delta, t2, d0 * t1mt);   // numerator
                den = __builtin_fmaf(s_, t1mt, delta);
            } else if constexpr (PART == 4) {
                const float r0 = __builtin_amdgcn_rcpf(den);
                const float r = __builtin_fmaf(__builtin_fmaf(-den, r0, 1.0f), r0, r0);
                const float q = t0 * r;
                y = ch0 + __builtin_fmaf(__builtin_fmaf(-q, den, t0), r, q);
                r_den = r;
            } else if constexpr (PART == 5) {
                float c = d0 * t1;
                c = __builtin_fmaf(delta + delta, t1mt, c);
                c = __builtin_fmaf(d1, t2, c);
                t5 = (delta * r_den) * c;
            } else {
                // log(delta^2 c / den^2) with the reciprocal of den the output needed anyway: one logarithm
                lad = __builtin_amdgcn_logf((delta * r_den) * t5) * 0.693147182464599609375f;
            }
        }
    }
};

template <bool INVERSE>
using FusedSteps8 = FusedSteps<INVERSE, 8>;

}  // namespace nfa
