// The per-element arithmetic of the float64 rational-quadratic functional (K5d) and of its gradient:
// plain double arithmetic on one spline, no wave-level operation, so that the same functions compile for
// the device (rqs_f64.hip) and -- as test infrastructure only -- for the host (tests/test_oracle_golden.py
// builds them with g++ and holds them to the reference's float64 autograd; the product never runs them there).
//
// Forward: splines/rational_quadratic.py:66-181 (softmax -> min + (1 - min K) p -> cumsum -> affine -> forced end
// knots, searchsorted with its +1e-6 on the last knot, the rational-quadratic map or the root of its quadratic).
// Backward: closed-form adjoints of those expressions.  The inverse direction differentiates the root
// implicitly (d root = (d y - sum_p F_p dp) / F_theta), which is the derivative of the closed-form root the
// reference's autograd walks through; non-differentiable steps (bin search, the forced end knots, the padded
// tail derivatives) carry no gradient there either.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define NFA_HD __host__ __device__ __forceinline__
#else
#define NFA_HD inline
#endif

namespace nfa {
namespace f64 {

struct Spec {
    int K, linear, inverse, nd;
    double left, right, bottom, top, min_w, min_h, min_d, beta, tail_logit, divisor;
};

NFA_HD double softplus(double x, double beta) {
    const double xb = x * beta;
    return xb > 20.0 ? x : log1p(exp(xb)) / beta;   // F.softplus(threshold = 20)
}
NFA_HD double softplus_slope(double x, double beta) {   // d softplus / dx
    const double xb = x * beta;
    return xb > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-xb));
}

// One axis (widths or heights) of one spline, from its K logits.
struct Axis {
    double m, sum;          // softmax: maximum and sum of exponentials of the scaled logits
    double knot_lo, knot_hi;
    double below, upto;     // sum of p_i for i < k and for i <= k (backward)
};

// knots of one axis on the fly.  SEARCH: the bin x falls into (count of knots <= x, minus one, the last
// knot moved up by 1e-6: torchutils.py:134-136), else the given bin; its two knots come back.
template <bool SEARCH>
NFA_HD void axis_bin(const double* u, int K, double divisor, double lo, double hi, double minbin, double x, int& k,
                     Axis& ax) {
    double m = -INFINITY;
    for (int i = 0; i < K; ++i) {
        const double v = divisor != 0.0 ? u[i] / divisor : u[i];
        m = v > m ? v : m;
    }
    double sum = 0.0;
    for (int i = 0; i < K; ++i) sum += exp((divisor != 0.0 ? u[i] / divisor : u[i]) - m);
    const double one_minus = 1.0 - minbin * K;
    const double span = hi - lo;
    double acc = 0.0, prev = lo, pacc = 0.0;
    int found = -1;
    for (int i = 0; i < K; ++i) {
        const double p = exp((divisor != 0.0 ? u[i] / divisor : u[i]) - m) / sum;
        acc += minbin + one_minus * p;
        const double next = i == K - 1 ? hi : span * acc + lo;
        const bool take = SEARCH ? (x >= prev) : (i == k);
        if (take) {
            found = i;
            ax.knot_lo = prev;
            ax.knot_hi = next;
            ax.below = pacc;
            ax.upto = pacc + p;
        }
        prev = next;
        pacc += p;
    }
    ax.m = m;
    ax.sum = sum;
    if (SEARCH) k = (x >= hi + 1e-6) ? K : found;
}

// Everything the two directions share once the bin is known.
struct Bin {
    int k;
    double cw0, cw1, ch0, ch1, d0, d1, u0, u1;
    bool u0_is_logit, u1_is_logit;   // false: the padded tail constant (no gradient)
    int i0, i1;                      // positions of the two derivative logits
};

// status bits as in nflows_amd.h (NFA_STATUS_OUTSIDE_DOMAIN = 1, NFA_STATUS_NEG_DISCRIMINANT = 2)
constexpr int kOutside = 1, kNegDisc = 2;

// locate the bin of x and gather its knots and derivatives; false (and a status bit) when x has no bin
NFA_HD bool locate(const Spec& a, double x, const double* uw, const double* uh, const double* ud, Bin& b, Axis& aw,
                   Axis& ah, int& status) {
    int k = -1;
    if (a.inverse) {
        axis_bin<true>(uh, a.K, a.divisor, a.bottom, a.top, a.min_h, x, k, ah);
        if (k >= 0 && k < a.K) axis_bin<false>(uw, a.K, a.divisor, a.left, a.right, a.min_w, x, k, aw);
    } else {
        axis_bin<true>(uw, a.K, a.divisor, a.left, a.right, a.min_w, x, k, aw);
        if (k >= 0 && k < a.K) axis_bin<false>(uh, a.K, a.divisor, a.bottom, a.top, a.min_h, x, k, ah);
    }
    b.k = k;   // (what searchsorted returns: K past the nudged last knot, -1 when no knot is <= x)
    if (k < 0 || k >= a.K) {
        status |= kOutside;
        return false;
    }
    b.cw0 = aw.knot_lo;
    b.cw1 = aw.knot_hi;
    b.ch0 = ah.knot_lo;
    b.ch1 = ah.knot_hi;
    // derivative logits: linear tails pad both ends with the tail constant (:33-36)
    if (a.linear) {
        b.u0_is_logit = k != 0;
        b.u1_is_logit = k < a.nd;
        b.i0 = k - 1;
        b.i1 = k;
    } else {
        b.u0_is_logit = b.u1_is_logit = true;
        b.i0 = k;
        b.i1 = k + 1;
    }
    b.u0 = b.u0_is_logit ? ud[b.i0] : a.tail_logit;
    b.u1 = b.u1_is_logit ? ud[b.i1] : a.tail_logit;
    b.d0 = a.min_d + softplus(b.u0, a.beta);
    b.d1 = a.min_d + softplus(b.u1, a.beta);
    return true;
}

NFA_HD bool inside_box(const Spec& a, double x, int& status) {
    if (a.linear) return x >= a.left && x <= a.right;   // (NaN falls outside: identity, :26, :38-39)
    const bool inside = !(x < a.left || x > a.right);
    if (!inside) status |= kOutside;                     // :81-82
    return inside;
}

// ---- forward -------------------------------------------------------------------------------------
// `bin` (optional): the searched bin, -1 for elements the reference never searches (tails / outside / NaN)
NFA_HD void forward_element(const Spec& a, double x, const double* uw, const double* uh, const double* ud, double& y,
                            double& lad, int& status, int* bin = nullptr) {
    y = x;
    lad = 0.0;
    if (bin) *bin = -1;
    if (!inside_box(a, x, status)) return;
    Bin b;
    Axis aw, ah;
    const bool located = locate(a, x, uw, uh, ud, b, aw, ah, status);
    if (bin) *bin = b.k;
    if (!located) return;
    const double d0 = b.d0, d1 = b.d1;
    const double in_w = b.cw1 - b.cw0, in_h = b.ch1 - b.ch0, delta = in_h / in_w;
    const double s = (d0 + d1) - 2.0 * delta;
    if (a.inverse) {   // :132-160
        const double yc = x - b.ch0;
        const double qa = yc * s + in_h * (delta - d0), qb = in_h * d0 - yc * s, qc = -delta * yc;
        const double disc = qb * qb - 4.0 * qa * qc;
        if (!(disc >= 0.0)) status |= kNegDisc;
        const double root = (2.0 * qc) / (-qb - sqrt(disc));
        y = root * in_w + b.cw0;
        const double t1mt = root * (1.0 - root), den = delta + s * t1mt, omr = 1.0 - root;
        const double dnum = (delta * delta) * ((d1 * (root * root) + (2.0 * delta) * t1mt) + d0 * (omr * omr));
        lad = -(log(dnum) - 2.0 * log(den));
    } else {           // :162-181
        const double theta = (x - b.cw0) / in_w, t1mt = theta * (1.0 - theta);
        const double num = in_h * (delta * (theta * theta) + d0 * t1mt), den = delta + s * t1mt;
        y = b.ch0 + num / den;
        const double omt = 1.0 - theta;
        const double dnum = (delta * delta) * ((d1 * (theta * theta) + (2.0 * delta) * t1mt) + d0 * (omt * omt));
        lad = log(dnum) - 2.0 * log(den);
    }
}

// ---- backward ------------------------------------------------------------------------------------
// adjoint of one axis: `g_lo`, `g_hi` are the adjoints of the bin's two knots; the forced end knots
// (k == 0 / k == K - 1) and the affine map knot = span * cumsum + lo are folded here; writes the K logit
// gradients.  w_i = minbin + (1 - minbin K) p_i, cumsum_j = sum_{i < j} w_i, p = softmax(u / divisor).
NFA_HD void axis_backward(const double* u, double* g, int K, double divisor, double lo, double hi, double minbin, int k,
                          const Axis& ax, double g_lo, double g_hi) {
    const double span = hi - lo, one_minus = 1.0 - minbin * K;
    const double a = k == 0 ? 0.0 : span * g_lo;          // adjoint of cumsum_k     (knot 0 is the constant lo)
    const double b = k == K - 1 ? 0.0 : span * g_hi;      // adjoint of cumsum_{k+1} (knot K is the constant hi)
    // g_w_i = [i < k] a + [i <= k] b;  g_p_i = one_minus g_w_i;  g_v_i = p_i (g_p_i - sum_j p_j g_p_j)
    const double dot = one_minus * (a * ax.below + b * ax.upto);
    const double back = divisor != 0.0 ? 1.0 / divisor : 1.0;
    for (int i = 0; i < K; ++i) {
        const double p = exp((divisor != 0.0 ? u[i] / divisor : u[i]) - ax.m) / ax.sum;
        const double gw = (i < k ? a : 0.0) + (i <= k ? b : 0.0);
        g[i] = p * (one_minus * gw - dot) * back;
    }
}

// gy, gl: upstream gradients of the output and of logabsdet.  Writes g_x and the K, K, nd logit gradients.
NFA_HD void backward_element(const Spec& a, double x, const double* uw, const double* uh, const double* ud, double gy,
                             double gl, double& g_x, double* g_uw, double* g_uh, double* g_ud, int& status) {
    for (int i = 0; i < a.nd; ++i) g_ud[i] = 0.0;
    Bin b;
    Axis aw, ah;
    int ignored = 0;
    const bool live = inside_box(a, x, ignored) && locate(a, x, uw, uh, ud, b, aw, ah, ignored);
    (void)status;   // the forward pass reported domain errors already
    if (!live) {    // identity: tails, NaN, or an input the forward pass flagged
        g_x = gy;
        for (int i = 0; i < a.K; ++i) {
            g_uw[i] = 0.0;
            g_uh[i] = 0.0;
        }
        return;
    }
    const double d0 = b.d0, d1 = b.d1;
    const double in_w = b.cw1 - b.cw0, in_h = b.ch1 - b.ch0, delta = in_h / in_w;
    const double s = (d0 + d1) - 2.0 * delta;
    double theta;
    if (a.inverse) {
        const double yc = x - b.ch0;
        const double qa = yc * s + in_h * (delta - d0), qb = in_h * d0 - yc * s, qc = -delta * yc;
        theta = (2.0 * qc) / (-qb - sqrt(qb * qb - 4.0 * qa * qc));
    } else {
        theta = (x - b.cw0) / in_w;
    }
    // F(theta; in_h, delta, d0, d1) = N / Dn (the map inside the bin, output minus ch0),
    // G(theta; delta, d0, d1) = log(delta^2 A) - 2 log Dn (its log-derivative w.r.t. the input)
    const double t = theta * (1.0 - theta), tp = 1.0 - 2.0 * theta, omt = 1.0 - theta;
    const double Nn = in_h * (delta * (theta * theta) + d0 * t), Dn = delta + s * t;
    const double A = (d1 * (theta * theta) + (2.0 * delta) * t) + d0 * (omt * omt);
    const double rD = 1.0 / Dn, rD2 = rD * rD;
    const double F_theta = (in_h * (2.0 * delta * theta + d0 * tp) * Dn - Nn * (s * tp)) * rD2;
    const double F_inh = (delta * (theta * theta) + d0 * t) * rD;
    const double F_delta = (in_h * (theta * theta) * Dn - Nn * (1.0 - 2.0 * t)) * rD2;
    const double F_d0 = (in_h * t * Dn - Nn * t) * rD2;
    const double F_d1 = -Nn * t * rD2;
    const double G_theta = (2.0 * d1 * theta + 2.0 * delta * tp - 2.0 * d0 * omt) / A - 2.0 * (s * tp) * rD;
    const double G_delta = 2.0 / delta + 2.0 * t / A - 2.0 * (1.0 - 2.0 * t) * rD;
    const double G_d0 = (omt * omt) / A - 2.0 * t * rD;
    const double G_d1 = (theta * theta) / A - 2.0 * t * rD;
    double g_inh, g_inw, g_delta, g_d0, g_d1, g_cw0, g_ch0;
    if (!a.inverse) {
        // y = ch0 + F(theta, ...), theta = (x - cw0) / in_w, lad = G
        const double g_theta = gy * F_theta + gl * G_theta;
        g_inh = gy * F_inh;
        g_delta = gy * F_delta + gl * G_delta;
        g_d0 = gy * F_d0 + gl * G_d0;
        g_d1 = gy * F_d1 + gl * G_d1;
        g_ch0 = gy;
        g_x = g_theta / in_w;
        g_cw0 = -g_theta / in_w;
        g_inw = -g_theta * theta / in_w;
    } else {
        // out = root in_w + cw0, F(root, ...) = x - ch0, lad = -G(root, ...)
        const double g_root = gy * in_w - gl * G_theta;
        const double r = g_root / F_theta;
        g_x = r;
        g_ch0 = -r;
        g_inh = -r * F_inh;
        g_delta = -r * F_delta - gl * G_delta;
        g_d0 = -r * F_d0 - gl * G_d0;
        g_d1 = -r * F_d1 - gl * G_d1;
        g_inw = gy * theta;
        g_cw0 = gy;
    }
    // delta = in_h / in_w
    g_inh += g_delta / in_w;
    g_inw -= g_delta * delta / in_w;
    // in_w = cw1 - cw0, in_h = ch1 - ch0
    axis_backward(uw, g_uw, a.K, a.divisor, a.left, a.right, a.min_w, b.k, aw, g_cw0 - g_inw, g_inw);
    axis_backward(uh, g_uh, a.K, a.divisor, a.bottom, a.top, a.min_h, b.k, ah, g_ch0 - g_inh, g_inh);
    if (b.u0_is_logit) g_ud[b.i0] += g_d0 * softplus_slope(b.u0, a.beta);
    if (b.u1_is_logit) g_ud[b.i1] += g_d1 * softplus_slope(b.u1, a.beta);
}

}  // namespace f64
}  // namespace nfa
