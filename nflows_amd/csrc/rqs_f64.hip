// K5d: the rational-quadratic spline functional in float64 --
// unconstrained_rational_quadratic_spline / rational_quadratic_spline (splines/rational_quadratic.py:13-63 /
// :66-181) on double tensors, so that `.double()` flows (the reference is dtype-generic) and the float64
// ground truth of the parity tests run on the device.  Not a fast path: one lane per element, the logits
// read straight from global memory in three passes per axis (max, sum of exponentials, knot walk), every
// step the reference's own formula in double (softmax -> min + (1 - min K) p -> cumsum -> affine -> forced
// end knot, searchsorted with its +1e-6 on the last knot, the quadratic root of the inverse).
#include "common.hpp"

#include <math.h>

namespace nfa {

struct Rqs64Args {
    const double *x, *uw, *uh, *ud;
    int64_t sw, sh, sd, n;
    int nd;
    double* y;
    double* lad;
    int32_t* status;
    int K, linear, inverse;
    double left, right, bottom, top, min_w, min_h, min_d, beta, tail_logit, divisor;
};

__device__ __forceinline__ double softplus64(double x, double beta) {
    const double xb = x * beta;
    return xb > 20.0 ? x : log1p(exp(xb)) / beta;   // F.softplus(threshold = 20)
}

// knots of one axis on the fly.  SEARCH: the bin x falls into (count of knots <= x, minus one, the last
// knot moved up by 1e-6: torchutils.py:134-136), else the given bin; its two knots come back.
template <bool SEARCH>
__device__ __forceinline__ void axis_bin(const double* u, int K, double divisor, double lo, double hi, double minbin,
                                         double x, int& k, double& knot_lo, double& knot_hi) {
    double m = -INFINITY;
    for (int i = 0; i < K; ++i) {
        const double v = divisor != 0.0 ? u[i] / divisor : u[i];
        m = v > m ? v : m;
    }
    double sum = 0.0;
    for (int i = 0; i < K; ++i) sum += exp((divisor != 0.0 ? u[i] / divisor : u[i]) - m);
    const double one_minus = 1.0 - minbin * K;
    const double span = hi - lo;
    double acc = 0.0, prev = lo;
    int found = -1;
    for (int i = 0; i < K; ++i) {
        const double p = exp((divisor != 0.0 ? u[i] / divisor : u[i]) - m) / sum;
        acc += minbin + one_minus * p;
        const double next = i == K - 1 ? hi : span * acc + lo;
        const bool take = SEARCH ? (x >= prev) : (i == k);
        if (take) {
            found = i;
            knot_lo = prev;
            knot_hi = next;
        }
        prev = next;
    }
    if (SEARCH) k = (x >= hi + 1e-6) ? K : found;
}

__global__ void __launch_bounds__(kBlock) rqs_elementwise_f64_kernel(const Rqs64Args a) {
    int my_status = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const double x = a.x[i];
        double y = x, lad = 0.0;
        bool inside;
        if (a.linear) {
            inside = x >= a.left && x <= a.right;           // (NaN falls outside: identity, :26, :38-39)
        } else {
            inside = !(x < a.left || x > a.right);
            if (!inside) my_status |= NFA_STATUS_OUTSIDE_DOMAIN;   // :81-82
        }
        if (inside) {
            const double* uw = a.uw + i * a.sw;
            const double* uh = a.uh + i * a.sh;
            const double* ud = a.ud + i * a.sd;
            int k = -1;
            double cw0 = 0, cw1 = 0, ch0 = 0, ch1 = 0;
            if (a.inverse) {
                axis_bin<true>(uh, a.K, a.divisor, a.bottom, a.top, a.min_h, x, k, ch0, ch1);
                if (k >= 0 && k < a.K) axis_bin<false>(uw, a.K, a.divisor, a.left, a.right, a.min_w, x, k, cw0, cw1);
            } else {
                axis_bin<true>(uw, a.K, a.divisor, a.left, a.right, a.min_w, x, k, cw0, cw1);
                if (k >= 0 && k < a.K) axis_bin<false>(uh, a.K, a.divisor, a.bottom, a.top, a.min_h, x, k, ch0, ch1);
            }
            if (k < 0 || k >= a.K) {
                my_status |= NFA_STATUS_OUTSIDE_DOMAIN;
            } else {
                // derivative logits: linear tails pad both ends with the tail constant (:33-36)
                double u0, u1;
                if (a.linear) {
                    u0 = k == 0 ? a.tail_logit : ud[k - 1];
                    u1 = k >= a.nd ? a.tail_logit : ud[k];
                } else {
                    u0 = ud[k];
                    u1 = ud[k + 1];
                }
                const double d0 = a.min_d + softplus64(u0, a.beta), d1 = a.min_d + softplus64(u1, a.beta);
                const double in_w = cw1 - cw0, in_h = ch1 - ch0, delta = in_h / in_w;
                const double s = (d0 + d1) - 2.0 * delta;
                if (a.inverse) {   // :132-160
                    const double yc = x - ch0;
                    const double qa = yc * s + in_h * (delta - d0), qb = in_h * d0 - yc * s, qc = -delta * yc;
                    const double disc = qb * qb - 4.0 * qa * qc;
                    if (!(disc >= 0.0)) my_status |= NFA_STATUS_NEG_DISCRIMINANT;
                    const double root = (2.0 * qc) / (-qb - sqrt(disc));
                    y = root * in_w + cw0;
                    const double t1mt = root * (1.0 - root), den = delta + s * t1mt, omr = 1.0 - root;
                    const double dnum = (delta * delta) * ((d1 * (root * root) + (2.0 * delta) * t1mt) + d0 * (omr * omr));
                    lad = -(log(dnum) - 2.0 * log(den));
                } else {           // :162-181
                    const double theta = (x - cw0) / in_w, t1mt = theta * (1.0 - theta);
                    const double num = in_h * (delta * (theta * theta) + d0 * t1mt), den = delta + s * t1mt;
                    y = ch0 + num / den;
                    const double omt = 1.0 - theta;
                    const double dnum = (delta * delta) * ((d1 * (theta * theta) + (2.0 * delta) * t1mt) + d0 * (omt * omt));
                    lad = log(dnum) - 2.0 * log(den);
                }
            }
        }
        a.y[i] = y;
        a.lad[i] = lad;
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_elementwise_f64(const double* inputs, const double* unnormalized_widths, int64_t stride_w,
                                       const double* unnormalized_heights, int64_t stride_h,
                                       const double* unnormalized_derivatives, int64_t stride_d,
                                       int32_t num_derivatives, double* outputs, double* logabsdet, int32_t* status,
                                       int64_t n, const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    if (!spec || n < 0) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->num_bins < 1 || spec->num_bins > 4096) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->tails != NFA_TAILS_NONE && spec->tails != NFA_TAILS_LINEAR) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->min_bin_width * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (spec->min_bin_height * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    const int K = spec->num_bins, linear = spec->tails == NFA_TAILS_LINEAR;
    if (num_derivatives < (linear ? K - 1 : K + 1)) return NFA_ERR_INVALID_ARGUMENT;
    if (n == 0) return NFA_OK;
    if (!inputs || !unnormalized_widths || !unnormalized_heights || !outputs || !logabsdet ||
        (num_derivatives > 0 && !unnormalized_derivatives))
        return NFA_ERR_INVALID_ARGUMENT;
    Rqs64Args a;
    a.x = inputs;
    a.uw = unnormalized_widths;
    a.uh = unnormalized_heights;
    a.ud = unnormalized_derivatives;
    a.sw = stride_w;
    a.sh = stride_h;
    a.sd = stride_d;
    a.n = n;
    a.nd = num_derivatives;
    a.y = outputs;
    a.lad = logabsdet;
    a.status = status;
    a.K = K;
    a.linear = linear;
    a.inverse = inverse ? 1 : 0;
    // (linear tails: the box is [-tail_bound, tail_bound]^2, carried in `right` like the fp32 kernels read it)
    a.left = linear ? -spec->right : spec->left;
    a.right = spec->right;
    a.bottom = linear ? -spec->right : spec->bottom;
    a.top = linear ? spec->right : spec->top;
    a.min_w = spec->min_bin_width;
    a.min_h = spec->min_bin_height;
    a.min_d = spec->min_derivative;
    a.beta = spec->softplus_beta;
    a.tail_logit = spec->tail_logit;
    a.divisor = spec->wh_divisor;
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)device_cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rqs_elementwise_f64_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
