// K5d: the rational-quadratic spline functional in float64 --
// unconstrained_rational_quadratic_spline / rational_quadratic_spline (splines/rational_quadratic.py:13-63 /
// :66-181) on double tensors, so that `.double()` flows (the reference is dtype-generic) and the float64
// ground truth of the parity tests run on the device.  Not a fast path: one lane per element, the logits
// read straight from global memory in three passes per axis (max, sum of exponentials, knot walk), every
// step the reference's own formula in double (softmax -> min + (1 - min K) p -> cumsum -> affine -> forced
// end knot, searchsorted with its +1e-6 on the last knot, the quadratic root of the inverse).
#include "common.hpp"
#include "rqs_f64_core.hpp"

namespace nfa {

struct Rqs64Args {
    const double *x, *uw, *uh, *ud;
    int64_t sw, sh, sd, n;
    double* y;
    double* lad;
    int32_t* bins;   // may be null
    int32_t* status;
    f64::Spec s;
};

__global__ void __launch_bounds__(kBlock) rqs_elementwise_f64_kernel(const Rqs64Args a) {
    int my_status = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        double y, lad;
        f64::forward_element(a.s, a.x[i], a.uw + i * a.sw, a.uh + i * a.sh, a.ud + i * a.sd, y, lad, my_status,
                             a.bins ? a.bins + i : nullptr);
        a.y[i] = y;
        a.lad[i] = lad;
    }
    if (my_status && a.status) atomicOr(a.status, my_status);
}

// K5d-backward: one lane per element, the spline rebuilt from its logits (three passes per axis, as in the
// forward kernel), the closed-form adjoints of rqs_f64_core.hpp, dense [n, K] / [n, K] / [n, nd] gradients.
struct Rqs64BwdArgs {
    Rqs64Args f;
    const double *gy, *gl;   // upstream gradients (either may be null = zeros)
    double *gx, *guw, *guh, *gud;
};

__global__ void __launch_bounds__(kBlock) rqs_elementwise_backward_f64_kernel(const Rqs64BwdArgs b) {
    const Rqs64Args& a = b.f;
    int ignored = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        double gx;
        f64::backward_element(a.s, a.x[i], a.uw + i * a.sw, a.uh + i * a.sh, a.ud + i * a.sd,
                              b.gy ? b.gy[i] : 0.0, b.gl ? b.gl[i] : 0.0, gx, b.guw + i * a.s.K, b.guh + i * a.s.K,
                              b.gud + i * a.s.nd, ignored);
        b.gx[i] = gx;
    }
}

// shared argument checks and the spec -> kernel-argument translation of the two entry points
static int fill_f64(Rqs64Args& a, const double* inputs, const double* unnormalized_widths, int64_t stride_w,
                    const double* unnormalized_heights, int64_t stride_h, const double* unnormalized_derivatives,
                    int64_t stride_d, int32_t num_derivatives, int64_t n, const nfa_rqs_spec* spec, int32_t inverse) {
    if (!spec || n < 0) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->num_bins < 1 || spec->num_bins > 4096) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->tails != NFA_TAILS_NONE && spec->tails != NFA_TAILS_LINEAR) return NFA_ERR_INVALID_ARGUMENT;
    if (spec->min_bin_width * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_WIDTH;
    if (spec->min_bin_height * spec->num_bins > 1.0) return NFA_ERR_MIN_BIN_HEIGHT;
    const int K = spec->num_bins, linear = spec->tails == NFA_TAILS_LINEAR;
    if (num_derivatives < (linear ? K - 1 : K + 1)) return NFA_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!inputs || !unnormalized_widths || !unnormalized_heights ||
                  (num_derivatives > 0 && !unnormalized_derivatives)))
        return NFA_ERR_INVALID_ARGUMENT;
    a.x = inputs;
    a.uw = unnormalized_widths;
    a.uh = unnormalized_heights;
    a.ud = unnormalized_derivatives;
    a.sw = stride_w;
    a.sh = stride_h;
    a.sd = stride_d;
    a.n = n;
    a.s.nd = num_derivatives;
    a.s.K = K;
    a.s.linear = linear;
    a.s.inverse = inverse ? 1 : 0;
    // (linear tails: the box is [-tail_bound, tail_bound]^2, carried in `right` like the fp32 kernels read it)
    a.s.left = linear ? -spec->right : spec->left;
    a.s.right = spec->right;
    a.s.bottom = linear ? -spec->right : spec->bottom;
    a.s.top = linear ? spec->right : spec->top;
    a.s.min_w = spec->min_bin_width;
    a.s.min_h = spec->min_bin_height;
    a.s.min_d = spec->min_derivative;
    a.s.beta = spec->softplus_beta;
    a.s.tail_logit = spec->tail_logit;
    a.s.divisor = spec->wh_divisor;
    return NFA_OK;
}

static unsigned grid_f64(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)device_cu_count() * 16;
    return (unsigned)(blocks > cap ? cap : blocks);
}

}  // namespace nfa

using namespace nfa;

extern "C" int nfa_rqs_elementwise_f64(const double* inputs, const double* unnormalized_widths, int64_t stride_w,
                                       const double* unnormalized_heights, int64_t stride_h,
                                       const double* unnormalized_derivatives, int64_t stride_d,
                                       int32_t num_derivatives, double* outputs, double* logabsdet, int32_t* bin_idx,
                                       int32_t* status, int64_t n, const nfa_rqs_spec* spec, int32_t inverse,
                                       void* stream) {
    Rqs64Args a;
    const int rc = fill_f64(a, inputs, unnormalized_widths, stride_w, unnormalized_heights, stride_h,
                            unnormalized_derivatives, stride_d, num_derivatives, n, spec, inverse);
    if (rc != NFA_OK) return rc;
    if (n == 0) return NFA_OK;
    if (!outputs || !logabsdet) return NFA_ERR_INVALID_ARGUMENT;
    a.y = outputs;
    a.lad = logabsdet;
    a.bins = bin_idx;
    a.status = status;
    hipLaunchKernelGGL(rqs_elementwise_f64_kernel, dim3(grid_f64(n)), dim3(kBlock), 0, (hipStream_t)stream, a);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}

extern "C" int nfa_rqs_elementwise_backward_f64(const double* inputs, const double* unnormalized_widths,
                                                int64_t stride_w, const double* unnormalized_heights, int64_t stride_h,
                                                const double* unnormalized_derivatives, int64_t stride_d,
                                                int32_t num_derivatives, const double* grad_outputs,
                                                const double* grad_logabsdet, double* grad_inputs,
                                                double* grad_widths, double* grad_heights, double* grad_derivatives,
                                                int64_t n, const nfa_rqs_spec* spec, int32_t inverse, void* stream) {
    Rqs64BwdArgs b;
    const int rc = fill_f64(b.f, inputs, unnormalized_widths, stride_w, unnormalized_heights, stride_h,
                            unnormalized_derivatives, stride_d, num_derivatives, n, spec, inverse);
    if (rc != NFA_OK) return rc;
    if (n == 0) return NFA_OK;
    if (!grad_inputs || !grad_widths || !grad_heights || (num_derivatives > 0 && !grad_derivatives))
        return NFA_ERR_INVALID_ARGUMENT;
    b.f.y = nullptr;
    b.f.lad = nullptr;
    b.f.bins = nullptr;
    b.f.status = nullptr;
    b.gy = grad_outputs;
    b.gl = grad_logabsdet;
    b.gx = grad_inputs;
    b.guw = grad_widths;
    b.guh = grad_heights;
    b.gud = grad_derivatives;
    hipLaunchKernelGGL(rqs_elementwise_backward_f64_kernel, dim3(grid_f64(n)), dim3(kBlock), 0, (hipStream_t)stream, b);
    NFA_HIP_CHECK(hipGetLastError());
    return NFA_OK;
}
