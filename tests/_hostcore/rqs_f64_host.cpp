// Test infrastructure: the per-element float64 spline arithmetic of the product (nflows_amd/csrc/rqs_f64_core.hpp,
// the functions the device kernels of rqs_f64.hip call) compiled for the HOST, so that the CPU suite can hold the
// forward values and the closed-form gradients to the reference's float64 results without a GPU.  Built by
// tests/test_oracle_golden.py with g++ into a temporary directory; nothing in the product loads it.
#include "rqs_f64_core.hpp"

using namespace nfa::f64;

extern "C" int host_rqs64(int backward, int64_t n, int K, int nd, int linear, int inverse, double left, double right,
                          double bottom, double top, double min_w, double min_h, double min_d, double beta,
                          double tail_logit, double divisor, const double* x, const double* uw, const double* uh,
                          const double* ud, const double* gy, const double* gl, double* out0, double* out1,
                          double* g_uw, double* g_uh, double* g_ud) {
    Spec s;
    s.K = K; s.nd = nd; s.linear = linear; s.inverse = inverse;
    s.left = left; s.right = right; s.bottom = bottom; s.top = top;
    s.min_w = min_w; s.min_h = min_h; s.min_d = min_d; s.beta = beta; s.tail_logit = tail_logit; s.divisor = divisor;
    int status = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (backward)
            backward_element(s, x[i], uw + i * K, uh + i * K, ud + i * nd, gy[i], gl[i], out0[i], g_uw + i * K,
                             g_uh + i * K, g_ud + i * nd, status);
        else
            forward_element(s, x[i], uw + i * K, uh + i * K, ud + i * nd, out0[i], out1[i], status);
    }
    return status;
}
