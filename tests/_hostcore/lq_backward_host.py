"""Test infrastructure: the bodies of the sibling splines' backward kernels (nflows_amd/csrc/splines_lq.hip:
linear_spline_backward_kernel, quadratic_spline_backward_kernel, cubic_spline_backward_kernel) turned into host functions by text substitution
-- one "lane", the LDS slot a static array, the device helpers of rqs_math.hpp (refined reciprocals, the custom
exponential) replaced by their IEEE counterparts -- and compiled with g++.  What this checks is the ALGEBRA of the
closed-form adjoints as written in the kernel source (against the reference's float64 autograd, to the same
4 x reference-fp32-error rule the GPU test applies); the device's own rounding is the GPU suite's business.
Nothing in the product loads this build."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HEADER = r'''
#include <math.h>
#include <stdint.h>
#include <string.h>
struct LqArgs { int K, nh, slot; float left, right, bottom, top, span_in, span_out, min_w, min_h, om_w, om_h, om_hk, divisor, rdivisor; };
struct LqBwdArgs { const float *x, *a0, *a1, *a2, *a3, *gy, *gl; float *gx, *g0, *g1, *g2, *g3; int64_t n; LqArgs f; };
static inline float div_with_rcp(float a, float b, float r) { return a / b; }
static inline float rcp_refined(float b) { return 1.0f / b; }
static inline float exp_noclamp(float x) { return expf(x); }
static inline float softplus_beta(float x, float beta) { return x > 20.f ? x : log1pf(expf(x)); }
static inline float sigmoid_of(float v) { return v > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-v)); }
template <int KT>
static inline void softmax_in_place(float* p, int Krt, float divisor, float rdivisor) {
    const int K = KT > 0 ? KT : Krt;
    float m = -INFINITY;
    for (int i = 0; i < K; ++i) { float u = p[i]; if (divisor != 0.0f) u = u / divisor; p[i] = u; m = fmaxf(m, u); }
    double s = 0.0;
    for (int i = 0; i < K; ++i) { const float e = expf(p[i] - m); p[i] = e; s += (double)e; }
    const float sum = (float)s;
    for (int i = 0; i < K; ++i) p[i] = p[i] / sum;
}
'''

TAIL = r'''
static void fill(LqBwdArgs& b, int K, int nh, int slot, float lo, float hi) {
    b.f.K = K; b.f.nh = nh; b.f.slot = slot; b.f.left = lo; b.f.right = hi; b.f.bottom = lo; b.f.top = hi;
    b.f.span_in = hi - lo; b.f.span_out = hi - lo; b.f.min_w = 1e-3f; b.f.min_h = 1e-3f;
    b.f.om_w = (float)(1.0 - 1e-3 * K); b.f.om_h = (float)(1.0 - 1e-3); b.f.om_hk = (float)(1.0 - 1e-3 * K);
    b.f.divisor = 0; b.f.rdivisor = 0;
}
extern "C" void linear(int inverse, int64_t n, int K, float lo, float hi, const float* x, const float* pdf, const float* gy,
                       const float* gl, float* gx, float* g0) {
    LqBwdArgs b; memset(&b, 0, sizeof b);
    b.x = x; b.a0 = pdf; b.gy = gy; b.gl = gl; b.gx = gx; b.g0 = g0; b.n = n;
    fill(b, K, 0, K | 1, lo, hi);
    if (inverse) linear_host<0, true>(b); else linear_host<0, false>(b);
}
extern "C" void quadratic(int inverse, int64_t n, int K, int nh, float lo, float hi, const float* x, const float* w,
                          const float* h, const float* gy, const float* gl, float* gx, float* g0, float* g1) {
    LqBwdArgs b; memset(&b, 0, sizeof b);
    b.x = x; b.a0 = w; b.a1 = h; b.gy = gy; b.gl = gl; b.gx = gx; b.g0 = g0; b.g1 = g1; b.n = n;
    fill(b, K, nh, (5 * K + 3) | 1, lo, hi);
    if (inverse) quadratic_host<0, false, true>(b); else quadratic_host<0, false, false>(b);
}
extern "C" void cubic(int inverse, int64_t n, int K, float lo, float hi, const float* x, const float* w, const float* h,
                      const float* dl, const float* dr, const float* gy, const float* gl, float* gx, float* g0, float* g1,
                      float* g2, float* g3) {
    LqBwdArgs b; memset(&b, 0, sizeof b);
    b.x = x; b.a0 = w; b.a1 = h; b.a2 = dl; b.a3 = dr; b.gy = gy; b.gl = gl; b.gx = gx; b.g0 = g0; b.g1 = g1; b.g2 = g2;
    b.g3 = g3; b.n = n;
    fill(b, K, K, (4 * K) | 1, lo, hi);
    if (inverse) cubic_host<0, true>(b); else cubic_host<0, false>(b);
}
'''


def _host_body(src, start, stop, kernel, host):
    body = src[src.index(start):src.index(stop)]
    decl = "__global__ void __launch_bounds__(kBlock) %s(const LqBwdArgs b) {" % kernel
    assert body.count(decl) == 1, kernel
    body = body.replace(decl, "void %s(const LqBwdArgs b) {" % host)
    body = body.replace("#pragma clang fp contract(off)", "")
    lds = "    extern __shared__ __attribute__((aligned(16))) float lds[];"
    assert body.count(lds) == 1
    body = body.replace(lds, "    static float lds[1 << 15];")
    body = body.replace("(int64_t)blockIdx.x * blockDim.x + threadIdx.x", "0").replace("threadIdx.x", "0")
    body = body.replace("i += (int64_t)gridDim.x * blockDim.x", "i += 1")
    assert "blockIdx" not in body and "__shfl" not in body and "__syncthreads" not in body
    return body


def build(out_dir):
    src = open(os.path.join(ROOT, "nflows_amd", "csrc", "splines_lq.hip")).read()
    h0 = src.index("// torchutils.cbrt (torchutils.py:139-141)")
    h1 = src.index("// splines/cubic.py:63-267.  w / h: K width / height logits (overwritten")
    helpers = src[h0:h1].replace("__device__ __forceinline__", "static inline").replace("#pragma clang fp contract(off)", "")
    lin = _host_body(src, "template <int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) linear_spline_backward_kernel",
                     "template <int KT, bool DERIVED, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) quadratic_spline_backward_kernel",
                     "linear_spline_backward_kernel", "linear_host")
    quad = _host_body(src, "template <int KT, bool DERIVED, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) quadratic_spline_backward_kernel",
                      "// Cubic spline (splines/cubic.py:63-267).  Only the searched bin", "quadratic_spline_backward_kernel",
                      "quadratic_host")
    cub = _host_body(src, "template <int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) cubic_spline_backward_kernel",
                     "static int launch_lq_backward", "cubic_spline_backward_kernel", "cubic_host")
    cpp = os.path.join(out_dir, "lq_backward_host.cpp")
    so = os.path.join(out_dir, "lq_backward_host.so")
    with open(cpp, "w") as f:
        f.write(HEADER + helpers + lin + quad + cub + TAIL)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", cpp, "-o", so])
    lib = ctypes.CDLL(so)
    p, f32, i32 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.linear.argtypes = [i32, ctypes.c_int64, i32, f32, f32] + [p] * 6
    lib.quadratic.argtypes = [i32, ctypes.c_int64, i32, i32, f32, f32] + [p] * 8
    lib.cubic.argtypes = [i32, ctypes.c_int64, i32, f32, f32] + [p] * 12
    return lib
