"""Test infrastructure: the sibling splines' arithmetic of the product (nflows_amd/csrc/splines_lq.hip) compiled for the
HOST on top of the real helpers of rqs_math.hpp (tests/_hostcore/rqs_f32_host.py's shim: `__device__` defined away, the
three gfx950 transcendental builtins mapped to libm):
  * the per-element forward functions `linear_eval` / `quadratic_eval` / `cubic_eval` as they are, driven by the same
    in-box / tails wrapper the kernel `spline_lq_kernel` applies, with `fill_common` building the arguments;
  * the bodies of the three backward kernels turned into host functions by text substitution (one "lane", the LDS
    slot a static array).
Built with g++ into a temporary directory by the CPU suite; nothing in the product loads it."""
import ctypes
import os
import subprocess

from _hostcore.rqs_f32_host import SHIM

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HARNESS = r'''
using namespace nfa;

// the element wrapper of spline_lq_kernel: inside the box the spline, outside the identity (linear tails) or the
// domain flag (constrained)
template <int KIND, int KT, bool INVERSE>
static int forward_all(LqArgs a, int64_t n, const float* x, const float* l0, const float* l1, const float* l2,
                       const float* l3, float* y, float* lad) {
    const int K = a.K;
    const int hshift = (KIND == kQuadratic && a.nh == K - 1) ? 1 : 0;
    int status = 0;
    static float slot[3 * 4096 + 8];
    for (int64_t i = 0; i < n; ++i) {
        memset(slot, 0, sizeof(float) * (2 * K + 4));
        memcpy(slot, l0 + i * K, sizeof(float) * K);
        if (KIND == kQuadratic) memcpy(slot + K + hshift, l1 + i * a.nh, sizeof(float) * a.nh);
        if (KIND == kCubic) memcpy(slot + K, l1 + i * K, sizeof(float) * K);
        float yy = x[i], ll = 0.0f;
        if (x[i] >= a.left && x[i] <= a.right) {
            if (KIND == kLinear) status |= linear_eval<KT, INVERSE>(x[i], slot, a, yy, ll);
            else if (KIND == kQuadratic) status |= quadratic_eval<KT, INVERSE>(x[i], slot, slot + K, a, yy, ll);
            else status |= cubic_eval<KT, INVERSE>(x[i], slot, slot + K, l2[i], l3[i], a, yy, ll);
        } else if (!a.unconstrained) {
            status |= NFA_STATUS_OUTSIDE_DOMAIN;
        }
        y[i] = yy;
        lad[i] = ll;
    }
    return status;
}

#define FWD(KIND_)                                                                                           \
    (kt == 8 ? (inverse ? forward_all<KIND_, 8, true>(a, n, x, l0, l1, l2, l3, y, lad)                       \
                        : forward_all<KIND_, 8, false>(a, n, x, l0, l1, l2, l3, y, lad))                     \
     : kt == 10 ? (inverse ? forward_all<KIND_, 10, true>(a, n, x, l0, l1, l2, l3, y, lad)                   \
                           : forward_all<KIND_, 10, false>(a, n, x, l0, l1, l2, l3, y, lad))                 \
                : (inverse ? forward_all<KIND_, 0, true>(a, n, x, l0, l1, l2, l3, y, lad)                    \
                           : forward_all<KIND_, 0, false>(a, n, x, l0, l1, l2, l3, y, lad)))

// kind: 0 linear, 1 quadratic, 2 cubic; kt: compile-time bin count of the instance (8, 10) or 0 = run-time K
extern "C" int lq_forward(int kind, int kt, int inverse, int64_t n, const nfa_rqs_spec* spec, int nh, const float* x,
                          const float* l0, const float* l1, const float* l2, const float* l3, float* y, float* lad) {
    LqArgs a;
    memset(&a, 0, sizeof a);
    if (fill_common(a, spec) != NFA_OK) return -1;
    a.nh = kind == kQuadratic ? nh : (kind == kCubic ? a.K : 0);
    return kind == kLinear ? FWD(kLinear) : (kind == kQuadratic ? FWD(kQuadratic) : FWD(kCubic));
}

extern "C" int lq_backward(int kind, int inverse, int64_t n, const nfa_rqs_spec* spec, int nh, const float* x,
                           const float* l0, const float* l1, const float* l2, const float* l3, const float* gy,
                           const float* gl, float* gx, float* g0, float* g1, float* g2, float* g3) {
    LqBwdArgs b;
    memset(&b, 0, sizeof b);
    if (fill_common(b.f, spec) != NFA_OK) return -1;
    const int K = b.f.K;
    b.f.nh = kind == kQuadratic ? nh : (kind == kCubic ? K : 0);
    b.f.slot = (kind == kLinear ? K : (kind == kCubic ? 4 * K : 5 * K + 3)) | 1;   // launch_lq_backward's
    b.x = x; b.a0 = l0; b.a1 = l1; b.a2 = l2; b.a3 = l3; b.gy = gy; b.gl = gl;
    b.gx = gx; b.g0 = g0; b.g1 = g1; b.g2 = g2; b.g3 = g3; b.n = n;
    if (kind == kLinear) { if (inverse) linear_host<0, true>(b); else linear_host<0, false>(b); }
    else if (kind == kQuadratic) { if (inverse) quadratic_host<0, false, true>(b); else quadratic_host<0, false, false>(b); }
    else { if (inverse) cubic_host<0, true>(b); else cubic_host<0, false>(b); }
    return 0;
}
'''


def _host_body(src, start, stop, kernel, host):
    body = src[src.index(start):src.index(stop)]
    decl = "__global__ void __launch_bounds__(kBlock) %s(const LqBwdArgs b) {" % kernel
    assert body.count(decl) == 1, kernel
    body = body.replace(decl, "void %s(const LqBwdArgs b) {" % host)
    lds = "    extern __shared__ __attribute__((aligned(16))) float lds[];"
    assert body.count(lds) == 1
    body = body.replace(lds, "    static float lds[1 << 15];")
    body = body.replace("(int64_t)blockIdx.x * blockDim.x + threadIdx.x", "0").replace("threadIdx.x", "0")
    body = body.replace("i += (int64_t)gridDim.x * blockDim.x", "i += 1")
    assert "blockIdx" not in body and "__shfl" not in body and "__syncthreads" not in body
    return body


def build(out_dir):
    csrc = os.path.join(ROOT, "nflows_amd", "csrc")
    math_src = open(os.path.join(csrc, "rqs_math.hpp")).read()
    math_src = math_src.replace('#include "common.hpp"', SHIM).replace("#pragma once", "", 1)
    src = open(os.path.join(csrc, "splines_lq.hip")).read()
    fwd = src[src.index("namespace nfa {\n\nenum { kLinear = 0"):src.index("template <int KIND, int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) spline_lq_kernel")]
    fill = src[src.index("static int fill_common(LqArgs& a, const nfa_rqs_spec* spec) {"):src.index("// ------------------------------------------------------------------------------------------\n// Backward of the linear and quadratic")]
    decls = src[src.index("struct LqBwdArgs {"):src.index("template <int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) linear_spline_backward_kernel")]
    lin = _host_body(src, "template <int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) linear_spline_backward_kernel",
                     "template <int KT, bool DERIVED, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) quadratic_spline_backward_kernel",
                     "linear_spline_backward_kernel", "linear_host")
    quad = _host_body(src, "template <int KT, bool DERIVED, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) quadratic_spline_backward_kernel",
                      "// Cubic spline (splines/cubic.py:63-267).  Only the searched bin", "quadratic_spline_backward_kernel",
                      "quadratic_host")
    cub = _host_body(src, "template <int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) cubic_spline_backward_kernel",
                     "static int launch_lq_backward", "cubic_spline_backward_kernel", "cubic_host")
    cpp = os.path.join(out_dir, "splines_lq_host.cpp")
    so = os.path.join(out_dir, "splines_lq_host.so")
    with open(cpp, "w") as f:
        f.write(math_src + fwd + fill + decls + lin + quad + cub + "\n}  // namespace nfa\n" + HARNESS)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), cpp, "-o", so])
    lib = ctypes.CDLL(so)
    p, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.lq_forward.argtypes = [i32, i32, i32, i64, p, i32] + [p] * 7
    lib.lq_backward.argtypes = [i32, i32, i64, p, i32] + [p] * 12
    lib.lq_forward.restype = lib.lq_backward.restype = i32
    return lib
