"""Test infrastructure: the fp32 spline arithmetic of the product compiled for the HOST -- nflows_amd/csrc/rqs_math.hpp
as it is (its `#include "common.hpp"` replaced by a dozen lines that define away `__device__` and map the three gfx950
builtins it uses -- v_rcp_f32, v_log_f32, v_exp_f32 -- onto 1/x, log2f, exp2f) plus the per-spline gradient function
`rqs_backward` cut out of rqs_bwd.hip, plus rqs_fused8.hpp (K8h's sliced evaluation) with its two lane-mask asm blocks
written out per lane.  The functions are the ones the kernels K1 / K5 (`rqs_eval`), the whole-layer kernels
(`rqs_eval_flat8`, `FlatSteps`, `FusedSteps`) and K1-backward (`rqs_backward`) call per lane; the hardware's transcendental approximations
(1 ulp) are the only thing the host build replaces.  Built with g++ into a temporary directory by the CPU suite;
nothing in the product loads it."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHIM = r'''
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "nflows_amd.h"
#define __device__
#define __host__
#define __forceinline__ inline
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
namespace nfa { constexpr int kBlock = 256; constexpr int kWave = 64; }
'''

HARNESS = r'''
using namespace nfa;

template <int KT, bool INVERSE, bool LINEAR>
static int forward_all(int64_t n, const RqsDev& sp, const float* x, const float* params, float* y, float* lad) {
    int status = 0;
    float slot[3 * 4096 + 2];
    for (int64_t i = 0; i < n; ++i) {
        memcpy(slot, params + i * sp.P, sizeof(float) * sp.P);
        status |= rqs_eval<KT, INVERSE, LINEAR>(x[i], slot, sp, y[i], lad[i]);
    }
    return status;
}

template <int KT, bool INVERSE, bool LINEAR>
static int backward_all(int64_t n, const RqsDev& sp, const float* x, const float* params, const float* gy, const float* gl,
                        float* gx, float* gparams) {
    int status = 0;
    for (int64_t i = 0; i < n; ++i) {
        float* slot = gparams + i * sp.P;   // the function overwrites the lane's logits with their gradients
        memcpy(slot, params + i * sp.P, sizeof(float) * sp.P);
        gx[i] = rqs_backward<KT, INVERSE, LINEAR>(x[i], slot, sp, gy[i], gl[i], status);
    }
    return status;
}

// the register-array form (REGS = true: K12's per-step evaluation, made_inverse.hip, and K1's wave-tile kernel, rqs.hip):
// same arithmetic, the two derivative logits picked by a select chain instead of a dynamic index
template <int KT, bool INVERSE>
static int forward_regs_all(int64_t n, const RqsDev& sp, const float* x, const float* params, float* y, float* lad) {
    int status = 0;
    for (int64_t i = 0; i < n; ++i) {
        float p[3 * KT - 1];
        memcpy(p, params + i * sp.P, sizeof p);
        status |= rqs_eval<KT, INVERSE, true, true>(x[i], p, sp, y[i], lad[i]);
    }
    return status;
}

extern "C" int host_rqs_forward_regs(int kt, int inverse, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                     const float* params, float* y, float* lad) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK || !sp.linear || sp.K != kt || sp.P != 3 * kt - 1) return -1;
#define REGS_CASE(KT_) case KT_: return inverse ? forward_regs_all<KT_, true>(n, sp, x, params, y, lad) : forward_regs_all<KT_, false>(n, sp, x, params, y, lad);
    switch (kt) {   // (every bin count the whole-layer kernels are built for: the exact kernel's plain loop runs this instance)
        REGS_CASE(2) REGS_CASE(3) REGS_CASE(4) REGS_CASE(5) REGS_CASE(6) REGS_CASE(7) REGS_CASE(8) REGS_CASE(9) REGS_CASE(10)
        REGS_CASE(11) REGS_CASE(12) REGS_CASE(13) REGS_CASE(14) REGS_CASE(15) REGS_CASE(16) REGS_CASE(20) REGS_CASE(24) REGS_CASE(32)
    }
#undef REGS_CASE
    return -1;
}

// the bin the search of rqs_eval chose (its `bin` output: what K1 / K5 store to `bin_idx`)
template <int KT, bool INVERSE, bool LINEAR>
static int bins_all(int64_t n, const RqsDev& sp, const float* x, const float* params, int32_t* bins) {
    int status = 0;
    float slot[3 * 4096 + 2];
    for (int64_t i = 0; i < n; ++i) {
        memcpy(slot, params + i * sp.P, sizeof(float) * sp.P);
        float y, lad;
        int b = -7;
        status |= rqs_eval<KT, INVERSE, LINEAR>(x[i], slot, sp, y, lad, &b);
        bins[i] = b;
    }
    return status;
}

#define DISPATCH(FN, ...)                                                                                     \
    do {                                                                                                      \
        const bool lin = sp.linear != 0;                                                                      \
        if (kt == 8) return inverse ? (lin ? FN<8, true, true>(__VA_ARGS__) : FN<8, true, false>(__VA_ARGS__)) \
                                    : (lin ? FN<8, false, true>(__VA_ARGS__) : FN<8, false, false>(__VA_ARGS__)); \
        if (kt == 4) return inverse ? (lin ? FN<4, true, true>(__VA_ARGS__) : FN<4, true, false>(__VA_ARGS__))   \
                                    : (lin ? FN<4, false, true>(__VA_ARGS__) : FN<4, false, false>(__VA_ARGS__)); \
        if (kt == 10) return inverse ? (lin ? FN<10, true, true>(__VA_ARGS__) : FN<10, true, false>(__VA_ARGS__)) \
                                     : (lin ? FN<10, false, true>(__VA_ARGS__) : FN<10, false, false>(__VA_ARGS__)); \
        return inverse ? (lin ? FN<0, true, true>(__VA_ARGS__) : FN<0, true, false>(__VA_ARGS__))              \
                       : (lin ? FN<0, false, true>(__VA_ARGS__) : FN<0, false, false>(__VA_ARGS__));           \
    } while (0)

// kt: the compile-time bin count of the instance to run (4, 8, 10) or 0 = the run-time-K instance
extern "C" int host_rqs_forward(int kt, int inverse, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                const float* params, float* y, float* lad) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK) return -1;
    DISPATCH(forward_all, n, sp, x, params, y, lad);
}

extern "C" int host_rqs_bins(int kt, int inverse, int64_t n, const nfa_rqs_spec* spec, const float* x,
                             const float* params, int32_t* bins) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK) return -1;
    DISPATCH(bins_all, n, sp, x, params, bins);
}

extern "C" int host_rqs_backward(int kt, int inverse, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                 const float* params, const float* gy, const float* gl, float* gx, float* gparams) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK) return -1;
    DISPATCH(backward_all, n, sp, x, params, gy, gl, gx, gparams);
}

// the whole-layer kernels' evaluation (8 bins, linear tails): rqs_eval_flat8, logits as the reference defines them
extern "C" int host_rqs_forward_flat8(int inverse, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                      const float* params, float* y, float* lad) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK || sp.K != 8 || !sp.linear) return -1;
    int status = 0;
    for (int64_t i = 0; i < n; ++i)
        status |= inverse ? rqs_eval_flat8<true>(x[i], params + i * sp.P, sp, y[i], lad[i])
                          : rqs_eval_flat8<false>(x[i], params + i * sp.P, sp, y[i], lad[i]);
    return status;
}

// the whole-layer kernel K8's sliced evaluation (rqs_math.hpp: FlatSteps, every slice in order): variant 0 = the
// reference's exact rounding sequence, 8 bins; 1 = the shorter sequence (FAST), 8 bins; 2 = FAST, 10 bins.  Logits as
// the kernel sees them: already divided by sqrt(hidden) (PRESCALED = 1; the fixtures carry no divisor).
template <class Steps, int KT>
static int flatsteps_all(int64_t n, RqsDev sp, const float* x, const float* params, float* y, float* lad) {
    sp.divisor = 0.0f;
    int status = 0;
    for (int64_t i = 0; i < n; ++i) {
        Steps f;
        memset(&f, 0, sizeof f);
        const float* p = params + i * sp.P;
        f.x = x[i];
        for (int j = 0; j < KT; ++j) {
            f.ew[j] = p[j];
            f.eh[j] = p[KT + j];
            if (j < KT - 1) f.sd[j] = p[2 * KT + j];
        }
        flat_steps_all(f, sp);
        y[i] = f.y;
        lad[i] = f.lad;
        status |= f.status;
    }
    return status;
}

extern "C" int host_rqs_forward_flatsteps(int variant, int inverse, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                          const float* params, float* y, float* lad) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK || !sp.linear || sp.K != (variant == 2 ? 10 : 8)) return -1;
    if (variant == 0) return inverse ? flatsteps_all<FlatSteps<true, 1, false, 8>, 8>(n, sp, x, params, y, lad)
                                     : flatsteps_all<FlatSteps<false, 1, false, 8>, 8>(n, sp, x, params, y, lad);
    if (variant == 1) return inverse ? flatsteps_all<FlatSteps<true, 1, true, 8>, 8>(n, sp, x, params, y, lad)
                                     : flatsteps_all<FlatSteps<false, 1, true, 8>, 8>(n, sp, x, params, y, lad);
    return inverse ? flatsteps_all<FlatSteps<true, 1, true, 10>, 10>(n, sp, x, params, y, lad)
                   : flatsteps_all<FlatSteps<false, 1, true, 10>, 10>(n, sp, x, params, y, lad);
}

// the whole-layer kernel K8h's sliced evaluation (rqs_fused8.hpp: FusedSteps, every slice in order) on logits scaled
// by 1 / kappa (a power of two), the way the kernel hands them over; 8 or 10 bins, linear tails, softplus beta = 1
template <class Steps, int KT>
static int fused_all(int64_t n, const RqsDev& sp, float kappa, const float* x, const float* params, float* y, float* lad,
                     int32_t* bins = nullptr) {
    int status = 0;
    const float inv_kappa = 1.0f / kappa;
    for (int64_t i = 0; i < n; ++i) {
        Steps f;
        memset(&f, 0, sizeof f);
        const float* p = params + i * sp.P;
        f.x = x[i];
        f.kappa = kappa;
        f.kl2e = 1.44269502162933349609375f * kappa;
        f.tail_s = sp.tail_logit * inv_kappa;
        for (int j = 0; j < KT; ++j) {
            f.ew[j] = p[j] * inv_kappa;
            f.eh[j] = p[KT + j] * inv_kappa;
            if (j < KT - 1) f.sd[j] = p[2 * KT + j] * inv_kappa;
        }
        flat_steps_all(f, sp);
        y[i] = f.y;
        lad[i] = f.lad;
        status |= f.status;
        if (bins) bins[i] = f.kbin;
    }
    return status;
}

// the diagnostic form of K8h's evaluation (FusedSteps<.., DBG = true>: rqs_resnet_f16_dbg.hip), 8 bins: values AND the bin
extern "C" int host_rqs_forward_fused_bins(int inverse, float kappa, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                           const float* params, float* y, float* lad, int32_t* bins) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK || !sp.linear || sp.K != 8) return -1;
    return inverse ? fused_all<FusedSteps<true, 8, true>, 8>(n, sp, kappa, x, params, y, lad, bins)
                   : fused_all<FusedSteps<false, 8, true>, 8>(n, sp, kappa, x, params, y, lad, bins);
}

extern "C" int host_rqs_forward_fused(int inverse, float kappa, int64_t n, const nfa_rqs_spec* spec, const float* x,
                                      const float* params, float* y, float* lad) {
    RqsDev sp;
    if (make_dev_spec(spec, &sp) != NFA_OK || !sp.linear) return -1;
#define FUSED_CASE(KT_) case KT_: return inverse ? fused_all<FusedSteps<true, KT_>, KT_>(n, sp, kappa, x, params, y, lad) \
                                                 : fused_all<FusedSteps<false, KT_>, KT_>(n, sp, kappa, x, params, y, lad);
    switch (sp.K) {
        FUSED_CASE(2) FUSED_CASE(3) FUSED_CASE(4) FUSED_CASE(5) FUSED_CASE(6) FUSED_CASE(7) FUSED_CASE(8) FUSED_CASE(9) FUSED_CASE(10)
        FUSED_CASE(11) FUSED_CASE(12) FUSED_CASE(13) FUSED_CASE(14) FUSED_CASE(15) FUSED_CASE(16) FUSED_CASE(20) FUSED_CASE(24) FUSED_CASE(32)
    }
#undef FUSED_CASE
    return -1;
}
'''

# rqs_fused8.hpp holds two inline-asm blocks that work on the wave's lane MASK (a compare written to an SGPR pair, six
# selects under it): per lane they are a comparison and six conditional moves, which is what the host build puts there
WALK_ASM_C = '''        const unsigned long long next = x >= next_lower ? 1ull : 0ull;
        if (take) {
            cw0 = kw;
            cw1 = kwn;
            ch0 = kh;
            ch1 = khn;
            u0 = cand_u0;
            u1 = cand_u1;
        }
        take = next;
'''


def _fused8_for_the_host(csrc):
    src = open(os.path.join(csrc, "rqs_fused8.hpp")).read()
    src = src.replace("#pragma once", "", 1).replace('#include "rqs_math.hpp"', "", 1)
    a = src.index("        unsigned long long next;\n        asm(\"v_cmp_ge_f32 %6, %7, %8")
    b = src.index("        take = next;\n", a) + len("        take = next;\n")
    src = src[:a] + WALK_ASM_C + src[b:]
    first = 'asm("v_cmp_ge_f32 %0, %1, %2" : "=s"(take) : "v"(x), "v"(INVERSE ? k1h : k1w));'
    assert src.count(first) == 1
    src = src.replace(first, "take = x >= (INVERSE ? k1h : k1w) ? 1ull : 0ull;")
    assert "asm(" not in src
    return src


ACT_HARNESS = r'''
// the conditioner blocks' activations as the whole-layer kernels evaluate them (fused_common.hpp: activate<ACT>)
extern "C" int host_activate(int act, int64_t n, const float* x, float* y) {
    for (int64_t i = 0; i < n; ++i) {
        switch (act) {
            case nfa::kActNone: y[i] = nfa::activate<nfa::kActNone>(x[i]); break;
            case nfa::kActRelu: y[i] = nfa::activate<nfa::kActRelu>(x[i]); break;
            case nfa::kActLeakyRelu: y[i] = nfa::activate<nfa::kActLeakyRelu>(x[i]); break;
            case nfa::kActElu: y[i] = nfa::activate<nfa::kActElu>(x[i]); break;
            case nfa::kActTanh: y[i] = nfa::activate<nfa::kActTanh>(x[i]); break;
            default: return -1;
        }
    }
    return 0;
}
'''


def _activations_for_the_host(csrc):
    """`enum { kActNone ... }` and `activate<ACT>` cut out of fused_common.hpp (the rest of that header is MFMA code)"""
    src = open(os.path.join(csrc, "fused_common.hpp")).read()
    a = src.index("enum : int { kActNone")
    b = src.index("template <int ACT>\n__device__ __forceinline__ float activate(float v)", a)
    end = src.index("\n}\n", b) + 3
    return "namespace nfa {\n" + src[a:end] + "\n}  // namespace nfa\n"


def build(out_dir):
    csrc = os.path.join(ROOT, "nflows_amd", "csrc")
    math_src = open(os.path.join(csrc, "rqs_math.hpp")).read()
    assert math_src.count('#include "common.hpp"') == 1
    bwd_src = open(os.path.join(csrc, "rqs_bwd.hip")).read()
    start = bwd_src.index("// adjoints of (x, a, b, c, e, d0, d1) for upstream (gy, gl)")
    stop = bwd_src.index("template <int KT, bool INVERSE>\n__global__ void __launch_bounds__(kBlock) rqs_coupling_backward_kernel")
    backward = "namespace nfa {\n" + bwd_src[start:stop] + "\n}  // namespace nfa\n"
    assert "__global__" not in backward and "__shfl" not in backward
    cpp = os.path.join(out_dir, "rqs_f32_host.cpp")
    so = os.path.join(out_dir, "rqs_f32_host.so")
    with open(cpp, "w") as f:
        f.write(math_src.replace('#include "common.hpp"', SHIM).replace("#pragma once", "", 1) + backward
                + _fused8_for_the_host(csrc) + HARNESS + _activations_for_the_host(csrc) + ACT_HARNESS)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), cpp, "-o", so])
    lib = ctypes.CDLL(so)
    p, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.host_rqs_forward.argtypes = [i32, i32, i64, p, p, p, p, p]
    lib.host_rqs_forward_regs.argtypes = [i32, i32, i64, p, p, p, p, p]
    lib.host_rqs_forward_regs.restype = i32
    lib.host_rqs_backward.argtypes = [i32, i32, i64, p, p, p, p, p, p, p]
    lib.host_rqs_forward_flat8.argtypes = [i32, i64, p, p, p, p, p]
    lib.host_rqs_forward_flatsteps.argtypes = [i32, i32, i64, p, p, p, p, p]
    lib.host_rqs_forward_fused.argtypes = [i32, ctypes.c_float, i64, p, p, p, p, p]
    lib.host_rqs_forward_fused_bins.argtypes = [i32, ctypes.c_float, i64, p, p, p, p, p, p]
    lib.host_rqs_forward_fused_bins.restype = i32
    lib.host_rqs_bins.argtypes = [i32, i32, i64, p, p, p, p]
    lib.host_rqs_bins.restype = i32
    lib.host_activate.argtypes = [i32, i64, p, p]
    lib.host_activate.restype = i32
    for fn in (lib.host_rqs_forward, lib.host_rqs_backward, lib.host_rqs_forward_flat8, lib.host_rqs_forward_flatsteps,
               lib.host_rqs_forward_fused):
        fn.restype = i32
    return lib
