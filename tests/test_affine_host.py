"""The per-element arithmetic of the affine kernels -- nflows_amd/csrc/affine_math.hpp: `scale_of`, `softplus1`,
`affine_element`, the functions K2 (affine_coupling_kernel), K2b (affine_ar_kernel) and K11 (affine_mlp_kernel) call
per element -- compiled for the HOST from the product's source at test time and held to the reference's affine /
additive coupling vectors (tests/golden/coupling.npz: coupling.py:212-269 run by the real reference in fp32 and fp64).
The host build replaces nothing but the device's libm (expf / logf / log1pf: <= 1 ulp); CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import LAD_TOL, OUT_TOL, assert_fp32_parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#define __device__
#define __forceinline__ inline
#include "nflows_amd.h"
#include "affine_math.hpp"
#include <stdint.h>
using namespace nfa;
// params [B, 2 d_t] = [shift | scale logit] halves (coupling.py:235-236); additive: [B, d_t] shifts
extern "C" void host_affine_layer(int64_t batch, int D, int dt, const int64_t* tidx, const float* x, const float* params,
                                  int activation, int inverse, float* y, float* lad) {
    const int pc = activation == NFA_SCALE_ADDITIVE ? dt : 2 * dt;
    for (int64_t b = 0; b < batch; ++b) {
        for (int c = 0; c < D; ++c) y[b * D + c] = x[b * D + c];
        float acc = 0.0f;
        for (int j = 0; j < dt; ++j) {
            const float xin = x[b * D + tidx[j]], shift = params[b * pc + j];
            float out, l;
            if (activation == NFA_SCALE_ADDITIVE) {
                out = inverse ? xin - shift : xin + shift;
                l = 0.0f;
            } else {
                const float sc = scale_of(params[b * pc + dt + j], activation);
                if (inverse) affine_element<true>(xin, shift, sc, out, l);
                else affine_element<false>(xin, shift, sc, out, l);
            }
            y[b * D + tidx[j]] = out;
            acc += l;
        }
        lad[b] = acc;
    }
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("affine_host")
    cpp, so = str(d / "affine_host.cpp"), str(d / "affine_host.so")
    open(cpp, "w").write(SRC)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "nflows_amd", "csrc"), cpp, "-o", so])
    lib = ctypes.CDLL(so)
    p = ctypes.c_void_p
    lib.host_affine_layer.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, p, p, p, ctypes.c_int, ctypes.c_int, p, p]
    lib.host_affine_layer.restype = None
    return lib


def test_affine_element_arithmetic_against_reference_vectors(lib, golden_dir):
    from nflows_amd import _native as N
    g = np.load(os.path.join(golden_dir, "coupling.npz"))
    codes = {"affine_default": N.SCALE_DEFAULT, "affine_general": N.SCALE_GENERAL, "affine_additive": N.SCALE_ADDITIVE}
    seen = 0
    for name, kind, cfg in g["meta"]:
        if not str(kind).startswith("affine"):
            continue
        x = np.ascontiguousarray(g[name + "/x"], dtype=np.float32)
        params = np.ascontiguousarray(g[name + "/params"], dtype=np.float32)
        tidx = np.ascontiguousarray(g[name + "/transform_idx"], dtype=np.int64)
        B, D = x.shape
        for direction, inverse in (("fwd", 0), ("inv", 1)):
            y, lad = np.empty_like(x), np.empty(B, dtype=np.float32)
            lib.host_affine_layer(B, D, tidx.size, tidx.ctypes.data, x.ctypes.data, params.ctypes.data, codes[str(kind)],
                                  inverse, y.ctypes.data, lad.ctypes.data)
            what = "%s %s" % (name, direction)
            assert_fp32_parity(y, g["%s/%s_y" % (name, direction)], g["%s/%s_y64" % (name, direction)], OUT_TOL, what + " y")
            assert_fp32_parity(lad, g["%s/%s_lad" % (name, direction)], g["%s/%s_lad64" % (name, direction)], LAD_TOL, what + " lad")
            # untouched columns are copies; the additive layer's log-determinant is exactly zero (coupling.py:263-269)
            ident = np.setdiff1d(np.arange(D), tidx)
            assert np.array_equal(y[:, ident].view(np.uint32), x[:, ident].view(np.uint32))
            if str(kind) == "affine_additive":
                assert np.all(lad == 0)
            seen += 1
    assert seen == 12
