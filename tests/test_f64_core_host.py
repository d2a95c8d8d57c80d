"""The per-element float64 spline arithmetic of the product (nflows_amd/csrc/rqs_f64_core.hpp -- the functions
the device kernels nfa_rqs_elementwise_f64 / nfa_rqs_elementwise_backward_f64 call per lane) compiled for the
host with g++ and held to the reference's float64 results (tests/golden/rqs_functional.npz: values,
tests/golden/grads.npz: the reference's float64 autograd through a spline coupling layer).  CPU only: this is the
check that the closed-form adjoints are the reference's gradients before the kernel ever runs on a GPU
(tests/test_gpu_grads.py repeats it through the C-ABI on the device).  Nothing in the product loads this build."""
import ctypes
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import parse_kwargs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("f64core") / "rqs_f64_host.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "nflows_amd", "csrc"),
                           os.path.join(ROOT, "tests", "_hostcore", "rqs_f64_host.cpp"), "-o", out])
    lib = ctypes.CDLL(out)
    d, p = ctypes.c_double, ctypes.c_void_p
    lib.host_rqs64.argtypes = [ctypes.c_int, ctypes.c_int64] + [ctypes.c_int] * 4 + [d] * 10 + [p] * 11
    lib.host_rqs64.restype = ctypes.c_int
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def run(lib, backward, x, uw, uh, ud, K, linear, inverse, box, divisor=0.0, gy=None, gl=None, mins=(1e-3, 1e-3, 1e-3),
        beta=1.0):
    x, uw, uh, ud = (np.ascontiguousarray(t, dtype=np.float64) for t in (x, uw, uh, ud))
    n, nd = x.size, ud.shape[-1]
    out0, out1 = np.empty(n), np.empty(n)
    g_uw, g_uh, g_ud = np.empty((n, K)), np.empty((n, K)), np.empty((n, nd))
    if backward:
        gy, gl = np.ascontiguousarray(gy, dtype=np.float64), np.ascontiguousarray(gl, dtype=np.float64)
    tail_logit = math.log(math.exp(1 - mins[2]) - 1)   # rational_quadratic.py:34
    status = lib.host_rqs64(int(backward), n, K, nd, int(linear), int(inverse), *box, *mins, beta, tail_logit, divisor,
                            _ptr(x), _ptr(uw), _ptr(uh), _ptr(ud), _ptr(gy), _ptr(gl), _ptr(out0), _ptr(out1),
                            _ptr(g_uw), _ptr(g_uh), _ptr(g_ud))
    return (out0, g_uw, g_uh, g_ud, status) if backward else (out0, out1, status)


def test_forward_values_match_the_reference_in_float64(core, golden_dir):
    G = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    seen = 0
    for name, inv, kw in G["meta"]:
        kw = parse_kwargs(kw)
        x, uw, uh, ud = (G[name + "/" + k].astype(np.float64) for k in ("x", "uw", "uh", "ud"))
        K = uw.shape[-1]
        linear = kw.get("tails") == "linear"
        if linear:
            tb = kw["tail_bound"]
            box = (-tb, tb, -tb, tb)
        else:
            box = (kw.get("left", 0.0), kw.get("right", 1.0), kw.get("bottom", 0.0), kw.get("top", 1.0))
        mins = (kw.get("min_bin_width", 1e-3), kw.get("min_bin_height", 1e-3), kw.get("min_derivative", 1e-3))
        beta = 1.0
        if kw.get("enable_identity_init"):
            beta = math.log(2) / (1 - mins[2])
        y, lad, st = run(core, False, x.reshape(-1), uw.reshape(-1, K), uh.reshape(-1, K),
                         ud.reshape(-1, ud.shape[-1]), K, linear, bool(int(inv)), box, mins=mins, beta=beta)
        ry, rl = G[name + "/y64"].reshape(-1), G[name + "/lad64"].reshape(-1)
        assert st == 0, name
        assert np.array_equal(np.isnan(y), np.isnan(ry)), name
        fin = np.isfinite(ry)
        assert np.abs(y[fin] - ry[fin]).max() <= 1e-10, name
        assert np.abs(lad[fin] - rl[fin]).max() <= 1e-10, name
        seen += 1
    assert seen >= 20


def test_gradients_match_the_reference_float64_autograd(core, golden_dir):
    """loss = <y, Wy> + <logabsdet, Wl> through a spline coupling layer whose conditioner output is a table
    (grads.npz): d loss / d inputs on the transformed columns and d loss / d table are the functional's
    gradients with gy = Wy[b, column], gl = Wl[b].  Linear tails with the conditioner's 1/sqrt(hidden) divisor
    (K = 8, 5) and the constrained spline (K = 4, K + 1 derivative logits), both directions."""
    G = np.load(os.path.join(golden_dir, "grads.npz"))
    done = 0
    for name, kind, cfg in G["meta"]:
        if kind != "rq":
            continue
        cfg = parse_kwargs(cfg)
        K, linear, tb, H = cfg["K"], cfg["tails"] == "linear", cfg["tail_bound"], cfg["hidden"]
        nd = K - 1 if linear else K + 1
        P = 2 * K + nd
        x, params = G[name + "/x"].astype(np.float64), G[name + "/params"].astype(np.float64)
        Wy, Wl, tidx = G[name + "/Wy"].astype(np.float64), G[name + "/Wl"].astype(np.float64), G[name + "/transform_idx"]
        B, dt = x.shape[0], len(tidx)
        pr = params.reshape(B * dt, P)
        box = (-tb, tb, -tb, tb) if linear else (0.0, 1.0, 0.0, 1.0)
        for inverse in (False, True):
            gx, g_uw, g_uh, g_ud, _ = run(core, True, x[:, tidx].reshape(-1), pr[:, :K], pr[:, K:2 * K], pr[:, 2 * K:],
                                          K, linear, inverse, box, divisor=math.sqrt(H) if H else 0.0,
                                          gy=Wy[:, tidx].reshape(-1), gl=np.repeat(Wl, dt))
            tag = name + ("/inv" if inverse else "/fwd")
            ref_gx = G[tag + "_gx64"][:, tidx].reshape(-1)
            ref_gp = G[tag + "_gp64"].reshape(B * dt, P)
            got_gp = np.concatenate([g_uw, g_uh, g_ud], axis=1)
            assert np.abs(gx - ref_gx).max() <= 1e-11 * (1 + np.abs(ref_gx).max()), tag
            assert np.abs(got_gp - ref_gp).max() <= 1e-11 * (1 + np.abs(ref_gp).max()), tag
            done += 1
    assert done == 6


def test_gradient_of_the_divisor_and_of_the_tails(core):
    """Logits divided by a divisor inside the functional = pre-divided logits with gradients scaled by 1/divisor;
    elements in the linear tails (and NaN) pass the upstream gradient through and give zero logit gradients."""
    rng = np.random.default_rng(5)
    n, K = 400, 8
    x = rng.normal(size=n) * 2.0
    x[:4] = (3.5, -3.01, np.nan, 2.9)
    uw, uh, ud = rng.normal(size=(n, K)) * 2, rng.normal(size=(n, K)) * 2, rng.normal(size=(n, K - 1))
    gy, gl = rng.normal(size=n), rng.normal(size=n)
    box = (-3.0, 3.0, -3.0, 3.0)
    for inverse in (False, True):
        a = run(core, True, x, uw, uh, ud, K, True, inverse, box, divisor=4.0, gy=gy, gl=gl)
        b = run(core, True, x, uw / 4.0, uh / 4.0, ud, K, True, inverse, box, divisor=0.0, gy=gy, gl=gl)
        assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-14, equal_nan=True)
        assert np.allclose(a[1], b[1] / 4.0, rtol=1e-12, atol=1e-14)
        assert np.allclose(a[2], b[2] / 4.0, rtol=1e-12, atol=1e-14)
        assert np.allclose(a[3], b[3], rtol=1e-12, atol=1e-14)
        for i in (0, 1, 2):   # tails and NaN
            assert a[0][i] == gy[i] and not a[1][i].any() and not a[2][i].any() and not a[3][i].any()
        assert a[1][3].any() and a[2][3].any()   # just inside the box: the last bin's logits get gradients


def test_gradients_against_central_differences_of_the_forward_core(core):
    """Independent of the fixtures: the adjoints against central differences of the same core's forward values
    (float64, step 1e-6), 10 bins, identity-initialised softplus (beta != 1), a non-default box."""
    rng = np.random.default_rng(11)
    n, K = 60, 10
    box = (-1.0, 5.0, 0.5, 4.0)   # (the reference checks inverse inputs against [left, right] too, :81-82)
    beta = math.log(2) / (1 - 1e-2)
    mins = (1e-2, 2e-2, 1e-2)
    uw, uh, ud = rng.normal(size=(n, K)), rng.normal(size=(n, K)), rng.normal(size=(n, K + 1))
    gy, gl = rng.normal(size=n), rng.normal(size=n)
    for inverse in (False, True):
        lo, hi = (box[2], box[3]) if inverse else (box[0], box[1])
        x = lo + (hi - lo) * (0.02 + 0.96 * rng.random(n))

        def loss(x_, uw_, uh_, ud_):
            y, lad, st = run(core, False, x_, uw_, uh_, ud_, K, False, inverse, box, mins=mins, beta=beta)
            assert st == 0
            return gy * y + gl * lad   # per element

        gx, g_uw, g_uh, g_ud, _ = run(core, True, x, uw, uh, ud, K, False, inverse, box, gy=gy, gl=gl, mins=mins, beta=beta)
        h = 1e-6
        num = (loss(x + h, uw, uh, ud) - loss(x - h, uw, uh, ud)) / (2 * h)
        assert np.abs(num - gx).max() <= 2e-6 * (1 + np.abs(gx).max())
        for arr, grad in ((uw, g_uw), (uh, g_uh), (ud, g_ud)):
            for j in range(arr.shape[1]):
                up, dn = arr.copy(), arr.copy()
                up[:, j] += h
                dn[:, j] -= h
                args_up = [uw, uh, ud]
                args_dn = [uw, uh, ud]
                idx = 0 if arr is uw else 1 if arr is uh else 2
                args_up[idx], args_dn[idx] = up, dn
                num = (loss(x, *args_up) - loss(x, *args_dn)) / (2 * h)
                assert np.abs(num - grad[:, j]).max() <= 2e-6 * (1 + np.abs(grad).max()), (inverse, idx, j)
