"""The closed-form adjoints of the quadratic and cubic spline backward kernels, as written in
nflows_amd/csrc/splines_lq.hip, run on the host (tests/_hostcore/lq_backward_host.py) against the reference's
autograd (tests/golden/splines_lq_grads.npz) -- the rule of the GPU test: the error against the float64 gradient
is at most 4 x the reference's own fp32 error + 2e-5 * scale.  CPU only; catches an algebra error in the kernel
source before a GPU sees it (the inverse quadratic root's implicit differentiation was developed this way)."""
import ctypes
import os
import shutil

import numpy as np
import pytest

from _hostcore import lq_backward_host


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    return lq_backward_host.build(str(tmp_path_factory.mktemp("lqhost")))


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "splines_lq_grads.npz"))


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def check(got, G, key, worst):
    truth, ref = G[key + "64"], G[key]
    err, ref_err = np.abs(got - truth).max(), np.abs(ref.astype(np.float64) - truth).max()
    limit = 4 * ref_err + 2e-5 * (1 + np.abs(truth).max())
    assert err <= limit, "%s: %.3e > %.3e" % (key, err, limit)
    worst.append(err / limit)


def test_linear_adjoints(lib, G):
    worst = []
    for name in ("lin_k8", "lin_k10", "lin_k4", "lin_k17", "ulin_k8", "ulin_k10", "ulin_k4", "ulin_k17"):
        x = G[name + "/x"].astype(np.float32)
        pdf = np.ascontiguousarray(G[name + "/logits0"])
        n, K = x.size, pdf.shape[1]
        lo, hi = (-3.0, 3.0) if name.startswith("u") else (0.0, 1.0)
        for inverse in (0, 1):
            gx, g0 = np.empty(n, np.float32), np.empty_like(pdf)
            lib.linear(inverse, n, K, lo, hi, P(x), P(pdf), P(G[name + "/wy"]), P(G[name + "/wl"]), P(gx), P(g0))
            pre = name + "/" + ("inv_" if inverse else "")
            check(gx, G, pre + "gx", worst)
            check(g0, G, pre + "glogits0", worst)
    assert len(worst) == 32


def test_quadratic_adjoints(lib, G):
    worst = []
    for name in ("quad_k8", "quad_k10", "quad_k4", "quad_k17", "uquad_k8", "uquad_k10", "uquad_k4", "uquad_k17"):
        x = G[name + "/x"].astype(np.float32)
        w, h = (np.ascontiguousarray(G[name + "/logits%d" % i]) for i in range(2))
        n, K, nh = x.size, w.shape[1], h.shape[1]
        lo, hi = (-3.0, 3.0) if name.startswith("u") else (0.0, 1.0)
        for inverse in (0, 1):
            gx, g0, g1 = np.empty(n, np.float32), np.empty_like(w), np.empty_like(h)
            lib.quadratic(inverse, n, K, nh, lo, hi, P(x), P(w), P(h), P(G[name + "/wy"]), P(G[name + "/wl"]), P(gx), P(g0), P(g1))
            pre = name + "/" + ("inv_" if inverse else "")
            for got, key in ((gx, "gx"), (g0, "glogits0"), (g1, "glogits1")):
                check(got, G, pre + key, worst)
    assert len(worst) == 48


def test_inverse_quadratic_root_is_differentiated_without_cancellation(lib, G):
    """Flat bins (equal heights at the two knots: qa -> 0) make (-1 + qb / r) cancel in the chain through the closed-form
    root; the kernel differentiates the root implicitly.  On the unconstrained fixtures the height-logit gradient of
    the inverse is then at least 10 x closer to float64 than the reference's own fp32 autograd."""
    for name in ("uquad_k8", "uquad_k10", "uquad_k4"):
        x = G[name + "/x"].astype(np.float32)
        w, h = (np.ascontiguousarray(G[name + "/logits%d" % i]) for i in range(2))
        n, K, nh = x.size, w.shape[1], h.shape[1]
        gx, g0, g1 = np.empty(n, np.float32), np.empty_like(w), np.empty_like(h)
        lib.quadratic(1, n, K, nh, -3.0, 3.0, P(x), P(w), P(h), P(G[name + "/wy"]), P(G[name + "/wl"]), P(gx), P(g0), P(g1))
        truth, ref = G[name + "/inv_glogits164"], G[name + "/inv_glogits1"]
        assert np.abs(g1 - truth).max() * 10 <= np.abs(ref - truth).max(), name


def test_cubic_adjoints(lib, G):
    worst = []
    for name in ("cub_k8", "cub_k10", "cub_k4", "ucub_k8", "ucub_k10", "ucub_k4"):
        x = G[name + "/x"].astype(np.float32)
        L = [np.ascontiguousarray(G[name + "/logits%d" % i]) for i in range(4)]
        n, K = x.size, L[0].shape[1]
        lo, hi = (-3.0, 3.0) if name.startswith("u") else (0.0, 1.0)
        for inverse in (0, 1):
            gx, g = np.empty(n, np.float32), [np.empty_like(t) for t in L]
            lib.cubic(inverse, n, K, lo, hi, P(x), P(L[0]), P(L[1]), P(L[2]), P(L[3]), P(G[name + "/wy"]), P(G[name + "/wl"]),
                      P(gx), P(g[0]), P(g[1]), P(g[2]), P(g[3]))
            pre = name + "/" + ("inv_" if inverse else "")
            for got, key in ((gx, "gx"), (g[0], "glogits0"), (g[1], "glogits1"), (g[2], "glogits2"), (g[3], "glogits3")):
                check(got, G, pre + key, worst)
    assert len(worst) == 60
