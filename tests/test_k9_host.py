"""The sibling splines' arithmetic of the PRODUCT on the CPU (nflows_amd/csrc/splines_lq.hip compiled for the host on the
real helpers of rqs_math.hpp: tests/_hostcore/splines_lq_host.py):
  * forward: `linear_eval` / `quadratic_eval` / `cubic_eval`, run-time-K and K = 8 / 10 instances, against the reference's
    vectors (splines_lq.npz, splines_cubic.npz) with the rule the oracle and the GPU kernels are held to;
  * backward: the bodies of the three backward kernels against the reference's autograd (splines_lq_grads.npz) -- the
    rule of the GPU test: error against the float64 gradient <= 4 x the reference's own fp32 error + 2e-5 x scale.
CPU only; catches an algebra error in the kernel source before a GPU sees it (the inverse quadratic root's implicit
differentiation was developed this way)."""
import ctypes
import os
import shutil

import numpy as np
import pytest

from _hostcore import splines_lq_host
from helpers import LAD_TOL, OUT_TOL, assert_sibling_spline_parity, parse_kwargs

KIND = {"linear": 0, "quadratic": 1, "cubic": 2}


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    return splines_lq_host.build(str(tmp_path_factory.mktemp("lqhost")))


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "splines_lq_grads.npz"))


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def spec_of(K, kw):
    from nflows_amd import ops
    kw = dict(kw)
    tails = kw.pop("tails", None)
    return ops.make_rqs_spec(K, tails if tails == "linear" else None, **kw)


def rows(t):
    return np.ascontiguousarray(t.reshape(-1, t.shape[-1]), dtype=np.float32)


def forward_case(lib, g, name, kind, logits, kw):
    x = g[name + "/x"]
    xs = np.ascontiguousarray(x.reshape(-1), dtype=np.float32)
    K = logits[0].shape[-1]
    spec = spec_of(K, kw)
    L = [rows(t) for t in logits[:2]] + [np.ascontiguousarray(t.reshape(-1), dtype=np.float32) for t in logits[2:]]
    L += [None] * (4 - len(L))
    nh = logits[1].shape[-1] if kind == "quadratic" else 0
    count = 0
    for inverse in (False, True):
        pre = name + ("/inv_" if inverse else "/")
        for kt in [0] + ([K] if K in (8, 10) else []):
            y, lad = np.empty_like(xs), np.empty_like(xs)
            status = lib.lq_forward(KIND[kind], kt, int(inverse), xs.size, ctypes.byref(spec), nh, P(xs), P(L[0]), P(L[1]),
                                    P(L[2]), P(L[3]), P(y), P(lad))
            assert status == 0, (name, inverse, kt)
            what = "%s%s [instance %d]" % (name, " inverse" if inverse else "", kt)
            assert_sibling_spline_parity(y.reshape(x.shape), g[pre + "y"], g[pre + "y64"], OUT_TOL, 5e-5, what + " y")
            assert_sibling_spline_parity(lad.reshape(x.shape), g[pre + "lad"], g[pre + "lad64"], LAD_TOL, 1e-3, what + " lad")
            if kw.get("tails") == "linear":
                tb = np.float32(kw["tail_bound"])
                outside = ~((xs >= -tb) & (xs <= tb))
                assert np.array_equal(y[outside].view(np.uint32), xs[outside].view(np.uint32)), what
                assert np.all(lad[outside] == 0), what
            count += 1
    return count


def test_linear_and_quadratic_forward_values(lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "splines_lq.npz"))
    count = 0
    for name, kind, kw in g["meta"]:
        logits = [g["%s/logits%d" % (name, i)] for i in range(1 if kind == "linear" else 2)]
        count += forward_case(lib, g, str(name), str(kind), logits, parse_kwargs(kw))
    assert count >= 28


def test_cubic_forward_values(lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "splines_cubic.npz"))
    count = 0
    for name, kind, kw in g["meta"]:
        logits = [g["%s/logits%d" % (name, i)] for i in range(4)]
        count += forward_case(lib, g, str(name), "cubic", logits, parse_kwargs(kw))
    assert count >= 4


def backward_case(lib, G, name, kind, worst):
    x = np.ascontiguousarray(G[name + "/x"].reshape(-1), dtype=np.float32)
    n_logits = {"linear": 1, "quadratic": 2, "cubic": 4}[kind]
    logits = [G["%s/logits%d" % (name, i)] for i in range(n_logits)]
    K = logits[0].shape[-1]
    spec = spec_of(K, dict((m[0], parse_kwargs(m[2])) for m in G["meta"])[name])
    L = [rows(t) for t in logits[:2]] + [np.ascontiguousarray(t.reshape(-1), dtype=np.float32) for t in logits[2:]]
    L += [None] * (4 - len(L))
    nh = logits[1].shape[-1] if kind == "quadratic" else 0
    gy = np.ascontiguousarray(G[name + "/wy"].reshape(-1), dtype=np.float32)
    gl = np.ascontiguousarray(G[name + "/wl"].reshape(-1), dtype=np.float32)
    for inverse in (0, 1):
        gx = np.empty_like(x)
        g = [np.empty_like(t) if t is not None else None for t in L]
        status = lib.lq_backward(KIND[kind], inverse, x.size, ctypes.byref(spec), nh, P(x), P(L[0]), P(L[1]), P(L[2]), P(L[3]),
                                 P(gy), P(gl), P(gx), P(g[0]), P(g[1]), P(g[2]), P(g[3]))
        assert status == 0
        pre = name + "/" + ("inv_" if inverse else "")
        for got, key in [(gx, "gx")] + [(g[i], "glogits%d" % i) for i in range(n_logits)]:
            truth, ref = G[pre + key + "64"], G[pre + key]
            got = got.reshape(truth.shape)
            err, ref_err = np.abs(got - truth).max(), np.abs(ref.astype(np.float64) - truth).max()
            limit = 4 * ref_err + 2e-5 * (1 + np.abs(truth).max())
            assert err <= limit, "%s%s: %.3e > %.3e" % (pre, key, err, limit)
            worst.append(err / limit)


def test_backward_adjoints(lib, G):
    worst = []
    for name, kind, _ in G["meta"]:
        backward_case(lib, G, str(name), str(kind), worst)
    assert len(worst) == 2 * (8 * 2 + 8 * 3 + 6 * 5 + 1 * 2 + 1 * 3)   # every case of the fixture, both directions


def test_inverse_quadratic_root_is_differentiated_without_cancellation(lib, G):
    """Flat bins (equal heights at the two knots: qa -> 0) make (-1 + qb / r) cancel in the chain through the closed-form
    root; the kernel differentiates the root implicitly.  On the unconstrained fixtures the height-logit gradient of
    the inverse is then at least 10 x closer to float64 than the reference's own fp32 autograd."""
    meta = dict((m[0], parse_kwargs(m[2])) for m in G["meta"])
    for name in ("uquad_k8", "uquad_k10", "uquad_k4"):
        x = np.ascontiguousarray(G[name + "/x"], dtype=np.float32)
        w, h = rows(G[name + "/logits0"]), rows(G[name + "/logits1"])
        spec = spec_of(w.shape[1], meta[name])
        gx, g0, g1 = np.empty_like(x), np.empty_like(w), np.empty_like(h)
        lib.lq_backward(1, 1, x.size, ctypes.byref(spec), h.shape[1], P(x), P(w), P(h), None, None, P(G[name + "/wy"]),
                        P(G[name + "/wl"]), P(gx), P(g0), P(g1), None, None)
        truth, ref = G[name + "/inv_glogits164"], G[name + "/inv_glogits1"]
        assert np.abs(g1 - truth).max() * 10 <= np.abs(ref - truth).max(), name
