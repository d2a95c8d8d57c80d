"""Parity of the HIP kernels (through the C ABI) against the CPU oracle and the committed golden
vectors from the real reference.  Needs an MI355X: run with `-m gpu`.

Tolerances: see tests/helpers.py (bit-exact for index / pass-through work; the stated fp32
model for everything else).
"""
import os

import numpy as np
import pytest
import torch

from helpers import LAD_TOL, OUT_TOL, assert_fp32_parity, assert_sibling_spline_parity, conditioning, parse_kwargs
from oracle import capi

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    import nflows_amd
    from nflows_amd import ops as o
    assert os.path.exists(nflows_amd.native_library_path())
    return o


def _spec_pair(ops, K, **kw):
    return ops.make_rqs_spec(K, **kw), capi.make_spec(K, **kw)


# ------------------------------------------------------------------------------- K5 vs golden
def test_rqs_elementwise_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    for name, inv, kw in g["meta"]:
        kw = parse_kwargs(kw)
        inv = bool(int(inv))
        x, uw, uh, ud = (g[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        K = uw.shape[-1]
        spec, _ = _spec_pair(ops, K, **kw)
        y, lad = ops.rqs_elementwise(dev(x), dev(uw), dev(uh), dev(ud), spec, inverse=inv)
        ops.check_status()
        y, lad = host(y), host(lad)
        _, ospec = _spec_pair(ops, K, **kw)
        cy, cl = conditioning(lambda *a: capi.rqs_elementwise(*a, ospec, inverse=inv)[:2], (x, uw, uh, ud), (0, 1, 2, 3))
        assert_fp32_parity(y, g[name + "/y"], g[name + "/y64"], OUT_TOL, name + " y", cond=cy)
        assert_fp32_parity(lad, g[name + "/lad"], g[name + "/lad64"], LAD_TOL, name + " lad", cond=cl)
        if kw.get("tails") == "linear":
            tb = np.float32(kw["tail_bound"])
            outside = ~((x >= -tb) & (x <= tb))
            assert np.array_equal(y[outside].view(np.uint32), x[outside].view(np.uint32)), name
            assert np.all(lad[outside] == 0), name


# ------------------------------------------------------------------------------- K5 vs oracle
@pytest.mark.parametrize("K", [1, 2, 3, 5, 8, 10, 16, 33])
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_elementwise_oracle(ops, K, inverse):
    rng = np.random.RandomState(100 + K)
    n = 5000 + K  # ragged: not a multiple of the 256-element tile
    tb = 3.0
    x = (tb * 0.9 * rng.randn(n)).astype(np.float32)
    uw = rng.randn(n, K).astype(np.float32)
    uh = rng.randn(n, K).astype(np.float32)
    ud = rng.randn(n, K - 1).astype(np.float32)
    spec, ospec = _spec_pair(ops, K, tails="linear", tail_bound=tb)
    y, lad = ops.rqs_elementwise(dev(x), dev(uw), dev(uh), dev(ud), spec, inverse=inverse)
    oy, ol, st = capi.rqs_elementwise(x, uw, uh, ud, ospec, inverse=inverse)
    ty, tl, _ = capi.rqs_elementwise(*(a.astype(np.float64) for a in (x, uw, uh, ud)), ospec, inverse=inverse)
    assert st == 0
    cy, cl = conditioning(lambda *a: capi.rqs_elementwise(*a, ospec, inverse=inverse)[:2], (x, uw, uh, ud), (0, 1, 2, 3))
    assert_fp32_parity(host(y), oy, ty, OUT_TOL, "K5 y K=%d inverse=%s" % (K, inverse), cond=cy)
    assert_fp32_parity(host(lad), ol, tl, LAD_TOL, "K5 lad K=%d inverse=%s" % (K, inverse), cond=cl)


def test_rqs_elementwise_strided_views(ops):
    """Logits given as views into one packed [N, P] buffer (the autoregressive layout,
    autoregressive.py:453-460) and as three separate tensors give the same result."""
    rng = np.random.RandomState(7)
    K, n = 8, 3001
    P = 3 * K - 1
    packed = rng.randn(n, P).astype(np.float32)
    x = (2.5 * rng.randn(n)).astype(np.float32)
    spec, _ = _spec_pair(ops, K, tails="linear", tail_bound=3.0)
    p = dev(packed)
    y1, l1 = ops.rqs_elementwise(dev(x), p[:, :K], p[:, K:2 * K], p[:, 2 * K:], spec)
    y2, l2 = ops.rqs_elementwise(dev(x), p[:, :K].contiguous(), p[:, K:2 * K].contiguous(),
                                 p[:, 2 * K:].contiguous(), spec)
    assert torch.equal(y1, y2) and torch.equal(l1, l2)
    # misaligned packed base (storage offset of 1 float)
    buf = torch.empty(n * P + 1, device=DEV)
    buf[1:] = p.view(-1)
    q = buf[1:].view(n, P)
    y3, l3 = ops.rqs_elementwise(dev(x), q[:, :K], q[:, K:2 * K], q[:, 2 * K:], spec)
    assert torch.equal(y1, y3) and torch.equal(l1, l3)


def test_rqs_elementwise_shapes_and_empty(ops):
    K = 4
    spec, _ = _spec_pair(ops, K, tails="linear", tail_bound=1.0)
    x = torch.randn(2, 3, 4, device=DEV)
    uw, uh, ud = torch.randn(2, 3, 4, K, device=DEV), torch.randn(2, 3, 4, K, device=DEV), torch.randn(2, 3, 4, K - 1, device=DEV)
    y, lad = ops.rqs_elementwise(x, uw, uh, ud, spec)
    assert y.shape == x.shape and lad.shape == x.shape
    y0, l0 = ops.rqs_elementwise(x[:0], uw[:0], uh[:0], ud[:0], spec)
    assert y0.numel() == 0 and l0.numel() == 0


def test_rqs_errors(ops):
    from nflows_amd import InputOutsideDomain
    from nflows_amd.transforms import splines
    x = torch.tensor([0.5, 1.5], device=DEV)
    z = torch.zeros(2, 4, device=DEV)
    with pytest.raises(InputOutsideDomain):
        splines.rational_quadratic_spline(x, z, z, torch.zeros(2, 5, device=DEV))
    with pytest.raises(ValueError, match="Minimal bin width too large"):
        splines.rational_quadratic_spline(x[:1], z[:1], z[:1], torch.zeros(1, 5, device=DEV), min_bin_width=0.3)
    with pytest.raises(ValueError, match="Minimal bin height too large"):
        splines.rational_quadratic_spline(x[:1], z[:1], z[:1], torch.zeros(1, 5, device=DEV), min_bin_height=0.3)
    with pytest.raises(RuntimeError, match="cubic tails are not implemented"):
        splines.unconstrained_rational_quadratic_spline(x[:1], z[:1], z[:1], torch.zeros(1, 3, device=DEV), tails="cubic")
    with pytest.raises(NotImplementedError):
        splines.unconstrained_rational_quadratic_spline(x[:1].cpu(), z[:1].cpu(), z[:1].cpu(), torch.zeros(1, 3))
    ops.check_status()  # clean


def test_identity_init(ops):
    """reference tests/transforms/splines/rational_quadratic_test.py:33-62 and :116-146, same
    inputs (the unconstrained one passes K+1 derivative logits, which the reference accepts)."""
    from nflows_amd.transforms import splines
    shape, K = [2, 3, 4], 10
    z = torch.zeros(*shape, K, device=DEV)
    zd = torch.zeros(*shape, K + 1, device=DEV)
    x = torch.rand(*shape, device=DEV)
    for inv in (False, True):
        y, lad = splines.rational_quadratic_spline(x, z, z, zd, inverse=inv, enable_identity_init=True)
        assert (y - x).abs().max().item() < 1e-6 and lad.abs().max().item() < 1e-6
    tail_bound = 1.0
    xo = torch.sign(torch.randn(*shape, device=DEV)) * (tail_bound + torch.rand(*shape, device=DEV))
    y, lad = splines.unconstrained_rational_quadratic_spline(xo, z, z, zd, inverse=False,
                                                             enable_identity_init=True)
    assert torch.equal(y, xo) and torch.equal(lad, torch.zeros_like(lad))
    y, lad = splines.unconstrained_rational_quadratic_spline(x, z, z, zd, inverse=True,
                                                             enable_identity_init=True)
    assert (y - x).abs().max().item() < 1e-6 and lad.abs().max().item() < 1e-6


# ------------------------------------------------------------------------------- K1 / K2 golden
def test_coupling_layers_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "coupling.npz"))
    from nflows_amd import _native as N
    for name, kind, cfg in g["meta"]:
        cfg = parse_kwargs(cfg)
        x, params, tidx = g[name + "/x"], g[name + "/params"], g[name + "/transform_idx"]
        for direction, inv in (("fwd", False), ("inv", True)):
            ry, rl = g["%s/%s_y" % (name, direction)], g["%s/%s_lad" % (name, direction)]
            ry64, rl64 = g["%s/%s_y64" % (name, direction)], g["%s/%s_lad64" % (name, direction)]
            if kind == "rq":
                H = cfg["hidden"]
                spec = ops.make_rqs_spec(cfg["K"], cfg["tails"], tail_bound=cfg["tail_bound"],
                                         wh_divisor=float(np.sqrt(H)) if H else 0.0)
                y, lad = ops.rqs_coupling(dev(x), dev(params), dev(tidx), spec, inverse=inv)
            else:
                act = {"affine_default": N.SCALE_DEFAULT, "affine_general": N.SCALE_GENERAL,
                       "affine_additive": N.SCALE_ADDITIVE}[kind]
                y, lad = ops.affine_coupling(dev(x), dev(params), dev(tidx), act, inverse=inv)
            ops.check_status()
            y, lad = host(y), host(lad)
            if kind == "rq":
                ospec = capi.make_spec(cfg["K"], tails=cfg["tails"], tail_bound=cfg["tail_bound"],
                                       wh_divisor=float(np.sqrt(H)) if H else 0.0)
                f64 = lambda xx, pp: capi.rqs_coupling(xx, pp, tidx, ospec, inverse=inv)[:2]      # noqa: E731
            else:
                oact = {"affine_default": capi.AFFINE_DEFAULT, "affine_general": capi.AFFINE_GENERAL,
                        "affine_additive": capi.AFFINE_ADDITIVE}[kind]
                f64 = lambda xx, pp: capi.affine_coupling(xx, pp, tidx, oact, inverse=inv)[:2]    # noqa: E731
            cy, cl = conditioning(f64, (x.astype(np.float64), params.astype(np.float64)), (0, 1))
            assert_fp32_parity(y, ry, ry64, OUT_TOL, name + direction + " y", cond=cy)
            assert_fp32_parity(lad, rl, rl64, LAD_TOL, name + direction + " lad", cond=cl)
            ident = np.setdiff1d(np.arange(x.shape[1]), tidx)
            assert np.array_equal(y[:, ident], x[:, ident]), name  # bit-exact pass-through
            if kind == "affine_additive":
                assert np.all(lad == 0), name


# ------------------------------------------------------------------------------- K1 vs oracle
@pytest.mark.parametrize("B,D,K,tails", [
    (1, 64, 8, "linear"), (7, 64, 8, "linear"), (4099, 64, 8, "linear"), (513, 6, 8, "linear"),
    (300, 64, 10, "linear"), (257, 9, 5, "linear"), (129, 33, 3, None), (64, 2, 8, "linear"),
    (50, 700, 8, "linear"), (3, 1500, 4, "linear"), (4, 3000, 8, "linear"), (5, 784, 8, "linear"),
    (2, 5000, 10, None),
    # shapes that take the software-pipelined kernel with other tile geometries
    (1000, 48, 8, "linear"), (777, 32, 8, "linear"), (300, 128, 8, "linear"), (2049, 16, 8, "linear"),
    (65, 8, 8, "linear"), (999, 64, 8, None),
])
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_coupling_oracle(ops, B, D, K, tails, inverse):
    rng = np.random.RandomState(B + D + K)
    mask = rng.rand(D) < 0.5
    if D in (8, 16, 32, 48, 64, 128):
        mask = np.arange(D) % 2 == 0  # half/half masks: aligned layouts (pipelined kernel)
    mask[0] = True
    tidx = np.nonzero(mask)[0].astype(np.int64)
    dt = tidx.size
    P = 3 * K - 1 if tails == "linear" else 3 * K + 1
    tb = 3.0
    if tails == "linear":
        x = (1.4 * rng.randn(B, D)).astype(np.float32)
    else:
        x = rng.rand(B, D).astype(np.float32)
    params = (1.5 * rng.randn(B, dt * P)).astype(np.float32)
    kw = dict(tails=tails, tail_bound=tb, wh_divisor=float(np.sqrt(32)))
    spec, ospec = _spec_pair(ops, K, **kw)
    y, lad = ops.rqs_coupling(dev(x), dev(params), dev(tidx), spec, inverse=inverse)
    ops.check_status()
    oy, ol, st = capi.rqs_coupling(x, params, tidx, ospec, inverse=inverse)
    ty, tl, _ = capi.rqs_coupling(x.astype(np.float64), params.astype(np.float64), tidx, ospec, inverse=inverse)
    assert st == 0
    y, lad = host(y), host(lad)
    cy, cl = conditioning(lambda xx, pp: capi.rqs_coupling(xx, pp, tidx, ospec, inverse=inverse)[:2],
                          (x.astype(np.float64), params.astype(np.float64)), (0, 1))
    what = "K1 B=%d D=%d K=%d tails=%s inverse=%s " % (B, D, K, tails, inverse)
    assert_fp32_parity(y, oy, ty, OUT_TOL, what + "y", cond=cy)
    assert_fp32_parity(lad, ol, tl, LAD_TOL, what + "lad", cond=cl)
    ident = np.nonzero(~mask)[0]
    assert np.array_equal(y[:, ident], x[:, ident])


@pytest.mark.parametrize("inverse", [False, True])
def test_coupling_fused_permutation_bit_exact(ops, inverse):
    """in_perm == Permutation.forward before the layer; out_scatter == Permutation.inverse after
    it: bit-identical to running the K4 kernel separately."""
    rng = np.random.RandomState(3)
    B, D, K = 1000, 64, 8
    tidx = np.arange(0, D, 2).astype(np.int64)
    P = 3 * K - 1
    x = dev((1.5 * rng.randn(B, D)).astype(np.float32))
    params = dev(rng.randn(B, tidx.size * P).astype(np.float32))
    perm = dev(rng.permutation(D).astype(np.int64))
    spec, _ = _spec_pair(ops, K, tails="linear", tail_bound=3.0, wh_divisor=float(np.sqrt(128)))
    t = dev(tidx)
    if not inverse:
        y_seq, l_seq = ops.rqs_coupling(ops.permute_cols(x, perm), params, t, spec)
        y_f, l_f = ops.rqs_coupling(x, params, t, spec, in_perm=perm)
    else:
        y0, l_seq = ops.rqs_coupling(x, params, t, spec, inverse=True)
        y_seq = ops.permute_cols(y0, torch.argsort(perm))
        y_f, l_f = ops.rqs_coupling(x, params, t, spec, inverse=True, out_scatter=perm)
    assert torch.equal(y_seq, y_f) and torch.equal(l_seq, l_f)


def test_affine_oracle_and_given_scale(ops):
    from nflows_amd import _native as N
    rng = np.random.RandomState(5)
    for B, D in [(1, 32), (1000, 32), (333, 7), (20, 300)]:
        mask = rng.rand(D) < 0.5
        mask[0] = True
        tidx = np.nonzero(mask)[0].astype(np.int64)
        dt = tidx.size
        x = rng.randn(B, D).astype(np.float32)
        for act, cols in ((N.SCALE_DEFAULT, 2 * dt), (N.SCALE_GENERAL, 2 * dt), (N.SCALE_ADDITIVE, dt)):
            params = (2 * rng.randn(B, cols)).astype(np.float32)
            for inv in (False, True):
                y, lad = ops.affine_coupling(dev(x), dev(params), dev(tidx), act, inverse=inv)
                oy, ol, _ = capi.affine_coupling(x, params, tidx, act, inverse=inv)
                ty, tl, _ = capi.affine_coupling(x.astype(np.float64), params.astype(np.float64), tidx, act, inverse=inv)
                cy, cl = conditioning(lambda xx, pp: capi.affine_coupling(xx, pp, tidx, act, inverse=inv)[:2],
                                      (x.astype(np.float64), params.astype(np.float64)), (0, 1))
                assert_fp32_parity(host(y), oy, ty, OUT_TOL, "K2 y act=%d inverse=%s" % (act, inv), cond=cy)
                assert_fp32_parity(host(lad), ol, tl, LAD_TOL, "K2 lad act=%d inverse=%s" % (act, inv), cond=cl)
        # arbitrary activation evaluated by the caller
        params = rng.randn(B, 2 * dt).astype(np.float32)
        scale = np.exp(0.3 * params[:, dt:]).astype(np.float32)
        y, lad = ops.affine_coupling(dev(x), dev(params), dev(tidx), N.SCALE_GIVEN, scale=dev(scale))
        oy, ol, _ = capi.affine_coupling(x, params, tidx, capi.AFFINE_GIVEN_SCALE, scale=scale)
        assert np.abs(host(y) - oy).max() <= 1e-6 and np.abs(host(lad) - ol).max() <= 1e-4


def test_affine_autoregressive_oracle(ops):
    rng = np.random.RandomState(9)
    for B, D in [(128, 2), (77, 100), (5, 784)]:
        x = rng.randn(B, D).astype(np.float32)
        params = rng.randn(B, D * 2).astype(np.float32)
        for inv in (False, True):
            y, lad = ops.affine_autoregressive(dev(x), dev(params), inverse=inv)
            oy, ol = capi.affine_autoregressive(x, params, inverse=inv)
            assert np.abs(host(y) - oy).max() <= 2e-6 * (1 + np.abs(oy).max())
            assert np.abs(host(lad) - ol).max() <= 1e-5 * max(1, D // 8)


# ------------------------------------------------------------------------------- K4 / K3
def test_permute_cols_bit_exact(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    out = ops.permute_cols(dev(g["perm_x"]), dev(g["perm"]))
    assert np.array_equal(host(out), g["perm_fwd"])
    out = ops.permute_cols(dev(g["perm_x"]), torch.argsort(dev(g["perm"])))
    assert np.array_equal(host(out), g["perm_inv"])
    rng = np.random.RandomState(1)
    for B, D in [(1, 1), (3, 2), (1000, 64), (4097, 7), (33, 1000), (2, 5000)]:
        x = rng.randint(-2 ** 31, 2 ** 31 - 1, size=(B, D)).astype(np.int32)  # any 4-byte pattern, incl. NaNs
        perm = rng.permutation(D).astype(np.int64)
        out = ops.permute_cols(dev(x), dev(perm))
        assert np.array_equal(host(out), x[:, perm])
    ops.check_status()
    with pytest.raises(IndexError):
        ops.permute_cols(dev(np.zeros((2, 3), np.float32)), dev(np.array([0, 1, 5], np.int64)))
        ops.check_status()


def test_rowsum_and_normal(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    assert np.abs(host(ops.rowsum(dev(g["rowsum_x"]))) - g["rowsum"]).max() <= 4e-6
    assert np.abs(host(ops.standard_normal_log_prob(dev(g["perm_x"]))) - g["normal_lp"]).max() <= 3e-5
    rng = np.random.RandomState(2)
    for B, D in [(1, 1), (5, 3), (1000, 64), (17, 1001)]:
        x = rng.randn(B, D).astype(np.float32)
        assert np.abs(host(ops.rowsum(dev(x))) - capi.rowsum(x)).max() <= 2e-6 * D
        lad = rng.randn(B).astype(np.float32)
        want = capi.standard_normal_log_prob(x) + lad
        got = host(ops.standard_normal_log_prob(dev(x), dev(lad)))
        assert np.abs(got - want).max() <= 1e-5 * (1 + np.abs(want).max())


# ------------------------------------------------------------------------------- size-independent
def test_full_size_round_trip_and_properties(ops):
    """BASELINE cfg-3/4 layer shape (B=65536, D=64, K=8): properties that need no oracle run."""
    B, D, K = 65536, 64, 8
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B, D, device=DEV, generator=g)
    P = 3 * K - 1
    tidx = torch.arange(0, D, 2, device=DEV)
    params = torch.randn(B, tidx.numel() * P, device=DEV, generator=g)
    spec, ospec = _spec_pair(ops, K, tails="linear", tail_bound=3.0, wh_divisor=float(np.sqrt(128)))
    y, lad = ops.rqs_coupling(x, params, tidx, spec)
    xr, lad_inv = ops.rqs_coupling(y, params, tidx, spec, inverse=True)
    ops.check_status()
    assert torch.equal(y[:, 1::2], x[:, 1::2])          # identity half bit-exact
    assert (xr - x).abs().max().item() < 1e-5           # per-layer fwd∘inv
    assert (lad + lad_inv).abs().max().item() < 2e-3     # sum of 32 per-feature log-derivatives
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()
    # monotone: ordering of two inputs under the same spline is preserved
    x2 = x.clone()
    x2[:, ::2] += 0.01
    y2, _ = ops.rqs_coupling(x2, params, tidx, spec)
    assert (y2[:, ::2] >= y[:, ::2]).all()
    # spot-check 2048 random rows against the oracle
    rows = torch.randint(0, B, (2048,), generator=torch.Generator().manual_seed(1))
    oy, ol, _ = capi.rqs_coupling(host(x[rows]), host(params[rows]), host(tidx), ospec)
    assert np.abs(host(y[rows]) - oy).max() <= 5e-6 and np.abs(host(lad[rows]) - ol).max() <= 5e-5


# ------------------------------------------------------------------ K9: linear / quadratic splines
def test_linear_and_quadratic_splines_golden(ops, golden_dir):
    """K9 through the drop-in functionals against the real reference's vectors
    (tests/golden/splines_lq.npz): constrained / linear tails, forward / inverse, shaped inputs."""
    from nflows_amd.transforms import splines
    g = np.load(os.path.join(golden_dir, "splines_lq.npz"))
    for name, kind, kw in g["meta"]:
        kw = parse_kwargs(kw)
        x = dev(g[name + "/x"])
        logits = [dev(g["%s/logits%d" % (name, i)]) for i in range(1 if kind == "linear" else 2)]
        unconstrained = kw.get("tails") == "linear"
        fn = {("linear", False): splines.linear_spline, ("linear", True): splines.unconstrained_linear_spline,
              ("quadratic", False): splines.quadratic_spline,
              ("quadratic", True): splines.unconstrained_quadratic_spline}[(str(kind), unconstrained)]
        for inverse in (False, True):
            pre = name + ("/inv_" if inverse else "/")
            y, lad = fn(x, *logits, inverse=inverse, **kw)
            ops.check_status()
            assert y.shape == x.shape and lad.shape == x.shape
            assert_sibling_spline_parity(host(y), g[pre + "y"], g[pre + "y64"], OUT_TOL, 5e-5, name + " y")
            assert_sibling_spline_parity(host(lad), g[pre + "lad"], g[pre + "lad64"], LAD_TOL, 1e-3, name + " lad")
            if unconstrained:
                tb = np.float32(kw["tail_bound"])
                xs = g[name + "/x"]
                outside = ~((xs >= -tb) & (xs <= tb))
                assert np.array_equal(host(y)[outside].view(np.uint32), xs[outside].view(np.uint32)), name
                assert np.all(host(lad)[outside] == 0), name


@pytest.mark.parametrize("K", [3, 8, 40])
def test_linear_and_quadratic_splines_oracle(ops, K):
    """Fresh inputs, packed and strided logit layouts, against the C oracle (float build)."""
    from nflows_amd.transforms import splines
    rng = np.random.RandomState(K)
    n = 3000
    x = (2.0 * rng.randn(n)).astype(np.float32)
    blob = (1.5 * rng.randn(n, 2 * K - 1 + 5)).astype(np.float32)   # widths | heights | unrelated columns
    spec = capi.make_spec(K, tails="linear", tail_bound=3.0)
    uw, uh = blob[:, :K], blob[:, K:2 * K - 1]
    for inverse in (False, True):
        oy, ol, st = capi.quadratic_spline(x, uw, uh, spec, inverse=inverse)
        assert st == 0
        t = dev(blob)
        y, lad = splines.unconstrained_quadratic_spline(dev(x), t[:, :K], t[:, K:2 * K - 1], inverse=inverse, tail_bound=3.0)
        packed = dev(np.ascontiguousarray(blob[:, :2 * K - 1]))
        y2, lad2 = splines.unconstrained_quadratic_spline(dev(x), packed[:, :K], packed[:, K:], inverse=inverse, tail_bound=3.0)
        assert torch.equal(y, y2) and torch.equal(lad, lad2)      # strided gather == packed tile path
        assert np.mean(np.abs(host(y) - oy) <= OUT_TOL * (1 + np.abs(oy))) >= 0.97
        assert np.mean(np.abs(host(lad) - ol) <= LAD_TOL * (1 + np.abs(ol))) >= 0.97
        oy, ol, st = capi.linear_spline(x, uw, spec, inverse=inverse)
        y, lad = splines.unconstrained_linear_spline(dev(x), t[:, :K], inverse=inverse, tail_bound=3.0)
        y2, lad2 = splines.unconstrained_linear_spline(dev(x), dev(np.ascontiguousarray(uw)), inverse=inverse, tail_bound=3.0)
        assert torch.equal(y, y2) and torch.equal(lad, lad2)
        assert np.mean(np.abs(host(y) - oy) <= OUT_TOL * (1 + np.abs(oy))) >= 0.97
        assert np.mean(np.abs(host(lad) - ol) <= LAD_TOL * (1 + np.abs(ol))) >= 0.97
    ops.check_status()
    # round trip
    xs = dev(np.clip(x, -2.999, 2.999))
    y, lad = splines.unconstrained_quadratic_spline(xs, t[:, :K], t[:, K:2 * K - 1], tail_bound=3.0)
    xr, lad_inv = splines.unconstrained_quadratic_spline(y, t[:, :K], t[:, K:2 * K - 1], inverse=True, tail_bound=3.0)
    assert (xr - xs).abs().median().item() < 1e-5 and (lad + lad_inv).abs().median().item() < 1e-4


def test_linear_and_quadratic_spline_errors(ops):
    from nflows_amd import InputOutsideDomain
    from nflows_amd.transforms import splines
    x = torch.tensor([0.5, 1.5], device=DEV)
    z = torch.zeros(2, 4, device=DEV)
    with pytest.raises(InputOutsideDomain):
        splines.linear_spline(x, z)
    with pytest.raises(InputOutsideDomain):
        splines.quadratic_spline(x, z, torch.zeros(2, 5, device=DEV))
    with pytest.raises(ValueError, match="Minimal bin width too large"):
        splines.quadratic_spline(x[:1], z[:1], torch.zeros(1, 5, device=DEV), min_bin_width=0.3)
    with pytest.raises(RuntimeError, match="cubic tails are not implemented"):
        splines.unconstrained_linear_spline(x[:1], z[:1], tails="cubic")
    with pytest.raises(AssertionError):
        splines.unconstrained_quadratic_spline(x[:1], z[:1], torch.zeros(1, 5, device=DEV))
    e = torch.zeros(0, device=DEV)
    y, lad = splines.linear_spline(e, torch.zeros(0, 4, device=DEV))
    assert y.shape == (0,) and lad.shape == (0,)
    ops.check_status()


def test_cubic_spline_golden(ops, golden_dir):
    """K9 (cubic) through the drop-in functionals against the real reference's vectors
    (tests/golden/splines_cubic.npz)."""
    from nflows_amd.transforms import splines
    g = np.load(os.path.join(golden_dir, "splines_cubic.npz"))
    for name, kind, kw in g["meta"]:
        kw = parse_kwargs(kw)
        x = dev(g[name + "/x"])
        logits = [dev(g["%s/logits%d" % (name, i)]) for i in range(4)]
        fn = splines.unconstrained_cubic_spline if kw.get("tails") == "linear" else splines.cubic_spline
        for inverse in (False, True):
            pre = name + ("/inv_" if inverse else "/")
            y, lad = fn(x, *logits, inverse=inverse, **kw)
            ops.check_status()
            assert_sibling_spline_parity(host(y), g[pre + "y"], g[pre + "y64"], OUT_TOL, 5e-5, name + " y")
            assert_sibling_spline_parity(host(lad), g[pre + "lad"], g[pre + "lad64"], LAD_TOL, 1e-3, name + " lad")
    # other K (runtime-K path), strided logits, round trip
    rng = np.random.RandomState(5)
    n, K = 2000, 6
    x = dev(np.clip(1.5 * rng.randn(n), -2.99, 2.99).astype(np.float32))
    blob = dev((0.7 * rng.randn(n, 2 * K + 2 + 3)).astype(np.float32))
    args = (blob[:, :K], blob[:, K:2 * K], blob[:, 2 * K:2 * K + 1], blob[:, 2 * K + 1:2 * K + 2])
    y, lad = splines.unconstrained_cubic_spline(x, *args, tail_bound=3.0)
    oy, ol, st = capi.cubic_spline(host(x), *[host(t) for t in args], capi.make_spec(K, tails="linear", tail_bound=3.0))
    assert st == 0
    assert np.mean(np.abs(host(y) - oy) <= OUT_TOL * (1 + np.abs(oy))) >= 0.97
    assert np.mean(np.abs(host(lad) - ol) <= LAD_TOL * (1 + np.abs(ol))) >= 0.97
    xr, lad_inv = splines.unconstrained_cubic_spline(y, *args, inverse=True, tail_bound=3.0)
    ops.check_status()
    assert (xr - x).abs().median().item() < 1e-5 and (lad + lad_inv).abs().median().item() < 1e-4


@pytest.mark.parametrize("K,D", [(8, 64), (10, 64), (8, 32), (8, 128), (8, 8)])
def test_wave_tile_coupling_kernel_is_bit_identical_to_the_pipelined_kernel(K, D):
    """K1's wave-tile form (round 4: LDS-DMA into a wave-private image, no workgroup barriers; linear tails, d_t a
    power of two, D = 2 d_t) evaluates every spline with the same rqs_eval as the register-pipelined kernel: outputs
    and log-determinants bit for bit, forward and inverse, with fused permutations on both sides, accumulation into a
    running total, a ragged batch (leftover rows go to the generic kernel either way), NaN / out-of-box inputs."""
    import os
    from nflows_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    B = 8192 + 37
    x = torch.randn(B, D, device=dev, generator=g) * 1.6
    x[3, 1] = float("nan")
    x[4, 0] = 9.0
    tidx = torch.arange(0, D, 2, device=dev)
    P = 3 * K - 1
    params = torch.randn(B, tidx.numel() * P, device=dev, generator=g) * 2.0
    perm = torch.randperm(D, device=dev, generator=g)
    scat = torch.randperm(D, device=dev, generator=g)
    spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(128)))
    running = torch.randn(B, device=dev, generator=g)
    results = {}
    saved = os.environ.get("NFA_K1_WAVETILE")
    try:
        for mode in ("1", "0"):
            os.environ["NFA_K1_WAVETILE"] = mode
            out = []
            for inverse in (False, True):
                for ip, osc in ((None, None), (perm, scat)):
                    y, lad = ops.rqs_coupling(x, params, tidx, spec, inverse=inverse, in_perm=ip, out_scatter=osc)
                    out += [y, lad, ops.last_layer_kernel()]
                acc = running.clone()
                y, lad = ops.rqs_coupling(x, params, tidx, spec, inverse=inverse, accumulate_into=acc)
                out += [y, acc]
            results[mode] = out
    finally:
        if saved is None:
            os.environ.pop("NFA_K1_WAVETILE", None)
        else:
            os.environ["NFA_K1_WAVETILE"] = saved
    ops.check_status()
    for a_, b_ in zip(results["1"], results["0"]):
        if isinstance(a_, str):
            # (the note is left by the last launch: the generic kernel on the 37 leftover rows)
            continue
        assert torch.equal(a_.view(torch.int32), b_.view(torch.int32)), "wave-tile and pipelined kernels differ"
    # the kernel under test really ran: a batch of whole tiles
    os.environ["NFA_K1_WAVETILE"] = "1"
    try:
        ops.rqs_coupling(x[:8192], params[:8192], tidx, spec)
        assert "wavetile" in ops.last_layer_kernel(), ops.last_layer_kernel()
    finally:
        if saved is None:
            os.environ.pop("NFA_K1_WAVETILE", None)
        else:
            os.environ["NFA_K1_WAVETILE"] = saved
