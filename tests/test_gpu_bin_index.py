"""The bin index on the device (SURVEY.md 8 row a3, 8c "bin index exact").

torchutils.searchsorted (utils/torchutils.py:134-136) appears in the product three ways, each held here to the
reference's known-answer test (tests/utils/torchutils_test.py:80-90), to the bin_idx the REAL reference computed
(tests/golden/rqs_bins.npz: caught inside rational_quadratic.py:115-118 by make_golden.py) and to the oracle:

  1. on its own: `nfa_searchsorted_f32` behind `nflows_amd.utils.searchsorted` -- integer work on given knots: bit-exact;
  2. fused into K5 / K1 (`rqs_eval`: the reference's knot recipe step by step): the kernels report the bin they chose
     through `bin_idx` -- exact on the reference's 24 functional cases; on 2^20 random elements equal to the oracle's
     except for an input between the two evaluations' versions of a knot (helpers.assert_bins_match: <= 4 per 2^20, one
     bin, within 6 ulp(span) of the knot);
  3. K8h / K8s keep NO index (one walk over fp32 running knot sums, csrc/rqs_fused8.hpp): their diagnostic twins store
     the bin of the last layer's evaluations; the fraction that differs from the reference evaluation of the same layer
     is reported and bounded, every difference is one bin at an input next to the knot, and at exactly those elements
     the outputs and the row's logabsdet stay within the parity tolerances (the spline is C1 across a knot).
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from helpers import LAD_TOL, OUT_TOL, assert_bins_match, parse_kwargs, steepen
from oracle import capi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def _report(rec):
    path = os.environ.get("NFA_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


@pytest.fixture(scope="module")
def ops():
    from nflows_amd import ops as o
    return o


# ------------------------------------------------------------------------------- 1. searchsorted on its own
def test_searchsorted_reference_known_answer(golden_dir):
    """tests/utils/torchutils_test.py:80-90 as the reference runs it (one row of knots broadcast over the inputs),
    through the drop-in `utils.searchsorted` on device tensors, and :92-97 (arbitrary leading shape)."""
    from nflows_amd import utils
    g = np.load(os.path.join(golden_dir, "searchsorted.npz"))
    knots = dev(g["knots"])
    before = knots.clone()
    for which in ("left", "right", "mid"):
        idx = utils.searchsorted(knots[None, :], dev(g[which + "_in"]))
        assert idx.dtype == torch.int64 and idx.is_cuda and tuple(idx.shape) == (9,)
        assert np.array_equal(host(idx), g[which + "_idx"]) and np.array_equal(host(idx), np.arange(9)), which
    assert torch.equal(knots, before)           # (the reference leaves += eps behind; this one does not)
    shape = [2, 3, 4]
    locations = torch.linspace(0, 1, 10).repeat(*shape, 1).to(DEV)
    inputs = torch.rand(*shape).to(DEV)
    idx = utils.searchsorted(locations, inputs)
    assert tuple(idx.shape) == tuple(shape)
    want = (host(inputs)[..., None] >= (host(locations) + np.float32([0] * 9 + [1e-6]))).sum(-1) - 1
    assert np.array_equal(host(idx), want)


def test_searchsorted_kernel_is_exact_on_a_million_elements(ops):
    """2^20 inputs: one shared row of knots against oracle.capi.searchsorted (the C restatement pinned to the
    reference's known answers), per-input rows (monotone, NON-monotone: the reference counts, it does not bisect; NaN
    and +-inf inputs) against the count written out in numpy.  Integer results: bit-exact."""
    rng = np.random.default_rng(11)
    n = 1 << 20
    knots = np.sort(rng.uniform(-3, 3, 33)).astype(np.float32)
    x = rng.uniform(-3.5, 3.5, n).astype(np.float32)
    x[:64] = np.repeat(knots[:32], 2)                       # inputs ON knots
    x[64:67] = [np.nan, np.inf, -np.inf]
    got = host(ops.searchsorted(dev(knots), dev(x)))
    assert np.array_equal(got, capi.searchsorted(knots, x))
    for nk, monotone in ((9, True), (11, True), (8, False), (130, True)):
        m = n if nk < 100 else 1 << 14
        rows = rng.standard_normal((m, nk)).astype(np.float32)
        if monotone:
            rows = np.sort(rows, axis=1)
        xs = rng.standard_normal(m).astype(np.float32)
        xs[:nk] = rows[0]                                      # on the first row's knots
        xs[nk:nk + 2] = [np.nan, np.inf]
        nudged = rows.copy()
        nudged[:, -1] += np.float32(1e-6)
        want = (xs[:, None] >= nudged).sum(1).astype(np.int64) - 1
        got = host(ops.searchsorted(dev(rows), dev(xs)))
        assert np.array_equal(got, want), (nk, monotone)
        # a strided view of a wider buffer (what `cumwidths[..., :K + 1]` of a padded tensor looks like)
        wide = torch.zeros(m, nk + 3, device=DEV)
        wide[:, :nk] = dev(rows)
        assert np.array_equal(host(ops.searchsorted(wide[:, :nk], dev(xs))), want), (nk, "strided")


# ------------------------------------------------------------------------------- 2. the search fused into K5 / K1
def test_k5_bin_index_equals_the_reference(ops, golden_dir):
    """The 24 functional cases: K5's bin_idx (float32 kernel, both instances, and the float64 kernel) == the bin_idx of
    the real reference, element for element (-1 = an element the reference does not search)."""
    G = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    B = np.load(os.path.join(golden_dir, "rqs_bins.npz"))
    for name, inv, kw in G["meta"]:
        kw = parse_kwargs(kw)
        x, uw, uh, ud = (G[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        spec = ops.make_rqs_spec(uw.shape[-1], kw.pop("tails", None), **kw)
        for cast in (torch.float32, torch.float64):
            args = [dev(a).to(cast) for a in (x, uw, uh, ud)]
            y, lad, bins = ops.rqs_elementwise(*args, spec, inverse=bool(int(inv)), return_bin_idx=True)
            y2, lad2 = ops.rqs_elementwise(*args, spec, inverse=bool(int(inv)))
            assert bins.dtype == torch.int32 and tuple(bins.shape) == x.shape
            if cast == torch.float32:
                assert np.array_equal(host(bins).astype(np.int64), B[name + "/bin_idx"]), name
            else:   # (float64 knots differ from the fp32 reference's in the 8th digit: an input ON a knot may flip)
                assert (host(bins) != B[name + "/bin_idx"]).mean() <= 0.002, name
            # asking for the index changes nothing else
            assert torch.equal(torch.nan_to_num(y), torch.nan_to_num(y2)) and torch.equal(torch.nan_to_num(lad), torch.nan_to_num(lad2)), name
    import nflows_amd
    try:
        nflows_amd.check_status()
    except Exception:
        pass


def test_k5_bin_index_on_the_reference_knots(ops, golden_dir):
    """Inputs ON the reference's own knots and one ulp beside them (rqs_bins.npz, knots_*): never more than one bin
    from the reference's choice, the tail decision identical, values within the per-element allowances."""
    from helpers import assert_fp32_parity, conditioning, knot_case_keep
    import nflows_amd
    B = np.load(os.path.join(golden_dir, "rqs_bins.npz"))
    for name, inv, kw in B["meta"]:
        kw = parse_kwargs(kw)
        inverse = bool(int(inv))
        x, uw, uh, ud = (B[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        K = uw.shape[-1]
        ospec = capi.make_spec(K, **kw)
        spec = ops.make_rqs_spec(K, kw.pop("tails", None), **kw)
        y, lad, bins = ops.rqs_elementwise(dev(x), dev(uw), dev(uh), dev(ud), spec, inverse=inverse, return_bin_idx=True)
        status = 0
        try:
            nflows_amd.check_status()
        except AssertionError as e:
            assert "negative discriminant" in str(e), str(e)
            status = 2
        y, lad, bins, ref = host(y), host(lad), host(bins).astype(np.int64), B[name + "/bin_idx"]
        assert np.abs(bins - ref).max() <= 1 and np.array_equal(bins == -1, ref == -1), name
        _report({"config": name, "what": "K5 bins vs the reference's on its own knots", "differ": float((bins != ref).mean())})
        assert (bins != ref).mean() < 0.2, name
        keep = knot_case_keep(y, B[name + "/y"], lad, B[name + "/lad"], status, inverse, name)
        cy, cl = conditioning(lambda *a: capi.rqs_elementwise(*a, ospec, inverse=inverse)[:2], (x, uw, uh, ud), (0, 1, 2, 3))
        assert_fp32_parity(y[keep], B[name + "/y"][keep], B[name + "/y64"][keep], OUT_TOL, name + " y", cond=cy[keep], bulk=0.99)
        assert_fp32_parity(lad[keep], B[name + "/lad"][keep], B[name + "/lad64"][keep], LAD_TOL, name + " lad", cond=cl[keep], bulk=0.99)


@pytest.mark.parametrize("K,scale", [(8, 1.0), (8, 3.0), (5, 2.0), (10, 2.0)])
@pytest.mark.parametrize("inverse", [False, True])
def test_k5_bin_index_on_a_million_random_elements(ops, K, scale, inverse):
    rng = np.random.default_rng(100 + K + int(10 * scale) + inverse)
    n = 1 << 20
    x = (rng.standard_normal(n) * 1.5).astype(np.float32)
    pr = (rng.standard_normal((n, 3 * K - 1)) * scale).astype(np.float32)
    ospec = capi.make_spec(K, tails="linear", tail_bound=3.0)
    spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0)
    ob = capi.rqs_elementwise(x, pr[:, :K], pr[:, K:2 * K], pr[:, 2 * K:], ospec, inverse=inverse, return_bins=True)[3]
    knots = capi.rqs_knots(pr[:, K:2 * K] if inverse else pr[:, :K], ospec, axis=int(inverse))
    p = dev(pr)
    bins = ops.rqs_elementwise(dev(x), p[:, :K], p[:, K:2 * K], p[:, 2 * K:], spec, inverse=inverse, return_bin_idx=True)[2]
    d = assert_bins_match(host(bins), ob, x, knots, "K5 K=%d scale=%g inverse=%d" % (K, scale, inverse))
    _report({"config": "k5_bins_K%d_scale%g_inv%d" % (K, scale, inverse), "elements": n, "differ": int(d.size)})
    try:
        import nflows_amd
        nflows_amd.check_status()
    except AssertionError:
        pass


# K1: which kernel, by shape -- (features, transformed features (first ones of an alternating / custom mask), bins, rows)
# (`kernel`: the LAST launch of the call -- a ragged batch ends with the generic kernel on the rows behind the last full tile)
K1_SHAPES = {
    "wavetile": (64, 32, 8, 32768, "rqs_coupling_wavetile<K=8"),
    "wavetile_ragged": (64, 32, 8, 8192 + 3, "rqs_coupling_kernel<K=8"),
    "wavetile_k10": (32, 16, 10, 16384, "rqs_coupling_wavetile<K=10"),
    "pipelined": (48, 24, 8, 16380, "rqs_coupling_pipelined<K=8"),          # (tiles of 10 rows: 1 638 full tiles)
    "pipelined_ragged": (48, 24, 8, 4096 + 5, "rqs_coupling_kernel<K=8"),
    "generic": (10, 5, 5, 40000, "rqs_coupling_kernel<K=0"),
}


@pytest.mark.parametrize("shape", list(K1_SHAPES))
@pytest.mark.parametrize("inverse", [False, True])
def test_k1_bin_index(ops, shape, inverse):
    """The fused layer kernels (wave-tile, register-pipelined, generic; ragged batches: the tail rows go to the generic
    kernel) with a fused permutation and the 1 / sqrt(hidden) scale: bin_idx [B, d_t] in transform_idx order against
    the oracle's, ~10^6 splines per case; outputs and logabsdet unchanged by the request."""
    D, dt, K, rows, kernel = K1_SHAPES[shape]
    rng = np.random.default_rng(7 + D + inverse)
    P = 3 * K - 1
    x = (rng.standard_normal((rows, D)) * 1.5).astype(np.float32)
    params = (rng.standard_normal((rows, dt * P)) * 12.0).astype(np.float32)   # (/ sqrt(128): logits ~ N(0, 1))
    tidx = np.sort(rng.permutation(D)[:dt]).astype(np.int64) if shape.startswith(("pipelined", "generic")) else np.arange(0, D, 2, dtype=np.int64)
    perm = rng.permutation(D).astype(np.int64)
    H = 128
    spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=math.sqrt(H))
    ospec = capi.make_spec(K, tails="linear", tail_bound=3.0, wh_divisor=math.sqrt(H))
    out, lad, bins = ops.rqs_coupling(dev(x), dev(params), dev(tidx), spec, inverse=inverse, in_perm=dev(perm), return_bin_idx=True)
    assert kernel in ops.last_layer_kernel(), ops.last_layer_kernel()
    out2, lad2 = ops.rqs_coupling(dev(x), dev(params), dev(tidx), spec, inverse=inverse, in_perm=dev(perm))
    assert torch.equal(out, out2) and torch.equal(lad, lad2)
    assert bins.dtype == torch.int32 and tuple(bins.shape) == (rows, dt)
    xt = x[:, perm][:, tidx].reshape(-1)
    pr = params.reshape(rows * dt, P)
    ob = capi.rqs_elementwise(xt, pr[:, :K], pr[:, K:2 * K], pr[:, 2 * K:], ospec, inverse=inverse, return_bins=True)[3]
    knots = capi.rqs_knots(pr[:, K:2 * K] if inverse else pr[:, :K], ospec, axis=int(inverse))
    d = assert_bins_match(host(bins), ob, xt, knots, "K1 %s inverse=%d" % (shape, inverse))
    _report({"config": "k1_bins_%s_inv%d" % (shape, inverse), "elements": int(xt.size), "differ": int(d.size),
             "kernel": ops.last_layer_kernel()})
    try:
        import nflows_amd
        nflows_amd.check_status()
    except AssertionError:
        pass


# ------------------------------------------------------------------------------- 3. K8h / K8s: no index on the data path
def _last_layer_reference(flow_cpu, x_mid, inverse):
    """The reference evaluation of the flow's LAST coupling layer on given inputs: bins, elementwise outputs and
    row logabsdet in fp32 (C oracle on the fp32 conditioner output: the reference's arithmetic) and fp64, plus the
    oracle's knots of the searched axis.  `x_mid`: the layer's input rows (before its permutation; for the inverse: the
    rows that enter the layer's inverse, i.e. after the inverse of nothing -- the last layer is the first to run)."""
    layers = list(flow_cpu._transform._transforms)
    perm_t, coup = layers[-2], layers[-1]
    perm = perm_t._permutation.numpy()
    idf, trf = coup.identity_features.numpy(), coup.transform_features.numpy()
    K, tb, H = coup.num_bins, coup.tail_bound, coup.transform_net.hidden_features
    res = {}
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xm = torch.from_numpy(x_mid).to(dt)
        # forward: Permutation then coupling (base.py:45-52); inverse: the coupling's inverse runs first, on the rows as given
        xin = xm[:, torch.from_numpy(perm)] if not inverse else xm
        with torch.no_grad():
            params = coup.transform_net.to(dt)(xin[:, torch.from_numpy(idf)]).numpy()
        coup.transform_net.float()
        B, dtc = xin.shape[0], len(trf)
        pr = params.reshape(B * dtc, 3 * K - 1)
        spec = capi.make_spec(K, tails="linear", tail_bound=tb, wh_divisor=math.sqrt(H))
        xt = np.ascontiguousarray(xin[:, torch.from_numpy(trf)].numpy().reshape(-1))
        y, lad, st, bins = capi.rqs_elementwise(xt, pr[:, :K], pr[:, K:2 * K], pr[:, 2 * K:], spec, inverse=inverse, return_bins=True)
        res["y" + tag], res["lad" + tag], res["bins" + tag] = y.reshape(B, dtc), lad.reshape(B, dtc).sum(1), bins.reshape(B, dtc)
        if tag == "32":
            res["knots"] = capi.rqs_knots(pr[:, K:2 * K] if inverse else pr[:, :K], spec, axis=int(inverse))
            res["xt"] = xt
    res["perm"], res["trf"] = perm, trf
    return res


@pytest.mark.parametrize("engine,rows", [("k8h_w8", 65536), ("k8h_w4", 16384), ("k8s_w8", 32768), ("k8s_w4", 8192)])
@pytest.mark.parametrize("inverse", [False, True])
def test_whole_layer_kernels_choose_the_reference_bin(golden_dir, engine, rows, inverse):
    """The two-layer steep flow of flows_steep.npz (BASELINE layer shape: D = 64, K = 8, ResidualNet 128 x 2) through K8h / K8s with the
    diagnostic twin: the bins of the launch's last layer against the reference evaluation of that layer on the SAME
    layer inputs (the kernel's own first-layer output, taken from a one-layer launch of the same kernel).  K8h's logits
    come out of its own GEMMs (~1e-6 relative from the reference's) and its knots are fp32 running sums, so a knot sits
    a little off the reference's and an input in that gap lands in the neighbouring bin: the fraction is reported and
    bounded by 2e-5 (measured: 0 - 2 of 2 097 152 elements, profiles/r5/bin_index.jsonl), the gap by 1e-5 x span
    (measured: <= 1.2e-7 x span), and at exactly those elements the output and the row's logabsdet are within the parity
    tolerances of the float64 result."""
    import copy
    import nflows_amd
    from nflows_amd import configs, ops
    from nflows_amd.transforms import CompositeTransform
    from helpers import steep_flow
    flow_cpu, _, cfg = steep_flow(golden_dir, "steep_nsf_k8")     # two layers, logits ~ N(0, 1.8 .. 3.3): as after training
    K, D, tb = cfg["K"], cfg["D"], cfg["tail_bound"]
    gen = torch.Generator().manual_seed(41 + inverse)
    x = torch.randn(rows, D, generator=gen) * 1.3
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    layers = list(flow._transform._transforms)
    saved, saved_c = ops.K8S_ENABLED, ops.K8C_ENABLED
    ops.K8S_ENABLED = engine.startswith("k8s")
    ops.K8C_ENABLED = False      # (K8s's diagnostic twin is the subject; K8c has none and would take the uncaptured launches)
    try:
        with torch.no_grad():
            if not inverse:
                # the last layer's inputs as the launch holds them: a one-layer launch of the same kernel (rows stay
                # fp32 in LDS between layers: the same values)
                x_mid, _ = CompositeTransform(layers[:2])(x.to(DEV))
                assert engine[:3] + "::" in ops.last_layer_kernel() and "waves=" + engine[-1] in ops.last_layer_kernel(), ops.last_layer_kernel()
                with ops.capture_last_layer_bins() as cap:
                    z, lad = flow._transform(x.to(DEV))
            else:
                # inverse: the flow's LAST layer runs FIRST, the launch's last layer is the flow's FIRST: the reference
                # below is therefore built for layers[0:2] on the rows the first inverse step produced
                x_mid, _ = CompositeTransform(layers[2:]).inverse(x.to(DEV))
                assert engine[:3] + "::" in ops.last_layer_kernel() and "waves=" + engine[-1] in ops.last_layer_kernel(), ops.last_layer_kernel()
                with ops.capture_last_layer_bins() as cap:
                    z, lad = flow._transform.inverse(x.to(DEV))
            label = ops.last_layer_kernel()
        assert cap.launches == 1 and cap.bins is not None
        assert engine[:3] + "::" in label and "waves=" + engine[-1] in label and ("inverse=%d" % inverse) in label, label
        assert int(cap.redo.sum()) == 0
        try:
            nflows_amd.check_status()
        except AssertionError as e:
            assert "negative discriminant" in str(e)
    finally:
        ops.K8S_ENABLED, ops.K8C_ENABLED = saved, saved_c
    ref_flow = copy.deepcopy(flow_cpu)
    if inverse:   # the launch's last layer = the flow's first pair
        ref_flow._transform._transforms = torch.nn.ModuleList(list(flow_cpu._transform._transforms)[:2])
    R = _last_layer_reference(ref_flow, host(x_mid), inverse)
    bins = host(cap.bins)[:, :R["bins32"].shape[1]]
    assert bins.min() >= -1 and bins.max() <= K - 1
    assert np.array_equal(bins == -1, R["bins32"] == -1)          # the tail decision: a compare of the same fp32 input with +-B
    n = bins.size
    d = np.nonzero(bins.reshape(-1) != R["bins32"].reshape(-1))[0]
    frac = d.size / n
    b_got, b_ref = bins.reshape(-1)[d], R["bins32"].reshape(-1)[d]
    kn = R["knots"][d, np.maximum(b_got, b_ref)] if d.size else np.zeros(0, np.float32)
    gap = np.abs(R["xt"][d].astype(np.float64) - kn) if d.size else np.zeros(0)
    # the layer's own results at those elements: forward: column trf[j] of z after nothing else; inverse: the flow's
    # first permutation is inverted after the coupling (x = y[:, argsort(perm)]), i.e. z[:, perm[trf[j]]]
    zt = host(z)
    cols = R["trf"] if not inverse else R["perm"][R["trf"]]
    got_y = zt[:, cols].reshape(-1)[d]
    e_got = np.abs(got_y.astype(np.float64) - R["y64"].reshape(-1)[d])
    e_ref = np.abs(R["y32"].reshape(-1)[d].astype(np.float64) - R["y64"].reshape(-1)[d])
    _report({"config": "%s_bins_inv%d" % (engine, inverse), "kernel": label, "elements": n, "differ": int(d.size), "fraction": frac,
             "max_gap_over_span": float(gap.max() / (2 * tb)) if d.size else 0.0,
             "max_output_error_at_differing": float(e_got.max()) if d.size else 0.0,
             "reference_fp32_error_there": float(e_ref.max()) if d.size else 0.0})
    assert frac <= 2e-5, frac
    if d.size:
        assert np.all(np.abs(b_got - b_ref) == 1)
        assert gap.max() <= 1e-5 * 2 * tb, float(gap.max())
        assert np.all(e_got <= 8 * OUT_TOL * (1 + np.abs(R["y64"].reshape(-1)[d])) + 4 * e_ref), float(e_got.max())
        # rows that hold a differing element: the layer's logabsdet (the launch's total minus the other layer's is not at
        # hand; the total of TWO layers is compared for the forward direction through the one-layer launch below)
    if not inverse:
        with torch.no_grad():
            _, lad1 = CompositeTransform(layers[:2])(x.to(DEV))
        lad_last = host(lad) - host(lad1)
        rows_d = np.unique(d // bins.shape[1])
        if rows_d.size:
            e_l = np.abs(lad_last[rows_d].astype(np.float64) - R["lad64"][rows_d])
            e_lref = np.abs(R["lad32"][rows_d].astype(np.float64) - R["lad64"][rows_d])
            assert np.all(e_l <= 8 * LAD_TOL * (1 + np.abs(R["lad64"][rows_d])) + 4 * e_lref), float(e_l.max())
