"""Pins the CPU oracle (oracle/nfa_oracle.c) against vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest

from helpers import LAD_TOL, OUT_TOL, assert_fp32_parity, assert_sibling_spline_parity, conditioning, knot_case_keep, parse_kwargs
from oracle import capi


@pytest.fixture(scope="module")
def rqs(golden_dir):
    return np.load(os.path.join(golden_dir, "rqs_functional.npz"))


def _cases(golden_dir, fname):
    g = np.load(os.path.join(golden_dir, fname))
    return g, [tuple(r) for r in g["meta"]]


def test_rqs_functional_fp64_matches_reference_fp64(rqs):
    """The double build of the oracle restates the algorithm: agrees with the reference run in
    float64 to 1e-10 on every case (this is the check that the restatement is the same maths)."""
    for name, inv, kw in rqs["meta"]:
        kw = parse_kwargs(kw)
        x, uw, uh, ud = (rqs[name + "/" + k].astype(np.float64) for k in ("x", "uw", "uh", "ud"))
        spec = capi.make_spec(uw.shape[-1], **kw)
        y, lad, st = capi.rqs_elementwise(x, uw, uh, ud, spec, inverse=bool(int(inv)))
        ry, rl = rqs[name + "/y64"], rqs[name + "/lad64"]
        assert st == 0, name
        assert np.array_equal(np.isnan(y), np.isnan(ry)), name
        fin = np.isfinite(ry)
        assert np.abs(y[fin] - ry[fin]).max() <= 1e-10, name
        assert np.abs(lad[fin] - rl[fin]).max() <= 1e-10, name
        assert np.array_equal(y[~fin & ~np.isnan(ry)], ry[~fin & ~np.isnan(ry)]), name


def test_rqs_functional_fp32_matches_reference_fp32(rqs):
    for name, inv, kw in rqs["meta"]:
        kw = parse_kwargs(kw)
        x, uw, uh, ud = (rqs[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        spec = capi.make_spec(uw.shape[-1], **kw)
        y, lad, st = capi.rqs_elementwise(x, uw, uh, ud, spec, inverse=bool(int(inv)))
        assert st == 0, name
        cy, cl = conditioning(lambda *a: capi.rqs_elementwise(*a, spec, inverse=bool(int(inv)))[:2], (x, uw, uh, ud), (0, 1, 2, 3))
        assert_fp32_parity(y, rqs[name + "/y"], rqs[name + "/y64"], OUT_TOL, name + " y", cond=cy)
        assert_fp32_parity(lad, rqs[name + "/lad"], rqs[name + "/lad64"], LAD_TOL, name + " lad", cond=cl)
        # pass-through elements are bit-exact, logabsdet exactly 0 there (A2)
        if kw.get("tails") == "linear":
            tb = np.float32(kw["tail_bound"])
            outside = ~((x >= -tb) & (x <= tb))
            assert np.array_equal(y[outside].view(np.uint32), x[outside].view(np.uint32)), name
            assert np.all(lad[outside] == 0), name


def test_searchsorted_known_answer(golden_dir):
    """reference tests/utils/torchutils_test.py:80-90"""
    g = np.load(os.path.join(golden_dir, "searchsorted.npz"))
    for which in ("left", "right", "mid"):
        idx = capi.searchsorted(g["knots"], g[which + "_in"])
        assert np.array_equal(idx, g[which + "_idx"])
        assert np.array_equal(idx, np.arange(9))


def test_bin_index_matches_the_reference(golden_dir):
    """`bin_idx` of rational_quadratic.py:115-118 -- caught inside the real reference by wrapping its
    torchutils.searchsorted (tests/golden/make_golden.py: bin_index_cases) -- against the oracle's `bins`: EXACT on all
    24 functional cases (-1 = an element the reference never searches).  On the adversarial cases (inputs ON a
    reference knot or one ulp beside it) two correct fp32 evaluations may disagree by one bin where their knots
    differ in the last bits (aten's vectorised softmax sum is not restated bit for bit): neighbours only, and only
    there; the values stay within the usual per-element allowances (the spline is C1 across a knot)."""
    B = np.load(os.path.join(golden_dir, "rqs_bins.npz"))
    G = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    for name, inv, kw in G["meta"]:
        kw = parse_kwargs(kw)
        x, uw, uh, ud = (G[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        spec = capi.make_spec(uw.shape[-1], **kw)
        bins = capi.rqs_elementwise(x, uw, uh, ud, spec, inverse=bool(int(inv)), return_bins=True)[3]
        assert np.array_equal(bins.astype(np.int64), B[name + "/bin_idx"]), name
    assert len(B["meta"]) == 6
    for name, inv, kw in B["meta"]:
        kw = parse_kwargs(kw)
        inverse = bool(int(inv))
        x, uw, uh, ud = (B[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        K = uw.shape[-1]
        spec = capi.make_spec(K, **kw)
        y, lad, st, bins = capi.rqs_elementwise(x, uw, uh, ud, spec, inverse=inverse, return_bins=True)
        ref = B[name + "/bin_idx"]
        off = bins.astype(np.int64) - ref
        assert np.abs(off).max() <= 1, name
        assert np.array_equal(bins == -1, ref == -1), name            # the tail decision is a compare with +-B: exact
        # a differing bin means the input sits within an ulp or two of the reference's knot between the two bins
        d = np.nonzero(off)[0]
        assert d.size < 0.2 * x.size, name
        kn = B[name + "/knots"][d, np.maximum(ref[d], bins[d])]
        span = np.float32(2 * kw["tail_bound"])   # (a knot is RN(span * cumsum + left): its versions differ by ulps of the span)
        assert np.all(np.abs(x[d].astype(np.float64) - kn) <= 4 * np.spacing(span)), name
        # the oracle's knots against the reference's own `cumwidths` / `cumheights` (recorded next to bin_idx): never
        # more than four ulp of the span apart (measured: 2 - 4), 71 - 78 % of them identical -- the reproducibility of
        # the reference's own knots by a restatement that does not copy aten's vectorised softmax sum
        inside = ref >= 0
        okn = capi.rqs_knots(uh if inverse else uw, spec, axis=int(inverse))[inside]
        rkn = B[name + "/knots"][inside]
        assert np.all(np.abs(okn.astype(np.float64) - rkn) <= 4 * np.spacing(span)), name
        assert (okn == rkn).mean() > 0.7, name
        cy, cl = conditioning(lambda *a: capi.rqs_elementwise(*a, spec, inverse=inverse)[:2], (x, uw, uh, ud), (0, 1, 2, 3))
        keep = knot_case_keep(y, B[name + "/y"], lad, B[name + "/lad"], st, inverse, name)
        assert_fp32_parity(y[keep], B[name + "/y"][keep], B[name + "/y64"][keep], OUT_TOL, name + " y", cond=cy[keep], bulk=0.99)
        assert_fp32_parity(lad[keep], B[name + "/lad"][keep], B[name + "/lad64"][keep], LAD_TOL, name + " lad", cond=cl[keep], bulk=0.99)


def test_coupling_layers(golden_dir):
    g, meta = _cases(golden_dir, "coupling.npz")
    for name, kind, cfg in meta:
        cfg = parse_kwargs(cfg)
        x, params, tidx = g[name + "/x"], g[name + "/params"], g[name + "/transform_idx"]
        for direction, inv in (("fwd", False), ("inv", True)):
            ry, rl = g["%s/%s_y" % (name, direction)], g["%s/%s_lad" % (name, direction)]
            ry64, rl64 = g["%s/%s_y64" % (name, direction)], g["%s/%s_lad64" % (name, direction)]
            for dt in (np.float32, np.float64):
                if kind == "rq":
                    H = cfg["hidden"]
                    spec = capi.make_spec(cfg["K"], tails=cfg["tails"], tail_bound=cfg["tail_bound"],
                                          wh_divisor=float(np.sqrt(H)) if H else 0.0)
                    y, lad, st = capi.rqs_coupling(x.astype(dt), params.astype(dt), tidx, spec, inverse=inv)
                else:
                    act = {"affine_default": capi.AFFINE_DEFAULT, "affine_general": capi.AFFINE_GENERAL,
                           "affine_additive": capi.AFFINE_ADDITIVE}[kind]
                    y, lad, st = capi.affine_coupling(x.astype(dt), params.astype(dt), tidx, act, inverse=inv)
                assert st == 0, name
                if dt is np.float64:
                    assert np.abs(y - ry64).max() <= 1e-10, (name, direction)
                    assert np.abs(lad - rl64).max() <= 1e-9, (name, direction)
                else:
                    if kind == "rq":
                        f64 = lambda xx, pp: capi.rqs_coupling(xx, pp, tidx, spec, inverse=inv)[:2]       # noqa: E731
                    else:
                        f64 = lambda xx, pp: capi.affine_coupling(xx, pp, tidx, act, inverse=inv)[:2]     # noqa: E731
                    cy, cl = conditioning(f64, (x.astype(np.float64), params.astype(np.float64)), (0, 1))
                    assert_fp32_parity(y, ry, ry64, OUT_TOL, name + direction + " y", cond=cy)
                    assert_fp32_parity(lad, rl, rl64, LAD_TOL, name + direction + " lad", cond=cl)
                    ident = np.setdiff1d(np.arange(x.shape[1]), tidx)
                    assert np.array_equal(y[:, ident], x[:, ident]), name
                    if kind == "affine_additive":
                        assert np.all(lad == 0), name


def test_misc(golden_dir):
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    out, st = capi.permute_cols(g["perm_x"], g["perm"])
    assert st == 0 and np.array_equal(out, g["perm_fwd"])
    out, _ = capi.permute_cols(g["perm_x"], np.argsort(g["perm"]))
    assert np.array_equal(out, g["perm_inv"])
    assert np.abs(capi.rowsum(g["rowsum_x"]) - g["rowsum"]).max() <= 4e-6
    assert np.abs(capi.standard_normal_log_prob(g["perm_x"]) - g["normal_lp"]).max() <= 3e-5


def test_fused_permutation_equals_sequence(golden_dir):
    """in_perm / out_scatter in the oracle are exactly Permutation then layer / layer then
    Permutation.inverse."""
    g, meta = _cases(golden_dir, "coupling.npz")
    name = "rq_d64_k8"
    cfg = parse_kwargs([m for m in meta if m[0] == name][0][2])
    x, params, tidx = g[name + "/x"], g[name + "/params"], g[name + "/transform_idx"]
    spec = capi.make_spec(cfg["K"], tails=cfg["tails"], tail_bound=cfg["tail_bound"],
                          wh_divisor=float(np.sqrt(cfg["hidden"])))
    perm = np.random.RandomState(0).permutation(x.shape[1])
    y_seq, lad_seq, _ = capi.rqs_coupling(x[:, perm], params, tidx, spec)
    y_f, lad_f, _ = capi.rqs_coupling(x, params, tidx, spec, in_perm=perm)
    assert np.array_equal(y_seq, y_f) and np.array_equal(lad_seq, lad_f)
    y0, lad0, _ = capi.rqs_coupling(x, params, tidx, spec, inverse=True)
    y_s, lad_s, _ = capi.rqs_coupling(x, params, tidx, spec, inverse=True, out_scatter=perm)
    assert np.array_equal(y0[:, np.argsort(perm)], y_s) and np.array_equal(lad0, lad_s)


# ------------------------------------------------------------------ the PyTorch-eager CPU port
def test_eager_port_is_bit_identical_to_reference(golden_dir):
    """oracle/eager.py (the cpu_baseline 'port' of bench.py) reproduces the reference's CPU
    outputs bit for bit: every functional case and whole-flow log_prob."""
    import torch
    from oracle import eager
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    for name, inv, kw in g["meta"]:
        kw = parse_kwargs(kw)
        tails = kw.pop("tails")
        x, uw, uh, ud = (torch.from_numpy(g[name + "/" + k]) for k in ("x", "uw", "uh", "ud"))
        fn = eager.rqs_unconstrained if tails == "linear" else eager.rqs_constrained
        y, lad = fn(x, uw, uh, ud, inverse=bool(int(inv)), **kw)
        assert np.array_equal(y.numpy(), g[name + "/y"], equal_nan=True), name
        assert np.array_equal(lad.numpy(), g[name + "/lad"], equal_nan=True), name
    from nflows_amd import configs
    gf = np.load(os.path.join(golden_dir, "flows.npz"))
    metas = dict((n, parse_kwargs(c)) for n, c in gf["meta"])
    for name in ("nsf_small", "nsf_d64"):
        cfg = metas[name]
        flow = configs.rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], 2, cfg["tail_bound"])
        prefix = name + "/sd/"
        flow.load_state_dict({k[len(prefix):]: torch.from_numpy(gf[k]) for k in gf.files if k.startswith(prefix)})
        with torch.no_grad():
            lp = eager.flow_log_prob(flow.eval(), torch.from_numpy(gf[name + "/x"]))
            z, lad = eager.flow_transform(flow, torch.from_numpy(gf[name + "/x"]))
            xi, ladi = eager.flow_transform(flow, torch.from_numpy(gf[name + "/noise"]), inverse=True)
        assert np.array_equal(lp.numpy(), gf[name + "/log_prob"]), name
        assert np.array_equal(z.numpy(), gf[name + "/z"]) and np.array_equal(lad.numpy(), gf[name + "/lad"]), name
        assert np.array_equal(xi.numpy(), gf[name + "/inv_x"]), name
        assert np.array_equal(ladi.numpy(), gf[name + "/inv_lad"]), name


def test_eager_port_bit_identical_on_steep_flows(golden_dir):
    """tests/golden/flows_steep.npz (round 4): flows with STEEP splines -- logits ~ N(0, 2) as after training --, forward
    and inverse of the real reference.  The weights rebuilt from seed + helpers.steepen carry the reference's checksums
    and the eager port reproduces z, logabsdet, log_prob and the inverse pass bit for bit: the GPU tests of
    tests/test_gpu_steep.py may extend the fixture's 512 rows with the port's own evaluation."""
    import torch
    from helpers import steep_flow
    from oracle import eager
    torch.set_num_threads(1)
    for name in ("steep_nsf_k8", "steep_nsf_k8_deep", "steep_nsf_k10", "steep_affine", "steep_ar_rq"):
        flow, g, cfg = steep_flow(golden_dir, name)
        with torch.no_grad():
            z, lad = eager.flow_transform(flow, torch.from_numpy(g[name + "/x"]))
            lp = eager.flow_log_prob(flow, torch.from_numpy(g[name + "/x"]))
            xi, ladi = eager.flow_transform(flow, torch.from_numpy(g[name + "/noise"]), inverse=True)
        for got, key in ((z, "z"), (lad, "lad"), (lp, "log_prob"), (xi, "inv_x"), (ladi, "inv_lad")):
            assert np.array_equal(got.numpy(), g[name + "/" + key]), (name, key)
        if "logit_std_wh_d_per_layer" in cfg:   # the fixture is what it claims to be (seed-0 weights give ~ 0.03)
            assert min(min(pair) for pair in cfg["logit_std_wh_d_per_layer"]) > (0.6 if name.endswith("deep") else 1.8), name


BIN_COUNTS = (2, 3, 4, 5, 6, 7, 9, 11, 12, 13, 16, 20, 24, 32)


def test_eager_port_bit_identical_on_other_bin_counts(golden_dir):
    """tests/golden/flows_bins.npz (round 4): steep two-layer coupling flows with 2 .. 16 bins (the counts the
    whole-layer kernels serve besides 8 and 10), forward and inverse of the real reference: the eager port reproduces
    every vector bit for bit, so tests/test_gpu_bins.py may extend the fixture's rows with the port's evaluation."""
    import torch
    from helpers import steep_flow
    from oracle import eager
    torch.set_num_threads(1)
    for K in BIN_COUNTS:
        name = "bins_k%d" % K
        flow, g, cfg = steep_flow(golden_dir, name, "flows_bins.npz")
        assert cfg["K"] == K
        with torch.no_grad():
            z, lad = eager.flow_transform(flow, torch.from_numpy(g[name + "/x"]))
            lp = eager.flow_log_prob(flow, torch.from_numpy(g[name + "/x"]))
            xi, ladi = eager.flow_transform(flow, torch.from_numpy(g[name + "/noise"]), inverse=True)
        for got, key in ((z, "z"), (lad, "lad"), (lp, "log_prob"), (xi, "inv_x"), (ladi, "inv_lad")):
            assert np.array_equal(got.numpy(), g[name + "/" + key]), (name, key)
        assert min(min(pair) for pair in cfg["logit_std_wh_d_per_layer"]) > 1.5, (name, cfg["logit_std_wh_d_per_layer"])


def test_eager_port_bit_identical_on_the_trained_flow(golden_dir):
    """tests/golden/flows_trained.npz (round 4): six coupling layers TRAINED with the reference for 400 Adam steps on a
    multimodal, skewed 16-dimensional density (loss 16.6 -> -4.6; derivative logits spread to N(0, 1 .. 4)).  The
    reference's state_dict loads strictly into the drop-in classes and the eager port reproduces the reference's
    forward and inverse vectors bit for bit."""
    import torch
    from helpers import trained_flow
    from oracle import eager
    torch.set_num_threads(1)
    flow, g, cfg = trained_flow(golden_dir)
    name = "trained_nsf"
    assert cfg["loss_last"] < cfg["loss_first"] - 15 and max(b for _, b in cfg["logit_std_wh_d_per_layer"]) > 3.0
    with torch.no_grad():
        z, lad = eager.flow_transform(flow, torch.from_numpy(g[name + "/x"]))
        lp = eager.flow_log_prob(flow, torch.from_numpy(g[name + "/x"]))
        xi, ladi = eager.flow_transform(flow, torch.from_numpy(g[name + "/noise"]), inverse=True)
    for got, key in ((z, "z"), (lad, "lad"), (lp, "log_prob"), (xi, "inv_x"), (ladi, "inv_lad")):
        assert np.array_equal(got.numpy(), g[name + "/" + key]), key


def test_eager_port_bit_identical_on_other_activations(golden_dir):
    """tests/golden/flows_acts.npz (round 4): conditioners built with F.leaky_relu / F.elu / torch.tanh
    (nn/nets/resnet.py:27), steep two-layer flows, forward and inverse of the real reference; the eager port -- which
    runs the flow's own conditioner modules -- reproduces every vector bit for bit."""
    import torch
    from helpers import steep_flow
    from oracle import eager
    torch.set_num_threads(1)
    for name in ("act_leaky_relu_k8", "act_elu_k8", "act_tanh_k8", "act_elu_k10", "act_tanh_k10"):
        flow, g, cfg = steep_flow(golden_dir, name, "flows_acts.npz")
        assert flow._transform._transforms[1].transform_net.blocks[0].activation is not torch.nn.functional.relu
        with torch.no_grad():
            z, lad = eager.flow_transform(flow, torch.from_numpy(g[name + "/x"]))
            lp = eager.flow_log_prob(flow, torch.from_numpy(g[name + "/x"]))
            xi, ladi = eager.flow_transform(flow, torch.from_numpy(g[name + "/noise"]), inverse=True)
        for got, key in ((z, "z"), (lad, "lad"), (lp, "log_prob"), (xi, "inv_x"), (ladi, "inv_lad")):
            assert np.array_equal(got.numpy(), g[name + "/" + key]), (name, key)


def test_eager_port_other_configs_bit_identical(golden_dir):
    """The eager port on the affine stack (configs[1]'s layer type) and on the autoregressive RQ layer
    (configs[4]) in both directions -- the latter's inverse is the reference's D-pass loop -- : bit-identical to the reference, in float32
    and float64 (the float64 evaluation of the port is the ground truth of the GPU parity tests)."""
    import torch
    from oracle import eager
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_flows import build
    gf = np.load(os.path.join(golden_dir, "flows.npz"))
    metas = dict((n, parse_kwargs(c)) for n, c in gf["meta"])
    for name in ("affine_small", "ar_rq_small", "nsf_d64"):
        flow = build(metas[name])
        prefix = name + "/sd/"
        flow.load_state_dict({k[len(prefix):]: torch.from_numpy(gf[k]) for k in gf.files if k.startswith(prefix)})
        flow = flow.eval()
        x, noise = torch.from_numpy(gf[name + "/x"]), torch.from_numpy(gf[name + "/noise"])
        with torch.no_grad():
            z, lad = eager.flow_transform(flow, x)
            lp = eager.flow_log_prob(flow, x)
            assert np.array_equal(z.numpy(), gf[name + "/z"]), name
            assert np.array_equal(lad.numpy(), gf[name + "/lad"]), name
            assert np.array_equal(lp.numpy(), gf[name + "/log_prob"]), name
            # (ar_rq_small: the reference's D-pass inverse loop, autoregressive.py:43-52)
            xi, ladi = eager.flow_transform(flow, noise, inverse=True)
            assert np.array_equal(xi.numpy(), gf[name + "/inv_x"]), name
            assert np.array_equal(ladi.numpy(), gf[name + "/inv_lad"]), name
            flow64 = flow.double()
            z64, lad64 = eager.flow_transform(flow64, x.double())
            assert np.abs(z64.numpy() - gf[name + "/z64"]).max() <= 1e-12, name
            assert np.abs(lad64.numpy() - gf[name + "/lad64"]).max() <= 1e-11, name
            xi64, ladi64 = eager.flow_transform(flow64, noise.double(), inverse=True)
            assert np.abs(xi64.numpy() - gf[name + "/inv_x64"]).max() <= 1e-12, name
            assert np.abs(ladi64.numpy() - gf[name + "/inv_lad64"]).max() <= 1e-10, name
            flow.float()


@pytest.mark.parametrize("case", [None, "ctx_k4", "ctx_k16", "ctx_k24", "ctx_elu_k10", "ctx_tanh_k8"])
def test_eager_port_with_context_is_bit_identical(golden_dir, case):
    """The eager port on a conditional flow (context embedded by a Linear, concatenated in front of every
    conditioner's initial layer, GLU gate per residual block) against the vectors the reference produced
    for it (tests/golden/flows_context.npz; round 5: other bin counts and block activations,
    flows_context_more.npz): bit-identical in float32, 1e-12 in float64."""
    import torch
    from helpers import golden_conditional_flow
    from oracle import eager
    torch.set_num_threads(1)
    flow, g, name = golden_conditional_flow(golden_dir, case)
    x, noise, ctx = (torch.from_numpy(g[name + "/" + k]) for k in ("x", "noise", "context"))
    with torch.no_grad():
        emb = flow._embedding_net(ctx)
        z, lad = eager.flow_transform(flow, x, context=emb)
        lp = eager.flow_log_prob(flow, x, context=ctx)
        xi, ladi = eager.flow_transform(flow, noise, inverse=True, context=emb)
        for got, key in ((z, "z"), (lad, "lad"), (lp, "log_prob"), (xi, "inv_x"), (ladi, "inv_lad")):
            assert np.array_equal(got.numpy(), g[name + "/" + key]), key
        flow64 = flow.double()
        z64, lad64 = eager.flow_transform(flow64, x.double(), context=flow64._embedding_net(ctx.double()))
        assert np.abs(z64.numpy() - g[name + "/z64"]).max() <= 1e-12
        assert np.abs(lad64.numpy() - g[name + "/lad64"]).max() <= 1e-11
        flow.float()


@pytest.mark.parametrize("case", ["realnvp_affine", "realnvp_additive", "realnvp_h64_d22", "realnvp_d64_b3", "realnvp_d80"])
def test_eager_port_on_the_reference_realnvp_is_bit_identical(golden_dir, case):
    """The reference's SimpleRealNVP (flows/realnvp.py:17-71: affine / additive couplings with ResidualNet conditioners on a
    flipping +-1 mask), built by the factory itself for tests/golden/flows_realnvp.npz and rebuilt here from the seed
    (same state_dict keys, same checksums): the eager port reproduces the reference's fp32 vectors bit for bit and its
    float64 ones to 1e-12 -- the yardstick of tests/test_gpu_realnvp.py."""
    import torch
    from helpers import golden_realnvp_flow
    from oracle import eager
    torch.set_num_threads(1)
    flow, g, cfg = golden_realnvp_flow(golden_dir, case)
    x, noise = (torch.from_numpy(g[case + "/" + k]) for k in ("x", "noise"))
    with torch.no_grad():
        z, lad = eager.flow_transform(flow, x)
        lp = eager.flow_log_prob(flow, x)
        xi, ladi = eager.flow_transform(flow, noise, inverse=True)
        for got, key in ((z, "z"), (lad, "lad"), (lp, "log_prob"), (xi, "inv_x"), (ladi, "inv_lad")):
            assert np.array_equal(got.numpy(), g[case + "/" + key]), key
        flow64 = flow.double()
        z64, lad64 = eager.flow_transform(flow64, x.double())
        assert np.abs(z64.numpy() - g[case + "/z64"]).max() <= 1e-12
        assert np.abs(lad64.numpy() - g[case + "/lad64"]).max() <= 1e-11
        flow.float()


def _h128_flow(golden_dir):
    """The flow of tests/golden/flows_h128.npz rebuilt from its seed (weights are not stored)."""
    import torch
    from nflows_amd import configs
    g = np.load(os.path.join(golden_dir, "flows_h128.npz"))
    name, cfg = g["meta"][0]
    cfg = parse_kwargs(cfg)
    flow = configs.rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], 2, cfg["tail_bound"], seed=cfg["seed"])
    with torch.no_grad():
        for n_, p in flow.named_parameters():
            if "final_layer" in n_:
                p.mul_(cfg["scale_final"])
            elif "linear_layers.1" in n_:
                p.mul_(cfg["scale_linear1"])
    sd = flow.state_dict()
    assert list(sd.keys()) == list(g[name + "/param_names"])
    sums = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
    assert np.array_equal(sums, g[name + "/param_checksums"]), "seeded weights differ from the reference's"
    return flow.eval(), g, name


def test_eager_port_bit_identical_at_baseline_layer_shape(golden_dir):
    """Same as above on the BASELINE layer shape (D = 64, K = 8, ResidualNet H = 128): the weights
    rebuilt from the seed equal the reference's, and the eager port's log_prob is bit-identical."""
    import torch
    from oracle import eager
    torch.set_num_threads(1)
    flow, g, name = _h128_flow(golden_dir)
    with torch.no_grad():
        lp = eager.flow_log_prob(flow, torch.from_numpy(g[name + "/x"]))
    assert np.array_equal(lp.numpy(), g[name + "/log_prob"])


def sibling_cases(golden_dir):
    """(name, kind, oracle function, logits, spec, reference arrays) for tests/golden/splines_lq.npz"""
    g = np.load(os.path.join(golden_dir, "splines_lq.npz"))
    for name, kind, kw in g["meta"]:
        kw = parse_kwargs(kw)
        logits = [g["%s/logits%d" % (name, i)] for i in range(1 if kind == "linear" else 2)]
        K = logits[0].shape[-1]
        spec_kw = dict(kw)
        if spec_kw.get("tails") != "linear":
            spec_kw["tails"] = None
        yield str(name), str(kind), logits, K, spec_kw, g


def test_linear_and_quadratic_splines_match_reference(golden_dir):
    """oracle/nfa_oracle.c linear_one / quadratic_one against the real reference: the double build
    to 1e-10 of its float64 results, the float build within the fp32 parity model."""
    for name, kind, logits, K, kw, g in sibling_cases(golden_dir):
        x = g[name + "/x"]
        for inverse in (False, True):
            pre = name + ("/inv_" if inverse else "/")
            for dt in (np.float64, np.float32):
                spec = capi.make_spec(K, **kw)
                args = [a.astype(dt) for a in logits]
                fn = capi.linear_spline if kind == "linear" else capi.quadratic_spline
                y, lad, st = fn(x.astype(dt), *args, spec, inverse=inverse)
                assert st == 0, (name, inverse)
                if dt is np.float64:
                    ry, rl = g[pre + "y64"], g[pre + "lad64"]
                    assert np.array_equal(np.isnan(y), np.isnan(ry)), name
                    fin = np.isfinite(ry)
                    assert np.abs(y[fin] - ry[fin]).max() <= 1e-9, (name, inverse)
                    assert np.abs(lad[fin] - rl[fin]).max() <= 1e-9, (name, inverse)
                else:
                    assert_sibling_spline_parity(y, g[pre + "y"], g[pre + "y64"], OUT_TOL, 5e-5, name + " y")
                    assert_sibling_spline_parity(lad, g[pre + "lad"], g[pre + "lad64"], LAD_TOL, 1e-3, name + " lad")
                    if kw.get("tails") == "linear":
                        tb = np.float32(kw["tail_bound"])
                        outside = ~((x >= -tb) & (x <= tb))
                        assert np.array_equal(y[outside].view(np.uint32), x[outside].view(np.uint32)), name
                        assert np.all(lad[outside] == 0), name


def test_cubic_spline_matches_reference(golden_dir):
    """oracle/nfa_oracle.c cubic_one against the real reference (tests/golden/splines_cubic.npz)."""
    g = np.load(os.path.join(golden_dir, "splines_cubic.npz"))
    for name, kind, kw in g["meta"]:
        kw = parse_kwargs(kw)
        x = g[name + "/x"]
        logits = [g["%s/logits%d" % (name, i)] for i in range(4)]
        K = logits[0].shape[-1]
        spec_kw = dict(kw)
        if spec_kw.get("tails") != "linear":
            spec_kw["tails"] = None
        for inverse in (False, True):
            pre = name + ("/inv_" if inverse else "/")
            for dt in (np.float64, np.float32):
                spec = capi.make_spec(K, **spec_kw)
                y, lad, st = capi.cubic_spline(x.astype(dt), *[a.astype(dt) for a in logits], spec, inverse=inverse)
                assert st == 0, (name, inverse)
                if dt is np.float64:
                    ry, rl = g[pre + "y64"], g[pre + "lad64"]
                    assert np.array_equal(np.isnan(y), np.isnan(ry)), name
                    fin = np.isfinite(ry) & np.isfinite(rl)
                    assert np.abs(y[fin] - ry[fin]).max() <= 1e-9, (name, inverse)
                    assert np.abs(lad[fin] - rl[fin]).max() <= 1e-8, (name, inverse)
                else:
                    assert_sibling_spline_parity(y, g[pre + "y"], g[pre + "y64"], OUT_TOL, 5e-5, name + " y")
                    assert_sibling_spline_parity(lad, g[pre + "lad"], g[pre + "lad64"], LAD_TOL, 1e-3, name + " lad")
