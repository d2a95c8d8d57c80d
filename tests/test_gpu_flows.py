"""Whole-flow parity on the GPU: the drop-in classes, loaded with the reference's weights
(state_dict keys are identical), against reference outputs stored in tests/golden/flows.npz.
Run with `-m gpu`."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import parse_kwargs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg):
    from nflows_amd import configs
    from nflows_amd.nn.nets import MLP
    from nflows_amd.transforms import AffineCouplingTransform, CompositeTransform, ReversePermutation
    from nflows_amd.flows import Flow
    from nflows_amd.distributions import StandardNormal
    from nflows_amd.utils import create_alternating_binary_mask
    kind = cfg["kind"]
    if kind == "rq_nsf":
        return configs.rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], 2, cfg["tail_bound"])
    if kind == "affine":
        class Wrapped(torch.nn.Module):  # same parameter names as the fixture's wrapper module
            def __init__(self, i, o, hidden):
                super().__init__()
                self.mlp = MLP([i], [o], hidden)

            def forward(self, x, context=None):
                return self.mlp(x)
        layers = []
        for i in range(cfg["L"]):
            layers.append(AffineCouplingTransform(
                create_alternating_binary_mask(cfg["D"], even=(i % 2 == 0)),
                lambda i_, o_: Wrapped(i_, o_, cfg["hidden"])))
            layers.append(ReversePermutation(cfg["D"]))
        return Flow(CompositeTransform(layers), StandardNormal([cfg["D"]]))
    if kind == "maf":
        return configs.moons_maf_flow(cfg["L"], cfg["D"], cfg["H"])
    if kind == "ar_rq":
        return configs.ar_rq_flow(cfg["D"], cfg["H"], cfg["K"], cfg["tail_bound"], cfg["num_blocks"])
    raise KeyError(kind)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "flows.npz"))


def load_state(flow, g, name):
    prefix = name + "/sd/"
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    missing, unexpected = flow.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def check(got, ref32, ref64, what, tol):
    """Reference vectors with a float64 truth: the maximum within 4 x the reference-fp32's own error + tol x scale, and
    (round 4) the mean and the 99.9 % quantile within 2 x (helpers.assert_error_ratio) -- a maximum alone is set by the
    reference's worst ill-conditioned element and let a mis-scaled Newton step through for two rounds."""
    from helpers import assert_error_ratio
    got = got.detach().cpu().numpy()
    scale = 1 + np.abs(ref64).max()
    assert_error_ratio(got, ref32, ref64, what, factor=2.0, max_factor=4.0, max_floor=tol * scale)


@pytest.mark.parametrize("fuse", [True, False])
def test_golden_flows(golden, fuse):
    for name, cfg in golden["meta"]:
        cfg = parse_kwargs(cfg)
        flow = build(cfg)
        load_state(flow, golden, name)
        flow._transform.fuse_permutations = fuse
        flow = flow.to(DEV).eval()
        x = torch.from_numpy(golden[name + "/x"]).to(DEV)
        noise = torch.from_numpy(golden[name + "/noise"]).to(DEV)
        with torch.no_grad():
            lp = flow.log_prob(x)
            z, lad = flow._transform(x)
            xs, lad_inv = flow._transform.inverse(noise)
        import nflows_amd
        nflows_amd.check_status()
        d = cfg.get("D", 2)
        check(z, golden[name + "/z"], golden[name + "/z64"], name + " z", 3e-6)
        check(lad, golden[name + "/lad"], golden[name + "/lad64"], name + " lad", 3e-6 * d)
        check(lp, golden[name + "/log_prob"], golden[name + "/log_prob64"], name + " log_prob", 3e-6 * d)
        check(xs, golden[name + "/inv_x"], golden[name + "/inv_x64"], name + " inv_x", 3e-6)
        check(lad_inv, golden[name + "/inv_lad"], golden[name + "/inv_lad64"], name + " inv_lad", 3e-6 * d)


def test_fused_permutation_is_bit_identical(golden, monkeypatch):
    """Folding a column permutation into the neighbouring layer kernel (K1) only changes which column the
    kernel reads / writes: bit-identical to running the two transforms one after the other.  (The
    whole-layer kernels are switched off here: a run of layers in one launch adds the layers'
    log-determinants in another order than layer-by-layer launches do.)"""
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    monkeypatch.setattr(RQ, "fuse_conditioner", False)
    name = "nsf_d64"
    cfg = parse_kwargs(dict((n, c) for n, c in golden["meta"])[name])
    flow = build(cfg)
    load_state(flow, golden, name)
    flow = flow.to(DEV).eval()
    x = torch.from_numpy(golden[name + "/x"]).to(DEV)
    with torch.no_grad():
        flow._transform.fuse_permutations = True
        z1, l1 = flow._transform(x)
        x1, li1 = flow._transform.inverse(x)
        flow._transform.fuse_permutations = False
        z2, l2 = flow._transform(x)
        x2, li2 = flow._transform.inverse(x)
    assert torch.equal(z1, z2) and torch.equal(l1, l2)
    assert torch.equal(x1, x2) and torch.equal(li1, li2)


def test_same_seed_same_weights_as_reference(golden):
    """The conditioners consume the RNG exactly like the reference's, so a user switching
    frameworks gets the same initial model from the same seed (fixture built with seed 0)."""
    name = "nsf_d64"
    cfg = parse_kwargs(dict((n, c) for n, c in golden["meta"])[name])
    flow = build(cfg)  # seed 0 inside
    sd = flow.state_dict()
    k = "_transform._transforms.0._permutation"
    assert np.array_equal(sd[k].numpy(), golden[name + "/sd/" + k])
    k = "_transform._transforms.1.transform_net.initial_layer.weight"
    assert np.array_equal(sd[k].numpy(), golden[name + "/sd/" + k])


def test_forward_inverse_consistency_and_sampling():
    """reference tests/transforms/coupling_test.py:237-254 (eps 1e-3) and flows/base_test.py:54-69."""
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=4, features=64, num_bins=8, hidden_features=128).to(DEV).eval()
    with torch.no_grad():
        x = torch.randn(4096, 64, device=DEV)
        z, lad = flow._transform(x)
        xr, lad_inv = flow._transform.inverse(z)
        assert (xr - x).abs().max().item() < 1e-4
        assert (lad + lad_inv).abs().max().item() < 1e-3
        samples, lp = flow.sample_and_log_prob(512)
        assert samples.shape == (512, 64) and lp.shape == (512,)
        assert (flow.log_prob(samples) - lp).abs().max().item() < 1e-3
        s2 = flow.sample(100, batch_size=30)
        assert s2.shape == (100, 64)
        noise = flow.transform_to_noise(x)
        assert torch.equal(noise, z)


def test_affine_real_nvp_stack():
    """configs[1] shape: 8 affine coupling layers, D=32, MLP conditioner, batch 16384."""
    from nflows_amd import configs
    flow = configs.affine_coupling_flow(8, 32, (128, 128)).to(DEV).eval()
    with torch.no_grad():
        x = torch.randn(16384, 32, device=DEV)
        z, lad = flow._transform(x)
        xr, lad_inv = flow._transform.inverse(z)
        assert (xr - x).abs().max().item() < 1e-5
        assert (lad + lad_inv).abs().max().item() < 1e-4
        assert flow.log_prob(x).shape == (16384,)


def test_grad_mode_uses_the_backward_kernels():
    """Parameters require grad and grad mode is on: log_prob builds a graph (tests/test_gpu_grads.py
    checks the gradients); under no_grad the inference path runs."""
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=1, features=8, num_bins=4, hidden_features=16).to(DEV)
    x = torch.randn(16, 8, device=DEV)
    lp = flow.log_prob(x)
    assert lp.requires_grad
    with torch.no_grad():
        lp2 = flow.log_prob(x)
    assert not lp2.requires_grad and (lp - lp2).abs().max().item() < 1e-5


def test_hip_graph_replay_is_bit_identical():
    """A whole log_prob / inverse pass captured into a HIP graph (nflows_amd/graphs.py): the
    library's launches are captured from PyTorch's current stream like any other kernel."""
    from nflows_amd import configs
    from nflows_amd.graphs import GraphedInverse, GraphedLogProb
    flow = configs.rq_nsf_flow(num_layers=3, features=64, num_bins=8, hidden_features=32).to(DEV).eval()
    x = torch.randn(2048, 64, device=DEV)
    g = GraphedLogProb(flow, x)
    gi = GraphedInverse(flow, x)
    for _ in range(2):
        x2 = torch.randn(2048, 64, device=DEV)
        with torch.no_grad():
            want = flow.log_prob(x2)
            wx, wl = flow._transform.inverse(x2)
        assert torch.equal(g(x2), want)
        gx, gl = gi(x2)
        assert torch.equal(gx, wx) and torch.equal(gl, wl)
    with pytest.raises(ValueError):
        g(x[:10])


def test_shared_parameter_cdf_and_unconditional_transform(golden_dir):
    """K6 (PiecewiseRationalQuadraticCDF, nonlinearities.py:386-467) and the spline coupling layer
    with apply_unconditional_transform=True (coupling.py:524-535) against reference fixtures."""
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import PiecewiseRationalQuadraticCDF, PiecewiseRationalQuadraticCouplingTransform
    from nflows_amd.utils import create_alternating_binary_mask
    g = np.load(os.path.join(golden_dir, "cdf.npz"))
    for name, cfg in g["meta"]:
        cfg = parse_kwargs(cfg)
        if name == "coupling_uncond":
            t = PiecewiseRationalQuadraticCouplingTransform(
                create_alternating_binary_mask(cfg["D"]), lambda i, o: ResidualNet(i, o, hidden_features=cfg["hidden"]),
                num_bins=cfg["K"], tails="linear", tail_bound=3.0, apply_unconditional_transform=True)
        else:
            t = PiecewiseRationalQuadraticCDF([cfg["F"]], num_bins=cfg["K"], tails=cfg["tails"],
                                              tail_bound=cfg["tail_bound"])
        prefix = name + "/sd/"
        t.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)})
        t = t.to(DEV).eval()
        x = torch.from_numpy(g[name + "/x"]).to(DEV)
        with torch.no_grad():
            for direction, fn in (("fwd", t.forward), ("inv", t.inverse)):
                y, lad = fn(x)
                check(y, g["%s/%s_y" % (name, direction)], g["%s/%s_y64" % (name, direction)], name + direction + " y", 3e-6)
                check(lad, g["%s/%s_lad" % (name, direction)], g["%s/%s_lad64" % (name, direction)],
                      name + direction + " lad", 1e-5)
        # grad mode goes through the differentiable functional and agrees with the kernel
        y2, lad2 = t.forward(x)
        assert y2.requires_grad and (y2 - t.forward(x)[0]).abs().max().item() == 0
        with torch.no_grad():
            y3, lad3 = t.forward(x)
        # (two different kernels: 1-ulp differences in the knots are amplified on steep bins)
        assert (y2 - y3).abs().max().item() < 2e-4 and (lad2 - lad3).abs().max().item() < 2e-3
        (y2.sum() + lad2.sum()).backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in t.parameters())
    import nflows_amd
    nflows_amd.check_status()


def test_columnwise_autoregressive_inverse_equals_reference_loop(golden):
    """f2: the O(D) column-wise inverse gives the reference loop's result (and the reference's
    fixtures), for the affine MAF and the autoregressive spline."""
    for name in ("moons_maf", "ar_rq_small"):
        cfg = parse_kwargs(dict((n, c) for n, c in golden["meta"])[name])
        flow = build(cfg)
        load_state(flow, golden, name)
        flow = flow.to(DEV).eval()
        noise = torch.from_numpy(golden[name + "/noise"]).to(DEV)
        with torch.no_grad():
            xs, lad = flow._transform.inverse(noise)                      # column-wise (default)
            for t in flow._transform._transforms:
                if hasattr(t, "columnwise_inverse"):
                    t.columnwise_inverse = False
            xr, ladr = flow._transform.inverse(noise)                     # reference loop
        d = cfg.get("D", 2)
        assert (xs - xr).abs().max().item() <= 2e-5 * (1 + xr.abs().max().item())
        assert (lad - ladr).abs().max().item() <= 2e-5 * d
        check(xs, golden[name + "/inv_x"], golden[name + "/inv_x64"], name + " inv_x", 3e-6)
        check(lad, golden[name + "/inv_lad"], golden[name + "/inv_lad64"], name + " inv_lad", 3e-6 * d)


def test_conditional_flow_with_context():
    """Context flows to the conditioner (resnet.py:92-100 concatenation + GLU gate) and through
    Flow._sample's merge/split of leading dims (flows/base.py:62-73), reference
    tests/flows/base_test.py:13-69 shapes."""
    from nflows_amd.distributions import StandardNormal
    from nflows_amd.flows import Flow
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import (CompositeTransform, PiecewiseRationalQuadraticCouplingTransform,
                                       RandomPermutation)
    from nflows_amd.utils import create_alternating_binary_mask
    torch.manual_seed(3)
    D, C = 6, 4
    layers = []
    for i in range(3):
        layers.append(RandomPermutation(D))
        layers.append(PiecewiseRationalQuadraticCouplingTransform(
            create_alternating_binary_mask(D, even=(i % 2 == 0)),
            lambda i_, o_: ResidualNet(i_, o_, hidden_features=32, context_features=8),
            num_bins=8, tails="linear", tail_bound=3.0))
    flow = Flow(CompositeTransform(layers), StandardNormal([D]), embedding_net=torch.nn.Linear(C, 8)).to(DEV).eval()
    x = torch.randn(50, D, device=DEV)
    ctx = torch.randn(50, C, device=DEV)
    with torch.no_grad():
        lp = flow.log_prob(x, context=ctx)
        assert lp.shape == (50,) and torch.isfinite(lp).all()
        lp_other = flow.log_prob(x, context=ctx.roll(1, 0))
        assert (lp - lp_other).abs().max().item() > 1e-4  # the context matters
        s = flow.sample(7, context=ctx[:5])
        assert s.shape == (5, 7, D)
        s2, lp2 = flow.sample_and_log_prob(7, context=ctx[:5])
        assert s2.shape == (5, 7, D) and lp2.shape == (5, 7)
        again = flow.log_prob(s2.reshape(35, D), context=ctx[:5].repeat_interleave(7, 0)).reshape(5, 7)
        assert (again - lp2).abs().max().item() < 1e-3
        noise = flow.transform_to_noise(x, context=ctx)
        assert noise.shape == x.shape
    with pytest.raises(ValueError):
        flow.log_prob(x, context=ctx[:10])


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_conditional_flow_against_reference_vectors(monkeypatch, golden_dir, engine):
    """A conditional flow at the whole-layer kernels' layer shape (H = 128, context embedded by a Linear)
    against the vectors the real reference produced for it (tests/golden/flows_context.npz; the eager
    oracle is pinned to the same vectors bit for bit in tests/test_oracle_golden.py): log_prob, the
    transform in both directions and the log-determinants, each within 4 x the reference-fp32's own
    error against its float64 result."""
    from helpers import golden_conditional_flow
    import nflows_amd
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ_
    monkeypatch.setattr(RQ_, "conditioner_engine", engine)
    flow, g, name = golden_conditional_flow(golden_dir)
    flow = flow.to(DEV)
    x, noise, ctx = (torch.from_numpy(g[name + "/" + k]).to(DEV) for k in ("x", "noise", "context"))
    with torch.no_grad():
        emb = flow._embedding_net(ctx)
        # the whole conditional flow is one run of the whole-layer kernel (K8h / K8 with a context)
        units, after = flow._transform._collect_run(list(flow._transform._transforms), 0, x, emb, inverse=False)
        assert len(units) == 3 and after == 6
        lp = flow.log_prob(x, context=ctx)
        z, lad = flow._transform(x, context=emb)
        xs, lad_inv = flow._transform.inverse(noise, context=emb)
        # ... and equals the layer-by-layer path (PyTorch conditioners + the spline kernel) to fp32 rounding
        from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
        try:
            RQ.fuse_conditioner = False
            z2, lad2 = flow._transform(x, context=emb)
            lp_ragged = flow.log_prob(x[:200], context=ctx[:200])
        finally:
            RQ.fuse_conditioner = True
        assert (z - z2).abs().max().item() < 2e-4 and (lad - lad2).abs().max().item() < 2e-3
        lp_ragged_fused = flow.log_prob(x[:200], context=ctx[:200])     # 200 rows padded to two full blocks
        assert (lp_ragged - lp_ragged_fused).abs().max().item() < 2e-3
    nflows_amd.check_status()
    d = x.shape[1]
    check(z, g[name + "/z"], g[name + "/z64"], "z", 3e-6)
    check(lad, g[name + "/lad"], g[name + "/lad64"], "lad", 3e-6 * d)
    check(lp, g[name + "/log_prob"], g[name + "/log_prob64"], "log_prob", 3e-6 * d)
    check(xs, g[name + "/inv_x"], g[name + "/inv_x64"], "inv_x", 3e-6)
    check(lad_inv, g[name + "/inv_lad"], g[name + "/inv_lad64"], "inv_lad", 3e-6 * d)


def _select_fused_path(path):
    """k8: whole ResidualNet in the spline kernel; k7b / k7: only the final Linear (split-bf16 /
    fp32 MFMA); none: PyTorch conditioner + K1."""
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    RQ.fuse_conditioner = path == "k8"
    RQ.fuse_final_linear = path != "none"
    RQ.final_linear_engine = "f32" if path == "k7" else "bf16x3"


@pytest.fixture
def restore_fused_path():
    yield
    _select_fused_path("k8")


@pytest.mark.parametrize("path", ["k8", "k7b", "k7"])
@pytest.mark.parametrize("B", [32, 1000, 4096])
def test_fused_conditioner_kernels_match_gemm_plus_k1(B, path, restore_fused_path):
    """K8 (whole ResidualNet conditioner inside the spline kernel, split-bf16 MFMA), K7b and K7
    (final Linear only; split-bf16 / fp32 MFMA) against the unfused path (hipBLASLt GEMMs + K1)
    and against the CPU eager port of the reference in float64: same arithmetic up to the GEMMs'
    summation order."""
    from nflows_amd import configs
    from oracle import eager
    import copy
    flow = configs.rq_nsf_flow(num_layers=3, features=64, num_bins=8, hidden_features=128, seed=5)
    with torch.no_grad():
        for n_, p in flow.named_parameters():  # non-trivial splines
            if "final_layer" in n_:
                p.mul_(4.0)
            elif "linear_layers.1" in n_:
                p.mul_(30.0)
    cpu = copy.deepcopy(flow).eval()
    flow = flow.to(DEV).eval()
    x = torch.randn(B, 64, device=DEV)
    with torch.no_grad():
        _select_fused_path(path)
        z1, l1 = flow._transform(x)
        x1, li1 = flow._transform.inverse(x)
        lp = flow.log_prob(x)
        _select_fused_path("none")
        z0, l0 = flow._transform(x)
        x0, li0 = flow._transform.inverse(x)
        cpu64 = cpu.double()
        lp_ref64 = eager.flow_log_prob(cpu64, x.cpu().double())
        _, l64 = eager.flow_transform(cpu64, x.cpu().double())
        _, li64 = eager.flow_transform(cpu64, x.cpu().double(), inverse=True)
    import nflows_amd
    nflows_amd.check_status()
    # (three sharpened layers amplify the GEMMs' different summation orders; the bulk agrees tightly)
    for got, want, tol in ((z1, z0, 2e-4), (l1, l0, 5e-3), (x1, x0, 2e-4), (li1, li0, 5e-3)):
        d = (got - want).abs()
        assert d.median().item() < tol / 50
        if tol == 2e-4:
            assert d.max().item() < tol
    # worst case of the log-determinants: the two fp32 paths differ by their own errors against float64 --
    # the fused path may be at most twice as far from the truth as the unfused one (+ 4 ulps of the value)
    for got, want, truth in ((l1, l0, l64), (li1, li0, li64)):
        e_got = (got.cpu().double() - truth).abs().max().item()
        e_want = (want.cpu().double() - truth).abs().max().item()
        assert e_got <= 2.0 * e_want + 4 * 2.0 ** -23 * truth.abs().max().item(), (e_got, e_want)
    assert (lp.cpu().double() - lp_ref64).abs().max().item() < 5e-3


@pytest.mark.parametrize("features,blocks,bins", [(16, 1, 8), (24, 3, 8), (64, 0, 8), (128, 2, 8),
                                                  (64, 2, 10), (128, 1, 10), (8, 0, 10)])
def test_whole_layer_kernel_shapes(features, blocks, bins, restore_fused_path):
    """K8 on other layer geometries (d_i = d_t = 8 .. 64), block counts,
    ragged batches, NaN / out-of-range inputs: equal to the unfused path within the GEMM noise,
    pass-through columns bit-exact, NaN pattern identical."""
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=2, features=features, num_bins=bins, hidden_features=128,
                               num_blocks=blocks, seed=11).to(DEV).eval()
    with torch.no_grad():
        assert flow._transform._transforms[1]._resnet_eligible(None)
        for n_, p in flow.named_parameters():
            if "final_layer" in n_:
                p.mul_(3.0)
            elif "linear_layers.1" in n_:
                p.mul_(20.0)
    x = torch.randn(128 * 3 + 37, features, device=DEV) * 1.5
    x[5, 0] = float("nan")
    x[6, 1] = 7.5
    clean = torch.nan_to_num(x, nan=0.25)  # (NaN parameters trip the reference's discriminant assert)
    with torch.no_grad():
        _select_fused_path("k8")
        z1, l1 = flow._transform(x)
        x1, li1 = flow._transform.inverse(clean)
        _select_fused_path("none")
        z0, l0 = flow._transform(x)
        x0, li0 = flow._transform.inverse(clean)
    import nflows_amd
    nflows_amd.check_status()
    assert torch.isnan(z0).any()
    for got, want, tol in ((z1, z0, 2e-4), (l1, l0, 5e-3), (x1, x0, 2e-4), (li1, li0, 5e-3)):
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        ok = torch.isfinite(want)
        d = (got[ok] - want[ok]).abs()
        assert d.max().item() < tol and d.median().item() < tol / 50
    layer = flow._transform._transforms[1]
    ident = layer.identity_features
    with torch.no_grad():
        _select_fused_path("k8")
        y, _ = layer(x)
    assert torch.equal(y[:, ident].isnan(), x[:, ident].isnan())
    keep = ~x[:, ident].isnan()
    assert torch.equal(y[:, ident][keep], x[:, ident][keep])


@pytest.mark.parametrize("path", ["k8", "k7b", "k7", "none"])
def test_baseline_layer_shape_against_reference_vectors(golden_dir, path, restore_fused_path):
    """tests/golden/flows_h128.npz: outputs of the REAL reference (fp32 and fp64) for a 2-layer
    RQ-NSF flow at the BASELINE layer shape (D = 64, K = 8, ResidualNet H = 128, 160 rows: one
    128-row block for the fused kernels + a ragged tail).  Every layer-kernel path must meet the
    fp32 parity model: error against the reference's fp64 result <= 4x the reference-fp32's own."""
    from test_oracle_golden import _h128_flow
    flow, g, name = _h128_flow(golden_dir)
    flow = flow.to(DEV)
    x = torch.from_numpy(g[name + "/x"]).to(DEV)
    noise = torch.from_numpy(g[name + "/noise"]).to(DEV)
    d = x.shape[1]
    _select_fused_path(path)
    with torch.no_grad():
        lp = flow.log_prob(x)
        z, lad = flow._transform(x)
        xs, lad_inv = flow._transform.inverse(noise)
    import nflows_amd
    nflows_amd.check_status()
    check(z, g[name + "/z"], g[name + "/z64"], path + " z", 3e-6)
    check(lad, g[name + "/lad"], g[name + "/lad64"], path + " lad", 3e-6 * d)
    check(lp, g[name + "/log_prob"], g[name + "/log_prob64"], path + " log_prob", 3e-6 * d)
    check(xs, g[name + "/inv_x"], g[name + "/inv_x64"], path + " inv_x", 3e-6)
    check(lad_inv, g[name + "/inv_lad"], g[name + "/inv_lad64"], path + " inv_lad", 3e-6 * d)
    # bulk agreement with the reference's fp32 output itself
    from helpers import bulk_fraction
    assert bulk_fraction(z.cpu().numpy(), g[name + "/z"], 2e-5) >= 0.99
    assert bulk_fraction(xs.cpu().numpy(), g[name + "/inv_x"], 2e-5) >= 0.99


def test_whole_layer_kernel_abi_contract(restore_fused_path):
    """K8 through the C ABI: unsupported shapes are refused before anything is enqueued, bad table
    entries are reported through the status word, a corrupted table cannot write out of bounds."""
    import ctypes
    from nflows_amd import _native as N, ops, configs
    flow = configs.rq_nsf_flow(num_layers=1, features=64, num_bins=8, hidden_features=128, seed=3).to(DEV).eval()
    layer = flow._transform._transforms[1]
    wp, bp = ops.pack_resnet_conditioner(layer.transform_net, 32, 23)
    tables = ops.coupling_layer_tables(64, layer.transform_features, layer.identity_features)
    spec = layer._spec()
    x = torch.randn(256, 64, device=DEV)
    y, lad = ops.rqs_coupling_resnet(x, wp, bp, tables, 32, 32, 2, spec)
    ops.check_status()
    assert torch.equal(y[:, layer.identity_features], x[:, layer.identity_features])
    # a ragged batch is padded to full blocks by the host wrapper (rows keep their results bit for bit);
    # the C entry point itself refuses it, and a bin count the kernel does not have
    y200, lad200 = ops.rqs_coupling_resnet(x[:200], wp, bp, tables, 32, 32, 2, spec)
    assert torch.equal(y200, y[:200]) and torch.equal(lad200, lad[:200])
    # (round 4: 2 .. 16 bins are served -- with blobs packed for that bin count; 17 and up are refused)
    spec17 = ops.make_rqs_spec(17, "linear", tail_bound=3.0)
    assert ops.rqs_coupling_resnet(x, wp, bp, tables, 32, 32, 2, spec17) is None
    lib = N.load()
    out, l2 = torch.empty_like(x), torch.empty(256, device=DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    def call(tab, flags=0, batch=256):
        return lib.nfa_rqs_coupling_resnet_f32(N.ptr(x), N.ptr(wp), N.ptr(bp), N.ptr(tab), N.ptr(out), N.ptr(l2),
                                               N.ptr(st), batch, 64, 32, 32, 128, 2, ctypes.byref(spec), flags,
                                               N.stream_handle(x.device))
    assert call(tables, flags=64) == N.ERR_INVALID_ARGUMENT
    assert call(None) == N.ERR_INVALID_ARGUMENT
    assert call(tables, batch=0) == N.OK
    assert call(tables, batch=200) == N.ERR_UNSUPPORTED
    bad = tables.clone()
    bad[64 + 5] = 64  # transformed feature 5 read from a slot outside the row
    assert call(bad) == N.OK
    torch.cuda.synchronize()
    assert st.item() & N.STATUS_BAD_INDEX


@pytest.mark.parametrize("layers", [1, 3])
def test_whole_layer_kernel_persistent_loop(layers, restore_fused_path):
    """More 128-row blocks than resident workgroups (2 per CU): every workgroup walks several
    blocks, re-starting the weight stream (and, for a run of layers, the layer tables) from the
    first layer each time.  Compared with the PyTorch conditioner + K1 path on the first, middle
    and last rows; forward and inverse."""
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=layers, features=64, num_bins=8, hidden_features=128, seed=2).to(DEV).eval()
    with torch.no_grad():
        for n_, p in flow.named_parameters():
            if "final_layer" in n_:
                p.mul_(4.0)
            elif "linear_layers.1" in n_:
                p.mul_(30.0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B = 128 * (2 * cus * 2 + 37)  # > 2 blocks per resident workgroup, not a multiple of the grid
    x = torch.randn(B, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    rows = torch.cat((torch.arange(0, 512), torch.arange(B // 2 - 256, B // 2 + 256), torch.arange(B - 512, B))).to(DEV)
    with torch.no_grad():
        _select_fused_path("k8")
        z1, l1 = flow._transform(x)
        x1, li1 = flow._transform.inverse(x)
        _select_fused_path("none")
        z0, l0 = flow._transform(x[rows])
        x0, li0 = flow._transform.inverse(x[rows])
    import nflows_amd
    nflows_amd.check_status()
    scale = layers * layers  # (sharpened layers amplify the GEMMs' different summation orders)
    for got, want, tol in ((z1[rows], z0, 2e-5), (l1[rows], l0, 5e-4), (x1[rows], x0, 2e-5), (li1[rows], li0, 5e-4)):
        d = (got - want).abs()
        assert d.max().item() < tol * scale and d.median().item() < tol, (d.max().item(), d.median().item())
    assert torch.isfinite(z1).all() and torch.isfinite(l1).all()
    # every row was written: round trip over the whole batch
    with torch.no_grad():
        _select_fused_path("k8")
        back, _ = flow._transform.inverse(z1)
    assert (back - x).abs().max().item() < 1e-4 * layers


def test_sibling_couplings_against_reference_vectors(golden_dir):
    """tests/golden/couplings_lq.npz, couplings_cubic.npz: piecewise-linear / -quadratic / -cubic
    coupling layers on [B, D] (two layers with permutations, linear tails, cases with
    apply_unconditional_transform) and spline couplings on [B, C, H, W] images with a
    ConvResidualNet conditioner, against the real reference's fp32 / fp64 outputs; reference
    state_dicts load unchanged."""
    from nflows_amd.nn.nets import ConvResidualNet, ResidualNet
    from nflows_amd.transforms import (CompositeTransform, PiecewiseCubicCouplingTransform,
                                       PiecewiseLinearCouplingTransform, PiecewiseQuadraticCouplingTransform,
                                       PiecewiseRationalQuadraticCouplingTransform, RandomPermutation)
    from nflows_amd.utils import create_alternating_binary_mask
    classes = {"linear": PiecewiseLinearCouplingTransform, "quadratic": PiecewiseQuadraticCouplingTransform,
               "quadratic_uncond": PiecewiseQuadraticCouplingTransform, "cubic": PiecewiseCubicCouplingTransform,
               "rq": PiecewiseRationalQuadraticCouplingTransform}
    files = [np.load(os.path.join(golden_dir, f)) for f in ("couplings_lq.npz", "couplings_cubic.npz")]
    for g, name, cfg in [(g_, n_, c_) for g_ in files for n_, c_ in g_["meta"]]:
        cfg = parse_kwargs(cfg)
        cls = classes[cfg["kind"]]
        if name.startswith("c2d_"):
            layers = []
            for i in range(cfg["L"]):
                layers.append(RandomPermutation(cfg["D"]))
                layers.append(cls(mask=create_alternating_binary_mask(cfg["D"], even=(i % 2 == 0)),
                                  transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=cfg["H"], num_blocks=1),
                                  num_bins=cfg["K"], tails="linear", tail_bound=cfg["tail_bound"],
                                  apply_unconditional_transform=cfg.get("apply_unconditional_transform", False)))
            t = CompositeTransform(layers)
        else:
            t = cls(mask=create_alternating_binary_mask(cfg["C"], even=True),
                    transform_net_create_fn=lambda i_, o_: ConvResidualNet(i_, o_, hidden_channels=cfg["hidden_channels"], num_blocks=1),
                    num_bins=cfg["K"], tails="linear", tail_bound=cfg["tail_bound"])
        load_state(t, g, name)
        t = t.to(DEV).eval()
        x = torch.from_numpy(g[name + "/x"]).to(DEV)
        noise = torch.from_numpy(g[name + "/noise"]).to(DEV)
        per_sample = x[0].numel()
        with torch.no_grad():
            z, lad = t(x)
            xs, lad_inv = t.inverse(noise)
        import nflows_amd
        nflows_amd.check_status()
        assert z.shape == x.shape and lad.shape == (x.shape[0],)
        check(z, g[name + "/z"], g[name + "/z64"], name + " z", 1e-5)
        check(lad, g[name + "/lad"], g[name + "/lad64"], name + " lad", 1e-5 * per_sample)
        check(xs, g[name + "/inv_x"], g[name + "/inv_x64"], name + " inv_x", 1e-5)
        check(lad_inv, g[name + "/inv_lad"], g[name + "/inv_lad64"], name + " inv_lad", 1e-5 * per_sample)
        # pass-through part bit-exact
        if not cfg.get("apply_unconditional_transform", False) and not name.startswith("c2d_"):
            ident = t.identity_features
            assert torch.equal(z[:, ident], x[:, ident])


@pytest.mark.parametrize("columnwise", [True, False])
def test_sibling_autoregressive_layers_against_reference_vectors(golden_dir, columnwise):
    """tests/golden/ar_siblings.npz: masked autoregressive layers on the linear / quadratic / cubic
    splines, forward and inverse (column-wise and the reference's D-pass loop), against the real
    reference; its state_dicts load unchanged."""
    from nflows_amd.transforms import (MaskedPiecewiseCubicAutoregressiveTransform,
                                       MaskedPiecewiseLinearAutoregressiveTransform,
                                       MaskedPiecewiseQuadraticAutoregressiveTransform)
    g = np.load(os.path.join(golden_dir, "ar_siblings.npz"))
    for name, cfg in g["meta"]:
        cfg = parse_kwargs(cfg)
        D, H, K = cfg["D"], cfg["H"], cfg["K"]
        t = {"ar_linear": lambda: MaskedPiecewiseLinearAutoregressiveTransform(K, D, H),
             "ar_quadratic": lambda: MaskedPiecewiseQuadraticAutoregressiveTransform(D, H, num_bins=K, tails="linear",
                                                                                     tail_bound=3.0),
             "ar_quadratic_box": lambda: MaskedPiecewiseQuadraticAutoregressiveTransform(D, H, num_bins=K),
             "ar_cubic": lambda: MaskedPiecewiseCubicAutoregressiveTransform(K, D, H)}[str(name)]()
        load_state(t, g, name)
        t = t.to(DEV).eval()
        t.columnwise_inverse = columnwise
        x = torch.from_numpy(g[name + "/x"]).to(DEV)
        noise = torch.from_numpy(g[name + "/noise"]).to(DEV)
        with torch.no_grad():
            z, lad = t(x)
            xs, lad_inv = t.inverse(noise)
        import nflows_amd
        nflows_amd.check_status()
        check(z, g[name + "/z"], g[name + "/z64"], name + " z", 1e-5)
        check(lad, g[name + "/lad"], g[name + "/lad64"], name + " lad", 1e-5 * D)
        check(xs, g[name + "/inv_x"], g[name + "/inv_x64"], name + " inv_x", 1e-5)
        check(lad_inv, g[name + "/inv_lad"], g[name + "/inv_lad64"], name + " inv_lad", 1e-5 * D)


@pytest.mark.parametrize("features,blocks", [(64, 2), (128, 2), (24, 1), (64, 0)])
def test_woven_final_layer_is_bit_identical_to_the_plain_loop(tmp_path, features, blocks):
    """K8 evaluates the splines between the MFMAs of its final layer (gemm_tile_pumped); with
    NFA_K8_PIPE=0 the evaluation follows the tiles as one block.  NFA_K8_PIPE=1: same operations in
    the same order, outputs and logabsdet agree bit for bit, forward and inverse, tails and NaN
    included.  (The switch is read once per process, hence the child processes.)"""
    import subprocess
    import sys
    script = tmp_path / "child.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import nflows_amd\n"
        "from nflows_amd import configs\n"
        "D, NB = int(sys.argv[2]), int(sys.argv[3])\n"
        "flow = configs.rq_nsf_flow(num_layers=5, features=D, num_bins=8, hidden_features=128, num_blocks=NB, seed=2).cuda().eval()\n"
        "x = 1.4 * torch.randn(1024, D, generator=torch.Generator().manual_seed(9)).cuda()\n"
        "x[:4, :8] = torch.tensor([3.0, -3.0, 3.5, float('nan'), 0.0, 2.9999998, -7.0, 1e-8]).cuda()\n"
        "with torch.no_grad():\n"
        "    y, lad = flow._transform(x)\n"
        "    xi, ladi = flow._transform.inverse(torch.nan_to_num(y))\n"
        "np.savez(sys.argv[1], y=y.cpu().numpy(), lad=lad.cpu().numpy(), xi=xi.cpu().numpy(), ladi=ladi.cpu().numpy())\n"
        % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for flag in ("0", "1", "2"):
        out = str(tmp_path / ("pipe%s.npz" % flag))
        subprocess.check_call([sys.executable, str(script), out, str(features), str(blocks)], env=dict(os.environ, NFA_K8_PIPE=flag))
        outs.append(np.load(out))
    for key in ("y", "lad", "xi", "ladi"):
        assert np.array_equal(outs[0][key], outs[1][key], equal_nan=True), key
    # NFA_K8_PIPE=2 (the default: woven, cheaper rounding sequence in the evaluation) is not
    # bit-identical; it stays within the noise the in-kernel GEMMs put on the logits anyway
    # (logabsdet: a sum over 5 layers x d_t features of values of order 1, i.e. a few ulp of ~30)
    for key, tol, typical in (("y", 2e-5, 1e-6), ("lad", 4e-4, 4e-5), ("xi", 2e-5, 1e-6), ("ladi", 4e-4, 4e-5)):
        a, b = outs[0][key], outs[2][key]
        assert np.array_equal(np.isnan(a), np.isnan(b)), key
        ok = ~np.isnan(a)
        d = np.abs(a[ok] - b[ok])
        assert d.max() <= tol and np.median(d) <= typical, (key, d.max(), np.median(d))


def test_whole_layer_kernel_random_geometries(restore_fused_path):
    """Seeded sweep over layer geometries the parametrised tests do not hit: random (non-alternating)
    masks with d_t a multiple of 4, d_i and d_t up to 64, 8 and 10 bins, 0..3 residual blocks, with
    and without neighbouring permutations, ragged batches.  K8 against the unfused path."""
    import nflows_amd
    from nflows_amd import transforms as T
    from nflows_amd.nn.nets import ResidualNet
    rng = np.random.RandomState(20260924)
    for case in range(12):
        D = int(rng.choice([8, 12, 20, 36, 64, 96, 128]))
        dt = int(rng.choice([v for v in range(4, min(D, 68), 4) if D - v <= 64 and D - v >= 1]))
        bins = int(rng.choice([8, 10]))
        blocks = int(rng.randint(0, 4))
        mask = np.zeros(D, dtype=np.int64)
        mask[rng.permutation(D)[:dt]] = 1
        torch.manual_seed(1000 + case)
        layers = []
        for i in range(3):
            if rng.rand() < 0.7:
                layers.append(T.RandomPermutation(D))
            m = torch.from_numpy(mask if i % 2 == 0 else np.roll(mask, 1))
            layers.append(T.PiecewiseRationalQuadraticCouplingTransform(
                m, lambda a, b, nb=blocks: ResidualNet(a, b, hidden_features=128, num_blocks=nb),
                num_bins=bins, tails="linear", tail_bound=3.0))
        t = T.CompositeTransform(layers).to(DEV).eval()
        B = int(rng.choice([128, 256 + 17, 1024, 1000]))
        x = torch.randn(B, D, device=DEV) * 1.3
        with torch.no_grad():
            _select_fused_path("k8")
            assert all(l._resnet_eligible(None) for l in layers if hasattr(l, "_resnet_eligible")), (D, dt)
            z1, l1 = t(x)
            x1, li1 = t.inverse(z1)
            _select_fused_path("none")
            z0, l0 = t(x)
            x0, li0 = t.inverse(z1)
        nflows_amd.check_status()
        what = "case %d: D=%d d_t=%d bins=%d blocks=%d B=%d" % (case, D, dt, bins, blocks, B)
        for got, want, tol in ((z1, z0, 1e-4), (l1, l0, 2e-3), (x1, x0, 1e-4), (li1, li0, 2e-3)):
            d = (got - want).abs()
            assert d.max().item() < tol and d.median().item() < tol / 30, (what, d.max().item(), d.median().item())


def test_f16_engine_hands_out_of_range_blocks_to_the_exact_kernel(monkeypatch):
    """K8h / K8s / K8c compute the conditioner on f16 pieces: a row block (128 rows; 64 under K8s's four-wave
    workgroups; 32 under K8c) in which an activation leaves the f16 range (here an identity feature of 1e6), or whose inputs
    are not finite, is not written by it but redone by the bf16x3 kernel right behind it -- those rows equal
    the bf16x3 engine's results bit for bit (NaN pattern included), every other block is the f16 engine's own
    result."""
    from nflows_amd import configs
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    import nflows_amd
    flow = configs.rq_nsf_flow(num_layers=3, features=64, num_bins=8, hidden_features=128, seed=0)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if "final_layer" in name:
                p.mul_(4.0)
            elif "linear_layers.1" in name:
                p.mul_(30.0)
    flow = flow.to(DEV).eval()
    x = torch.randn(640, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    x[130, 5] = 1.0e6
    x[300, 7] = float("nan")
    x[301, 9] = float("inf")
    results = {}
    for engine in ("f16x2", "bf16x3"):
        monkeypatch.setattr(RQ, "conditioner_engine", engine)
        with torch.no_grad():
            z, lad = flow._transform(x)
            if engine == "f16x2":
                from nflows_amd import ops
                f16_label = ops.last_layer_kernel()
            nflows_amd.check_status()
            xi, ladi = flow._transform.inverse(x)
            # (a NaN input fails the inverse's discriminant check, as it fails the reference's
            # assert, rational_quadratic.py:142: the status word reports it for either engine)
            with pytest.raises(AssertionError):
                nflows_amd.check_status()
        results[engine] = [t.cpu().numpy() for t in (z, lad, xi, ladi)]
    # the f16 engine hands over what its workgroup covers: 128 rows under K8h, 64 under K8s's four-wave form, 32 under K8c
    # (round 6: this batch's kernel) -- the blocks around row 130 (overflow) and rows 300, 301 (non-finite inputs) are the
    # exact kernel's bits, every other row is the f16 engine's own result
    gran = 32 if "rows=32" in f16_label else 64 if ("waves=4" in f16_label or "rows=64" in f16_label) else 128
    redone = np.zeros(640, dtype=bool)
    for r in (130, 300, 301):
        redone[r // gran * gran:r // gran * gran + gran] = True
    for got, want in zip(results["f16x2"], results["bf16x3"]):
        assert np.array_equal(got[redone], want[redone], equal_nan=True)
        assert np.isfinite(got[~redone]).all()
        assert np.abs(got[~redone] - want[~redone]).max() <= 2e-4 * (1 + np.abs(want[~redone]).max())
    # the two engines are different computations: somewhere outside the redone blocks they differ
    assert not np.array_equal(results["f16x2"][0][:128], results["bf16x3"][0][:128])


def test_functional_call_sees_the_supplied_weights():
    """torch.func.functional_call swaps the modules' `_parameters` entries without registering anything: the
    packed-weight caches must notice (EMA evaluation, ensembles, hypernetworks) -- the result equals that of a
    flow that really carries the other weights, and the original weights are back afterwards."""
    import copy
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=4, features=64, num_bins=8, hidden_features=128, seed=0).to(DEV).eval()
    x = torch.randn(512, 64, generator=torch.Generator().manual_seed(5)).to(DEV)
    gen = torch.Generator().manual_seed(6)
    other = {n: (p.detach().cpu() * (1.0 + 0.2 * torch.randn(p.shape, generator=gen))).to(DEV)
             for n, p in flow._transform.named_parameters()}
    twin = copy.deepcopy(flow)
    twin._transform.load_state_dict(other, strict=False)
    with torch.no_grad():
        z0, lad0 = flow._transform(x)
        z1, lad1 = torch.func.functional_call(flow._transform, other, (x,))
        zt, ladt = twin._transform(x)
        z2, lad2 = flow._transform(x)
    assert torch.equal(z1, zt) and torch.equal(lad1, ladt)
    assert torch.equal(z2, z0) and torch.equal(lad2, lad0)
    assert not torch.equal(z1, z0)


@pytest.mark.parametrize("engine", ["f16x2", "f16x3", "bf16x3", "k11"])
def test_a_write_through_data_is_served_on_the_next_call(monkeypatch, engine):
    """The whole-layer kernels read packed copies of the conditioner weights; the reference reads
    `self.transform_net`'s parameters on every call (coupling.py:85).  A write through `.data` (EMA swap,
    `dist.broadcast(p.data)`) advances no version counter -- until round 5 the kernels then kept the OLD packs until a
    periodic checksum raised, up to 255 evaluations later.  Round 6: with NFA_VERIFY_WEIGHTS at its DEFAULT, the very next
    `log_prob` after `p.data.mul_(2)` gives the new weights' result -- bit for bit what a fresh copy of the flow (fresh
    packs) gives --, nothing is raised, and calls without such a write keep their cached plan."""
    import copy
    import nflows_amd
    from nflows_amd import configs
    from nflows_amd.transforms import coupling as C
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    assert C.VERIFY_WEIGHTS_EVERY == int(os.environ.get("NFA_VERIFY_WEIGHTS", "256") or 0)
    if engine == "k11":
        flow = configs.affine_coupling_flow(6, 32, (128, 128), seed=0).to(DEV).eval()
        D = 32
    else:
        monkeypatch.setattr(RQ, "conditioner_engine", engine)
        flow = configs.rq_nsf_flow(num_layers=6, features=64, num_bins=8, hidden_features=128, seed=0).to(DEV).eval()
        D = 64
    x = torch.randn(1024, D, generator=torch.Generator().manual_seed(5)).to(DEV)
    T = flow._transform
    with torch.no_grad():
        lp0 = flow.log_prob(x)
        assert torch.equal(flow.log_prob(x), lp0)
        plans = dict(T.__dict__.get("_run_plans", {}))
        params = list(flow.parameters())
        params[-2].data.mul_(2.0)                                   # one layer's final weight, through `.data`
        lp1 = flow.log_prob(x)                                      # ... and IMMEDIATELY the new weights' result
        fresh = copy.deepcopy(flow)
        assert torch.equal(fresh.log_prob(x), lp1) and not torch.equal(lp1, lp0)
        assert torch.equal(flow.log_prob(x), lp1)
        # an EMA-style swap of every parameter
        shadow = [p.detach().clone() * 0.97 for p in params]
        for p, s_ in zip(params, shadow):
            p.data.copy_(s_)
        lp2 = flow.log_prob(x)
        assert torch.equal(copy.deepcopy(flow).log_prob(x), lp2) and not torch.equal(lp2, lp1)
        # the inverse pass plans its own run: the same contents
        z, _ = flow._transform(x)
        params[3].data.add_(0.01)
        xr, _ = flow._transform.inverse(z)
        xr_fresh, _ = copy.deepcopy(flow)._transform.inverse(z)
        assert torch.equal(xr, xr_fresh)
        # reading `.data` (logging a norm) compares contents once and changes nothing
        before = {k: v for k, v in T.__dict__.get("_run_plans", {}).items()}
        _ = float(sum(p.data.abs().sum() for p in params))
        lp3 = flow.log_prob(x)
        after = T.__dict__.get("_run_plans", {})
        assert all(after.get(k) is v for k, v in before.items() if k in after) and torch.isfinite(lp3).all()
        # a write announced by nothing (raw storage) is what NFA_VERIFY_WEIGHTS remains for
        monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 1)
        flow.log_prob(x)
        torch._C.TensorBase.data.__get__(params[-2]).mul_(1.05)
        with pytest.raises(C.StalePackedWeights):
            for _ in range(3):
                flow.log_prob(x)
        nflows_amd.invalidate_packed_weights()
        lp4 = flow.log_prob(x)
        assert torch.equal(copy.deepcopy(flow).log_prob(x), lp4)
    nflows_amd.check_status()
    del plans


def test_f16_engine_leaves_non_finite_weights_to_the_exact_kernel(monkeypatch):
    """K8h's ReLU is v_max_f32, which does not propagate a NaN the way torch.relu does: a conditioner with a
    non-finite weight (a diverged training run) must not take the f16 engine.  The packer notices
    (ops.build_f16_stream) and the run goes to the bf16x3 kernel: same bits as with that engine selected,
    NaN pattern of the reference (every transformed feature of the poisoned layer's rows)."""
    from nflows_amd import configs, ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    flow = configs.rq_nsf_flow(num_layers=3, features=64, num_bins=8, hidden_features=128, seed=0)
    with torch.no_grad():
        flow._transform._transforms[3].transform_net.blocks[0].linear_layers[0].weight[5, 7] = float("nan")
    flow = flow.to(DEV).eval()
    x = torch.randn(256, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    results = {}
    for engine in ("f16x2", "bf16x3"):
        monkeypatch.setattr(RQ, "conditioner_engine", engine)
        with torch.no_grad():
            z, lad = flow._transform(x)
            y1, lad1 = flow._transform._transforms[3](x)       # the layer alone (single-layer path)
        results[engine] = [t.cpu().numpy() for t in (z, lad, y1, lad1)]
    ops.check_status()
    for got, want in zip(results["f16x2"], results["bf16x3"]):
        assert np.array_equal(got, want, equal_nan=True)
    assert np.isnan(results["f16x2"][1]).all() and np.isnan(results["f16x2"][3]).all()


@pytest.mark.parametrize("engine,bins", [("f16x2", 8), ("bf16x3", 8), ("bf16x3", 10)])
def test_standard_normal_density_folded_into_the_last_layer(monkeypatch, engine, bins):
    """Flow.log_prob of a flow that is one run of whole-layer kernels over a StandardNormal base: the
    last layer's kernel adds the base density (normal.py:31-33) to the log-determinant and skips the
    z store (NFA_FLAG_STANDARD_NORMAL_LOG_PROB | NFA_FLAG_SKIP_OUTPUTS).  Equal to the two-step route
    -- transform, then base density + logabsdet (flows/base.py:42-49) -- to the rounding of one
    64-term fp32 sum taken in a different order; rows the f16 engine hands to the exact kernel
    (overflow, NaN, inf) included, with the reference's NaN pattern."""
    from nflows_amd import configs, ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow = configs.rq_nsf_flow(num_layers=4, features=64, num_bins=bins, hidden_features=128, seed=3).to(DEV).eval()
    x = torch.randn(1024, 64, generator=torch.Generator().manual_seed(5)).to(DEV)
    x[130, 5] = 1.0e6
    x[300, 7] = float("nan")
    x[301, 9] = float("inf")
    with torch.no_grad():
        assert flow._transform.standard_normal_log_prob(x) is not None       # the folded route is taken
        lp = flow.log_prob(x)
        z, lad = flow._transform(x)
        two_step = ops.standard_normal_log_prob(z, lad)
        lp_ragged = flow.log_prob(x[:1000])                                  # ragged batch: padded, same kernel
    lp, two_step, lp_ragged = lp.cpu().numpy(), two_step.cpu().numpy(), lp_ragged.cpu().numpy()
    assert np.array_equal(np.isnan(lp), np.isnan(two_step))
    assert np.isnan(lp[300]) and np.isnan(lp[301]) and np.isfinite(lp[130])
    fin = np.isfinite(two_step)
    assert np.abs(lp[fin] - two_step[fin]).max() <= 2e-5 * (1 + np.abs(two_step[fin]).max())
    assert np.array_equal(lp_ragged, lp[:1000], equal_nan=True)   # a row's result does not depend on the batch
    # a base that is not the standard normal, or a composite that is not one run, never takes the fold
    from nflows_amd.transforms import CompositeTransform, ReversePermutation
    mixed = CompositeTransform(list(flow._transform._transforms) + [ReversePermutation(64).to(DEV)])
    with torch.no_grad():
        assert mixed.standard_normal_log_prob(x) is None


def test_sum_count_is_the_float64_sum():
    from nflows_amd import ops, parallel
    for n in (1, 1000, 65536, 100001):
        v = (torch.randn(n, generator=torch.Generator().manual_seed(n)) * 50 - 150).to(DEV)
        got = ops.sum_count(v).cpu()
        want = v.double().sum().cpu()
        assert got[1].item() == n
        assert abs(got[0].item() - want.item()) <= 1e-12 * max(1.0, abs(want.item()))
        assert torch.equal(parallel.reduce_log_likelihood(v).cpu(), got)


@pytest.mark.parametrize("kind", ["default", "general", "additive"])
@pytest.mark.parametrize("features,hidden_sizes,permute", [(32, (128, 128), False), (32, (128,), True),
                                                           (100, (128, 128, 128), True), (12, (128, 128), False)])
def test_affine_run_in_one_kernel_matches_the_layer_by_layer_path(monkeypatch, kind, features, hidden_sizes, permute):
    """K11 (csrc/affine_mlp.hip): a run of affine / additive coupling layers with their MLP conditioners
    in one launch against the same flow evaluated layer by layer (conditioner GEMMs by the library, then
    K2).  The per-element arithmetic is identical; the conditioner outputs differ by the rounding of
    the GEMM sums (split-bf16 products, fp32 accumulation in another order)."""
    from nflows_amd.nn.nets import MLP
    from nflows_amd.transforms import (AdditiveCouplingTransform, AffineCouplingTransform, CompositeTransform,
                                       RandomPermutation, ReversePermutation)
    from nflows_amd.utils.torchutils import create_alternating_binary_mask
    from nflows_amd.flows.base import Flow
    from nflows_amd.distributions.normal import StandardNormal
    torch.manual_seed(11)
    layers = []
    for i in range(6):
        if permute:
            layers.append(RandomPermutation(features) if i % 2 else ReversePermutation(features))
        kwargs = {}
        cls = AffineCouplingTransform
        if kind == "general":
            kwargs["scale_activation"] = AffineCouplingTransform.GENERAL_SCALE_ACTIVATION
        elif kind == "additive":
            cls = AdditiveCouplingTransform
        layers.append(cls(mask=create_alternating_binary_mask(features, even=(i % 2 == 0)),
                          transform_net_create_fn=lambda a, b: MLP([a], [b], list(hidden_sizes)), **kwargs))
    flow = Flow(CompositeTransform(layers), StandardNormal([features])).to(DEV).eval()
    x = torch.randn(1000, features, generator=torch.Generator().manual_seed(3)).to(DEV)
    z_in = torch.randn(1000, features, generator=torch.Generator().manual_seed(4)).to(DEV)   # both routes invert the same rows
    coupling = [t for t in flow._transform._transforms if isinstance(t, AffineCouplingTransform)]
    results = {}
    for fused in (True, False):
        for c in coupling:
            monkeypatch.setattr(c, "fuse_conditioner", fused, raising=False)
        with torch.no_grad():
            units, _ = flow._transform._collect_run(list(flow._transform._transforms), 0, x, None, inverse=False)
            assert bool(units) == fused
            z, lad = flow._transform(x)
            xr, ladi = flow._transform.inverse(z_in)
            lp = flow.log_prob(x[:896])
            back, _ = flow._transform.inverse(z)
            assert (back - x).abs().max().item() < 1e-4
        results[fused] = [t.cpu().numpy() for t in (z, lad, xr, ladi, lp)]
    for name, got, want in zip(("z", "logabsdet", "inverse", "inverse logabsdet", "log_prob"), results[True], results[False]):
        assert np.isfinite(got).all(), name
        tol = 2e-5 * (1 + np.abs(want).max())
        assert np.abs(got - want).max() <= tol, (name, np.abs(got - want).max(), tol)
    if kind == "additive":
        assert not results[True][1].any() and not results[True][3].any()   # log-determinants exactly zero
    # (the 1000 rows run as eight full blocks, the last one padded with zero rows)


@pytest.mark.parametrize("features,hidden,blocks,residual,random_mask,bins,batch", [
    (20, 30, 5, True, False, 10, 10),      # the reference test's shape: hidden degrees cycle (several units per step)
    (20, 30, 2, False, True, 8, 37),       # feed-forward blocks, random masks
    (64, 48, 2, True, False, 8, 100),      # H < D - 1: the tail of independent features
    (784, 256, 2, True, False, 8, 64),     # BASELINE configs[4] shape
])
def test_persistent_autoregressive_inverse_equals_the_step_by_step_loop(monkeypatch, features, hidden, blocks, residual,
                                                                       random_mask, bins, batch):
    """K12 (csrc/made_inverse.hip): the sequential features of the autoregressive spline inverse in one
    persistent kernel -- hidden units evaluated once, when their last input is found -- against the
    column-wise host loop (itself equal to the reference's D-iteration loop, autoregressive.py:43-52:
    test_columnwise_autoregressive_inverse_equals_reference_loop) and against the forward pass."""
    from nflows_amd.transforms import MaskedPiecewiseRationalQuadraticAutoregressiveTransform as AR
    import nflows_amd
    torch.manual_seed(features + hidden)
    t = AR(features=features, hidden_features=hidden, num_bins=bins, tails="linear", tail_bound=3.0, num_blocks=blocks,
           use_residual_blocks=residual, random_mask=random_mask).to(DEV).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    z = (2.0 * torch.randn(batch, features, generator=torch.Generator().manual_seed(1))).to(DEV)
    results = {}
    for fused in (True, False):
        monkeypatch.setattr(AR, "fuse_sequential_inverse", fused)
        with torch.no_grad():
            assert (t._sequential_kernel(z, None, min(features, t._sequential_steps())) is not None) == fused
            x, lad = t.inverse(z)
            nflows_amd.check_status()
        results[fused] = (x, lad)
    (x, lad), (x_ref, lad_ref) = results[True], results[False]
    assert torch.isfinite(x).all() and torch.isfinite(lad).all()
    assert (x - x_ref).abs().max().item() <= 2e-5 * (1 + x_ref.abs().max().item())
    assert (lad - lad_ref).abs().max().item() <= 1e-4 * (1 + lad_ref.abs().max().item())
    with torch.no_grad():
        zz, lad_fwd = t(x)
        zz_ref, lad_fwd_ref = t(x_ref)
    # forward(inverse(z)) = z as well as the step-by-step loop manages it (scaled weights: steep bins)
    assert (zz - z).abs().max().item() <= 2 * (zz_ref - z).abs().max().item() + 1e-5
    # (with these scaled weights a 1-ulp difference in one found feature moves the forward pass's
    # log-determinant of that row by 5e-4 -- tools/k12_determinism_probe.py -- so the sum of the two
    # log-determinants is held to 1e-3 beyond what the step-by-step loop reaches)
    assert (lad + lad_fwd).abs().max().item() <= 2 * (lad_ref + lad_fwd_ref).abs().max().item() + 1e-3


@pytest.mark.parametrize("features,hidden,blocks,residual,random_mask,batch", [
    (10, 32, 1, True, False, 128),        # features not a multiple of four, one chunk
    (23, 64, 2, False, True, 1000),       # feed-forward blocks, random masks, a ragged batch
    (64, 48, 2, True, False, 100),        # H < D - 1: the inverse has a tail of independent features
    (105, 256, 2, True, False, 300),      # two chunks (26 + 2 groups of four features)
    (784, 256, 2, True, False, 512),      # BASELINE configs[4] shape: eight chunks
])
def test_made_output_layer_inside_the_spline_kernel(monkeypatch, features, hidden, blocks, residual, random_mask, batch):
    """K13 (csrc/made_output.hip): the MADE's masked output layer, the spline of every feature and the per-sample
    logabsdet sum in one kernel -- forward pass (autoregressive.py:38-41) and the last pass of the inverse
    (:43-52: the features behind the last sequential one) -- against the GEMM + spline-kernel path
    (`fuse_output_layer = False`), which the golden-vector tests tie to the reference."""
    from nflows_amd import ops
    from nflows_amd.transforms import MaskedPiecewiseRationalQuadraticAutoregressiveTransform as AR
    import nflows_amd
    torch.manual_seed(features + hidden)
    t = AR(features=features, hidden_features=hidden, num_bins=8, tails="linear", tail_bound=3.0, num_blocks=blocks,
           use_residual_blocks=residual, random_mask=random_mask).to(DEV).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = (1.5 * torch.randn(batch, features, generator=torch.Generator().manual_seed(2))).to(DEV)
    real = ops.made_output_spline
    calls = []

    def counting(*a, **k):
        r = real(*a, **k)
        calls.append(r is not None)
        return r
    monkeypatch.setattr(ops, "made_output_spline", counting)
    results = {}
    for fused in (True, False):
        monkeypatch.setattr(AR, "fuse_output_layer", fused)
        del calls[:]
        with torch.no_grad():
            y, lad = t(x)
            xi, ladi = t.inverse(x)
            y2, _ = t(x)
        nflows_amd.check_status()
        sequential = min(features, t._sequential_steps())
        assert calls == ([True, True, True] if sequential < features else [True, True]) if fused else calls == []
        assert torch.equal(y, y2)
        results[fused] = (y, lad, xi, ladi)
    for got, want, tol in zip(results[True], results[False], (2e-5, 1e-4, 2e-5, 1e-4)):
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() <= tol * (1 + want.abs().max().item()), (got - want).abs().max().item()


@pytest.mark.parametrize("hidden", [30, 64, 96])
@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_whole_layer_kernels_take_narrower_conditioners(monkeypatch, hidden, engine):
    """Conditioners narrower than the kernels' 128 hidden units run in them zero-padded (padding units have
    zero weights and biases on both sides and stay at relu(0) = 0): spline flows on both engines and an
    affine flow, against the layer-by-layer path.  The 1/sqrt(hidden) scale of the width / height logits
    (coupling.py:554-556) uses the network's own width."""
    from nflows_amd import configs
    from nflows_amd.flows.base import Flow
    from nflows_amd.distributions.normal import StandardNormal
    from nflows_amd.nn.nets import MLP
    from nflows_amd.transforms import (AffineCouplingTransform, CompositeTransform,
                                       PiecewiseRationalQuadraticCouplingTransform as RQ)
    from nflows_amd.utils.torchutils import create_alternating_binary_mask
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow = configs.rq_nsf_flow(num_layers=4, features=32, num_bins=8, hidden_features=hidden, seed=5)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if "final_layer" in name:
                p.mul_(4.0)
            elif "linear_layers.1" in name:
                p.mul_(30.0)
    flow = flow.to(DEV).eval()
    torch.manual_seed(6)
    affine = Flow(CompositeTransform([
        AffineCouplingTransform(create_alternating_binary_mask(32, even=(i % 2 == 0)),
                                lambda a, b: MLP([a], [b], [hidden, hidden])) for i in range(4)]),
        StandardNormal([32])).to(DEV).eval()
    x = torch.randn(512, 32, generator=torch.Generator().manual_seed(7)).to(DEV)
    for f, cls in ((flow, RQ), (affine, AffineCouplingTransform)):
        results = {}
        for fused in (True, False):
            monkeypatch.setattr(cls, "fuse_conditioner", fused)
            with torch.no_grad():
                units, _ = f._transform._collect_run(list(f._transform._transforms), 0, x, None, inverse=False)
                assert bool(units) == fused
                z, lad = f._transform(x)
                lp = f.log_prob(x)
                xr, _ = f._transform.inverse(z)
                assert (xr - x).abs().max().item() < 2e-4
            results[fused] = (z, lad, lp)
        for got, want in zip(results[True], results[False]):
            assert torch.isfinite(got).all()
            assert (got - want).abs().max().item() <= 5e-5 * (1 + want.abs().max().item())


def test_user_hook_overrides_run_the_reference_sequence_on_the_device():
    """A subclass that overrides one of the reference's hooks takes coupling.py:73-130 call for call (`_user_hooks`,
    transforms/coupling.py): with an override that changes nothing the layer agrees with the library class's fused
    kernels (same weights) to fp32 rounding, forward and inverse, alone and inside a CompositeTransform with
    permutations; with an override that doubles the outputs the doubling is THERE (the fused kernels would have dropped
    it)."""
    import copy
    from nflows_amd import configs
    from nflows_amd.transforms import CompositeTransform, PiecewiseRationalQuadraticCouplingTransform as RQ, AffineCouplingTransform as AC

    class SameRQ(RQ):
        def _coupling_transform_forward(self, inputs, transform_params):
            return super()._coupling_transform_forward(inputs, transform_params)

    class DoubledRQ(RQ):
        def _coupling_transform_forward(self, inputs, transform_params):
            y, lad = super()._coupling_transform_forward(inputs, transform_params)
            return 2 * y, lad + math.log(2.0) * inputs.shape[1]

    class SameAffine(AC):
        def _scale_and_shift(self, transform_params):
            return super()._scale_and_shift(transform_params)

    flow = configs.rq_nsf_flow(3, 64, 8, 128, 2, 3.0, seed=5).to("cuda:0").eval()
    x = torch.randn(4096, 64, generator=torch.Generator().manual_seed(6)).to("cuda:0")
    lib_layers = list(flow._transform._transforms)
    with torch.no_grad():
        want, want_lad = flow._transform(x)
        mine = copy.deepcopy(lib_layers)
        for t in mine:
            if isinstance(t, RQ):
                t.__class__ = SameRQ
        assert all(t._user_hooks for t in mine if isinstance(t, RQ))
        got, got_lad = CompositeTransform(mine)(x)
        assert (got - want).abs().max().item() <= 2e-5 and (got_lad - want_lad).abs().max().item() <= 2e-4
        back, back_lad = CompositeTransform(mine).inverse(want)
        ref_back, ref_back_lad = flow._transform.inverse(want)
        assert (back - ref_back).abs().max().item() <= 2e-5 and (back_lad - ref_back_lad).abs().max().item() <= 2e-4
        one, one_lad = lib_layers[1](x)
        dbl = copy.deepcopy(lib_layers[1])
        dbl.__class__ = DoubledRQ
        y2, lad2 = dbl(x)
        tf = lib_layers[1].transform_features
        idf = lib_layers[1].identity_features
        assert torch.equal(y2[:, idf], x[:, idf])
        assert (y2[:, tf] - 2 * one[:, tf]).abs().max().item() <= 2e-5
        assert (lad2 - one_lad - math.log(2.0) * tf.numel()).abs().max().item() <= 2e-4
        aff = configs.affine_coupling_flow(2, 32, (64, 64), seed=3).to("cuda:0").eval()
        xa = torch.randn(1024, 32, generator=torch.Generator().manual_seed(7)).to("cuda:0")
        wa, wa_lad = aff._transform(xa)
        mine_a = copy.deepcopy(list(aff._transform._transforms))
        for t in mine_a:
            t.__class__ = SameAffine
        ga, ga_lad = CompositeTransform(mine_a)(xa)
        assert (ga - wa).abs().max().item() <= 2e-6 and (ga_lad - wa_lad).abs().max().item() <= 2e-5


def test_device_float64_port_is_the_reference_float64(golden_dir):
    """helpers.eager_oracle's float64 half on the device -- oracle/eager.py run by stock PyTorch in float64 on cuda:0, the
    TRUTH of the whole-flow parity tests since round 5 -- against the REAL reference's float64 vectors of the steep
    fixtures (flows_steep.npz: coupling flows with 8 and 10 bins, two and four layers, the affine flow, the autoregressive
    layer; forward, log_prob and inverse): 1e-10 (float64 rounding of other aten kernels in another order).  The fp32
    half stays on the CPU and bit-identical to the reference (tests/test_oracle_golden.py)."""
    from helpers import eager_oracle, steep_flow
    for case in ("steep_nsf_k8", "steep_nsf_k8_deep", "steep_nsf_k10", "steep_affine", "steep_ar_rq"):
        flow_cpu, g, cfg = steep_flow(golden_dir, case)
        x, noise = torch.from_numpy(g[case + "/x"]), torch.from_numpy(g[case + "/noise"])
        o = eager_oracle(flow_cpu, x, noise, fp64_device="cuda:0")
        for mine, theirs in (("z", "z"), ("lad", "lad"), ("lp", "log_prob"), ("xi", "inv_x"), ("ladi", "inv_lad")):
            want = g["%s/%s64" % (case, theirs)]
            got = o[mine + "64"]
            assert got.dtype == np.float64 and got.shape == want.shape
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), fin), (case, mine)
            assert np.abs(got[fin] - want[fin]).max() <= 1e-10 * (1 + np.abs(want[fin]).max()), (case, mine, float(np.abs(got[fin] - want[fin]).max()))
            # and the CPU fp32 half is the reference's fp32 (bit for bit in the build container, tests/test_oracle_golden.py;
            # on this box's CPU aten may vectorise exp / softmax in another width: agreement, not identity, is asserted here)
            want32 = g["%s/%s" % (case, theirs)]
            f32 = np.isfinite(want32) & np.isfinite(o[mine + "32"])
            close = np.abs(o[mine + "32"][f32].astype(np.float64) - want32[f32]) <= 1e-4 * (1 + np.abs(want32[f32]))
            assert f32.mean() >= 0.999 and close.mean() >= 0.99, (case, mine, float(close.mean()))


def test_float64_flows_run_on_the_device(golden, golden_dir):
    """`.double()` flows (the reference is dtype-generic): the float64 functional kernel (K5d) + device tensor
    operations reproduce the reference's float64 results -- the spline coupling flows and the affine flow of
    tests/golden/flows.npz and the conditional flow of flows_context.npz, log_prob and both directions."""
    import nflows_amd
    from helpers import golden_conditional_flow
    for name, cfg in golden["meta"]:
        cfg = parse_kwargs(cfg)
        if cfg["kind"] not in ("rq_nsf", "affine"):
            continue
        flow = build(cfg)
        load_state(flow, golden, name)
        flow = flow.double().to(DEV).eval()
        x = torch.from_numpy(golden[name + "/x"]).double().to(DEV)
        noise = torch.from_numpy(golden[name + "/noise"]).double().to(DEV)
        with torch.no_grad():
            lp = flow.log_prob(x)
            z, lad = flow._transform(x)
            xs, lad_inv = flow._transform.inverse(noise)
        nflows_amd.check_status()
        for got, key in ((lp, "log_prob64"), (z, "z64"), (lad, "lad64"), (xs, "inv_x64"), (lad_inv, "inv_lad64")):
            assert got.dtype == torch.float64 and got.is_cuda
            ref = golden[name + "/" + key]
            assert np.abs(got.cpu().numpy() - ref).max() <= 1e-10 * (1 + np.abs(ref).max()), (name, key)
    flow, g, name = golden_conditional_flow(golden_dir)
    flow = flow.double().to(DEV)
    x, noise, ctx = (torch.from_numpy(g[name + "/" + k]).double().to(DEV) for k in ("x", "noise", "context"))
    with torch.no_grad():
        emb = flow._embedding_net(ctx)
        lp = flow.log_prob(x, context=ctx)
        z, lad = flow._transform(x, context=emb)
        xs, lad_inv = flow._transform.inverse(noise, context=emb)
    for got, key in ((lp, "log_prob64"), (z, "z64"), (lad, "lad64"), (xs, "inv_x64"), (lad_inv, "inv_lad64")):
        ref = g[name + "/" + key]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-9 * (1 + np.abs(ref).max()), key


def test_float64_functional_matches_the_reference_vectors(golden_dir):
    """K5d against the float64 results of the reference's functional on the 24 committed cases (edge values,
    NaN / inf, extreme logits, identity init, constrained boxes)."""
    from nflows_amd import ops
    g = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    for name, inv, kw in g["meta"]:
        kw = parse_kwargs(kw)
        x, uw, uh, ud = (torch.from_numpy(g[name + "/" + k]).double().to(DEV) for k in ("x", "uw", "uh", "ud"))
        spec = ops.make_rqs_spec(uw.shape[-1], **kw)
        y, lad = ops.rqs_elementwise(x, uw, uh, ud, spec, inverse=bool(int(inv)))
        ops.check_status()
        y, lad = y.cpu().numpy(), lad.cpu().numpy()
        ry, rl = g[name + "/y64"], g[name + "/lad64"]
        assert np.array_equal(np.isnan(y), np.isnan(ry)), name
        fin = np.isfinite(ry)
        assert np.abs(y[fin] - ry[fin]).max() <= 1e-10, name
        assert np.abs(lad[fin] - rl[fin]).max() <= 1e-9, name
        assert np.array_equal(y[~fin & ~np.isnan(ry)], ry[~fin & ~np.isnan(ry)]), name


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_non_finite_inputs_propagate_like_the_reference(monkeypatch, engine):
    """Rows with an infinite, NaN or huge identity / transformed feature through a run of whole-layer kernels
    against the eager oracle (bit-identical to the reference on the CPU): the same NaN pattern and the same
    infinities in z, logabsdet and log_prob; finite rows unaffected by their neighbours in the block."""
    import copy
    from nflows_amd import configs
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    from oracle import eager
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow_cpu = configs.rq_nsf_flow(num_layers=3, features=16, num_bins=8, hidden_features=128, seed=9).eval()
    flow = copy.deepcopy(flow_cpu).to(DEV)
    x = torch.randn(256, 16, generator=torch.Generator().manual_seed(10))
    x[3, 0] = float("inf")
    x[5, 1] = float("-inf")
    x[7, 2] = float("nan")
    x[9, 3] = 1.0e30
    x[140, 4] = float("inf")
    x[141, 9] = float("nan")
    with torch.no_grad():
        z_ref, lad_ref = eager.flow_transform(flow_cpu, x)
        lp_ref = eager.flow_log_prob(flow_cpu, x)
        units, _ = flow._transform._collect_run(list(flow._transform._transforms), 0, x.to(DEV), None, inverse=False)
        assert units
        z, lad = flow._transform(x.to(DEV))
        lp = flow.log_prob(x.to(DEV))
    z, lad, lp = z.cpu().numpy(), lad.cpu().numpy(), lp.cpu().numpy()
    z_ref, lad_ref, lp_ref = z_ref.numpy(), lad_ref.numpy(), lp_ref.numpy()
    for got, want, name in ((z, z_ref, "z"), (lad, lad_ref, "logabsdet"), (lp, lp_ref, "log_prob")):
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        inf = np.isinf(want)
        assert np.array_equal(got[inf], want[inf]), name
        fin = np.isfinite(want)
        assert np.abs(got[fin] - want[fin]).max() <= 1e-4 * (1 + np.abs(want[fin]).max()), name


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_non_finite_and_huge_contexts_propagate_like_the_reference(monkeypatch, engine):
    """The same for a conditional flow: non-finite inputs, a non-finite context value, and context values beyond
    the f16 range (K8h gives those row blocks up, the exact kernel with a context redoes them) against the
    eager oracle: same NaN pattern and infinities, finite rows to fp32 accuracy."""
    import copy
    from nflows_amd import configs
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    from oracle import eager
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow_cpu = configs.conditional_rq_nsf_flow(num_layers=3, features=16, num_bins=8, hidden_features=128,
                                               raw_context=5, context_features=12, seed=13).eval()
    flow = copy.deepcopy(flow_cpu).to(DEV)
    gen = torch.Generator().manual_seed(14)
    x = torch.randn(384, 16, generator=gen)
    emb = torch.randn(384, 12, generator=gen)      # the embedded context, handed to the transform directly
    x[3, 0] = float("inf")
    x[7, 2] = float("nan")
    emb[130, 4] = float("nan")
    emb[131, 5] = float("inf")
    emb[260, 1] = 3.0e5                             # beyond f16: block 2 is redone by the exact kernel
    emb[261, 7] = -1.0e9
    with torch.no_grad():
        z_ref, lad_ref = eager.flow_transform(flow_cpu, x, context=emb)
        z, lad = flow._transform(x.to(DEV), context=emb.to(DEV))
    z, lad = z.cpu().numpy(), lad.cpu().numpy()
    z_ref, lad_ref = z_ref.numpy(), lad_ref.numpy()
    for got, want, name in ((z, z_ref, "z"), (lad, lad_ref, "logabsdet")):
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        inf = np.isinf(want)
        assert np.array_equal(got[inf], want[inf]), name
        fin = np.isfinite(want)
        assert np.abs(got[fin] - want[fin]).max() <= 1e-4 * (1 + np.abs(want[fin]).max()), name
    assert np.isfinite(z_ref[260]).all() and np.isfinite(z_ref[261]).all()   # (huge contexts saturate the gates)


@pytest.mark.parametrize("features", [6, 20, 21, 43, 63])
@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_whole_layer_kernels_take_any_feature_count(monkeypatch, features, engine):
    """Shapes whose feature counts are not multiples of four (tabular data: 6, 21, 43, 63 columns; the
    reference tests' 20 with a 10 / 10 split) run in the whole-layer kernels padded (ops.fused_geometry: pad
    columns holding a constant outside the spline's box, surplus transformed features reading it and leaving
    it alone): spline flows on both engines and an affine flow against the layer-by-layer path, batch 300."""
    from nflows_amd import configs
    from nflows_amd.flows.base import Flow
    from nflows_amd.distributions.normal import StandardNormal
    from nflows_amd.nn.nets import MLP
    from nflows_amd.transforms import (AffineCouplingTransform, CompositeTransform, RandomPermutation,
                                       PiecewiseRationalQuadraticCouplingTransform as RQ)
    from nflows_amd.utils.torchutils import create_alternating_binary_mask
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow = configs.rq_nsf_flow(num_layers=4, features=features, num_bins=8, hidden_features=64, seed=features)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if "final_layer" in name:
                p.mul_(4.0)
            elif "linear_layers.1" in name:
                p.mul_(30.0)
    flow = flow.to(DEV).eval()
    torch.manual_seed(features + 1)
    layers = []
    for i in range(4):
        layers.append(RandomPermutation(features))
        layers.append(AffineCouplingTransform(create_alternating_binary_mask(features, even=(i % 2 == 0)),
                                              lambda a, b: MLP([a], [b], [128, 128])))
    affine = Flow(CompositeTransform(layers), StandardNormal([features])).to(DEV).eval()
    x = torch.randn(300, features, generator=torch.Generator().manual_seed(7)).to(DEV)
    for f, cls in ((flow, RQ), (affine, AffineCouplingTransform)):
        results = {}
        for fused in (True, False):
            monkeypatch.setattr(cls, "fuse_conditioner", fused)
            with torch.no_grad():
                units, _ = f._transform._collect_run(list(f._transform._transforms), 0, x, None, inverse=False)
                # (an odd feature count under alternating masks gives layers of two splits: the spline run takes
                # one padded geometry for both; the affine kernel needs equal splits and leaves those flows to
                # the layer-by-layer path)
                assert len(units) == (4 if fused and (cls is RQ or features % 2 == 0) else 0)
                z, lad = f._transform(x)
                lp = f.log_prob(x)
                if units:   # the base density is folded into the launch, pad columns left out of its sum
                    folded = f._transform.standard_normal_log_prob(x, None)
                    assert folded is not None and torch.equal(folded, lp)
                xr, lad_inv = f._transform.inverse(z)
                assert z.shape == x.shape and (xr - x).abs().max().item() < 2e-4
                # (sharpened layers: the worst of 300 rows is an ill-conditioned element -- 1.5e-3 on K8h's
                #  32x32x16 tiles, 2.7e-3 on K8s's 16x16x32 tiles, which serve this batch size)
                assert (lad + lad_inv).abs().max().item() < 5e-3
            results[fused] = (z, lad, lp)
        for got, want in zip(results[True], results[False]):
            assert torch.isfinite(got).all()
            assert (got - want).abs().max().item() <= 5e-5 * (1 + want.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("features,num_bins,hidden,context_features", [
    (3, 10, 50, 12),     # a posterior over three parameters given an embedded observation: 10 bins, 50 hidden units
    (7, 10, 50, 20),
    (21, 8, 64, 8),      # odd count: the layers of the run have two splits
    (32, 10, 128, 12),
])
@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_conditional_flows_of_any_shape_in_the_whole_layer_kernel(monkeypatch, engine, features, num_bins, hidden,
                                                                  context_features):
    """Conditional spline flows (context in the initial layer + the gate of every block, resnet.py:9-52,
    :92-100) with 8 or 10 bins, feature counts that need the padded geometry and conditioners narrower than 128
    run as one launch of the whole-layer kernel: against the eager oracle in float64 (the reference's
    sequence) and against the layer-by-layer path, forward, inverse and log_prob, ragged batch."""
    import copy
    from oracle import eager
    from nflows_amd import configs
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow = configs.conditional_rq_nsf_flow(num_layers=5, features=features, num_bins=num_bins, hidden_features=hidden,
                                           raw_context=6, context_features=context_features, seed=features)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if "final_layer" in name:
                p.mul_(3.0)
    flow = flow.eval()
    flow64 = copy.deepcopy(flow).double()
    flow = flow.to(DEV)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(333, features, generator=gen) * 1.5
    ctx = torch.randn(333, 6, generator=gen)
    with torch.no_grad():
        emb64 = flow64._embedding_net(ctx.double())
        z64, lad64 = eager.flow_transform(flow64, x.double(), context=emb64)
        lp64 = eager.flow_log_prob(flow64, x.double(), context=ctx.double())
        xd, cd = x.to(DEV), ctx.to(DEV)
        emb = flow._embedding_net(cd)
        units, _ = flow._transform._collect_run(list(flow._transform._transforms), 0, xd, emb, inverse=False)
        assert len(units) == 5 and units[0][0]._use_f16() == (engine == "f16x2")
        z, lad = flow._transform(xd, context=emb)
        lp = flow.log_prob(xd, context=cd)
        xr, lad_inv = flow._transform.inverse(z, context=emb)
        try:
            RQ.fuse_conditioner = False
            z2, lad2 = flow._transform(xd, context=emb)
        finally:
            RQ.fuse_conditioner = True
    import nflows_amd
    nflows_amd.check_status()
    assert (z.cpu().double() - z64).abs().max().item() < 2e-5 * (1 + z64.abs().max().item())
    assert (lad.cpu().double() - lad64).abs().max().item() < 5e-5 * (1 + lad64.abs().max().item())
    assert (lp.cpu().double() - lp64).abs().max().item() < 5e-5 * (1 + lp64.abs().max().item())
    assert (z - z2).abs().max().item() < 5e-5 and (lad - lad2).abs().max().item() < 5e-4
    assert (xr - xd).abs().max().item() < 2e-4 and (lad + lad_inv).abs().max().item() < 2e-3


@pytest.mark.parametrize("tails", ["linear", None])
def test_whole_layer_kernel_beside_an_unconditional_transform(tails, restore_fused_path):
    """apply_unconditional_transform=True (coupling.py:524-538, :90-94, :114-118): the identity half goes through a
    batch-shared spline of its own.  The conditioned half still runs in the whole-layer kernel (round 6) -- the conditioner
    sees the identity features before (forward) / after (inverse) that spline --; results against the layer-by-layer path,
    alone and behind a fused permutation."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import CompositeTransform, RandomPermutation
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    from nflows_amd.utils import torchutils
    import nflows_amd
    torch.manual_seed(11)
    D = 16
    kw = dict(tails="linear", tail_bound=3.0) if tails == "linear" else dict(tails=None)
    layers = []
    for i in range(2):
        layers.append(RandomPermutation(D))
        layers.append(RQ(torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                         lambda i_, o_: ResidualNet(i_, o_, hidden_features=128, num_blocks=2), num_bins=8,
                         apply_unconditional_transform=True, **kw))
    t = CompositeTransform(layers).to(DEV).eval()
    with torch.no_grad():
        for n, p in t.named_parameters():
            if "final_layer" in n or "unconditional_transform" in n:
                p.mul_(4.0) if "final_layer" in n else p.normal_(0.0, 1.0)
    gen = torch.Generator().manual_seed(3)
    x = (torch.rand(1000, D, generator=gen) * 0.96 + 0.02 if tails is None else torch.randn(1000, D, generator=gen) * 1.5).to(DEV)
    res = {}
    for path in ("k8", "none"):
        _select_fused_path(path)
        with torch.no_grad():
            z, lad = t(x)
            label = ops.last_layer_kernel()
            xr, ladr = t.inverse(z)
            label_i = ops.last_layer_kernel()
        nflows_amd.check_status()
        if path == "k8":
            assert "resnet" in label and "inverse=0" in label and "resnet" in label_i and "inverse=1" in label_i, (label, label_i)
        res[path] = (z, lad, xr, ladr)
    a, b = res["k8"], res["none"]
    # (steep random splines, log-determinants of ~30: two correct fp32 evaluations differ on a few ill-conditioned elements --
    #  means and 99 % quantiles, and the round trip against the layer-by-layer path's own)
    def close(u, v, mean_tol, q_tol):
        d = (u - v).abs().flatten().float()
        return float(d.mean()) < mean_tol and float(torch.quantile(d, 0.99)) < q_tol
    assert close(a[0], b[0], 2e-6, 5e-5) and close(a[1], b[1], 1e-4, 2e-3)
    assert close(a[2], b[2], 2e-5, 5e-4) and close(a[3], b[3], 1e-3, 1e-2)
    assert float((a[2] - x).abs().mean()) <= 2.0 * float((b[2] - x).abs().mean()) + 1e-6


@pytest.mark.parametrize("engine,tails", [("f16x3", "linear"), ("f16x2", None)])
def test_whole_layer_kernel_random_geometries_round_6(engine, tails, restore_fused_path, monkeypatch):
    """The sweep of test_whole_layer_kernel_random_geometries for what round 6 added: the three-piece engine K8x (every
    whole-layer bin count, d_i on both sides of 32, ragged batches) and tails=None couplings (K8's constrained-spline
    instances, whatever the engine switch says), against the layer-by-layer path."""
    import nflows_amd
    from nflows_amd import ops
    from nflows_amd import transforms as T
    from nflows_amd.nn.nets import ResidualNet
    monkeypatch.setattr(T.PiecewiseRationalQuadraticCouplingTransform, "conditioner_engine", engine)
    rng = np.random.RandomState(20260930 + (0 if tails else 1))
    for case in range(14):
        D = int(rng.choice([8, 12, 20, 36, 64, 96, 128]))
        dt = int(rng.choice([v for v in range(4, min(D, 68), 4) if D - v <= 64 and D - v >= 1]))
        bins = int(rng.choice([3, 4, 8, 8, 10, 12, 16, 24]))
        blocks = int(rng.randint(0, 4))
        mask = np.zeros(D, dtype=np.int64)
        mask[rng.permutation(D)[:dt]] = 1
        torch.manual_seed(3000 + case)
        layers = []
        for i in range(3):
            if rng.rand() < 0.7:
                layers.append(T.RandomPermutation(D))
            m = torch.from_numpy(mask if i % 2 == 0 else np.roll(mask, 1))
            kw = dict(tails="linear", tail_bound=3.0) if tails else dict(tails=None)
            layers.append(T.PiecewiseRationalQuadraticCouplingTransform(
                m, lambda a, b, nb=blocks: ResidualNet(a, b, hidden_features=128, num_blocks=nb), num_bins=bins, **kw))
        t = T.CompositeTransform(layers).to(DEV).eval()
        B = int(rng.choice([128, 256 + 17, 1024, 1000, 4096]))
        x = (torch.randn(B, D, device=DEV) * 1.3) if tails else (torch.rand(B, D, device=DEV) * 0.98 + 0.01)
        with torch.no_grad():
            _select_fused_path("k8")
            assert all(l._resnet_eligible(None) for l in layers if hasattr(l, "_resnet_eligible")), (D, dt)
            z1, l1 = t(x)
            label = ops.last_layer_kernel()
            x1, li1 = t.inverse(z1)
            _select_fused_path("none")
            z0, l0 = t(x)
            x0, li0 = t.inverse(z1)
        nflows_amd.check_status()
        what = "case %d: D=%d d_t=%d bins=%d blocks=%d B=%d %s" % (case, D, dt, bins, blocks, B, label)
        assert ("k8x::" in label) if tails else ("tails=none" in label), what
        for got, want, tol in ((z1, z0, 1e-4), (l1, l0, 2e-3), (x1, x0, 1e-4), (li1, li0, 2e-3)):
            d = (got - want).abs()
            assert d.max().item() < tol and d.median().item() < tol / 30, (what, d.max().item(), d.median().item())
