"""Conditional flows in the whole-layer kernels beyond 8 / 10 bins with ReLU (round 5).

`ResidualNet(..., context_features=E)` conditioners (nn/nets/resnet.py:9-52: the context concatenated to the initial
layer's input, a GLU gate per block; :92-100) ran in one launch only at 8 / 10 bins with ReLU blocks; every other served
bin count and the other block activations fell to GEMMs + K1 as soon as a context was given ("missing"
3).  K8h (csrc/rqs_resnet_f16_ctx_{a,b}.hip) and K8 (csrc/rqs_resnet_ctx.hip) now have the context instances for
2 .. 16 / 20 / 24 / 32 bins (ReLU) and for leaky ReLU / ELU / tanh blocks (8 / 10 bins).

Rule of round 4: a fixture from the REAL reference first.  tests/golden/flows_context_more.npz (make_golden.py
`context_more`): ten three-layer conditional flows (H = 128, 12 context features embedded from 5, sharpened weights), 256
rows each, forward / inverse / log_prob in fp32 and fp64.  The eager port reproduces them bit for bit
(tests/test_oracle_golden.py), so the 16 384 rows behind the fixture's 256 are held to the port:
  * the 256 fixture rows: log_prob, z, logabsdet, the inverse and its logabsdet within the golden rule of
    tests/test_gpu_flows.py (`check`: mean / q999 at 2 x the reference-fp32's own error against float64, max at 4 x);
  * 16 384 rows, both engines (K8h: f16 x 2 pieces; K8: bf16 x 3), the kernel that ran read back: the same rule
    (helpers.assert_error_ratio) against the port's fp32 on the CPU and its float64 on the device;
  * ragged batch (200 rows -> two padded blocks) equals the layer-by-layer path to fp32 rounding.
"""
import copy

import numpy as np
import pytest
import torch

from helpers import CONTEXT_MORE_CASES, assert_error_ratio, eager_oracle, golden_conditional_flow, parse_kwargs
from test_gpu_flows import check
from test_gpu_headline_parity import _report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROWS = 16384


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("case", CONTEXT_MORE_CASES)
def test_conditional_flows_of_other_bin_counts_and_activations(monkeypatch, golden_dir, case, engine):
    import nflows_amd
    from nflows_amd import ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow_cpu, g, name = golden_conditional_flow(golden_dir, case)
    cfg = parse_kwargs(dict((str(n), str(c)) for n, c in g["meta"])[name])
    flow = copy.deepcopy(flow_cpu).to(DEV)
    x, noise, ctx = (torch.from_numpy(g[name + "/" + k]).to(DEV) for k in ("x", "noise", "context"))
    want_kernel = ("k8h::rqs_resnet_f16_kernel<", "K=%d" % cfg["K"], "ctx=1") if engine == "f16x2" else \
                  ("rqs_resnet_kernel<", "K=%d" % cfg["K"], "ctx=1")
    with torch.no_grad():
        emb = flow._embedding_net(ctx)
        units, after = flow._transform._collect_run(list(flow._transform._transforms), 0, x, emb, inverse=False)
        assert len(units) == 3 and after == 6, "the conditional flow is not one run of the whole-layer kernel"
        lp = flow.log_prob(x, context=ctx)
        z, lad = flow._transform(x, context=emb)
        ran = ops.last_layer_kernel()
        assert all(s in ran for s in want_kernel), ran
        xs, lad_inv = flow._transform.inverse(noise, context=emb)
        assert all(s in ops.last_layer_kernel() for s in want_kernel), ops.last_layer_kernel()
        try:
            RQ.fuse_conditioner = False
            z2, lad2 = flow._transform(x, context=emb)
            lp_ragged = flow.log_prob(x[:200], context=ctx[:200])
        finally:
            RQ.fuse_conditioner = True
        assert (z - z2).abs().max().item() < 5e-4 and (lad - lad2).abs().max().item() < 5e-3
        assert (lp_ragged - flow.log_prob(x[:200], context=ctx[:200])).abs().max().item() < 5e-3
    nflows_amd.check_status()
    d = x.shape[1]
    check(z, g[name + "/z"], g[name + "/z64"], case + " z", 3e-6)
    check(lad, g[name + "/lad"], g[name + "/lad64"], case + " lad", 3e-6 * d)
    check(lp, g[name + "/log_prob"], g[name + "/log_prob64"], case + " log_prob", 3e-6 * d)
    check(xs, g[name + "/inv_x"], g[name + "/inv_x64"], case + " inv_x", 3e-6)
    check(lad_inv, g[name + "/inv_lad"], g[name + "/inv_lad64"], case + " inv_lad", 3e-6 * d)

    # 16 384 rows behind the fixture's: the port's fp32 on the CPU (= the reference's bits) and its float64 on the device
    gen = torch.Generator().manual_seed(11)
    xb = 1.2 * torch.randn(ROWS, d, generator=gen)
    nb = torch.randn(ROWS, d, generator=gen)
    cb = torch.randn(ROWS, ctx.shape[1], generator=gen)
    o = eager_oracle(flow_cpu, xb, nb, cb, fp64_device=DEV)
    with torch.no_grad():
        emb = flow._embedding_net(cb.to(DEV))
        z, lad = flow._transform(xb.to(DEV), context=emb)
        ran = ops.last_layer_kernel()
        xi, ladi = flow._transform.inverse(nb.to(DEV), context=emb)
    nflows_amd.check_status()
    figures = {}
    for what, got, k in (("z", z, "z"), ("lad", lad, "lad"), ("inv_x", xi, "xi"), ("inv_lad", ladi, "ladi")):
        scale = 1 + np.abs(o[k + "64"]).max()
        figures[what] = assert_error_ratio(got.cpu().numpy(), o[k + "32"], o[k + "64"], "%s %s %s" % (case, engine, what),
                                           factor=2.0, max_factor=4.0, max_floor=3e-6 * scale * (d if "lad" in what else 1))
    _report({"config": "context_%s_%s" % (case, engine), "kernel": ran, "rows": ROWS,
             "mean_error_ratio": {k: v["got"]["mean"] / max(v["reference"]["mean"], 1e-30) for k, v in figures.items() if v}})


@pytest.mark.parametrize("case", ["ctx_k4", "ctx_k16", "ctx_elu_k10", "ctx_tanh_k8"])
def test_eight_wave_context_instances_equal_the_four_wave_ones(golden_dir, case):
    """K8h picks four waves per workgroup up to 16 384 rows and eight above; a wave runs the same instruction stream in
    both, so a 65 536-row launch (eight waves: the instances the test above never reaches) must reproduce, bit for bit,
    the same rows pushed through in 16 384-row pieces (four waves) -- forward and inverse."""
    from nflows_amd import ops
    flow_cpu, g, name = golden_conditional_flow(golden_dir, case)
    flow = copy.deepcopy(flow_cpu).to(DEV)
    gen = torch.Generator().manual_seed(5)
    x = (1.2 * torch.randn(65536, 16, generator=gen)).to(DEV)
    ctx = torch.randn(65536, 5, generator=gen).to(DEV)
    with torch.no_grad():
        emb = flow._embedding_net(ctx)
        for fn in (flow._transform, flow._transform.inverse):
            big, big_lad = fn(x, context=emb)
            assert "waves=8" in ops.last_layer_kernel() and "ctx=1" in ops.last_layer_kernel(), ops.last_layer_kernel()
            parts = [fn(x[i:i + 16384], context=emb[i:i + 16384]) for i in range(0, 65536, 16384)]
            assert "waves=4" in ops.last_layer_kernel(), ops.last_layer_kernel()
            assert torch.equal(big, torch.cat([p[0] for p in parts])) and torch.equal(big_lad, torch.cat([p[1] for p in parts]))
            assert torch.isfinite(big).all()


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("num_bins,activation", [(4, "relu"), (12, "relu"), (8, "tanh")])
def test_wider_conditional_flows_against_the_port(monkeypatch, num_bins, activation, engine):
    """48 features (24 identity + 12 context columns: more than 32, K8's four-k-step initial layer, which the 16-feature
    fixtures never reach) against the eager port -- pinned to the reference bit for bit on the fixtures' structure
    (tests/test_oracle_golden.py) -- under the rule of the test above, 16 384 rows."""
    import nflows_amd
    from nflows_amd import configs, ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    F = torch.nn.functional
    act = {"relu": F.relu, "tanh": torch.tanh}[activation]
    flow_cpu = configs.conditional_rq_nsf_flow(3, 48, num_bins, 128, 5, 12, 3.0, seed=31, activation=act).eval()
    s_final, s_lin1 = (4.0, 30.0) if activation == "relu" else (8.0, 6.0)
    with torch.no_grad():
        for n_, p in flow_cpu.named_parameters():
            if "final_layer" in n_:
                p.mul_(s_final)
            elif "linear_layers.1" in n_:
                p.mul_(s_lin1)
            elif "context_layer" in n_:
                p.mul_(3.0)
    flow = copy.deepcopy(flow_cpu).to(DEV)
    gen = torch.Generator().manual_seed(23)
    xb = 1.2 * torch.randn(ROWS, 48, generator=gen)
    nb = torch.randn(ROWS, 48, generator=gen)
    cb = torch.randn(ROWS, 5, generator=gen)
    o = eager_oracle(flow_cpu, xb, nb, cb, fp64_device=DEV)
    with torch.no_grad():
        emb = flow._embedding_net(cb.to(DEV))
        z, lad = flow._transform(xb.to(DEV), context=emb)
        ran = ops.last_layer_kernel()
        want = ("k8h::", "ctx=1", "K=%d" % num_bins) if engine == "f16x2" else ("rqs_resnet_kernel<", "init_ks=4", "ctx=1", "K=%d" % num_bins)
        assert all(s in ran for s in want), ran
        xi, ladi = flow._transform.inverse(nb.to(DEV), context=emb)
    nflows_amd.check_status()
    for what, got, k in (("z", z, "z"), ("lad", lad, "lad"), ("inv_x", xi, "xi"), ("inv_lad", ladi, "ladi")):
        scale = 1 + np.abs(o[k + "64"]).max()
        assert_error_ratio(got.cpu().numpy(), o[k + "32"], o[k + "64"], "D48 K%d %s %s %s" % (num_bins, activation, engine, what),
                           factor=2.0, max_factor=4.0, max_floor=3e-6 * scale * (48 if "lad" in what else 1))


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_every_served_bin_count_with_a_context_agrees_with_the_layer_by_layer_path(monkeypatch, engine):
    """The fixtures cover 4, 6, 9, 12, 16 and 24 bins; the context instances exist for every bin count the whole-layer
    kernels serve.  All sixteen of them (and 8 / 10), 4 096 rows forward and inverse, against the layer-by-layer path
    (PyTorch-ROCm GEMMs + K1, held to the reference for every bin count in tests/test_gpu_bins.py): agreement to fp32
    rounding of three sharpened layers, the kernel that ran read back."""
    import nflows_amd
    from nflows_amd import configs, ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    gen = torch.Generator().manual_seed(41)
    x = (1.2 * torch.randn(4096, 16, generator=gen)).to(DEV)
    noise = torch.randn(4096, 16, generator=gen).to(DEV)
    ctx = torch.randn(4096, 5, generator=gen).to(DEV)
    worst = {}
    for K in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 24, 32):
        flow = configs.conditional_rq_nsf_flow(3, 16, K, 128, 5, 12, 3.0, seed=100 + K).to(DEV).eval()
        with torch.no_grad():
            for n_, p in flow.named_parameters():
                if "final_layer" in n_:
                    p.mul_(4.0)
                elif "linear_layers.1" in n_:
                    p.mul_(30.0)
                elif "context_layer" in n_:
                    p.mul_(3.0)
            emb = flow._embedding_net(ctx)
            z, lad = flow._transform(x, context=emb)
            ran = ops.last_layer_kernel()
            assert "ctx=1" in ran and "K=%d" % K in ran and ("k8h::" in ran) == (engine == "f16x2"), ran
            xi, ladi = flow._transform.inverse(noise, context=emb)
            try:
                RQ.fuse_conditioner = False
                z2, lad2 = flow._transform(x, context=emb)
                xi2, ladi2 = flow._transform.inverse(noise, context=emb)
            finally:
                RQ.fuse_conditioner = True
        nflows_amd.check_status()
        worst[K] = (float((z - z2).abs().max()), float((lad - lad2).abs().max()), float((xi - xi2).abs().max()),
                    float((ladi - ladi2).abs().max()))
        # (the log-determinant of an inverse that lands next to a knot is ill-conditioned -- one element of 4 096 at 15
        #  bins differs by 5.6e-3 between the two fp32 paths --: the 99.9 % quantile carries the bound, the maximum 10 x it)
        q999 = [float(torch.quantile((a - b).abs().flatten().float(), 0.999)) for a, b in ((lad, lad2), (ladi, ladi2))]
        assert worst[K][0] < 5e-4 and worst[K][2] < 5e-4 and max(q999) < 5e-3 and max(worst[K][1], worst[K][3]) < 5e-2, (K, worst[K], q999)
        assert float(lad.abs().mean()) > 0.5, (K, "the layers are not trivial")
    _report({"config": "context_all_bin_counts_%s" % engine, "max_abs_difference_to_layer_by_layer (z, lad, inv_x, inv_lad)": worst})
