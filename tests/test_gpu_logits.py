"""The conditioner's GEMM arithmetic of every whole-layer engine, IN FRONT of the spline (round 6).

Every other engine test compares results behind the spline, where the spline's own conditioning dominates the error and
hides what the GEMM engine does.  Here the quantity compared is what `transform_net(identity_split)` returns in the
reference (coupling.py:85; nn/nets/resnet.py:92-100: five F.linear on fp32), read out of the kernels by their diagnostic
twins (nfa_rqs_flow_resnet_*_logits_f32: the final Linear's accumulators as the spline evaluation reads them), and the
yardstick is the error an fp32 LIBRARY GEMM chain (stock PyTorch on the device: hipBLASLt / rocBLAS) has against the
float64 evaluation of the same network on the same inputs:

    err(engine vs float64)  <=  2 x err(library fp32 vs float64)      mean and 99.9 % quantile, 4 x on the maximum

for K8x (three f16 pieces, five products), K8h (two f16 pieces, three products), K8 (three bf16 pieces, six products) on
the BASELINE layer shape -- D = 64, d_i = d_t = 32, H = 128, two blocks, 736 logits per row, 65 536 rows -- with seed
weights, with the wide-dynamic-range weights of test_f16_engine_over_a_wide_dynamic_range, and with small activations
(inputs x 0.01: where two f16 pieces at scale 1 fall back to an absolute error and three at scale 16 do not).  K11
(affine_mlp.hip; bf16 x 3) has no diagnostic twin and needs none: an ADDITIVE coupling layer on zero inputs returns its
conditioner's output bit for bit (0 + shift), for MLP and ResidualNet conditioners alike.
"""
import copy
import math

import numpy as np
import pytest
import torch

from helpers import error_stats
from test_gpu_headline_parity import _report, _spread_rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROWS = 65536
ENGINES = {"k8x": ("f16x3", "k8x::"), "k8h": ("f16x2", "k8h::"), "k8": ("bf16x3", "rqs_resnet_kernel<")}


def _layer(case):
    """one RQ coupling layer of the BASELINE shape (alternating mask), eval mode, on the CPU"""
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    torch.manual_seed(5)
    mask = torch.ones(64)
    mask[::2] = -1
    layer = RQ(mask, lambda i, o: ResidualNet(i, o, hidden_features=128, num_blocks=2), num_bins=8, tails="linear",
               tail_bound=3.0).eval()
    net = layer.transform_net
    gen = torch.Generator().manual_seed(99)
    if case == "wide_weights":
        _spread_rows(net.initial_layer.weight, 4.0, gen)
        with torch.no_grad():
            net.initial_layer.weight.mul_(0.3)
        for b_i, block in enumerate(net.blocks):
            for l_i, lin in enumerate(block.linear_layers):
                _spread_rows(lin.weight, 4.0, gen)
                with torch.no_grad():
                    lin.weight.mul_((0.1, 1.0, 0.03, 0.5)[2 * b_i + l_i])
        _spread_rows(net.final_layer.weight, 3.0, gen)
        with torch.no_grad():
            net.final_layer.weight.mul_(0.5)
    else:
        with torch.no_grad():   # (the reference initialises the final layer near zero: give the logits some size)
            net.final_layer.weight.mul_(30.0)
            net.final_layer.bias.normal_(0.0, 0.5, generator=gen)
    return layer


def _references(layer, x):
    """(float64 truth, library fp32) logits [rows, d_t, 23] of `layer`'s conditioner on the device, width / height
    entries divided by sqrt(hidden_features) as the kernels hold them (coupling.py:554-556)"""
    xi = x[:, layer.identity_features.to(x.device)]
    with torch.no_grad():
        net64 = copy.deepcopy(layer.transform_net).double().to(DEV)
        net32 = copy.deepcopy(layer.transform_net).float().to(DEV)
        t64 = net64(xi.double()).view(x.shape[0], -1, 23)
        t32 = net32(xi).view(x.shape[0], -1, 23)
    for t in (t64, t32):
        t[..., :16] /= math.sqrt(128.0)
    return t64, t32


def _ratio_rule(config, got, lib32, truth, factor=2.0, max_factor=4.0):
    e_got = error_stats((got.double() - truth).abs().reshape(-1).cpu().numpy())
    e_lib = error_stats((lib32.double() - truth).abs().reshape(-1).cpu().numpy())
    ratio = {k: e_got[k] / max(e_lib[k], 1e-300) for k in ("mean", "q999", "max")}
    _report({"config": config, "what": "logits", "elements": int(truth.numel()), "engine_vs_fp64": e_got,
             "library_fp32_vs_fp64": e_lib, "ratio": ratio, "max_abs_logit": float(truth.abs().max())})
    for k, f in (("mean", factor), ("q999", factor), ("max", max_factor)):
        assert e_got[k] <= f * e_lib[k], "%s: %s error of the logits vs float64 %.3e exceeds %.0f x the library fp32 GEMMs' %.3e" % (
            config, k, e_got[k], f, e_lib[k])
    return ratio


@pytest.mark.parametrize("case", ["seed_weights", "wide_weights", "small_activations"])
@pytest.mark.parametrize("engine", list(ENGINES))
def test_last_layer_logits_against_a_library_fp32_gemm(monkeypatch, engine, case):
    import nflows_amd
    from nflows_amd import ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    layer_cpu = _layer(case)
    layer = copy.deepcopy(layer_cpu).to(DEV)
    x = torch.randn(ROWS, 64, generator=torch.Generator().manual_seed(1234)).to(DEV)
    if case == "small_activations":
        x = x * 0.01
    truth, lib32 = _references(layer, x)
    name, kernel = ENGINES[engine]
    monkeypatch.setattr(RQ, "conditioner_engine", name)
    monkeypatch.setattr(ops, "K8S_ENABLED", False)
    with torch.no_grad():
        z_plain, lad_plain = layer(x)
        assert kernel in ops.last_layer_kernel(), ops.last_layer_kernel()
        with ops.capture_last_layer_logits() as cap:
            z, lad = layer(x)
    nflows_amd.check_status()
    assert cap.launches == 1 and cap.logits is not None, "the diagnostic twin did not run"
    if cap.redo is not None:
        assert int((cap.redo != 0).sum()) == 0, "row blocks were handed to the exact kernel: the engine under test did not produce them"
    # the twin is the same kernel with one more store: same results, bit for bit (K8's twin is its plain final-layer
    # loop -- the same products in the same order, the spline evaluated by the longer rounding sequence)
    if engine == "k8":
        assert (z - z_plain).abs().max().item() < 2e-4 and ((lad - lad_plain).abs() / (1 + lad_plain.abs())).max().item() < 1e-3
    else:
        assert torch.equal(z, z_plain) and torch.equal(lad, lad_plain)
    logits = cap.logits[:, :32]
    assert logits.shape == truth.shape and torch.isfinite(logits).all()
    # the rule of this file; two f16 pieces at scale 1 lose the low piece of small activations to f16's subnormal
    # spacing (an ABSOLUTE 2^-25 per operand: rqs_resnet_f16.hip): K8h is given that floor on `small_activations`, the
    # engines with reference-width operands are not
    ratio = None
    if engine == "k8h" and case == "small_activations":
        e_got = error_stats((logits.double() - truth).abs().reshape(-1).cpu().numpy())
        _report({"config": "logits_%s_%s" % (engine, case), "engine_vs_fp64": e_got, "note": "absolute floor of two f16 pieces at scale 1"})
        assert e_got["max"] < 1e-5 and e_got["mean"] < 4e-7
    else:
        ratio = _ratio_rule("logits_%s_%s" % (engine, case), logits, lib32, truth)


@pytest.mark.parametrize("conditioner", ["mlp", "resnet"])
def test_k11_conditioner_output_through_an_additive_layer(conditioner):
    """K11: z[:, transformed] of an additive coupling layer on rows whose transformed features are zero IS the
    conditioner's output (0 + shift, exact): held to the same rule."""
    import nflows_amd
    from nflows_amd import ops
    from nflows_amd.nn.nets import MLP, ResidualNet
    from nflows_amd.transforms import AdditiveCouplingTransform, CompositeTransform
    torch.manual_seed(7)
    D = 32
    mask = torch.ones(D)
    mask[::2] = -1

    def make(i, o):
        if conditioner == "mlp":
            return MLP((i,), (o,), hidden_sizes=[128, 128])
        return ResidualNet(i, o, hidden_features=128, num_blocks=2)
    # (a run needs two layers: the second one sees the first one's output; its conditioner is the one observed)
    first, second = AdditiveCouplingTransform(mask, make), AdditiveCouplingTransform(-mask, make)
    with torch.no_grad():
        for p_ in second.transform_net.parameters():
            p_.mul_(3.0)
        for p_ in first.transform_net.parameters():     # the first layer adds nothing: its output = its input
            p_.zero_()
    flow = CompositeTransform([first, second]).to(DEV).eval()
    x = torch.randn(ROWS, D, generator=torch.Generator().manual_seed(4321)).to(DEV)
    x[:, second.transform_features.to(DEV)] = 0.0
    with torch.no_grad():
        z, lad = flow(x)
    nflows_amd.check_status()
    assert "affine_mlp_kernel" in ops.last_layer_kernel(), ops.last_layer_kernel()
    assert float(lad.abs().max()) == 0.0
    got = z[:, second.transform_features.to(DEV)]
    xi = x[:, second.identity_features.to(DEV)]
    with torch.no_grad():
        truth = copy.deepcopy(second.transform_net).double()(xi.double())
        lib32 = second.transform_net(xi)
    _ratio_rule("k11_output_%s" % conditioner, got, lib32, truth)
