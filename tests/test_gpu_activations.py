"""The whole-layer kernels on conditioners built with another activation than ReLU (round 4).

`ResidualNet(..., activation=F.elu)` (nn/nets/resnet.py:27; applied in front of both Linears of every block, :44, :47) used
to leave the one-launch kernels -- K8h / K8 apply ReLU while they convert accumulators into the next GEMM's pieces.  That
conversion is now templated on the activation (csrc/fused_common.hpp `activate<ACT>`: F.leaky_relu and F.elu with their
default parameters, tanh), K8h for 8 and 10 bins, K8 -- its second pass and the bf16x3 engine -- through its plain loop.

  * tests/golden/flows_acts.npz: the REAL reference on steep two-layer flows with leaky-ReLU / ELU / tanh conditioners
    (8 and 10 bins), forward and inverse, fp32 and fp64; the eager port (which runs the flow's own modules) reproduces it
    bit for bit (tests/test_oracle_golden.py), so the rows behind the fixture's 128 are held to the port: 65 536 rows per engine (round 5);
  * engines driven explicitly, the kernel that ran read back: K8h eight-wave (65 536 rows), K8h four-wave, K8, and the
    layer-by-layer path these layers took before (conditioner modules + the final Linear fused with the spline / K1);
  * the headline rule: error against float64 at most 2 x the reference-fp32's own on the mean and on the 99.9 % quantile,
    at most three elements above 4 x the reference's maximum (see tests/test_gpu_bins.py);
  * the per-element arithmetic of `activate<ACT>` runs in the CPU suite against torch (tests/test_rqs_f32_host.py).
"""
import copy

import pytest
import torch

from helpers import steep_flow
from test_gpu_headline_parity import _report
from test_gpu_steep import _batch, _check_all, _checked, _status, engine_switches  # noqa: F401  (fixture)
from test_gpu_bins import _oracle, ROWS
from test_gpu_steep import _chunked

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = {"act_leaky_relu_k8": "leaky_relu", "act_elu_k8": "elu", "act_tanh_k8": "tanh", "act_elu_k10": "elu",
         "act_tanh_k10": "tanh"}
CODES = {"leaky_relu": 1, "elu": 2, "tanh": 3}


def _engines(K, act):
    return {
        "k8h_w8": (dict(path="k8", engine="f16x2"), 65536, ("k8h::", "waves=8", "K=%d," % K, "act=%s>" % act)),
        "k8h_w4": (dict(path="k8", engine="f16x2"), 16384, ("k8h::", "waves=4", "K=%d," % K, "act=%s>" % act)),
        "k8": (dict(path="k8", engine="bf16x3"), 16384, ("rqs_resnet_kernel<", "pipe=0", "K=%d," % K, "act=%d>" % CODES[act])),
        "layer_by_layer": (dict(path="none", engine="f16x2"), 16384, ("rqs_coupling",)),
    }


@pytest.mark.parametrize("case,engine", [(c, e) for c in CASES for e in ("k8h_w8", "k8h_w4", "k8", "layer_by_layer")])
def test_other_activations_on_every_engine(golden_dir, engine_switches, case, engine):
    import nflows_amd
    from nflows_amd import ops
    flow_cpu, g, cfg = steep_flow(golden_dir, case, "flows_acts.npz")
    K, act = cfg["K"], CASES[case]
    switches, rows, expect = _engines(K, act)[engine]
    x = _batch(g, case, "x", 65536, cfg["D"])
    noise = _batch(g, case, "noise", 65536, cfg["D"])
    o = _oracle(case, flow_cpu, x, noise)
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    engine_switches(switches["path"], switches["engine"], True)
    _status(case, clear=True)
    ran = {}
    redo = {"f": 0, "i": 0}

    def counted(fn, key):
        def run(t):
            out = fn(t)
            if engine.startswith("k8h"):
                redo[key] += ops.last_redo_blocks()
            return out
        return run
    with torch.no_grad():
        z, lad = _chunked(counted(flow._transform, "f"), x, rows)
        ran["forward"] = ops.last_layer_kernel()
        lp = _chunked(flow.log_prob, x, rows)
        xi, ladi = _chunked(counted(flow._transform.inverse, "i"), noise, rows)
        ran["inverse"] = ops.last_layer_kernel()
    redo_f, redo_i = redo["f"], redo["i"]
    for direction, label in ran.items():
        for piece in expect:
            assert piece in label, "%s %s ran %r, expected %r" % (engine, direction, label, expect)
    _report({"config": "%s_%s" % (case, engine), "kernels": ran, "rows": rows, "redo_blocks": [redo_f, redo_i]})
    _checked("%s_%s" % (case, engine), flow, x, rows, z,
             lambda: _check_all("%s_%s" % (case, engine), case, g, o, z, lad, lp, xi, ladi, rows=ROWS))
    assert redo_f + redo_i <= max(1, ROWS // 128 // 100), "the f16 engine handed %d + %d row blocks to the exact kernel" % (redo_f, redo_i)
    _status("%s_%s" % (case, engine))


@pytest.mark.parametrize("act", ["leaky_relu", "elu", "tanh"])
def test_exact_kernel_redoes_overflowing_blocks_with_the_activation(act):
    """A row block whose activations leave the f16 range is redone by the exact kernel (K8) WITH the layer's activation:
    inputs scaled until K8h gives blocks up (reported by the library), results still equal to the eager port's."""
    import nflows_amd
    from nflows_amd import configs, ops
    from oracle import eager
    F = torch.nn.functional
    fn = {"leaky_relu": F.leaky_relu, "elu": F.elu, "tanh": torch.tanh}[act]
    flow_cpu = configs.rq_nsf_flow(2, 16, 8, 128, 2, 3.0, seed=77, activation=fn).eval()
    with torch.no_grad():
        for t in flow_cpu._transform._transforms:
            if hasattr(t, "transform_net"):   # hidden pre-activations of ~1e5: beyond f16 (65 504) in some rows
                t.transform_net.initial_layer.weight.mul_(3.0e4)
                t.transform_net.final_layer.weight.mul_(1.0e-4)
    x = torch.randn(4096, 16, generator=torch.Generator().manual_seed(3)) * 1.5
    flow = copy.deepcopy(flow_cpu).to(DEV)
    with torch.no_grad():
        z, lad = flow._transform(x.to(DEV))
        label, redo = ops.last_layer_kernel(), ops.last_redo_blocks()
        z64, lad64 = eager.flow_transform(flow_cpu.double(), x.double())
        flow_cpu.float()
        z32, lad32 = eager.flow_transform(flow_cpu, x)
    nflows_amd.check_status()
    assert "k8h::" in label or "k8s::" in label, label
    _report({"config": "redo_" + act, "kernel": label, "redo_blocks": redo})
    if act != "tanh":   # (tanh bounds the GEMM operands: only the residual stream can leave the range)
        assert redo > 0, "the case was built to overflow the f16 range"
    e_got, e_ref = (z.cpu().double() - z64).abs(), (z32.double() - z64).abs()
    assert e_got.mean().item() <= 2.0 * e_ref.mean().item() + 1e-9, (e_got.mean().item(), e_ref.mean().item())
    l_got, l_ref = (lad.cpu().double() - lad64).abs(), (lad32.double() - lad64).abs()
    assert l_got.mean().item() <= 2.0 * l_ref.mean().item() + 1e-9
