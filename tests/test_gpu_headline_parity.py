"""The HIP path at the BASELINE.json configurations, in its DEFAULT mode, against the oracle.

For every sized configuration the drop-in flow is evaluated on the GPU at
the stated batch size and compared with oracle/eager.py (bit-identical to the reference's CPU path,
tests/test_oracle_golden.py) evaluated in float32 and in float64 on the same weights and rows:

    err(HIP fp32 vs float64)  <=  2 x err(reference fp32 vs float64)  + floor

for the maximum, the mean and the 99.9 % quantile of the absolute error of z, logabsdet and
log_prob (at depth 32 the reference's own fp32 error is 1e-4 .. 1e-3, so its
error against the float64 evaluation of the same flow is the yardstick, not a fixed tolerance).
The floor (round 3) is FOUR fp32 ULPS of the largest |truth| on the maximum only -- a value of
magnitude 100 cannot be asked to agree better than its own spacing -- and ZERO on the mean and the
quantile: those two really assert "at most twice the reference's own error" (round 2 used
tol (1 + max |truth|) on all three, which was 8-11 x the reference's maximum error for log-densities).
Achieved figures are printed (pytest -s) and appended to gpurun_out/parity_report.jsonl.

Rows the oracle does not visit (it takes seconds per 16 384 rows) are covered by a size-independent
property: rows are independent, so evaluating a strided subset alone must reproduce the full-batch
results of those rows bit for bit.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import LAD_TOL, OUT_TOL, bulk_fraction

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FACTOR = 2.0  # native error <= 2 x the reference's own fp32 error


def _report(entry):
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(entry) + "\n")
    except OSError:
        pass
    print("\n[parity] " + json.dumps(entry))


def _stats(err):
    err = err.reshape(-1)
    return {"max": float(err.max()), "mean": float(err.mean()), "q999": float(np.quantile(err, 0.999))}


def compare(config, what, got, ref32, truth, tol, max_factor=FACTOR, q_factor=FACTOR, max_count=None):
    """got / ref32: float32 arrays, truth: float64.  Asserts the 2x bound on max, mean and the
    99.9 % quantile (`max_factor`: the bound on the maximum alone; `q_factor`: on the quantile alone); returns the
    figures.
    `max_count` (round 5, the steep fixtures at 65 536 rows): the maximum of a heavy-tailed error -- the reference's own
    is 1e3 .. 1e4 x its mean there -- is ONE element's luck and was bounded by factors of 8 .. 32 in round 4, which
    bounds nothing.  With `max_count` the maximum is not compared; instead the NUMBER of elements whose error exceeds
    4 x the reference-fp32's own maximum (+ four ulps of the largest value) must not exceed `max_count`: a correct
    implementation puts an element there once in a while, a defect puts many."""
    got64 = got.astype(np.float64)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref32)), "%s %s: non-finite pattern differs" % (config, what)
    fin = np.isfinite(truth)
    err_got = np.abs(got64 - truth)[fin]
    e_got = _stats(err_got)
    e_ref = _stats(np.abs(ref32.astype(np.float64) - truth)[fin])
    floor = 4.0 * 2.0 ** -23 * float(np.abs(truth[fin]).max())   # four ulps of the largest value, maximum only
    over = int((err_got > 4.0 * e_ref["max"] + floor).sum())
    entry = {"config": config, "what": what, "rows": int(got.shape[0]), "elements": int(err_got.size), "hip_vs_fp64": e_got,
             "reference_fp32_vs_fp64": e_ref, "floor_on_max": floor, "elements_above_4x_reference_max": over,
             "ratio": {k: e_got[k] / max(e_ref[k], 1e-300) for k in e_got},
             "bulk_within_tol_of_reference_fp32": bulk_fraction(got, ref32, tol), "tol": tol}
    _report(entry)
    for k in ("max", "mean", "q999"):
        if k == "max" and max_count is not None:
            assert over <= max_count, ("%s %s: %d elements with an error above 4 x the reference fp32's maximum %.3e (allowed: %d)"
                                       % (config, what, over, e_ref["max"], max_count))
            # (the few elements the count rule lets through are still bounded -- an isolated O(1) error, one
            #  wrong bin or one stale lane, is not "an element above 4 x once in a while")
            assert e_got["max"] <= 64.0 * e_ref["max"] + floor, (
                "%s %s: max error vs float64 %.3e exceeds 64 x the reference fp32's %.3e" % (config, what, e_got["max"], e_ref["max"]))
            continue
        bound = (max_factor if k == "max" else q_factor if k == "q999" else FACTOR) * e_ref[k] + (floor if k == "max" else 0.0)
        assert e_got[k] <= bound, (
            "%s %s: %s error vs float64 %.3e exceeds %.1f x the reference fp32's %.3e%s"
            % (config, what, k, e_got[k], max_factor if k == "max" else q_factor if k == "q999" else FACTOR, e_ref[k],
               " (+ %.1e)" % floor if k == "max" else ""))
    return entry


def oracle_eval(flow_cpu, x_cpu, need=("z", "lad", "lp"), context=None):
    """float32 (on the host: the reference's arithmetic) and float64 (the same port run by stock PyTorch on the device)
    evaluation of the eager port: helpers.eager_oracle (`context`: the raw context rows of a conditional flow, embedded
    by the flow's own embedding net in the same precision)."""
    from helpers import eager_oracle
    return eager_oracle(flow_cpu, x_cpu, context=context, fp64_device=DEV)


def hip_eval(flow_cpu, x_cpu, context=None):
    import copy
    import nflows_amd
    flow = copy.deepcopy(flow_cpu).float().to(DEV).eval()
    x = x_cpu.to(DEV)
    with torch.no_grad():
        if context is None:
            z, lad = flow._transform(x)
            lp = flow.log_prob(x)
        else:
            ctx = context.to(DEV)
            z, lad = flow._transform(x, context=flow._embedding_net(ctx))
            lp = flow.log_prob(x, context=ctx)
    nflows_amd.check_status()
    return flow, z, lad, lp


def check_flow(config, flow_cpu, x_cpu, oracle_rows, context=None):
    """Full batch on the GPU; the oracle on `oracle_rows` (an index tensor); row independence for
    the rest."""
    flow, z, lad, lp = hip_eval(flow_cpu, x_cpu, context)
    sub = x_cpu[oracle_rows]
    o = oracle_eval(flow_cpu, sub, context=None if context is None else context[oracle_rows])
    idx = oracle_rows.to(DEV)
    zs, lads, lps = (t[idx].cpu().numpy() for t in (z, lad, lp))
    d = x_cpu.shape[1]
    compare(config, "z", zs, o["z32"], o["z64"], OUT_TOL)
    compare(config, "logabsdet", lads, o["lad32"], o["lad64"], LAD_TOL)
    compare(config, "log_prob", lps, o["lp32"], o["lp64"], LAD_TOL)
    if len(oracle_rows) < x_cpu.shape[0]:
        # size-independent property: the same rows evaluated alone (another batch size, other
        # positions in the launch grid) give the same bits
        # (within ONE kernel family: batches that give a CU at most one 128-row block take the 16-sample-tile kernel
        #  K8s, whose sums run in another order -- it is held to the oracle by its own tests below)
        from nflows_amd import ops
        big = x_cpu.shape[0] > 32768
        saved = ops.K8S_ENABLED
        try:
            if big:
                ops.K8S_ENABLED = False
            with torch.no_grad():
                emb = None if context is None else flow._embedding_net(context[oracle_rows].to(DEV))
                z2, lad2 = flow._transform(sub.to(DEV), context=emb)
        finally:
            ops.K8S_ENABLED = saved
        assert torch.equal(z2, z[idx]) and torch.equal(lad2, lad[idx]), config + ": rows are not independent of the batch"
    return flow


def bench_rows(batch=65536, features=64):
    """bench.py's own input batch on rank 0."""
    return torch.randn(batch, features, generator=torch.Generator().manual_seed(1234))


def test_headline_32_layer_flow_default_mode():
    """configs[3] / north-star at one GPU: 32 x (RandomPermutation + RQ coupling), D = 64, K = 8,
    ResidualNet H = 128, seed-0 weights, B = 65 536 of bench.py's own rows; the oracle visits the
    first 16 384."""
    from nflows_amd import configs
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = bench_rows()
    check_flow("cfg4_32layer_d64_k8_b65536", flow_cpu, x, torch.arange(16384))


def test_config3_16_layer_flow():
    """configs[2]: 16 layers, B = 65 536; the oracle visits every fourth row."""
    from nflows_amd import configs
    flow_cpu = configs.rq_nsf_flow(num_layers=16, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = bench_rows()
    check_flow("cfg3_16layer_d64_k8_b65536", flow_cpu, x, torch.arange(0, 65536, 4))


def test_config2_affine_stack():
    """configs[1]: 8 x AffineCouplingTransform, D = 32, MLP [128, 128] conditioner, B = 16 384 (all
    rows through the oracle)."""
    from nflows_amd import configs
    flow_cpu = configs.affine_coupling_flow(num_layers=8, features=32, hidden_sizes=(128, 128), seed=0).eval()
    x = torch.randn(16384, 32, generator=torch.Generator().manual_seed(1234))
    check_flow("cfg2_8layer_affine_d32_b16384", flow_cpu, x, torch.arange(16384))


def test_config5_autoregressive_forward():
    """configs[4] forward pass: MaskedPiecewiseRationalQuadraticAutoregressiveTransform, D = 784,
    K = 8, H = 256, B = 4 096; the oracle visits the first 1 024 rows."""
    from nflows_amd import configs
    flow_cpu = configs.ar_rq_flow(features=784, hidden_features=256, num_bins=8, tail_bound=3.0, seed=0).eval()
    x = torch.randn(4096, 784, generator=torch.Generator().manual_seed(1234))
    check_flow("cfg5_ar_rq_d784_k8_b4096_forward", flow_cpu, x, torch.arange(1024))


def test_ten_bin_flow():
    """The reference's default bin count (10) on the 32-layer flow, 8 192 rows."""
    from nflows_amd import configs
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=10, hidden_features=128, seed=0).eval()
    x = bench_rows(8192)
    check_flow("32layer_d64_k10_b8192", flow_cpu, x, torch.arange(8192))


def test_conditional_32_layer_flow():
    """The 32-layer flow with conditioners that take a context (12 features embedded from 5 raw ones; K8h with a
    context), B = 32 768; the oracle visits every fourth row."""
    from nflows_amd import configs
    flow_cpu = configs.conditional_rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128,
                                               raw_context=5, context_features=12, seed=0).eval()
    x = bench_rows(32768)
    ctx = torch.randn(32768, 5, generator=torch.Generator().manual_seed(4321))
    check_flow("conditional_32layer_d64_k8_ctx12_b32768", flow_cpu, x, torch.arange(0, 32768, 4), context=ctx)


def test_conditional_ten_bin_flow_in_the_eight_wave_kernel():
    """Context + 10 bins at D = 64 (the reference's default bin count on a conditional flow): 2 080 parameter
    words per layer.  Round 2 gave every layer two 8 KB parameter blocks per parameter stage, which left no
    room for eight row tiles -- four-wave workgroups, one wave per SIMD; since round 3 a block takes the words it
    uses and the shape runs in the eight-wave kernel.  B = 65 536, 8 layers; the oracle visits every 16th row,
    the same rows evaluated alone (four-wave kernel) give the same bits."""
    from nflows_amd import configs
    flow_cpu = configs.conditional_rq_nsf_flow(num_layers=8, features=64, num_bins=10, hidden_features=128,
                                               raw_context=5, context_features=12, seed=0).eval()
    x = bench_rows(65536)
    ctx = torch.randn(65536, 5, generator=torch.Generator().manual_seed(4321))
    check_flow("conditional_8layer_d64_k10_ctx12_b65536", flow_cpu, x, torch.arange(0, 65536, 16), context=ctx)


def test_small_batch_kernel_on_a_32768_row_shard():
    """K8s, the 16-sample-tile form of the whole-layer kernel (csrc/rqs_resnet_f16s.hip), serves the batches that
    give a CU at most one 128-row block: config 4's per-GPU shard of an 8-GPU run (32 768 rows of bench.py's rank 0).
    log_prob, z and logabsdet against the oracle on every 8th row; inverse pass; rows evaluated alone (512 rows: the
    same kernel) bit-identical; a ragged batch; equal to K8h (NFA_K8S off) within the parity class; twice the same bits."""
    from nflows_amd import configs, ops
    import copy
    import nflows_amd
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = torch.randn(32768, 64, generator=torch.Generator().manual_seed(1234))
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    xd = x.to(DEV)
    assert ops.use_tile16(32768, 8, None, xd.device) == 1
    saved_c = ops.K8C_ENABLED
    ops.K8C_ENABLED = False       # (this test is K8s's: the 512- and 1 000-row launches below would be K8c's otherwise)
    with torch.no_grad():
        lp = flow.log_prob(xd)
        z, lad = flow._transform(xd)
        assert ops.last_redo_blocks() == 0
        lp2 = flow.log_prob(xd)
        xr, ladi = flow._transform.inverse(z)
        rows_solo = torch.arange(100, 32768, 64)
        zs, lads = flow._transform(xd[rows_solo.to(DEV)])
        zr, ladr = flow._transform(xd[:1000])                 # ragged: padded to full blocks inside ops
        saved = ops.K8S_ENABLED
        try:
            ops.K8S_ENABLED = False
            z_h, lad_h = flow._transform(xd)
        finally:
            ops.K8S_ENABLED = saved
    ops.K8C_ENABLED = saved_c
    nflows_amd.check_status()
    assert torch.equal(lp, lp2)
    assert torch.equal(zs, z[rows_solo.to(DEV)]) and torch.equal(lads, lad[rows_solo.to(DEV)])
    assert torch.equal(zr, z[:1000]), "ragged batch: %d outputs differ" % int((zr != z[:1000]).sum())
    assert torch.equal(ladr, lad[:1000]), "ragged batch: %d log-determinants differ, max %.3e" % (
        int((ladr != lad[:1000]).sum()), float((ladr - lad[:1000]).abs().max()))
    assert not torch.equal(z_h, z)                            # another kernel: other bits somewhere ...
    assert float((z_h - z).abs().max()) < 2e-3 and float((lad_h - lad).abs().max()) < 5e-3   # ... the same numbers
    # pass-through columns of the last layer are copies either way
    rows = torch.arange(0, 32768, 8)
    o = oracle_eval(flow_cpu, x[rows])
    idx = rows.to(DEV)
    compare("k8s_32layer_b32768", "z", z[idx].cpu().numpy(), o["z32"], o["z64"], OUT_TOL)
    compare("k8s_32layer_b32768", "logabsdet", lad[idx].cpu().numpy(), o["lad32"], o["lad64"], LAD_TOL)
    compare("k8s_32layer_b32768", "log_prob", lp[idx].cpu().numpy(), o["lp32"], o["lp64"], LAD_TOL)
    err = (xr - xd).abs()
    _report({"config": "k8s_32layer_b32768", "what": "|inv(fwd(x)) - x|", "max": float(err.max()), "mean": float(err.mean())})
    assert float(err.mean()) < 2e-5 and float(err.max()) < 2e-2


def test_small_batch_kernel_four_wave_blocks_on_a_16384_row_shard():
    """Batches with no more 64-row blocks than CUs run K8s with four-wave workgroups (one wave per SIMD, 64 rows):
    config 4's per-GPU shard of a 16-GPU run.  The oracle on every 4th row; the same bits as the eight-wave form
    (the rows as the first half of a 32 768-row batch); two workgroups share one redo flag: a batch whose rows
    64..127 leave the f16 range has exactly one block redone and rows 0..63 of it are still the oracle's."""
    from nflows_amd import configs, ops
    import copy
    import nflows_amd
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = torch.randn(32768, 64, generator=torch.Generator().manual_seed(1234))
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    xd = x.to(DEV)
    saved_c = ops.K8C_ENABLED
    ops.K8C_ENABLED = False       # (K8s's four-wave form: without this switch these batches are K8c's)
    with torch.no_grad():
        z_full, lad_full = flow._transform(xd)
        z, lad = flow._transform(xd[:16384])
        assert ops.last_redo_blocks() == 0
        lp = flow.log_prob(xd[:16384])
        xr, _ = flow._transform.inverse(z)
        x_hot = xd[:16384].clone()
        x_hot[64:128] *= 3e4                       # identity features of 3e4 .. 1e5: hidden activations beyond 65 504
        z_hot, lad_hot = flow._transform(x_hot)
        redo_hot = ops.last_redo_blocks()
    ops.K8C_ENABLED = saved_c
    nflows_amd.check_status()
    assert torch.equal(z, z_full[:16384]) and torch.equal(lad, lad_full[:16384]), "four- and eight-wave blocks differ"
    rows = torch.arange(0, 16384, 4)
    o = oracle_eval(flow_cpu, x[rows])
    idx = rows.to(DEV)
    compare("k8s_32layer_b16384", "z", z[idx].cpu().numpy(), o["z32"], o["z64"], OUT_TOL)
    compare("k8s_32layer_b16384", "logabsdet", lad[idx].cpu().numpy(), o["lad32"], o["lad64"], LAD_TOL)
    compare("k8s_32layer_b16384", "log_prob", lp[idx].cpu().numpy(), o["lp32"], o["lp64"], LAD_TOL)
    err = (xr - xd[:16384]).abs()
    assert float(err.mean()) < 2e-5 and float(err.max()) < 2e-2
    assert redo_hot == 1, "%d blocks redone" % redo_hot
    assert torch.equal(z_hot[128:], z[128:]) and torch.equal(lad_hot[128:], lad[128:])
    o_hot = oracle_eval(flow_cpu, x_hot[:128].cpu())
    compare("k8s_shared_redo_flag", "z", z_hot[:128].cpu().numpy(), o_hot["z32"], o_hot["z64"], OUT_TOL, max_factor=4.0)
    compare("k8s_shared_redo_flag", "logabsdet", lad_hot[:128].cpu().numpy(), o_hot["lad32"], o_hot["lad64"], LAD_TOL, max_factor=4.0)


def test_column_split_kernel_on_small_batches():
    """K8c (csrc/rqs_resnet_f16c.hip, round 6): K8s's whole-layer kernel with every GEMM split by columns over the four waves
    of a 64-row workgroup, weights straight from global memory into registers -- the form for batches with no more 64-row
    blocks than CUs (`Flow.sample(n)` / `log_prob` of a few thousand rows, a 16-GPU shard of config 4).  The same products in
    the same order as K8s: z is K8s's BIT FOR BIT (log-determinants are summed in another order: rounding only); the oracle
    on every 4th row; inverse; a ragged batch and rows evaluated alone give the same bits; workgroups are 32 rows and a
    128-row block's redo word carries a bit per quarter: a hot quarter is redone by the exact kernel, its three siblings keep
    their bits; D = 100 (two k-steps in the initial layer, 13 groups: a last round with idle waves)."""
    from nflows_amd import configs, ops
    import copy
    import nflows_amd
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = torch.randn(16384, 64, generator=torch.Generator().manual_seed(1234))
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    xd = x.to(DEV)
    assert ops.K8C_ENABLED and ops.use_tile16(16384, 8, None, xd.device) == 2 and ops.use_tile16(32768, 8, None, xd.device) == 1
    with torch.no_grad():
        z, lad = flow._transform(xd)
        label = ops.last_layer_kernel()
        assert ops.last_redo_blocks() == 0
        lp = flow.log_prob(xd)
        lp2 = flow.log_prob(xd)
        xr, _ = flow._transform.inverse(z)
        label_inv = ops.last_layer_kernel()
        rows_solo = torch.arange(100, 16384, 32)
        zs, lads = flow._transform(xd[rows_solo.to(DEV)])
        zr, ladr = flow._transform(xd[:1000])                 # ragged: padded to full blocks inside ops
        x_hot = xd.clone()
        x_hot[64:128] *= 3e4                       # identity features of 3e4 .. 1e5: hidden activations beyond 65 504
        z_hot, lad_hot = flow._transform(x_hot)
        redo_hot = ops.last_redo_blocks()
        saved = ops.K8C_ENABLED
        try:
            ops.K8C_ENABLED = False
            z_s, lad_s = flow._transform(xd)
            assert "k8s::" in ops.last_layer_kernel()
        finally:
            ops.K8C_ENABLED = saved
    nflows_amd.check_status()
    assert "k8c::" in label and "inverse=0" in label and "rows=32" in label and "k8c::" in label_inv and "inverse=1" in label_inv, (label, label_inv)
    assert torch.equal(lp, lp2)
    assert torch.equal(z, z_s), "%d outputs differ from K8s's" % int((z != z_s).sum())
    assert float((lad - lad_s).abs().max()) < 5e-4
    assert torch.equal(zs, z[rows_solo.to(DEV)]) and torch.equal(lads, lad[rows_solo.to(DEV)])
    assert torch.equal(zr, z[:1000]) and torch.equal(ladr, lad[:1000])
    rows = torch.arange(0, 16384, 4)
    o = oracle_eval(flow_cpu, x[rows])
    idx = rows.to(DEV)
    compare("k8c_32layer_b16384", "z", z[idx].cpu().numpy(), o["z32"], o["z64"], OUT_TOL)
    compare("k8c_32layer_b16384", "logabsdet", lad[idx].cpu().numpy(), o["lad32"], o["lad64"], LAD_TOL)
    compare("k8c_32layer_b16384", "log_prob", lp[idx].cpu().numpy(), o["lp32"], o["lp64"], LAD_TOL)
    err = (xr - xd).abs()
    _report({"config": "k8c_32layer_b16384", "what": "|inv(fwd(x)) - x|", "max": float(err.max()), "mean": float(err.mean())})
    assert float(err.mean()) < 2e-5 and float(err.max()) < 2e-2
    assert redo_hot == 1, "%d blocks redone" % redo_hot
    assert torch.equal(z_hot[128:], z[128:]) and torch.equal(lad_hot[128:], lad[128:])
    assert torch.equal(z_hot[:64], z[:64]) and torch.equal(lad_hot[:64], lad[:64])      # (quarters 0, 1 were not touched)
    o_hot = oracle_eval(flow_cpu, x_hot[:128].cpu())
    compare("k8c_shared_redo_flag", "z", z_hot[:128].cpu().numpy(), o_hot["z32"], o_hot["z64"], OUT_TOL, max_factor=4.0)
    compare("k8c_shared_redo_flag", "logabsdet", lad_hot[:128].cpu().numpy(), o_hot["lad32"], o_hot["lad64"], LAD_TOL, max_factor=4.0)
    # D = 100: 50 identity features (two 32-wide k-steps), 52 transformed after padding: 13 groups, four rounds
    flow_w = configs.rq_nsf_flow(num_layers=4, features=100, num_bins=8, hidden_features=128, seed=3).to(DEV).eval()
    xw = torch.randn(4100, 100, generator=torch.Generator().manual_seed(7)).to(DEV)
    with torch.no_grad():
        zw, ladw = flow_w._transform(xw)
        assert "k8c::" in ops.last_layer_kernel() and "init_ks=2" in ops.last_layer_kernel(), ops.last_layer_kernel()
        xwr, ladwr = flow_w._transform.inverse(zw)
        try:
            ops.K8C_ENABLED = False
            zw_s, ladw_s = flow_w._transform(xw)
        finally:
            ops.K8C_ENABLED = saved
    nflows_amd.check_status()
    assert torch.equal(zw, zw_s) and float((ladw - ladw_s).abs().max()) < 1e-4
    assert float((xwr - xw).abs().max()) < 1e-3 and float((ladw + ladwr).abs().max()) < 1e-3


def test_forward_inverse_consistency_against_the_reference():
    """Second half of the metric: max |inv(fwd(x)) - x| of the 32-layer composite on the 8 192 rows
    bench.py uses, next to the reference's own fp32 figure on the same rows and weights, and the
    inverse pass itself (z -> x) against the float64 inverse."""
    from nflows_amd import configs
    from oracle import eager
    import nflows_amd
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    xs = bench_rows()[:8192]
    with torch.no_grad():
        z32, _ = eager.flow_transform(flow_cpu, xs)
        xr32, lad_inv32 = eager.flow_transform(flow_cpu, z32, inverse=True)
        f64 = flow_cpu.double()
        xr64, lad_inv64 = eager.flow_transform(f64, z32.double(), inverse=True)  # truth of the inverse pass on z32
        flow_cpu.float()
    ref = (xr32 - xs).abs().numpy()
    import copy
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    with torch.no_grad():
        z, _ = flow._transform(xs.to(DEV))
        xr, _ = flow._transform.inverse(z)
        xi, lad_inv = flow._transform.inverse(z32.to(DEV))
    nflows_amd.check_status()
    got = (xr.cpu() - xs).abs().numpy()
    e_got, e_ref = _stats(got), _stats(ref)
    _report({"config": "cfg4_fwd_inv_8192_rows", "what": "|inv(fwd(x)) - x|", "hip": e_got, "reference_fp32": e_ref})
    floor = 4.0 * 2.0 ** -23 * float(xs.abs().max())
    for k in ("max", "mean", "q999"):
        assert e_got[k] <= FACTOR * e_ref[k] + (floor if k == "max" else 0.0), (
            "fwd/inv consistency: %s %.3e exceeds %.1f x the reference fp32's %.3e" % (k, e_got[k], FACTOR, e_ref[k]))
    compare("cfg4_inverse_pass_8192_rows", "x", xi.cpu().numpy(), xr32.numpy(), xr64.numpy(), OUT_TOL)
    compare("cfg4_inverse_pass_8192_rows", "logabsdet", lad_inv.cpu().numpy(), lad_inv32.numpy(), lad_inv64.numpy(), LAD_TOL)


def _redo_blocks():
    from nflows_amd import ops
    return ops.last_redo_blocks()


def test_bench_instance_262144_rows_log_prob():
    """The kernel instance bench.py times: Flow.log_prob of the 32-layer flow on bench.py's own 262 144 rows
    (8-wave workgroups, four row blocks per workgroup, standard-normal epilogue, z never written).  The
    oracle visits every 31st row (8 457 rows, every lane position of every wave, every row block); the same
    rows evaluated alone give the same bits; no row block is handed to the exact kernel."""
    from nflows_amd import configs
    import nflows_amd
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = torch.randn(262144, 64, generator=torch.Generator().manual_seed(1234))   # bench.py, rank 0
    import copy
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    xd = x.to(DEV)
    with torch.no_grad():
        lp = flow.log_prob(xd)
        redo = _redo_blocks()
        lp_again = flow.log_prob(xd)
    nflows_amd.check_status()
    assert redo == 0, "%d row blocks left the f16 range on the bench's own data" % redo
    assert torch.equal(lp, lp_again)
    rows = torch.arange(0, 262144, 31)
    o = oracle_eval(flow_cpu, x[rows])
    got = lp[rows.to(DEV)].cpu().numpy()
    compare("bench_instance_262144_rows", "log_prob", got, o["lp32"], o["lp64"], LAD_TOL)
    from nflows_amd import ops
    saved = ops.K8S_ENABLED
    try:
        ops.K8S_ENABLED = False               # (the same kernel family for the solo evaluation: see check_flow)
        with torch.no_grad():
            solo = flow.log_prob(xd[rows.to(DEV)])
    finally:
        ops.K8S_ENABLED = saved
    with torch.no_grad():
        z, lad = flow._transform(xd)          # the same launch geometry with outputs written
    assert torch.equal(solo, lp[rows.to(DEV)]), "rows are not independent of the batch"
    zs, lads = z[rows.to(DEV)].cpu().numpy(), lad[rows.to(DEV)].cpu().numpy()
    compare("bench_instance_262144_rows", "z", zs, o["z32"], o["z64"], OUT_TOL)
    compare("bench_instance_262144_rows", "logabsdet", lads, o["lad32"], o["lad64"], LAD_TOL)
    # the density epilogue adds -(1/2) sum z^2 - (D/2) log 2 pi to the same logabsdet
    lp_from_parts = (-0.5 * (z.double() ** 2).sum(1) - 32.0 * np.log(2 * np.pi) + lad.double())
    assert float((lp.double() - lp_from_parts).abs().max()) <= LAD_TOL * (1 + float(lp.abs().max()))


def test_whole_layer_kernel_is_bit_deterministic_across_fresh_flows():
    """K8h's correctness rests on counted waits (LDS-DMA ring paced by vmcnt(n), weight fragments awaited
    with lgkmcnt(2)): a wait that is one short shows up as a rare deviation, not as a wrong answer every
    time.  Twenty fresh flow objects (fresh packed streams at fresh addresses) on the bench's 262 144 rows,
    each evaluated twice, forward / inverse on 8 192 rows: every result bit-identical to the first."""
    import copy
    from nflows_amd import configs
    flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = torch.randn(262144, 64, generator=torch.Generator().manual_seed(1234)).to(DEV)
    xs = x[:8192]
    first = None
    deviations = []
    for it in range(20):
        flow = copy.deepcopy(flow_cpu).to(DEV).eval()
        with torch.no_grad():
            lp1 = flow.log_prob(x)
            lp2 = flow.log_prob(x)
            z, lad = flow._transform(xs)
            xr, ladi = flow._transform.inverse(z)
        out = (lp1, z, lad, xr, ladi)
        if first is None:
            first = tuple(t.clone() for t in out)
        if not torch.equal(lp1, lp2):
            deviations.append((it, "second call", int((lp1 != lp2).sum())))
        for name, a, b in zip(("log_prob", "z", "lad", "x", "lad_inv"), out, first):
            if not torch.equal(a, b):
                deviations.append((it, name, int((a != b).sum())))
    _report({"config": "k8h_determinism_20_fresh_flows_262144_rows", "deviations": deviations})
    assert not deviations, deviations


def test_config5_autoregressive_inverse_against_the_reference_loop(golden_dir):
    """configs[4]'s NAMED path: the inverse (sampling direction) of the autoregressive RQ layer at D = 784,
    H = 256, B = 4 096 -- degree-ordered evaluation + the persistent kernel K12 -- against the REAL reference's
    784-pass loop (autoregressive.py:43-52) in float32 and float64 on the first 64 rows: tests/golden/config5_inverse.npz
    (make_golden.py `cfg5`; until round 4 this test ran the loop through the port on the GPU box's CPU: 147 s of the
    suite).  The flow is rebuilt from its seed; the fixture's per-parameter checksums say it is the reference's."""
    from nflows_amd import configs
    import copy
    import nflows_amd
    g = np.load(os.path.join(golden_dir, "config5_inverse.npz"))
    flow_cpu = configs.ar_rq_flow(features=784, hidden_features=256, num_bins=8, tail_bound=3.0, seed=0).eval()
    sums = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in flow_cpu.state_dict().values()])
    assert sums.shape == g["cfg5/param_checksums"].shape and np.allclose(sums, g["cfg5/param_checksums"], rtol=1e-12, atol=0)
    z = torch.randn(4096, 784, generator=torch.Generator().manual_seed(4321))
    assert np.array_equal(z[:64].numpy(), g["cfg5/z"])
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    with torch.no_grad():
        x, lad = flow._transform.inverse(z.to(DEV))
        x_solo, lad_solo = flow._transform.inverse(z[:64].to(DEV))
    nflows_amd.check_status()
    assert torch.equal(x_solo, x[:64]) and torch.equal(lad_solo, lad[:64])
    compare("cfg5_ar_rq_d784_k8_b4096_inverse", "x", x[:64].cpu().numpy(), g["cfg5/inv_x"], g["cfg5/inv_x64"], OUT_TOL)
    compare("cfg5_ar_rq_d784_k8_b4096_inverse", "logabsdet", lad[:64].cpu().numpy(), g["cfg5/inv_lad"], g["cfg5/inv_lad64"], LAD_TOL)


def _spread_rows(weight, decades, gen):
    """rows of a weight matrix multiplied by 10^U(-decades/2, decades/2)"""
    with torch.no_grad():
        f = 10.0 ** ((torch.rand(weight.shape[0], generator=gen) - 0.5) * decades)
        weight.mul_(f[:, None] if weight.dim() == 2 else f)


@pytest.mark.parametrize("case", ["wide_weights", "large_activations"])
def test_f16_engine_over_a_wide_dynamic_range(case):
    """K8h splits every fp32 operand into two f16 pieces after a per-GEMM power-of-two scaling: weights far
    below their matrix's maximum keep fewer bits, activations beyond 65 504 overflow and must be caught.
    `wide_weights`: every conditioner matrix gets row magnitudes spread over four decades (1e4) and the
    matrices of one network differ by another two; inputs as in the bench.  `large_activations`: the same
    with identity features up to +-3e3 and hidden activations up to ~1e5, and a band of rows scaled down to
    1e-4.  Both against the float64 oracle with the 2 x rule; the number of row blocks handed to the exact
    kernel is reported, and in the first case must be zero."""
    import copy
    from nflows_amd import configs
    import nflows_amd
    gen = torch.Generator().manual_seed(99)
    flow_cpu = configs.rq_nsf_flow(num_layers=6, features=64, num_bins=8, hidden_features=128, seed=3).eval()
    for t in flow_cpu._transform._transforms:
        net = getattr(t, "transform_net", None)
        if net is None:
            continue
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
    B = 16384
    x = torch.randn(B, 64, generator=gen)
    if case == "large_activations":
        with torch.no_grad():
            x[:4096] *= 10.0 ** (torch.rand(4096, 1, generator=gen) * 3.0)        # up to ~3e3 (tails: identity)
            x[4096:6144] *= 1e-4
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    with torch.no_grad():
        z, lad = flow._transform(x.to(DEV))
        redo = _redo_blocks()
        lp = flow.log_prob(x.to(DEV))
    nflows_amd.check_status()
    rows = torch.arange(0, B, 2)
    o = oracle_eval(flow_cpu, x[rows])
    _report({"config": "f16_engine_" + case, "redo_blocks": redo, "of": B // 128,
             "max_abs_z": float(z.abs().max()), "max_abs_lad": float(lad.abs().max())})
    idx = rows.to(DEV)
    # (mean and 99.9 % quantile at the 2 x rule; the single worst element of these deliberately ill-conditioned
    #  networks -- the reference's own fp32 result is off by 0.05 .. 0.5 there -- at 4 x, the worst-case rule of
    #  tests/helpers.py)
    compare("f16_engine_" + case, "z", z[idx].cpu().numpy(), o["z32"], o["z64"], OUT_TOL, max_factor=4.0)
    compare("f16_engine_" + case, "logabsdet", lad[idx].cpu().numpy(), o["lad32"], o["lad64"], LAD_TOL, max_factor=4.0)
    compare("f16_engine_" + case, "log_prob", lp[idx].cpu().numpy(), o["lp32"], o["lp64"], LAD_TOL, max_factor=4.0)
    if case == "wide_weights":
        assert redo == 0, "%d row blocks left the f16 range with moderate activations" % redo
    else:
        assert redo < B // 128, "every block was redone: the f16 engine handled nothing"


def _batch_norm_flow(seed=11, layers=6, features=64, dropout=0.2):
    """RQ-NSF flow whose ResidualNet conditioners use batch norm (resnet.py:24-27) and dropout, with running
    statistics, gains and shifts away from their initial values."""
    from nflows_amd.flows.base import Flow
    from nflows_amd.distributions.normal import StandardNormal
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import (CompositeTransform, RandomPermutation,
                                       PiecewiseRationalQuadraticCouplingTransform as RQ)
    from nflows_amd.utils.torchutils import create_alternating_binary_mask
    torch.manual_seed(seed)
    parts = []
    for i in range(layers):
        parts.append(RandomPermutation(features))
        parts.append(RQ(create_alternating_binary_mask(features, even=(i % 2 == 0)),
                        lambda a, b: ResidualNet(a, b, hidden_features=128, num_blocks=2, use_batch_norm=True,
                                                 dropout_probability=dropout),
                        num_bins=8, tails="linear", tail_bound=3.0))
    flow = Flow(CompositeTransform(parts), StandardNormal([features]))
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in flow.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 1.5 + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
        for name, p in flow.named_parameters():
            if "final_layer" in name:
                p.mul_(4.0)
            elif "linear_layers.1" in name:
                p.mul_(30.0)
    return flow.eval()


@pytest.mark.parametrize("engine", ["f16x2", "bf16x3"])
def test_batch_norm_conditioners_run_in_the_whole_layer_kernels(monkeypatch, engine):
    """Eval-mode batch norm (and dropout) in the ResidualNet blocks is folded into the packed weights and biases
    (ops.fold_batch_norm): the flow still runs as ONE launch, and agrees with the float64 evaluation of the
    reference sequence (batch norm as torch applies it) within twice the reference's own fp32 error.  The fold
    follows the running statistics (a buffer update is seen), does not exist for a negative gain in front of the
    ReLU or in training mode -- those run layer by layer, with the same results."""
    import copy
    from nflows_amd import ops
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    import nflows_amd
    monkeypatch.setattr(RQ, "conditioner_engine", engine)
    flow_cpu = _batch_norm_flow()
    x_cpu = torch.randn(4096, 64, generator=torch.Generator().manual_seed(3))

    def check(f_cpu, expect_run, what):
        flow = copy.deepcopy(f_cpu).to(DEV).eval() if not isinstance(f_cpu, tuple) else f_cpu[0]
        layers = list(flow._transform._transforms)
        x = x_cpu.to(DEV)
        calls = []
        real = ops.rqs_coupling_resnet_f16 if engine == "f16x2" else ops.rqs_coupling_resnet
        name = "rqs_coupling_resnet_f16" if engine == "f16x2" else "rqs_coupling_resnet"
        monkeypatch.setattr(ops, name, lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.no_grad():
            units, _ = flow._transform._collect_run(layers, 0, x, None, inverse=False)
            assert bool(units) == expect_run, what
            z, lad = flow._transform(x)
            xr, _ = flow._transform.inverse(z)
        monkeypatch.setattr(ops, name, real)
        nflows_amd.check_status()
        if expect_run:
            assert len(calls) == 2, "forward and inverse: one launch each, got %d" % len(calls)
        o = oracle_eval(f_cpu if not isinstance(f_cpu, tuple) else f_cpu[1], x_cpu)
        config = "batch_norm_%s_%s" % (engine, what.replace(" ", "_"))
        compare(config, "z", z.cpu().numpy(), o["z32"], o["z64"], OUT_TOL, max_factor=4.0)
        compare(config, "logabsdet", lad.cpu().numpy(), o["lad32"], o["lad64"], LAD_TOL, max_factor=4.0)
        assert (xr - x).abs().max().item() < 5e-3 and (xr - x).abs().mean().item() < 2e-5
        return flow

    flow = check(flow_cpu, True, "batch norm folded")
    # the running statistics move (a training step elsewhere): the packed weights follow
    with torch.no_grad():
        for f in (flow, flow_cpu):
            bn = list(f._transform._transforms)[1].transform_net.blocks[0].batch_norm_layers[0]
            bn.running_mean.add_(0.25)
            bn.running_var.mul_(1.5)
    check((flow, flow_cpu), True, "after a statistics update")
    # a negative gain in front of the ReLU: no fold, layer by layer
    neg = copy.deepcopy(flow_cpu)
    with torch.no_grad():
        list(neg._transform._transforms)[3].transform_net.blocks[1].batch_norm_layers[0].weight[5] = -0.7
    check(neg, False, "negative gain")
    # training mode: batch statistics, active dropout -- never the whole-layer kernel
    flow.train()
    with torch.no_grad():
        units, _ = flow._transform._collect_run(list(flow._transform._transforms), 0, x_cpu.to(DEV), None, inverse=False)
    assert not units
