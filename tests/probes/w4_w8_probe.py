"""Round 5: the 65 536-row comparisons show the four-wave form of K8h (16 384-row launches) with 2 x the error of the
eight-wave form on the tanh / 10-bin fixture.  Same instruction stream per wave: the two must agree bit for bit on the
same rows.  Where do they differ, is it deterministic, which instances?  Usage: python tests/probes/w4_w8_probe.py"""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import steep_flow
from test_gpu_steep import _batch
from nflows_amd import ops
DEV = "cuda:0"
golden = os.path.join(ROOT, "tests", "golden")

def chunks(fn, t, rows):
    outs = [fn(t[i:i + rows].to(DEV)) for i in range(0, t.shape[0], rows)]
    return tuple(torch.cat([o[j] for o in outs], 0) for j in range(2))

for case, fixture in (("act_tanh_k10", "flows_acts.npz"), ("act_tanh_k8", "flows_acts.npz"), ("act_elu_k10", "flows_acts.npz"),
                      ("act_leaky_relu_k8", "flows_acts.npz"), ("steep_nsf_k10", "flows_steep.npz"), ("steep_nsf_k8", "flows_steep.npz"),
                      ("bins_k9", "flows_bins.npz")):
    flow_cpu, g, cfg = steep_flow(golden, case, fixture)
    x = _batch(g, case, "x", 65536, cfg["D"])
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    saved = ops.K8S_ENABLED
    ops.K8S_ENABLED = False
    with torch.no_grad():
        z8, l8 = flow._transform(x.to(DEV)); k8 = ops.last_layer_kernel()
        z4, l4 = chunks(flow._transform, x, 16384); k4 = ops.last_layer_kernel()
        z4b, l4b = chunks(flow._transform, x, 16384)
        z2, l2 = chunks(flow._transform, x, 8192)
        z8b, l8b = flow._transform(x.to(DEV))
    ops.K8S_ENABLED = saved
    d = (z8 != z4)
    rows_bad = d.any(1).nonzero().flatten()
    print(case, "|", k8.split("<")[1][:60], "|", k4.split("<")[1][:60])
    print("   w8 vs w4: elements differing", int(d.sum()), "rows", int(rows_bad.numel()), "max |diff| %.3e" % float((z8 - z4).abs().max()),
          "| per 16384-chunk rows:", [int(((rows_bad >= i) & (rows_bad < i + 16384)).sum()) for i in range(0, 65536, 16384)])
    print("   w4 run-to-run equal:", bool(torch.equal(z4, z4b) and torch.equal(l4, l4b)), "| w8 run-to-run equal:", bool(torch.equal(z8, z8b)),
          "| w4(16384) vs w4(8192) equal:", bool(torch.equal(z4, z2)), "| redo", ops.last_redo_blocks())
    if rows_bad.numel():
        r = rows_bad[:5].tolist()
        print("   first differing rows", r, "row %% 128:", [i % 128 for i in r], "row %% 32:", [i % 32 for i in r])
