"""Round 5: replay of tests/test_gpu_activations.py's sequence for act_tanh_k10 (oracle on the device first, then the
eight-wave engine on one flow object, then the four-wave engine in chunks on another), comparing the two engines' z."""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import steep_flow, eager_oracle
from test_gpu_steep import _batch, _chunked
from nflows_amd import ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
golden = os.path.join(ROOT, "tests", "golden")
case = sys.argv[1] if len(sys.argv) > 1 else "act_tanh_k10"
flow_cpu, g, cfg = steep_flow(golden, case, "flows_acts.npz")
x = _batch(g, case, "x", 65536, cfg["D"])
noise = _batch(g, case, "noise", 65536, cfg["D"])
use_oracle = os.environ.get("PROBE_ORACLE", "1") == "1"
if use_oracle:
    o = eager_oracle(flow_cpu, x, noise, fp64_device=DEV)
res = {}
for engine, rows in (("w8", 65536), ("w4", 16384), ("w4_again", 16384), ("w8_again", 65536)):
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    with torch.no_grad():
        z, lad = _chunked(flow._transform, x, rows)
        k = ops.last_layer_kernel()
        lp = _chunked(flow.log_prob, x, rows)
        xi, ladi = _chunked(flow._transform.inverse, noise, rows)
    res[engine] = (z.cpu(), lad.cpu(), lp.cpu(), xi.cpu(), ladi.cpu())
    print(engine, k.split("<")[1][:70], "redo", ops.last_redo_blocks())
    if use_oracle:
        e = np.abs(z.cpu().numpy().astype(np.float64) - o["z64"])
        er = np.abs(o["z32"].astype(np.float64) - o["z64"])
        print("    z mean err %.3e (ref %.3e) max %.3e (ref %.3e)" % (e.mean(), er.mean(), e.max(), er.max()))
for a, b in (("w8", "w4"), ("w4", "w4_again"), ("w8", "w8_again")):
    for i, name in enumerate(("z", "lad", "lp", "xi", "ladi")):
        ta, tb = res[a][i], res[b][i]
        d = (ta != tb)
        if d.any():
            rows_bad = (d.any(1) if d.dim() == 2 else d).nonzero().flatten()
            print(a, "vs", b, name, "differ:", int(d.sum()), "elements in", int(rows_bad.numel()), "rows; first", rows_bad[:8].tolist(),
                  "max |diff| %.3e" % float((ta - tb).abs().max()))
        else:
            print(a, "vs", b, name, "identical")
