"""Stress: the bench sequence (full-batch launches, then forward/inverse on 8 192 rows) repeated in
one process with fresh flow objects; reports every deviation from the first outcome."""
import os, sys, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs, parallel
flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
x = torch.randn(65536, 64, generator=torch.Generator().manual_seed(1234)).cuda()
xs = x[:8192]
ref = None
bad = 0
reps = int(os.environ.get("REPS", "12"))
for it in range(reps):
    flow = copy.deepcopy(flow_cpu).cuda()
    with torch.no_grad():
        for _ in range(6):
            lp = flow.log_prob(x)
        z, lad = flow._transform(xs)
        xr, ladi = flow._transform.inverse(z)
        layer = flow._transform._transforms[1]
        y1, _ = layer(xs); x1, _ = layer.inverse(y1)
    out = (lp.double().sum().item(), (xr - xs).abs().max().item(), (x1 - xs).abs().max().item(), z.double().sum().item(), xr.double().sum().item())
    if ref is None:
        ref = out
    if out != ref:
        bad += 1
        print("iteration %d deviates:" % it, out, "vs", ref)
print("stress: %d of %d iterations deviate; first outcome %s" % (bad, reps, ref))
