import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
x = torch.randn(65536, 64, generator=torch.Generator().manual_seed(1234)).cuda()
xs = x[:8192]
layer = flow._transform._transforms[1]
for engine in ("bf16x3", "f16x2"):
    RQ.conditioner_engine = engine
    with torch.no_grad():
        y1, l1 = layer(xs)
        x1, l2 = layer.inverse(y1)
    e = (x1 - xs).abs()
    i = np.unravel_index(e.argmax().item(), e.shape)
    print(engine, "single layer max err %.3e at %s: x %.7f y %.7f x' %.7f  lad sum err %.2e; count>2e-6: %d" % (
        e.max().item(), i, xs[i].item(), y1[i].item(), x1[i].item(), (l1 + l2).abs().max().item(), (e > 2e-6).sum().item()))
    bad = (e > 2e-6).nonzero()[:10].cpu().numpy()
    for r, c in bad:
        print("   row %d col %d: x %.7f y %.7f x' %.7f" % (r, c, xs[r, c].item(), y1[r, c].item(), x1[r, c].item()))
print("--- after full-batch log_prob calls (bench's sequence)")
RQ.conditioner_engine = "f16x2"
with torch.no_grad():
    for _ in range(3):
        flow.log_prob(x)
    y1, l1 = layer(xs)
    x1, l2 = layer.inverse(y1)
    e = (x1 - xs).abs()
    print("single layer max err %.3e; count>2e-6: %d; rows affected: %s" % (e.max().item(), (e > 2e-6).sum().item(), torch.unique((e > 2e-6).nonzero()[:, 0] // 128).cpu().numpy()[:20]))
    z, _ = flow._transform(xs)
    xr, _ = flow._transform.inverse(z)
    print("composite max err %.3e" % (xr - xs).abs().max().item())
