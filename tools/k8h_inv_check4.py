import os, sys, copy
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nflows_amd
from nflows_amd import configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
x = torch.randn(65536, 64, generator=torch.Generator().manual_seed(1234)).cuda()
xs = x[:8192]
layer = flow._transform._transforms[1]
with torch.no_grad():
    for _ in range(3):
        flow.log_prob(x)          # NW = 8 kernels first (bench's order)
    y1, _ = layer(xs); x1, _ = layer.inverse(y1)
    print("f16x2 after 65536-row launches: single layer %.3e" % (x1 - xs).abs().max().item())
    RQ.conditioner_engine = "bf16x3"
    yb, _ = layer(xs); xb, _ = layer.inverse(y1)
    print("  forward vs bf16x3 %.3e   inverse(of the same y) vs bf16x3 %.3e" % ((y1 - yb).abs().max().item(), (x1 - xb).abs().max().item()))
    RQ.conditioner_engine = "f16x2"
    y2, _ = layer(xs); x2, _ = layer.inverse(y2)
    print("  second call: single layer %.3e; forward repeat-identical %s inverse %s" % ((x2 - xs).abs().max().item(), torch.equal(y1, y2), torch.equal(x1, x2)))
    e = (x1 - xb).abs().max(dim=1).values
    print("  rows with inverse diff > 3e-6: %d, blocks %s" % ((e > 3e-6).sum().item(), torch.unique((e > 3e-6).nonzero()[:, 0] // 128).cpu().numpy()[:16]))
