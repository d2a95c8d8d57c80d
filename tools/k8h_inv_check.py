#!/usr/bin/env python3
"""Debug aid: K8h inverse against the bf16x3 engine and forward/inverse consistency, per batch size."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
for layers in (1, 32):
    flow = configs.rq_nsf_flow(num_layers=layers, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
    for B in (8192, 65536):
        x = torch.randn(B, 64, generator=torch.Generator().manual_seed(1)).cuda()
        res = {}
        for engine in ("bf16x3", "f16x2"):
            RQ.conditioner_engine = engine
            with torch.no_grad():
                z, lad = flow._transform(x)
                xi, ladi = flow._transform.inverse(x)
                xr, _ = flow._transform.inverse(z)
            res[engine] = (z, xi, (xr - x).abs().max().item())
        print("layers %2d B %6d: fwd diff %.2e  inv diff %.2e  | fwd/inv consistency bf16x3 %.2e  f16x2 %.2e" % (
            layers, B, (res["bf16x3"][0] - res["f16x2"][0]).abs().max().item(), (res["bf16x3"][1] - res["f16x2"][1]).abs().max().item(),
            res["bf16x3"][2], res["f16x2"][2]))
