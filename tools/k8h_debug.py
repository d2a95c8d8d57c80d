#!/usr/bin/env python3
"""Debug aid: K8h (f16x2 engine) against the bf16x3 engine over batch sizes; which rows differ."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ

layers = int(os.environ.get("LAYERS", "4"))
flow = configs.rq_nsf_flow(num_layers=layers, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
for B in [int(b) for b in os.environ.get("BATCHES", "8192,32768,32896,65536,131072").split(",")]:
    x = torch.randn(B, 64, generator=torch.Generator().manual_seed(1)).cuda()
    out = {}
    for engine in ("bf16x3", "f16x2", "f16x2"):
        RQ.conditioner_engine = engine
        with torch.no_grad():
            z, lad = flow._transform(x)
        torch.cuda.synchronize()
        out.setdefault(engine, []).append((z.cpu().numpy(), lad.cpu().numpy()))
    za, zb = out["bf16x3"][0][0], out["f16x2"][0][0]
    zc = out["f16x2"][1][0]
    d = np.abs(za - zb).max(axis=1)
    badrows = np.nonzero(d > 1e-4)[0]
    print("B=%6d max|diff| %.3e  rows off %d  repeat-identical %s" % (B, d.max(), badrows.size, np.array_equal(zb, zc)))
    if badrows.size:
        quads = np.unique(badrows // 128)
        waves = np.unique((badrows % 128) // 32)
        print("   quads off: %d of %d (first %s)  waves %s  rows-in-wave sample %s" % (quads.size, B // 128, quads[:12], waves, (badrows % 32)[:16]))
        cols = np.nonzero(np.abs(za - zb)[badrows[0]] > 1e-4)[0]
        print("   first bad row %d: cols %s" % (badrows[0], cols[:20]))
