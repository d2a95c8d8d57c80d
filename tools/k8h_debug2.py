#!/usr/bin/env python3
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
layers = int(os.environ.get("LAYERS", "2"))
B = int(os.environ.get("B", "65536"))
flow = configs.rq_nsf_flow(num_layers=layers, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
x = torch.randn(B, 64, generator=torch.Generator().manual_seed(1)).cuda()
RQ.conditioner_engine = "bf16x3"
with torch.no_grad():
    zr, lr = flow._transform(x)
zr, lr = zr.cpu().numpy(), lr.cpu().numpy()
RQ.conditioner_engine = "f16x2"
for attempt in range(6):
    with torch.no_grad():
        z, l = flow._transform(x)
    z, l = z.cpu().numpy(), l.cpu().numpy()
    d = np.abs(z - zr).max(axis=1)
    bad = np.nonzero(d > 1e-4)[0]
    if bad.size:
        break
print("attempt", attempt, "bad rows", bad.size)
if bad.size:
    w0 = bad[0] // 32 * 32
    rows = np.arange(w0, w0 + 32)
    print("wave rows", w0, "..", w0 + 31)
    perm_first = None
    for t in flow._transform._transforms:
        if type(t).__name__.endswith("Permutation"):
            perm_first = t._permutation.cpu().numpy(); break
    for r_ in rows:
        nbad = int((np.abs(z[r_] - zr[r_]) > 1e-4).sum())
        # does the bad row equal another row's reference result?
        match = ""
        if nbad:
            cand = np.nonzero(np.abs(zr[w0 - 96: w0 + 128] - z[r_]).max(axis=1) < 1e-4)[0]
            match = " == ref rows %s" % (cand + w0 - 96) if cand.size else ""
        print("row %6d (r=%2d): cols off %2d  max|dz| %.2e  dlad %.2e%s" % (r_, r_ % 32, nbad, np.abs(z[r_] - zr[r_]).max(), abs(l[r_] - lr[r_]), match))
    r_ = bad[0]
    cols = np.nonzero(np.abs(z[r_] - zr[r_]) > 1e-4)[0]
    print("bad cols of first bad row:", cols)
    print("z   :", z[r_, cols[:8]])
    print("zref:", zr[r_, cols[:8]])
    print("x   :", x[r_].cpu().numpy()[cols[:8]])
