import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs, ops
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval().cuda()
x = torch.randn(32768, 64, generator=torch.Generator().manual_seed(1234)).cuda()


def diff(tag, a, b):
    d = (a != b)
    rows = d.any(1).nonzero().flatten() if d.dim() == 2 else d.nonzero().flatten()
    print("%s: %d differ, rows %s (blocks %s), max %.3e" % (tag, int(d.sum()), rows[:6].tolist(), sorted(set((rows // 128).tolist()))[:8],
                                                          float((a - b).abs().max())), flush=True)


with torch.no_grad():
    z, lad = flow._transform(x)
    print("redo", ops.last_redo_blocks())
    for i in range(6):
        z2, lad2 = flow._transform(x)
        r = ops.last_redo_blocks()
        diff("fwd repeat %d (redo %s) z" % (i, r), z2, z)
        diff("fwd repeat %d lad" % i, lad2, lad)
    xr, ladi = flow._transform.inverse(z)
    print("inverse redo", ops.last_redo_blocks(), "fwd-inv max err", float((xr - x).abs().max()))
    for i in range(3):
        xr2, ladi2 = flow._transform.inverse(z)
        diff("inv repeat %d x" % i, xr2, xr)
    for i in range(3):
        z2, lad2 = flow._transform(x)
        diff("fwd after inverse %d (redo %s) z" % (i, ops.last_redo_blocks()), z2, z)
