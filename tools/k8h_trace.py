#!/usr/bin/env python3
"""Phase timeline of K8h waves (debug): cycle-counter stamps of wave 0 of every workgroup over the
first two layers; workgroups sharing a CU are paired by HW_ID."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import _native, configs
dev = "cuda:0"
B = int(os.environ.get("B", "65536"))
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).to(dev).eval()
x = torch.randn(B, 64, device=dev)
with torch.no_grad():
    for _ in range(3): flow._transform(x)
    nb = B // 128
    buf = torch.zeros(nb * 64, dtype=torch.int64, device=dev)
    lib = _native.load(); lib.nfa_debug_k7_trace.argtypes = [ctypes.c_void_p]
    lib.nfa_debug_k7_trace(ctypes.c_void_p(buf.data_ptr()))
    flow._transform(x)
    torch.cuda.synchronize(); lib.nfa_debug_k7_trace(None)
t = buf.cpu().numpy().reshape(nb, 64)
hw = t[:, 0] & 0xffffffff
xcc = (t[:, 0] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
key = xcc * 1000 + se * 100 + sh * 50 + cu
names = ["ident", "init", "b0.L0", "b0.L1", "b1.L0", "b1.L1"] + ["g%d" % g for g in range(8)] + ["drain"]
per_layer = len(names) + 1
def show(b):
    s = t[b, 1:1 + 2 * per_layer]
    print("  wg %3d (xcc %d se %d cu %2d simd %d) layer0 start +%6d:" % (b, xcc[b], se[b], cu[b], simd[b], s[0] - t[:, 1].min()),
          " ".join("%s %d" % (n, d) for n, d in zip(names, np.diff(s[:per_layer]))), "| layer total", s[per_layer] - s[0], "| layer1 total", s[2 * per_layer - 1] - s[per_layer])
pairs = {}
for b in range(nb):
    pairs.setdefault(int(key[b]), []).append(b)
shown = 0
for k, bs in sorted(pairs.items()):
    if len(bs) >= 2 and shown < 3:
        print("CU key", k, "workgroups", bs)
        for b in bs: show(b)
        shown += 1
tot = t[:, 1 + per_layer] - t[:, 1]
print("layer-0 duration over workgroups: mean %.0f min %d max %d cycles" % (tot.mean(), tot.min(), tot.max()))
d = np.diff(t[:, 1:1 + per_layer], axis=1)
print("mean phase durations:", " ".join("%s %.0f" % (n, v) for n, v in zip(names, d.mean(axis=0))))
if t[:, 36:60].any():
    f = t[:, 36:60].reshape(nb, 8, 3)
    ok = f[:, 0, 0] > 0
    f = f[ok]
    print("fine trace, block 0 second Linear, per k-step [request+reads .. MFMAs done | barrier wait]: ",
          " ".join("%d|%d" % ((f[:, k, 1] - f[:, k, 0]).mean(), (f[:, k, 2] - f[:, k, 1]).mean()) for k in range(8)),
          " gaps between k-steps:", " ".join("%d" % (f[:, k + 1, 0] - f[:, k, 2]).mean() for k in range(7)))
