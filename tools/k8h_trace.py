#!/usr/bin/env python3
"""Phase timeline of K8h (debug): cycle-counter stamps of wave 0 of every workgroup over the first layers
of its first row block.  Stamps per layer: start | initial layer done | per block: first / second Linear
done | final layer done."""
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
t = t[t[:, 1] > 0]                      # workgroups that ran (8-wave workgroups: B / 256 of them)
names = ["param+gather+init", "b0.L0", "b0.L1", "b1.L0", "b1.L1", "final"]
per = len(names) + 1
layers = (62 // per)
s = t[:, 1:1 + layers * per].reshape(t.shape[0], layers, per).astype(np.float64)
d = np.diff(s, axis=2)                   # [wg, layer, phase]
print("workgroups traced: %d; layers per trace: %d" % (t.shape[0], layers))
print("mean cycles per phase over workgroups and layers 1..%d:" % (layers - 1))
m = d[:, 1:, :].mean(axis=(0, 1))
for n, v in zip(names, m):
    print("  %-20s %8.0f  (%4.1f %%)" % (n, v, 100 * v / m.sum()))
gap = (s[:, 1:, 0] - s[:, :-1, -1]).mean()
print("  %-20s %8.0f" % ("layer-to-layer gap", gap))
print("layer total %.0f cycles (mean), min %.0f max %.0f" % (m.sum() + gap, (d[:, 1:, :].sum(axis=2) ).min(), (d[:, 1:, :].sum(axis=2)).max()))
