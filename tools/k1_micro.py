#!/usr/bin/env python3
"""Micro-benchmark of the K1 kernel alone (fused RQ coupling layer, BASELINE layer shape):
B=65536, D=64, d_t=32, K=8, linear tails.  Prints avg us per launch and algorithmic GB/s.

    NFLOWS_AMD_LIB=/path/to/variant.so python tools/k1_micro.py [--inverse] [--batch B] [--reps N]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--features", type=int, default=64)
ap.add_argument("--bins", type=int, default=8)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--inverse", action="store_true")
ap.add_argument("--perm", action="store_true")
a = ap.parse_args()

dev = "cuda:0"
B, D, K = a.batch, a.features, a.bins
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, D, device=dev, generator=g)
tidx = torch.arange(0, D, 2, device=dev)
P = 3 * K - 1
# distinct parameter buffers per rep so that nothing is served from the 256 MiB Infinity Cache
nbuf = 4
params = [torch.randn(B, tidx.numel() * P, device=dev, generator=g) for _ in range(nbuf)]
perm = torch.randperm(D, device=dev) if a.perm else None
spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(128)))
for i in range(5):
    y, lad = ops.rqs_coupling(x, params[i % nbuf], tidx, spec, inverse=a.inverse, in_perm=perm)
torch.cuda.synchronize()
evs = []
for i in range(a.reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    y, lad = ops.rqs_coupling(x, params[i % nbuf], tidx, spec, inverse=a.inverse, in_perm=perm)
    e.record()
    evs.append((s, e))
torch.cuda.synchronize()
raw = [s.elapsed_time(e) for s, e in evs]
if os.environ.get("K1_MICRO_PER_BUFFER"):
    for b in range(nbuf):
        sub = sorted(raw[b::nbuf])
        print("   buffer %d (%x): median %.1f us min %.1f max %.1f" % (b, params[b].data_ptr(), sub[len(sub) // 2] * 1e3, sub[0] * 1e3, sub[-1] * 1e3))
    print("   first 12 reps:", " ".join("%.0f" % (v * 1e3) for v in raw[:12]))
ms = sorted(raw)
med = ms[len(ms) // 2]
nbytes = 4 * (B * D + B * tidx.numel() * P + B * D + B)
print("%s lib=%s %s B=%d  median %.1f us  min %.1f  p10 %.1f  p90 %.1f us  -> %.0f GB/s algorithmic (median)  [%s]" % (
    "inverse" if a.inverse else "forward", os.path.basename(os.environ.get("NFLOWS_AMD_LIB", "default")),
    " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("NFA_K1")), B,
    med * 1e3, ms[0] * 1e3, ms[len(ms) // 10] * 1e3, ms[(9 * len(ms)) // 10] * 1e3, nbytes / (med * 1e-3) / 1e9,
    ops.last_layer_kernel()))
