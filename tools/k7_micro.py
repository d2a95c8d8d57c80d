#!/usr/bin/env python3
"""Micro-benchmark of K7 (final Linear + spline layer in one kernel) at the BASELINE layer shape."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops
dev = "cuda:0"
B, D, K, H = 65536, 64, 8, 128
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, D, device=dev, generator=g)
tidx = torch.arange(0, D, 2, device=dev)
hid = [torch.randn(B, H, device=dev, generator=g) for _ in range(3)]
W = torch.randn(32 * 23, H, device=dev, generator=g) * 0.05
b = torch.randn(32 * 23, device=dev, generator=g) * 0.1
ENGINE = os.environ.get("NFA_K7_ENGINE", "bf16x3")
wp, bp = ops.pack_final_linear(W, b, 32, 23, split_bf16=(ENGINE == "bf16x3"))
spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(H)))
for i in range(3):
    ops.rqs_coupling_fused_linear(x, hid[i % 3], wp, bp, tidx, spec)
torch.cuda.synchronize()
evs = []
for i in range(30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y, lad = ops.rqs_coupling_fused_linear(x, hid[i % 3], wp, bp, tidx, spec); e.record()
    evs.append((s, e))
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in evs)
med = ms[len(ms) // 2]
print("K7 engine=%s lib=%s median %.1f us min %.1f us -> %.1f TFLOP/s (2*B*128*736)" % (
    ENGINE, os.path.basename(os.environ.get("NFLOWS_AMD_LIB", "default")), med * 1e3, ms[0] * 1e3, 2.0 * B * H * 736 / med / 1e9))
if "--check" in sys.argv:
    h = hid[(30 - 1) % 3]
    params = torch.addmm(b, h, W.t())
    y0, l0 = ops.rqs_coupling(x, params, tidx, spec)
    print("  vs GEMM+K1: max |dy| %.2e  max |dlad| %.2e" % ((y - y0).abs().max().item(), (lad - l0).abs().max().item()))
    # against parameters computed in float64 (rounded to fp32 once): which path is closer?
    p64 = torch.addmm(b.double(), h.double(), W.double().t()).float()
    y1, l1 = ops.rqs_coupling(x, p64, tidx, spec)
    print("  vs fp64-GEMM+K1: K7 max |dy| %.2e |dlad| %.2e   fp32-GEMM+K1 max |dy| %.2e |dlad| %.2e" % (
        (y - y1).abs().max().item(), (lad - l1).abs().max().item(),
        (y0 - y1).abs().max().item(), (l0 - l1).abs().max().item()))
