#!/usr/bin/env python3
"""Phase timeline of one K7 wave (debug): cycles between s_memtime stamps."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops, _native
dev = "cuda:0"
B, D, K, H = 65536, 64, 8, 128
x = torch.randn(B, D, device=dev); tidx = torch.arange(0, D, 2, device=dev)
hid = torch.randn(B, H, device=dev)
W = torch.randn(32 * 23, H, device=dev) * 0.05; b = torch.randn(32 * 23, device=dev) * 0.1
ENGINE = os.environ.get("NFA_K7_ENGINE", "bf16x3")
wp, bp = ops.pack_final_linear(W, b, 32, 23, split_bf16=(ENGINE == "bf16x3"))
spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(H)))
for _ in range(3): ops.rqs_coupling_fused_linear(x, hid, wp, bp, tidx, spec)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
lib = _native.load(); lib.nfa_debug_k7_trace.argtypes = [ctypes.c_void_p]
lib.nfa_debug_k7_trace(ctypes.c_void_p(buf.data_ptr()))
ops.rqs_coupling_fused_linear(x, hid, wp, bp, tidx, spec)
torch.cuda.synchronize(); lib.nfa_debug_k7_trace(None)
t = buf.cpu().numpy()
if ENGINE == "bf16x3":
    base = min(t[0], t[256])
    for blk in (0, 1):
        s = t[blk * 256: blk * 256 + 250]; s = s[s > 0]
        d = np.diff(s)
        print("block %d: %d stamps, first at +%d, total %d cycles" % (blk * 256, len(s), s[0] - base, s[-1] - s[0]))
        print("  hidden load + split:", d[0])
        per_group = d[1:1 + 8 * 7].reshape(8, 7)
        print("  per group [mfma0, stage0, mfma1, stage1, mfma2, stage2, spline] (stage = LDS write of the next tile + barrier):")
        for g in range(8): print("   ", per_group[g].tolist())
        print("  output assembly:", d[1 + 56:].tolist())
        print("  sums: mfma %d  stage %d  spline %d" % (per_group[:, 0:6:2].sum(), per_group[:, 1:6:2].sum(), per_group[:, 6].sum()))
    sys.exit(0)
for blk in (0, 1):
    s = t[blk * 256: blk * 256 + 250]; s = s[s > 0]
    d = np.diff(s)
    print("block %d: %d stamps, total %d cycles" % (blk * 256, len(s), s[-1] - s[0]))
    print("  prologue (pass-through, A load):", d[:2].tolist())
    body = d[2:]
    per_group = body[:8 * 7].reshape(8, 7)
    print("  per group [mfma0, epi0, mfma1, epi1, mfma2, epi2, spline]:")
    for g in range(8): print("   ", per_group[g].tolist())
    print("  sums: mfma %d  epilogue %d  spline %d" % (per_group[:, 0::2][:, :3].sum(), per_group[:, 1::2].sum(), per_group[:, 6].sum()))
