#!/usr/bin/env python3
"""Phase timeline of one wave of K8 with a context (debug): cycles between cycle-counter stamps."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import _native
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
from nflows_amd.nn.nets import ResidualNet
from nflows_amd.utils import create_alternating_binary_mask
dev = "cuda:0"
B, D, CE = 65536, 64, int(os.environ.get("CE", "12"))
torch.manual_seed(0)
RQ.conditioner_engine = "bf16x3"
layer = RQ(create_alternating_binary_mask(D, even=True),
           lambda i, o: ResidualNet(i, o, hidden_features=128, num_blocks=2, context_features=CE),
           num_bins=8, tails="linear", tail_bound=3.0).to(dev)
x = torch.randn(B, D, device=dev)
ctx = torch.randn(B, CE, device=dev)
with torch.no_grad():
    for _ in range(3): layer(x, ctx)
    buf = torch.zeros(512, dtype=torch.int64, device=dev)
    lib = _native.load(); lib.nfa_debug_k7_trace.argtypes = [ctypes.c_void_p]
    lib.nfa_debug_k7_trace(ctypes.c_void_p(buf.data_ptr()))
    layer(x, ctx)
    torch.cuda.synchronize(); lib.nfa_debug_k7_trace(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): layer(x, ctx)
    e1.record(); torch.cuda.synchronize()
    print("one layer with context %d: %.1f us" % (CE, e0.elapsed_time(e1) * 50))
t = buf.cpu().numpy()
for blk in (0, 1):
    s = t[blk * 256: blk * 256 + 250]; s = s[s > 0]
    if len(s) < 10:
        continue   # (the grid had no workgroup 256)
    d = np.diff(s)
    print("workgroup %d: %d stamps, total %d cycles" % (blk * 256, len(s), s[-1] - s[0]))
    print("  rows -> LDS, input pieces: %d   initial layer (+ pieces, bias staging): %d" % (d[0], d[1]))
    for b in range(2):
        o = 2 + 3 * b
        print("  block %d: first Linear + relu pieces %d   second Linear %d   gate + skip + pieces %d" % (b, d[o], d[o + 1], d[o + 2]))
    print("  rest (final layer groups, output):", d[8:].tolist())
