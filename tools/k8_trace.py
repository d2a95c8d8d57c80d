#!/usr/bin/env python3
"""Phase timeline of one K8 wave (debug): cycles between cycle-counter stamps."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import _native
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
from nflows_amd.nn.nets import ResidualNet
from nflows_amd.utils import create_alternating_binary_mask
dev = "cuda:0"
B, D = 65536, 64
torch.manual_seed(0)
layer = RQ(create_alternating_binary_mask(D, even=True), lambda i, o: ResidualNet(i, o, hidden_features=128, num_blocks=2),
           num_bins=8, tails="linear", tail_bound=3.0).to(dev)
x = torch.randn(B, D, device=dev)
with torch.no_grad():
    for _ in range(3): layer(x)
    buf = torch.zeros(512, dtype=torch.int64, device=dev)
    lib = _native.load(); lib.nfa_debug_k7_trace.argtypes = [ctypes.c_void_p]
    lib.nfa_debug_k7_trace(ctypes.c_void_p(buf.data_ptr()))
    layer(x)
    torch.cuda.synchronize(); lib.nfa_debug_k7_trace(None)
t = buf.cpu().numpy()
base = min(t[0], t[256])
for blk in (0, 1):
    s = t[blk * 256: blk * 256 + 250]; s = s[s > 0]
    d = np.diff(s)
    print("block %d: %d stamps, first at +%d, total %d cycles" % (blk * 256, len(s), s[0] - base, s[-1] - s[0]))
    print("  rows -> LDS tile, identity pieces: %d   initial layer (+ pieces): %d" % (d[0], d[1]))
    for b in range(2):
        o = 2 + 2 * b
        print("  block %d: first Linear (tile-major, + relu pieces) %d   skip + second Linear (k-major) + pieces %d" % (b, d[o], d[o + 1]))
    o = 6
    g = d[o:o + 16].reshape(8, 2)
    print("  final groups [3 tiles mfma, 2 splines]:", g.tolist())
    print("  output rows + logabsdet:", d[o + 16:].tolist())
    print("  sums: hidden gemms %d  final mfma %d  spline %d" % (d[2:6].sum(), g[:, 0].sum(), g[:, 1].sum()))
