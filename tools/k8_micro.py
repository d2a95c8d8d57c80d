#!/usr/bin/env python3
"""Micro-benchmark / check of K8 (ResidualNet conditioner + spline layer in one kernel) at the
BASELINE layer shape, against the PyTorch conditioner + K1 (fp32 and fp64 conditioner)."""
import copy, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
from nflows_amd.nn.nets import ResidualNet
from nflows_amd.utils import create_alternating_binary_mask

dev = "cuda:0"
B, D, K, H = int(os.environ.get("B", 65536)), 64, 8, 128
torch.manual_seed(0)
layer = RQ(create_alternating_binary_mask(D, even=True),
           lambda i, o: ResidualNet(i, o, hidden_features=H, num_blocks=2),
           num_bins=K, tails="linear", tail_bound=3.0).to(dev)
with torch.no_grad():  # make every weight matter (the reference zero-initialises the block outputs)
    for p in layer.parameters():
        p.copy_(torch.randn_like(p) * (0.3 if p.dim() == 1 else 1.5 / np.sqrt(p.shape[1])))
x = torch.randn(B, D, device=dev) * 1.5
inv = "--inverse" in sys.argv
fn = layer.inverse if inv else layer.forward
with torch.no_grad():
    RQ.fuse_conditioner = True
    for _ in range(3): y, lad = fn(x)
    torch.cuda.synchronize()
    evs = []
    for i in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); y, lad = fn(x); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    print("K8 %s lib=%s: median %.1f us  min %.1f us" % ("inverse" if inv else "forward",
          os.path.basename(os.environ.get("NFLOWS_AMD_LIB", "default")), ms[len(ms) // 2] * 1e3, ms[0] * 1e3))
    if "--check" in sys.argv:
        RQ.fuse_conditioner = False
        RQ.fuse_final_linear = False
        y0, l0 = fn(x)  # PyTorch fp32 conditioner + K1
        cols = layer.identity_features
        net64 = copy.deepcopy(layer.transform_net).double()
        p64 = net64(x.index_select(1, cols).double()).float()
        y1, l1 = ops.rqs_coupling(x, p64, layer.transform_features, layer._spec(), inverse=inv)
        def d(a, b): return (a - b).abs().max().item()
        print("  vs fp32 torch conditioner + K1: max |dy| %.2e  max |dlad| %.2e" % (d(y, y0), d(lad, l0)))
        print("  vs fp64 conditioner + K1:  K8 |dy| %.2e |dlad| %.2e    fp32 path |dy| %.2e |dlad| %.2e" % (
            d(y, y1), d(lad, l1), d(y0, y1), d(l0, l1)))
        ident = layer.identity_features
        print("  identity columns bit-exact:", bool(torch.equal(y[:, ident], x[:, ident])))
