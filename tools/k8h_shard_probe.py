#!/usr/bin/env python3
"""Times the 32-layer flow on small per-GPU shards (four-wave workgroups: one wave per SIMD)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import configs
dev = "cuda:0"


def timed(fn, n=50, warm=300):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).to(dev).eval()
    for rows in (32768, 16384, 4096):
        x = torch.randn(rows, 64, device=dev)
        print("32 layers, 8 bins, %6d rows: %.4f ms" % (rows, timed(lambda: flow.log_prob(x))))
    flow10 = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=10, hidden_features=128, seed=0).to(dev).eval()
    x = torch.randn(32768, 64, device=dev)
    print("32 layers, 10 bins, 32768 rows: %.4f ms" % timed(lambda: flow10.log_prob(x)))
