#!/usr/bin/env python3
"""(measurement script, not collected by pytest; lives under tests/ because it runs the oracle)
Secondary baseline: the reference's own op sequence (its bit-identical eager port,
oracle/eager.py) run as PyTorch-eager ON THE SAME MI355X, next to the fused kernels.  Informational."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nflows_amd import configs
from oracle import eager
dev = "cuda:0"
flow = configs.rq_nsf_flow(32, 64, 8, 128).to(dev).eval()
for B in (65536, 16384):
    x = torch.randn(B, 64, device=dev)
    def timed(fn, reps):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    with torch.no_grad():
        t_fused = timed(lambda: flow.log_prob(x), 10)
        t_eager = timed(lambda: eager.flow_log_prob(flow, x), 3)
        d = (flow.log_prob(x) - eager.flow_log_prob(flow, x)).abs().max().item()
    print("B=%d: fused %.2f ms (%.2f M/s)   reference op sequence, eager on the same GPU %.2f ms (%.3f M/s)   "
          "ratio %.1fx   max |dlog_prob| %.2e" % (B, t_fused * 1e3, B / t_fused / 1e6, t_eager * 1e3,
                                                   B / t_eager / 1e6, t_eager / t_fused, d), flush=True)
