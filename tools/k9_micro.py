#!/usr/bin/env python3
"""Micro-benchmark of K9 (linear / quadratic spline functionals) and, for comparison, K5 (RQ
functional) on N = 2 097 152 elements (= 65 536 samples x 32 transformed features), packed logits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd.transforms import splines
dev = "cuda:0"
N, K = 65536 * 32, 8
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, device=dev, generator=g) * 1.5


def timeit(fn, nbytes, label):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    print("%-34s %7.1f us  %6.0f GB/s of algorithmic bytes (%.1f MB)" % (label, ms * 1e3, nbytes / ms / 1e6, nbytes / 1e6))


with torch.no_grad():
    for inv in (False, True):
        tag = "inverse" if inv else "forward"
        p = torch.randn(N, K, device=dev, generator=g)
        timeit(lambda: splines.unconstrained_linear_spline(x, p, inverse=inv, tail_bound=3.0), 4 * N * (K + 3), "linear K=8 " + tag)
        q = torch.randn(N, 2 * K - 1, device=dev, generator=g)
        timeit(lambda: splines.unconstrained_quadratic_spline(x, q[:, :K], q[:, K:], inverse=inv, tail_bound=3.0),
               4 * N * (2 * K - 1 + 3), "quadratic K=8 " + tag)
        r = torch.randn(N, 3 * K - 1, device=dev, generator=g)
        timeit(lambda: splines.unconstrained_rational_quadratic_spline(x, r[:, :K], r[:, K:2 * K], r[:, 2 * K:], inverse=inv,
                                                                       tail_bound=3.0), 4 * N * (3 * K - 1 + 3), "rational-quadratic K=8 (K5) " + tag)
