"""Wall time per Flow.log_prob of the BASELINE flow (32 layers, D = 64, 8 bins) at small batches, launched from the host
and replayed from a HIP graph (the GPU's own time): shows how much of a small-batch step is the HOST's.
    python tools/small_batch_probe.py [rows ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import configs, parallel
from nflows_amd.graphs import GraphedLogProb

rows = [int(a) for a in sys.argv[1:]] or [2048, 8192, 16384, 32768]
flow = configs.rq_nsf_flow(32, 64, 8, 128, seed=0).eval().cuda()
for B in rows:
    x = torch.randn(B, 64, generator=torch.Generator().manual_seed(977 + B)).cuda()

    def step():
        with torch.no_grad():
            return parallel.reduce_log_likelihood(flow.log_prob(x))

    def wall(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    eager = wall(step)
    g = GraphedLogProb(flow, x)
    graphed = wall(lambda: parallel.reduce_log_likelihood(g(x)))
    # host only: the same calls without waiting for the GPU (enqueue time per step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    host = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    print("rows %6d: eager %.3f ms/step | HIP-graph replay %.3f | host enqueue %.3f" % (B, eager, graphed, host), flush=True)
    del g
