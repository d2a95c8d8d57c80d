#!/usr/bin/env python3
"""K10 (`nfa_linear_wgrad_f32`) against the library GEMM for the conditioner layers of the BASELINE
flow at B = 65536: time per call (10 calls in one HIP graph, median of 20 replays)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops
dev = "cuda:0"


def timeit(fn, reps=20, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(inner):
            fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); graph.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / inner)
    return sorted(ts)[reps // 2]


B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
print("| layer (B=%d) | K10 us | library GEMM + column sum us | K10 fp32 TFLOP/s | K10 input GB/s |" % B)
print("|---|---|---|---|---|")
for I, O in ((32, 128), (128, 128), (128, 736)):
    x = torch.randn(B, I, device=dev); gy = torch.randn(B, O, device=dev)
    t_k = timeit(lambda: ops.linear_wgrad(x, gy))
    t_l = timeit(lambda: (gy.t() @ x, gy.sum(0)))
    print("| %d -> %d | %.1f | %.1f | %.1f | %.0f |" % (I, O, t_k, t_l, 2.0 * B * I * O / t_k / 1e6, 4.0 * B * (I + O) / t_k / 1e3))
# round 4: the four 128 x 128 layers of a conditioner in ONE launch pair (nfa_linear_wgrad_batched_f32)
probs = [(torch.randn(B, 128, device=dev), torch.randn(B, 128, device=dev)) for _ in range(4)]
t4 = timeit(lambda: [ops.linear_wgrad(x_, gy_) for x_, gy_ in probs])
tb = timeit(lambda: ops.linear_wgrad_batched(probs))
print("four 128 -> 128 layers: one by one %.1f us, batched %.1f us (%.1f fp32 TFLOP/s)" % (t4, tb, 4 * 2.0 * B * 128 * 128 / tb / 1e6))
