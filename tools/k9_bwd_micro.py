#!/usr/bin/env python3
"""GPU time of the sibling splines' backward kernels (K9: linear / quadratic / cubic, K = 8 and 10 -- the
instances compiled for a constant bin count -- and K = 9, the generic instance) and of the float64 functional
(K5d forward / backward), 2.1 M elements, HIP-graph replay (10 calls per graph, median of 30).
    python tools/k9_bwd_micro.py > profiles/r3/k9_backward.txt"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops, _native as NA

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
N = 65536 * 32


def timeit(fn, reps=30, inner=10):
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
    graph.replay()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); graph.replay(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3 / inner


def report(what, us, nbytes):
    print("%-44s %8.1f us  %6.0f GB/s (%.0f %% of 8 TB/s)" % (what, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80.0), flush=True)


lib = NA.load()
S = lambda: NA.stream_handle(torch.device(dev))   # (evaluated per call: the capture stream under graph capture)
xe = torch.randn(N, device=dev, generator=g) * 1.5
gy, gl = torch.randn(N, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
gx = torch.empty_like(xe)
with torch.no_grad():
    for K in (8, 10, 9):
        spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0)
        p = torch.randn(N, K, device=dev, generator=g)
        gp = torch.empty_like(p)
        report("K9 linear backward, K=%d" % K, timeit(lambda: NA.check(lib.nfa_linear_spline_backward_f32(
            NA.ptr(xe), NA.ptr(p), NA.ptr(gy), NA.ptr(gl), NA.ptr(gx), NA.ptr(gp), N, ctypes.byref(spec), 0, S()))),
            4 * N * (2 * K + 4))
        qh = torch.randn(N, K - 1, device=dev, generator=g)
        gqh = torch.empty_like(qh)
        for inverse in (0, 1):
            report("K9 quadratic backward, K=%d%s" % (K, ", inverse" if inverse else ""), timeit(lambda: NA.check(
                lib.nfa_quadratic_spline_backward_f32(NA.ptr(xe), NA.ptr(p), NA.ptr(qh), K - 1, NA.ptr(gy), NA.ptr(gl),
                                                      NA.ptr(gx), NA.ptr(gp), NA.ptr(gqh), N, ctypes.byref(spec), inverse, S()))),
                4 * N * (2 * (2 * K - 1) + 4))
        ch = torch.randn(N, K, device=dev, generator=g)
        cl, cr = torch.randn(N, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
        gch, gcl, gcr = torch.empty_like(ch), torch.empty_like(cl), torch.empty_like(cr)
        for inverse in (0, 1):
            report("K9 cubic backward, K=%d%s" % (K, ", inverse" if inverse else ""), timeit(lambda: NA.check(
                lib.nfa_cubic_spline_backward_f32(NA.ptr(xe), NA.ptr(p), NA.ptr(ch), NA.ptr(cl), NA.ptr(cr), NA.ptr(gy),
                                                  NA.ptr(gl), NA.ptr(gx), NA.ptr(gp), NA.ptr(gch), NA.ptr(gcl), NA.ptr(gcr),
                                                  N, ctypes.byref(spec), inverse, S()))),
                4 * N * (2 * (2 * K + 2) + 4))
        del p, gp, qh, gqh, ch, gch
    # float64 functional, K = 8 (8-byte elements)
    K = 8
    spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0)
    x64 = xe.double()
    uw, uh = (torch.randn(N, K, device=dev, generator=g).double() for _ in range(2))
    ud = torch.randn(N, K - 1, device=dev, generator=g).double()
    y, lad = torch.empty_like(x64), torch.empty_like(x64)
    status = ops._status_word(torch.device(dev))
    report("K5d float64 functional, K=8", timeit(lambda: NA.check(lib.nfa_rqs_elementwise_f64(
        NA.ptr(x64), NA.ptr(uw), K, NA.ptr(uh), K, NA.ptr(ud), K - 1, K - 1, NA.ptr(y), NA.ptr(lad), None, NA.ptr(status), N,
        ctypes.byref(spec), 0, S()))), 8 * N * (3 * K - 1 + 3))
    gy64, gl64 = gy.double(), gl.double()
    gx64, guw, guh, gud = torch.empty_like(x64), torch.empty_like(uw), torch.empty_like(uh), torch.empty_like(ud)
    report("K5d-backward float64, K=8", timeit(lambda: NA.check(lib.nfa_rqs_elementwise_backward_f64(
        NA.ptr(x64), NA.ptr(uw), K, NA.ptr(uh), K, NA.ptr(ud), K - 1, K - 1, NA.ptr(gy64), NA.ptr(gl64), NA.ptr(gx64),
        NA.ptr(guw), NA.ptr(guh), NA.ptr(gud), N, ctypes.byref(spec), 0, S()))), 8 * N * (2 * (3 * K - 1) + 4))
