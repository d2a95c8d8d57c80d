#!/usr/bin/env python3
"""One table of every kernel of the library at a representative size on one MI355X: launch time
(median of 30, HIP events) and the rate of ALGORITHMIC bytes (or flops) against the chip's peak.
Writes markdown to stdout:  python tools/all_kernels.py > profiles/r1_kernel_table.md"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nflows_amd
from nflows_amd import ops, configs
from nflows_amd.transforms import splines
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
rows = []


def timeit(fn, reps=30, inner=10):
    """Per-call GPU time: `inner` calls captured into one HIP graph (the host is out of the loop, so
    kernels of a few microseconds are not hidden behind Python's launch overhead), median of `reps`
    replays divided by `inner`."""
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
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3 / inner  # us


def timeit_eager(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3


def add(kernel, what, us, nbytes=None, flops=None, peak_note=""):
    if nbytes is not None:
        rate = "%.0f GB/s (%.0f %% of 8 TB/s)" % (nbytes / us / 1e3, nbytes / us / 1e3 / 80.0)
    else:
        rate = "%.0f TFLOP/s %s" % (flops / us / 1e6, peak_note)
    rows.append("| %s | %s | %.1f | %s |" % (kernel, what, us, rate))


with torch.no_grad():
    B, D, K, H = 65536, 64, 8, 128
    x = torch.randn(B, D, device=dev, generator=g)
    tidx = torch.arange(0, D, 2, device=dev)
    P = 3 * K - 1
    params = torch.randn(B, 32 * P, device=dev, generator=g)
    spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(H)))
    k1_bytes = 4 * (2 * B * D + B * 32 * P + B)
    add("K1 `rqs_coupling_pipelined`", "RQ coupling layer, B=65536 D=64 K=8", timeit(lambda: ops.rqs_coupling(x, params, tidx, spec)), k1_bytes)
    add("K1 inverse", "same", timeit(lambda: ops.rqs_coupling(x, params, tidx, spec, inverse=True)), k1_bytes)
    gx = torch.randn(B, D, device=dev, generator=g); gl = torch.randn(B, device=dev, generator=g)
    gin, gpar = torch.empty_like(x), torch.empty_like(params)
    from nflows_amd import _native as NA
    import ctypes
    def bwd():  # the C entry point itself (what nflows_amd.autograd.RqsCoupling.backward calls)
        NA.check(NA.load().nfa_rqs_coupling_backward_f32(
            NA.ptr(x), NA.ptr(params), NA.ptr(tidx), None, None, NA.ptr(gx), NA.ptr(gl), NA.ptr(gin), NA.ptr(gpar),
            NA.ptr(ops._status_word(x.device)), B, D, tidx.numel(), ctypes.byref(spec), 0, NA.stream_handle(x.device)))
    add("K1-backward `rqs_coupling_backward_pipelined`", "same layer, grads wrt inputs and params", timeit(bwd), 2 * k1_bytes)

    N = B * 32
    xe = torch.randn(N, device=dev, generator=g) * 1.5
    r = torch.randn(N, P, device=dev, generator=g)
    add("K5 `rqs_elementwise_kernel`", "RQ functional, 2.1 M elements, K=8", timeit(lambda: splines.unconstrained_rational_quadratic_spline(xe, r[:, :K], r[:, K:2 * K], r[:, 2 * K:], tail_bound=3.0)), 4 * N * (P + 3))
    p = torch.randn(N, K, device=dev, generator=g)
    add("K9 linear", "linear spline functional, 2.1 M elements, K=8", timeit(lambda: splines.unconstrained_linear_spline(xe, p, tail_bound=3.0)), 4 * N * (K + 3))
    q = torch.randn(N, 2 * K - 1, device=dev, generator=g)
    add("K9 quadratic", "quadratic spline functional, same", timeit(lambda: splines.unconstrained_quadratic_spline(xe, q[:, :K], q[:, K:], tail_bound=3.0)), 4 * N * (2 * K - 1 + 3))
    c = torch.randn(N, 2 * K + 2, device=dev, generator=g)
    add("K9 cubic", "cubic spline functional, same", timeit(lambda: splines.unconstrained_cubic_spline(xe, c[:, :K], c[:, K:2 * K], c[:, 2 * K:2 * K + 1], c[:, 2 * K + 1:], tail_bound=3.0)), 4 * N * (2 * K + 2 + 3))
    add("K9 cubic inverse", "same", timeit(lambda: splines.unconstrained_cubic_spline(xe, c[:, :K], c[:, K:2 * K], c[:, 2 * K:2 * K + 1], c[:, 2 * K + 1:], inverse=True, tail_bound=3.0)), 4 * N * (2 * K + 2 + 3))

    # backward kernels of the sibling splines (C entry points, dense logit rows)
    lib = NA.load()
    lspec = ops.make_rqs_spec(K, "linear", tail_bound=3.0)
    gy1, gl1 = torch.randn(N, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
    gxe = torch.empty_like(xe)
    gp_ = torch.empty_like(p)
    add("K9 linear backward", "grads wrt inputs and logits, same elements", timeit(lambda: NA.check(lib.nfa_linear_spline_backward_f32(
        NA.ptr(xe), NA.ptr(p), NA.ptr(gy1), NA.ptr(gl1), NA.ptr(gxe), NA.ptr(gp_), N, ctypes.byref(lspec), 0, NA.stream_handle(xe.device)))),
        4 * N * (2 * K + 4))
    qw, qh = q[:, :K].contiguous(), q[:, K:].contiguous()
    gqw, gqh = torch.empty_like(qw), torch.empty_like(qh)
    add("K9 quadratic backward", "same", timeit(lambda: NA.check(lib.nfa_quadratic_spline_backward_f32(
        NA.ptr(xe), NA.ptr(qw), NA.ptr(qh), K - 1, NA.ptr(gy1), NA.ptr(gl1), NA.ptr(gxe), NA.ptr(gqw), NA.ptr(gqh), N,
        ctypes.byref(lspec), 0, NA.stream_handle(xe.device)))), 4 * N * (2 * (2 * K - 1) + 4))
    cw, ch, cl, cr = (c[:, :K].contiguous(), c[:, K:2 * K].contiguous(), c[:, 2 * K].contiguous(), c[:, 2 * K + 1].contiguous())
    gcw, gch, gcl, gcr = torch.empty_like(cw), torch.empty_like(ch), torch.empty_like(cl), torch.empty_like(cr)
    add("K9 cubic backward", "same", timeit(lambda: NA.check(lib.nfa_cubic_spline_backward_f32(
        NA.ptr(xe), NA.ptr(cw), NA.ptr(ch), NA.ptr(cl), NA.ptr(cr), NA.ptr(gy1), NA.ptr(gl1), NA.ptr(gxe), NA.ptr(gcw), NA.ptr(gch),
        NA.ptr(gcl), NA.ptr(gcr), N, ctypes.byref(lspec), 0, NA.stream_handle(xe.device)))), 4 * N * (2 * (2 * K + 2) + 4))

    uw = torch.rand(32, K, device=dev, generator=g); ud = torch.rand(32, K - 1, device=dev, generator=g)
    xs = torch.randn(B, 32, device=dev, generator=g)
    add("K6 `rqs_shared_kernel`", "batch-shared RQ CDF, B=65536 F=32", timeit(lambda: ops.rqs_shared(xs, uw, uw, ud, ops.make_rqs_spec(K, "linear", tail_bound=3.0), False)), 4 * (2 * B * 32 + B))

    for I, O in ((128, 128), (128, 736)):
        xa = torch.randn(B, I, device=dev, generator=g); ga = torch.randn(B, O, device=dev, generator=g)
        add("K10 `wgrad_partial_kernel` + `wgrad_reduce_kernel`", "weight + bias gradient of Linear(%d -> %d), B=65536" % (I, O),
            timeit(lambda: ops.linear_wgrad(xa, ga)), flops=2.0 * B * I * O, peak_note="(fp32 matrix peak 157); library GEMM + column sum: %.0f us" % timeit(lambda: (ga.t() @ xa, ga.sum(0))))

    B2, D2 = 16384, 32
    x2 = torch.randn(B2, D2, device=dev, generator=g); p2 = torch.randn(B2, 32, device=dev, generator=g)
    t2 = torch.arange(0, D2, 2, device=dev)
    add("K2 `affine_coupling_kernel`", "affine coupling layer, B=16384 D=32", timeit(lambda: ops.affine_coupling(x2, p2, t2, nflows_amd._native.SCALE_DEFAULT)), 4 * (2 * B2 * D2 + B2 * 32 + B2))
    perm = torch.randperm(D, device=dev, generator=g)
    add("K4 `permute_cols_kernel`", "column permutation, B=65536 D=64", timeit(lambda: ops.permute_cols(x, perm)), 8 * B * D)
    add("K3 `rowsum_kernel`", "row sum, B=65536 D=64", timeit(lambda: ops.rowsum(x)), 4 * (B * D + B))
    add("K3 normal log-prob", "base density + logabsdet, B=65536 D=64", timeit(lambda: ops.standard_normal_log_prob(x, gl)), 4 * (B * D + 2 * B))

    # fused conditioner kernels on one layer and on the 32-layer run
    flow = configs.rq_nsf_flow(num_layers=32, features=D, num_bins=K, hidden_features=H, seed=0).to(dev).eval()
    layer = flow._transform._transforms[1]
    macs_final, macs_all = 32 * P * H, 32 * H + 4 * H * H + 32 * P * H
    for path, name in (("k7", "K7 `rqs_fused_linear_kernel` (fp32 MFMA)"), ("k7b", "K7b `rqs_fused_linear_bf16_kernel`")):
        RQ.fuse_conditioner, RQ.fuse_final_linear, RQ.final_linear_engine = False, True, ("f32" if path == "k7" else "bf16x3")
        hid = layer.transform_net.hidden(x.index_select(1, layer.identity_features))
        wp, bp = layer._packed_final_linear(layer.transform_net.final_layer)
        us = timeit(lambda: ops.rqs_coupling_fused_linear(x, hid, wp, bp, layer.transform_features, layer._spec()))
        if path == "k7":
            add(name, "final Linear + spline layer, B=65536", us, flops=2.0 * B * macs_final, peak_note="(fp32 matrix peak 157)")
        else:
            add(name, "same, split-bf16", us, flops=12.0 * B * macs_final, peak_note="bf16 (peak 2 500); fp32-equivalent %.0f" % (2.0 * B * macs_final / us / 1e6))
    RQ.fuse_conditioner, RQ.fuse_final_linear, RQ.final_linear_engine = True, True, "bf16x3"
    for engine, name, products, pipe in (("bf16x3", "K8 `rqs_resnet_kernel` (three bf16 pieces, 6 products)", 6, "bf16"),
                                         ("f16x2", "K8h `rqs_resnet_f16_kernel` (two f16 pieces, 3 products)", 3, "f16")):
        RQ.conditioner_engine = engine
        nflows_amd.invalidate_packed_weights()
        us = timeit(lambda: layer(x))
        add(name + ", 1 layer", "ResidualNet conditioner + spline layer, B=65536", us, flops=products * 2.0 * B * macs_all,
            peak_note="%s (peak 2 500); fp32-equivalent %.0f" % (pipe, 2.0 * B * macs_all / us / 1e6))
        us = timeit(lambda: flow._transform(x), reps=10, inner=2)
        add(name.split(" ")[0] + ", run of 32 layers", "the whole BASELINE transform in one launch, B=65536", us,
            flops=32 * products * 2.0 * B * macs_all,
            peak_note="%s (peak 2 500); fp32-equivalent %.0f" % (pipe, 32 * 2.0 * B * macs_all / us / 1e6))
    # K11: a run of affine layers with MLP conditioners (BASELINE configs[1]); K12: the autoregressive inverse (configs[4])
    aff = configs.affine_coupling_flow(8, 32, (128, 128)).to(dev).eval()
    xa = torch.randn(16384, 32, device=dev, generator=g)
    us = timeit(lambda: aff._transform(xa))
    add("K11 `affine_mlp_kernel`, run of 8 layers", "8 affine coupling layers + MLP conditioners, B=16384 D=32", us,
        flops=8 * 6 * 2.0 * 16384 * (16 * 128 + 128 * 128 + 128 * 32), peak_note="bf16 (peak 2 500)")
    ar = configs.ar_rq_flow(784, 256, 8, 3.0, 2).to(dev).eval()
    za = torch.randn(4096, 784, device=dev, generator=g)
    t_ar = ar._transform._transforms[0]
    us = timeit(lambda: t_ar.inverse(za), reps=10, inner=2)
    rows.append("| K12 `made_rqs_inverse_kernel` + tail | autoregressive RQ inverse, D=784 H=256 B=4096 (256 sequential steps) | %.1f | %.2f us per step and 16 samples: latency-bound |"
                % (us, us / 256))
    # K13: the MADE's output layer inside the spline kernel (the layer's forward pass; the hidden layers are library GEMMs)
    us = timeit(lambda: t_ar(za), reps=10, inner=2)
    rows.append("| K13 `rqs_made_output_kernel` + the MADE's hidden layers | autoregressive RQ forward (density), D=784 H=256 B=4096 | %.1f | %.0f TFLOP/s bf16 over the whole call (6 products x 37.8 GFLOP in the kernel); %.1f M samples/s |"
                % (us, 6 * 2.0 * 4096 * 256 * 784 * 23 / us / 1e6, 4096 / us))
    # K8s: the 32-layer flow on the small-batch kernel (128-row blocks at 32 768 rows, four-wave 64-row blocks below)
    RQ.conditioner_engine = "f16x2"
    nflows_amd.invalidate_packed_weights()
    flow32 = configs.rq_nsf_flow(32, 64, 8, 128).to(dev).eval()
    macs = 32 * 128 + 4 * 128 * 128 + 128 * 32 * 24
    for rows_s in (32768, 16384, 8192):
        xs = torch.randn(rows_s, 64, device=dev, generator=g)
        us = timeit(lambda: flow32._transform(xs), reps=10, inner=2)
        rows.append("| K8s `k8s::rqs_resnet_f16s_kernel` (%s), run of 32 layers | the whole BASELINE transform, B=%d | %.1f | %.0f TFLOP/s f16 (peak 2 500); %.1f M samples/s |"
                    % ("eight waves x 16 rows" if rows_s > 16384 else "four waves x 16 rows", rows_s, us,
                       32 * 3 * 2.0 * rows_s * macs / us / 1e6, rows_s / us))
    # K14: the conditioner under training (forward with the Linears' inputs saved, chain of input gradients, packer)
    from nflows_amd.nn.nets import ResidualNet
    net14 = ResidualNet(32, 736, 128, num_blocks=2).to(dev)
    blocks14 = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias)
                for b in net14.blocks]
    final14 = (net14.final_layer.weight, net14.final_layer.bias)
    pack14 = lambda: ops.pack_resnet_hidden_train(net14.initial_layer.weight, net14.initial_layer.bias, blocks14, final14)
    fw14, fb14, bw14, fbias14 = pack14()
    x14 = torch.randn(B, 32, device=dev, generator=g)
    g14 = torch.randn(B, 128, device=dev, generator=g)
    _, saved14, _ = ops.resnet_hidden_forward(x14, fw14, fb14, 2, fbias14, 736)
    fl14 = 2.0 * B * (32 * 128 + 4 * 128 * 128)
    us = timeit(lambda: ops.resnet_hidden_forward(x14, fw14, fb14, 2, fbias14, 736))
    rows.append("| K14 `resnet_hidden_forward_kernel<2>` | ResidualNet conditioner under training (32 -> 128 x 2 blocks -> 736), forward + the Linears' inputs saved, B=65536 | %.1f | %.0f TFLOP/s bf16 (6 products; peak 2 500) |"
                % (us, 6 * (fl14 + 2.0 * B * 128 * 736) / us / 1e6))
    us = timeit(lambda: ops.resnet_hidden_backward(g14, bw14, saved14, 32))
    rows.append("| K14 `resnet_hidden_backward_kernel<2>` | chain of input gradients through the two blocks and the initial layer, B=65536 | %.1f | %.0f TFLOP/s bf16 (6 products) |"
                % (us, 6 * fl14 / us / 1e6))
    rows.append("| K14 `pack_resnet_hidden_kernel` | both weight streams of one conditioner (80 + 34 stages) | %.1f | launch-bound |" % timeit(pack14))
    # K5d: the float64 functional and its gradient (correctness path)
    x64 = xe.double()
    uw64, uh64, ud64 = r[:, :K].double().contiguous(), r[:, K:2 * K].double().contiguous(), r[:, 2 * K:].double().contiguous()
    y64, l64 = torch.empty_like(x64), torch.empty_like(x64)
    s64 = ops.make_rqs_spec(K, "linear", tail_bound=3.0)
    rows.append("| K5d `rqs_elementwise_f64_kernel` | RQ functional in float64, 2.1 M elements, K=8 | %.1f | %.0f GB/s (the correctness path) |"
                % ((lambda u: (u, 8 * N * (P + 3) / u / 1e3))(timeit(lambda: NA.check(lib.nfa_rqs_elementwise_f64(
                    NA.ptr(x64), NA.ptr(uw64), K, NA.ptr(uh64), K, NA.ptr(ud64), K - 1, K - 1, NA.ptr(y64), NA.ptr(l64),
                    None, NA.ptr(ops._status_word(x64.device)), N, ctypes.byref(s64), 0, NA.stream_handle(x64.device)))))))
    gy64, gl64 = gy1.double(), gl1.double()
    gx64, guw64, guh64, gud64 = torch.empty_like(x64), torch.empty_like(uw64), torch.empty_like(uh64), torch.empty_like(ud64)
    rows.append("| K5d-backward `rqs_elementwise_backward_f64_kernel` | its gradient, same elements | %.1f | %.0f GB/s |"
                % ((lambda u: (u, 8 * N * (2 * P + 4) / u / 1e3))(timeit(lambda: NA.check(lib.nfa_rqs_elementwise_backward_f64(
                    NA.ptr(x64), NA.ptr(uw64), K, NA.ptr(uh64), K, NA.ptr(ud64), K - 1, K - 1, NA.ptr(gy64), NA.ptr(gl64),
                    NA.ptr(gx64), NA.ptr(guw64), NA.ptr(guh64), NA.ptr(gud64), N, ctypes.byref(s64), 0,
                    NA.stream_handle(x64.device)))))))
    nflows_amd.check_status()

print("# Kernel table (round 3, 1 x MI355X; `python tools/all_kernels.py`)\n")
print("GPU time per call: 10 calls captured in one HIP graph, median of 30 replays / 10 (the backward through")
print("C entry point is called directly); rates are ALGORITHMIC bytes or flops per launch over that time.  The helper")
print("kernels next to a call (output allocation is free, a status-word memset is not) are included.\n")
print("| kernel | workload | µs | rate |")
print("|---|---|---|---|")
print("\n".join(rows))
