#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one MI355X (informational; bench.py is the
contract benchmark).  Prints one JSON object per config."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nflows_amd  # noqa: E402
from nflows_amd import configs  # noqa: E402

dev = "cuda:0"


def timed(fn, reps, warm=50):
    for _ in range(warm):          # (long enough for the clocks to come up: the first milliseconds run slow)
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def report(name, seconds, batch, **extra):
    print(json.dumps(dict(config=name, ms=seconds * 1e3, samples_per_s=batch / seconds, batch=batch, **extra)),
          flush=True)


with torch.no_grad():
    # configs[0]: the README moons flow (2 x MAF + RandomPermutation), batch 1024
    flow = configs.moons_maf_flow().to(dev).eval()
    x = torch.randn(1024, 2, device=dev)
    report("configs[0] moons MAF x2 D=2 log_prob", timed(lambda: flow.log_prob(x), 200), 1024)
    report("configs[0] moons MAF x2 D=2 sample", timed(lambda: flow.sample(1024), 200), 1024)

    # configs[1]: 8 affine coupling layers, D=32, MLP conditioner, batch 16384 (K11: one launch)
    flow = configs.affine_coupling_flow(8, 32, (128, 128)).to(dev).eval()
    x = torch.randn(16384, 32, device=dev)
    report("configs[1] affine x8 D=32 log_prob", timed(lambda: flow.log_prob(x), 200, warm=500), 16384)
    z, _ = flow._transform(x)
    xr, _ = flow._transform.inverse(z)
    report("configs[1] affine x8 D=32 inverse", timed(lambda: flow._transform.inverse(z), 200, warm=500), 16384,
           fwd_inv_max_err=(xr - x).abs().max().item())

    # configs[2]: 16 RQ coupling layers, D=64, K=8, batch 65536
    flow = configs.rq_nsf_flow(16, 64, 8, 128).to(dev).eval()
    x = torch.randn(65536, 64, device=dev)
    report("configs[2] RQ-NSF x16 log_prob", timed(lambda: flow.log_prob(x), 50), 65536)
    z, _ = flow._transform(x)
    xr, _ = flow._transform.inverse(z)
    report("configs[2] RQ-NSF x16 sample path (inverse)", timed(lambda: flow._transform.inverse(z), 50), 65536,
           fwd_inv_max_err=(xr - x).abs().max().item())

    # configs[3] single-GPU share: 32 layers, 32768 rows (the per-GPU shard of batch 262144 over 8), and the
    # whole batch on one GPU
    flow = configs.rq_nsf_flow(32, 64, 8, 128).to(dev).eval()
    x = torch.randn(32768, 64, device=dev)
    report("configs[3] RQ-NSF x32 log_prob, one 32768-row shard", timed(lambda: flow.log_prob(x), 50), 32768)
    x = torch.randn(262144, 64, device=dev)
    report("configs[3] RQ-NSF x32 log_prob, all 262144 rows on one GPU", timed(lambda: flow.log_prob(x), 10, warm=5), 262144)

    # the reference's default of 10 bins on the same flow (K8h; the bf16x3 engine K8 takes 4.0 ms)
    flow = configs.rq_nsf_flow(32, 64, 10, 128).to(dev).eval()
    x = torch.randn(65536, 64, device=dev)
    report("32-layer RQ-NSF with num_bins = 10, log_prob", timed(lambda: flow.log_prob(x), 20, warm=10), 65536)

    # a conditional flow (context 12, embedded from 5): K8 with a context
    flow = configs.conditional_rq_nsf_flow(32, 64, 8, 128, 5, 12).to(dev).eval()
    ctx = torch.randn(65536, 5, device=dev)
    report("32-layer conditional RQ-NSF (context 12), log_prob", timed(lambda: flow.log_prob(x, context=ctx), 20, warm=10), 65536)

    # the same with the reference's default of 10 bins (round 3: the parameter blocks take the words they use, so
    # this shape keeps the eight-wave K8h kernel; round 2 ran it on the bf16x3 kernel in 5.8 ms)
    flow = configs.conditional_rq_nsf_flow(32, 64, 10, 128, 5, 12).to(dev).eval()
    report("32-layer conditional RQ-NSF (context 12), num_bins = 10, log_prob",
           timed(lambda: flow.log_prob(x, context=ctx), 20, warm=10), 65536)

    # configs[4]: autoregressive RQ spline, D=784, K=8, batch 4096
    flow = configs.ar_rq_flow(784, 256, 8, 3.0, 2).to(dev).eval()
    x = torch.randn(4096, 784, device=dev)
    t = flow._transform._transforms[0]
    report("configs[4] AR-RQ D=784 forward (log_prob)", timed(lambda: flow.log_prob(x), 50), 4096)
    z = torch.randn(4096, 784, device=dev)
    xs, _ = t.inverse(z)
    zz, _ = t(xs)
    report("configs[4] AR-RQ D=784 FULL inverse (sampling): K12 + tail", timed(lambda: t.inverse(z), 20, warm=5), 4096,
           fwd_of_inverse_max_err=(zz - z).abs().max().item())
    nflows_amd.check_status()
