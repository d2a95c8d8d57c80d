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


def timed(fn, reps, warm=3):
    for _ in range(warm):
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
    # configs[1]: 8 affine coupling layers, D=32, MLP conditioner, batch 16384
    flow = configs.affine_coupling_flow(8, 32, (128, 128)).to(dev).eval()
    x = torch.randn(16384, 32, device=dev)
    report("cfg1 affine x8 D=32 log_prob", timed(lambda: flow.log_prob(x), 50), 16384)
    z, _ = flow._transform(x)
    xr, _ = flow._transform.inverse(z)
    report("cfg1 affine x8 D=32 inverse", timed(lambda: flow._transform.inverse(z), 50), 16384,
           fwd_inv_max_err=(xr - x).abs().max().item())

    # configs[2]: 16 RQ coupling layers, D=64, K=8, batch 65536
    flow = configs.rq_nsf_flow(16, 64, 8, 128).to(dev).eval()
    x = torch.randn(65536, 64, device=dev)
    report("cfg2 RQ-NSF x16 log_prob", timed(lambda: flow.log_prob(x), 20), 65536)
    z, _ = flow._transform(x)
    xr, _ = flow._transform.inverse(z)
    report("cfg2 RQ-NSF x16 sample path (inverse)", timed(lambda: flow._transform.inverse(z), 20), 65536,
           fwd_inv_max_err=(xr - x).abs().max().item())

    # configs[3] single-GPU share: 32 layers, 32768 rows (the per-GPU shard of batch 262144 over 8)
    flow = configs.rq_nsf_flow(32, 64, 8, 128).to(dev).eval()
    x = torch.randn(32768, 64, device=dev)
    report("cfg3 RQ-NSF x32 log_prob, one 32768-row shard", timed(lambda: flow.log_prob(x), 20), 32768)

    # configs[4]: autoregressive RQ spline, D=784, K=8, batch 4096
    flow = configs.ar_rq_flow(784, 256, 8, 3.0, 2).to(dev).eval()
    x = torch.randn(4096, 784, device=dev)
    t = flow._transform._transforms[0]
    report("cfg4 AR-RQ D=784 forward (log_prob)", timed(lambda: flow.log_prob(x), 10), 4096)
    params = t.autoregressive_net(x)
    one = timed(lambda: t._elementwise_inverse(x, t.autoregressive_net(x)), 10)
    report("cfg4 AR-RQ D=784 inverse, ONE of 784 reference iterations", one, 4096,
           extrapolated_full_inverse_s=one * 784)
    z = torch.randn(4096, 784, device=dev)
    full = timed(lambda: t.inverse(z), 2, warm=1)
    report("cfg4 AR-RQ D=784 FULL inverse (sampling), column-wise", full, 4096)
    nflows_amd.check_status()
