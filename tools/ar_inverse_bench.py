#!/usr/bin/env python3
"""BASELINE configs[4] (autoregressive RQ spline, D=784, H=256, K=8, batch 4096): the full inverse
(sampling direction) and its agreement with the forward pass."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nflows_amd
from nflows_amd import configs
dev = "cuda:0"
with torch.no_grad():
    flow = configs.ar_rq_flow(784, 256, 8, 3.0, 2).to(dev).eval()
    t = flow._transform._transforms[0]
    z = torch.randn(4096, 784, device=dev)
    x, lad = t.inverse(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = int(os.environ.get("REPS", "3"))
    for _ in range(reps):
        x, lad = t.inverse(z)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    zz, ladf = t(x)
    nflows_amd.check_status()
    print(json.dumps({"config": "cfg5 AR-RQ D=784 H=256 K=8 B=4096 FULL inverse", "ms": ms,
                      "sequential_steps": t._sequential_steps(),
                      "fwd_of_inverse_max_err": (zz - z).abs().max().item(),
                      "logabsdet_sum_max_err": (lad + ladf).abs().max().item()}))
