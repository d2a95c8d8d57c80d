#!/usr/bin/env python3
"""Gradients of one RQ coupling layer from the generic and from the software-pipelined backward
kernel on the same inputs (the choice is read once per process from NFA_K1_BWD_PIPELINE, so the script
re-runs itself): prints the largest difference.   python tools/bwd_compare.py [B D inverse]"""
import os, subprocess, sys, numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nflows_amd import ops
    B, D, inverse, out = int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4])), sys.argv[5]
    g = torch.Generator().manual_seed(5)
    K, P, dt = 8, 23, D // 2
    x = (1.3 * torch.randn(B, D, generator=g)).cuda().requires_grad_(True)
    p = torch.randn(B, dt * P, generator=g).cuda().requires_grad_(True)
    wy, wl = torch.randn(B, D, generator=g).cuda(), torch.randn(B, generator=g).cuda()
    spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(32)))
    y, lad = ops.rqs_coupling(x, p, torch.arange(0, D, 2).cuda(), spec, inverse=inverse)
    ((y * wy).sum() + (lad * wl).sum()).backward()
    ops.check_status()
    np.savez(out, gx=x.grad.cpu().numpy(), gp=p.grad.cpu().numpy())
    sys.exit(0)
B, D, inverse = (sys.argv[1:4] + ["1030", "128", "1"][len(sys.argv) - 1:])[:3]
res = []
for flag in ("0", "1"):
    out = "/tmp/bwd_%s.npz" % flag
    subprocess.check_call([sys.executable, __file__, "--child", B, D, inverse, out], env=dict(os.environ, NFA_K1_BWD_PIPELINE=flag))
    res.append(np.load(out))
for k in ("gx", "gp"):
    d = np.abs(res[0][k] - res[1][k])
    print("%s: max |generic - pipelined| = %.3e  (max |value| %.3e, bit-identical: %s)" % (k, d.max(), np.abs(res[0][k]).max(), bool((res[0][k] == res[1][k]).all())))
