#!/usr/bin/env python3
"""K8 with the spline evaluation woven into the final layer's MFMAs (NFA_K8_PIPE=1) against the
plain K8 (the switch is read once per process, so the script re-runs itself): outputs must agree bit
for bit; prints both run times of the 32-layer transform at B = 65536."""
import os, subprocess, sys, numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import nflows_amd
    from nflows_amd import configs
    out = sys.argv[2]
    flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
    x = torch.randn(65536, 64, generator=torch.Generator().manual_seed(3)).cuda()
    x[:4, :8] = torch.tensor([3.0, -3.0, 3.5, float("nan"), 0.0, 2.9999998, -7.0, 1e-8]).cuda()
    with torch.no_grad():
        y, lad = flow._transform(x)
        xi, ladi = flow._transform.inverse(y)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); flow._transform(x); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    nflows_amd.check_status()
    np.savez(out, y=y.cpu().numpy(), lad=lad.cpu().numpy(), xi=xi.cpu().numpy(), ladi=ladi.cpu().numpy(), ms=np.array(sorted(ts)[2]))
    sys.exit(0)
res = []
for flag in ("0", "1"):
    out = "/tmp/k8_pipe_%s.npz" % flag
    subprocess.check_call([sys.executable, __file__, "--child", out], env=dict(os.environ, NFA_K8_PIPE=flag))
    res.append(np.load(out))
for k in ("y", "lad", "xi", "ladi"):
    a, b = res[0][k], res[1][k]
    same = np.array_equal(a, b, equal_nan=True)
    print("%-5s bit-identical: %s   max |diff| %.3e" % (k, same, np.nanmax(np.abs(a - b))))
print("32-layer forward, B=65536: plain %.3f ms, woven %.3f ms" % (float(res[0]["ms"]), float(res[1]["ms"])))
