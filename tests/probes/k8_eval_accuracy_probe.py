#!/usr/bin/env python3
"""(measurement script, not collected by pytest; lives here because it uses the oracle)
Accuracy of K8's evaluation variants against the float64 truth (oracle/eager.py in float64 on the
same weights and inputs): NFA_K8_PIPE=0 plain, 1 woven (bit-identical to plain), 2 woven with the
cheaper rounding sequence.  The switch is read once per process, so the script re-runs itself."""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    import nflows_amd
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    x = torch.randn(4096, 64, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        y, lad = flow.cuda()._transform(x.cuda())
        lp = flow.log_prob(x.cuda())
    nflows_amd.check_status()
    np.savez(sys.argv[2], y=y.cpu().numpy(), lad=lad.cpu().numpy(), lp=lp.cpu().numpy())
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
from nflows_amd import configs
from oracle import eager
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
x = torch.randn(4096, 64, generator=torch.Generator().manual_seed(3))
with torch.no_grad():
    lp32 = eager.flow_log_prob(flow.float(), x)
    lp64 = eager.flow_log_prob(flow.double(), x.double())
print("reference fp32 (eager port) vs float64: log_prob max %.3e  mean %.3e" % ((lp32.double() - lp64).abs().max(), (lp32.double() - lp64).abs().mean()))
for flag in ("0", "1", "2"):
    out = "/tmp/k8_eval_%s.npz" % flag
    subprocess.check_call([sys.executable, __file__, "--child", out], env=dict(os.environ, NFA_K8_PIPE=flag))
    r = np.load(out)
    d = np.abs(r["lp"].astype(np.float64) - lp64.numpy())
    print("NFA_K8_PIPE=%s vs float64: log_prob max %.3e  mean %.3e  99.9%% %.3e" % (flag, d.max(), d.mean(), np.quantile(d, 0.999)))
