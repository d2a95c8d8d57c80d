import os, sys, copy, ctypes
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nflows_amd
from nflows_amd import configs, ops, parallel, _native
dev = torch.device("cuda", 0)
flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, num_blocks=2, tail_bound=3.0, seed=0).eval()
flow = copy.deepcopy(flow_cpu).to(dev)
x = torch.randn(65536, 64, generator=torch.Generator().manual_seed(1234)).to(dev)
xs = x[:8192]
def check(tag):
    with torch.no_grad():
        layer = flow._transform._transforms[1]
        y1, _ = layer(xs); x1, _ = layer.inverse(y1)
        z, _ = flow._transform(xs); xr, _ = flow._transform.inverse(z)
    print(tag, "single %.3e composite %.3e" % ((x1 - xs).abs().max().item(), (xr - xs).abs().max().item()))
if os.environ.get("FRESH"): check("fresh")
def step():
    with torch.no_grad():
        lp = flow.log_prob(x)
    return parallel.reduce_log_likelihood(lp)
for _ in range(5): step()
torch.cuda.synchronize()
if os.environ.get("FRESH"): check("after warmup")
_native.check(_native.load().nfa_profile_enable(32 * 20))
for _ in range(20): step()
torch.cuda.synchronize()
if os.environ.get("FRESH"): check("after profiled steps (profiling still on)")
buf = (ctypes.c_float * 640)(); n = ctypes.c_int32(0)
_native.check(_native.load().nfa_profile_collect(buf, 640, ctypes.byref(n)))
_native.check(_native.load().nfa_profile_enable(0))
check("after profile off")
nflows_amd.check_status()
