"""configs[4] forward pass (AR-RQ, D = 784, H = 256, K = 8, B = 4096) in a loop: for rocprofv3 --kernel-trace --stats
(which kernels the 0.37 ms are made of).   python tools/cfg5_forward_probe.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import configs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
flow = configs.ar_rq_flow(features=784, hidden_features=256, num_bins=8, tail_bound=3.0, seed=0).to("cuda:0").eval()
x = torch.randn(4096, 784, generator=torch.Generator().manual_seed(1234)).to("cuda:0")
with torch.no_grad():
    for _ in range(10):
        flow.log_prob(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        flow.log_prob(x)
    torch.cuda.synchronize()
print("configs[4] forward: %.4f ms per log_prob" % ((time.perf_counter() - t0) / reps * 1e3))
