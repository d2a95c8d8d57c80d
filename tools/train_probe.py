#!/usr/bin/env python3
"""Training-step timing (forward + backward + Adam) of the 32-layer RQ-NSF flow; informational."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import configs
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
flow = configs.rq_nsf_flow(32, 64, 8, 128).to(dev).train()
opt = torch.optim.Adam(flow.parameters(), lr=1e-4)
x = torch.randn(B, 64, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean()
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("train step B=%d: %.2f ms  (%.0f samples/s)  loss %.4f" % (B, dt * 1e3, B / dt, l.item()))
