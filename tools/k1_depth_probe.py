"""K1 (the unfused spline coupling kernel) at the bench's layer shape, B rows: time per launch (parameters rotated
over four buffers: nothing served from the Infinity Cache) and a checksum of the results.  Round 3 used it to compare
the kernel with a variant carrying TWO tiles in registers (NFA_K1_DEPTH=2, since removed: slower, same checksum --
profiles/r3/k1_two_tile_prefetch.txt).   python tools/k1_depth_probe.py [rows]"""
import hashlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
D, K, dt = 64, 8, 32
P = 3 * K - 1
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = (1.3 * torch.randn(B, D, generator=g)).to(dev)
tidx = torch.arange(0, D, 2, device=dev)
spec = ops.make_rqs_spec(K, tails="linear", tail_bound=3.0, wh_divisor=float(np.sqrt(128)))
# four parameter buffers in rotation: nothing is served from the Infinity Cache
params = [torch.randn(B, dt * P, generator=g).to(dev) for _ in range(4)]
for i in range(4):
    y, lad = ops.rqs_coupling(x, params[i], tidx, spec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 40
for i in range(reps):
    y, lad = ops.rqs_coupling(x, params[i % 4], tidx, spec)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nbytes = 4 * (B * D + B * dt * P + B * D + B)
y0, lad0 = ops.rqs_coupling(x, params[0], tidx, spec)
h = hashlib.sha256(y0.cpu().numpy().tobytes() + lad0.cpu().numpy().tobytes()).hexdigest()[:16]
print("K1 depth %s rows %d: %.1f us per launch = %.0f GB/s (%.3f of 8 TB/s); sha256 %s"
      % (os.environ.get("NFA_K1_DEPTH", "1"), B, ms * 1e3, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000, h))
