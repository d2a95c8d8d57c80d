"""Round 5: the reference's SimpleRealNVP composition (affine couplings + ResidualNet conditioners, flows/realnvp.py:17-71)
in one launch (K11's residual form) against the layer-by-layer path (PyTorch-ROCm GEMMs + K2).
    python tools/realnvp_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import configs, ops
from nflows_amd.transforms import AffineCouplingTransform

DEV = "cuda:0"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for features, layers, blocks in ((32, 8, 2), (64, 16, 2)):
    flow = configs.simple_realnvp_flow(features, 128, layers, blocks, seed=0).to(DEV).eval()
    for rows in (16384, 262144):
        x = torch.randn(rows, features, device=DEV)
        with torch.no_grad():
            fused = timed(lambda: flow.log_prob(x))
            kernel = ops.last_layer_kernel()
            AffineCouplingTransform.fuse_conditioner = False
            unfused = timed(lambda: flow.log_prob(x))
            AffineCouplingTransform.fuse_conditioner = True
        print("SimpleRealNVP(features=%d, hidden=128, layers=%d, blocks=%d) log_prob %7d rows: one launch %.3f ms (%s), layer by layer %.3f ms: %.1f x"
              % (features, layers, blocks, rows, fused, kernel, unfused, unfused / fused), flush=True)
