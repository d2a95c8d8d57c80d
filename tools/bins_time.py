"""Times Flow.log_prob and the inverse pass of the BASELINE flow shape (32 layers, D = 64, ResidualNet H = 128 x 2 blocks)
for every bin count the whole-layer kernels serve, on the whole-layer path (K8h: the run of layers in one launch) and
on the path those layers took before round 4 (conditioner GEMMs + K1, layer by layer):
    python tools/bins_time.py [rows] > profiles/r4/bins_time.txt"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs, ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
x = torch.randn(rows, 64, generator=torch.Generator().manual_seed(1234)).cuda()


def ms(fn, steps=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


print("rows %d, 32 layers, D = 64: ms per Flow.log_prob | ms per inverse pass | kernel" % rows)
for K in [int(a) for a in sys.argv[2:]] or (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 20, 24, 32):
    flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=K, hidden_features=128, seed=0).eval().cuda()
    out = []
    for fused in (True, False):
        RQ.fuse_conditioner = fused
        RQ.fuse_final_linear = fused
        t_f = ms(lambda: flow.log_prob(x))
        label = ops.last_layer_kernel()
        t_i = ms(lambda: flow._transform.inverse(x))
        out.append((t_f, t_i, label))
    RQ.fuse_conditioner = RQ.fuse_final_linear = True
    (a, b, la), (c, d, lb) = out
    print("K=%2d  whole-layer %.3f | %.3f  [%s]   GEMMs + K1 %.3f | %.3f  [%s]   speed-up %.2f x | %.2f x"
          % (K, a, b, la.split("<")[0], c, d, lb.split("<")[0], c / a, d / b), flush=True)
