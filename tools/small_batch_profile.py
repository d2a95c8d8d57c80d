"""rocprofv3 --kernel-trace --stats target (profiles/r6/k8c_kernel_stats.csv): the small-batch kernels on the 32-layer flow --
K8c (column split, 32-row workgroups) at 8 192 and 16 384 rows beside K8s, and the tails=None default flow in K8."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nflows_amd import configs, ops
from nflows_amd.transforms import CompositeTransform, RandomPermutation, PiecewiseRationalQuadraticCouplingTransform as RQ
from nflows_amd.nn.nets import ResidualNet
from nflows_amd.utils import torchutils
DEV = "cuda:0"
RQ.conditioner_engine = "f16x2"
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).to(DEV).eval()
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    for rows in (8192, 16384):
        x = torch.randn(rows, 64, device=DEV)
        for k8c in (True, False):
            ops.K8C_ENABLED = k8c
            ms = timed(lambda: flow.log_prob(x))
            print("%6d rows  %-28s %.3f ms" % (rows, ops.last_layer_kernel().split("<")[0], ms), flush=True)
    ops.K8C_ENABLED = True
    torch.manual_seed(0)
    t = CompositeTransform(sum([[RandomPermutation(64), RQ(torchutils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
                                lambda a, b: ResidualNet(a, b, hidden_features=128, num_blocks=2), num_bins=10, tails=None)]
                                for i in range(32)], [])).to(DEV).eval()
    x = torch.rand(65536, 64, device=DEV) * 0.98 + 0.01
    print(" 65536 rows  tails=None, K=10: %.3f ms  %s" % (timed(lambda: t(x), 10), ops.last_layer_kernel()[:80]), flush=True)
