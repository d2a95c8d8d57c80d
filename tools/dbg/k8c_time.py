import copy, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nflows_amd import configs, ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
def timed(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
RQ.conditioner_engine = "f16x2"
os.environ["NFA_K8H_NOREDO"] = "1"
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=5).to(DEV).eval()
out = []
for rows in (2048, 8192, 16384):
    x = torch.randn(rows, 64, device=DEV)
    for name, k8c in (("k8c", True), ("k8s", False)):
        ops.K8C_ENABLED = k8c
        with torch.no_grad():
            ms = timed(lambda: flow.log_prob(x))
        out.append("%s@%d %.3f" % (name, rows, ms))
print(os.environ.get("NFLOWS_AMD_LIB", "product").split("/")[-1], " ".join(out), flush=True)
