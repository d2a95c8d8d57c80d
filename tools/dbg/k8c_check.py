import copy, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nflows_amd import configs, ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
def bits():
    w = ops._status_word(torch.device(DEV)); b = int(w.item()); w.zero_(); return b
def timed(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
RQ.conditioner_engine = "f16x2"
for features, rows, layers in ((64, 128, 1), (64, 256, 2), (64, 8192, 32), (64, 16384, 32), (64, 1000, 4), (100, 640, 4), (24, 512, 3), (22, 1000, 4)):
    flow_cpu = configs.rq_nsf_flow(num_layers=layers, features=features, num_bins=8, hidden_features=128, seed=5).eval()
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    x = torch.randn(rows, features, generator=torch.Generator().manual_seed(features)).to(DEV)
    res = {}
    for name, k8c, k8s in (("k8c", True, True), ("k8s", False, True), ("k8h", False, False)):
        ops.K8C_ENABLED, ops.K8S_ENABLED = k8c, k8s
        with torch.no_grad():
            bits()
            z, lad = flow._transform(x); b1 = bits(); kf = ops.last_layer_kernel()[:34]; r1 = ops.last_redo_blocks()
            lp = flow.log_prob(x)
            xr, ladr = flow._transform.inverse(res["k8c"][0] if res else z); b3 = bits(); r3 = ops.last_redo_blocks()
            ms = timed(lambda: flow.log_prob(x)) if rows >= 8192 else 0.0
        res[name] = (z, lad, xr, ladr, lp)
        print(features, rows, layers, name, kf, "status %x %x redo %d %d  log_prob %.3f ms" % (b1 & 0xffffffff, b3 & 0xffffffff, r1, r3, ms), flush=True)
    for other in ("k8s", "k8h"):
        print("   k8c vs %s: max diffs z %.2e lad %.2e xr %.2e ladr %.2e lp %.2e" % ((other,) + tuple(float((u - v).abs().max()) for u, v in zip(res["k8c"], res[other]))), flush=True)
    print("   round trip k8c %.2e" % float((res["k8c"][2] - x).abs().max()))
