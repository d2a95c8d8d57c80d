import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nflows_amd import configs, ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
def bits():
    w = ops._status_word(torch.device(DEV)); b = int(w.item()); w.zero_(); return b
for features, rows, layers in ((100, 4096, 4), (100, 128, 4), (100, 128, 1), (100, 256, 1), (128, 128, 1), (72, 128, 1), (66, 128, 2), (128, 4096, 4)):
    flow_cpu = configs.rq_nsf_flow(num_layers=layers, features=features, num_bins=8, hidden_features=128, seed=5).eval()
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    x = torch.randn(rows, features, generator=torch.Generator().manual_seed(features)).to(DEV)
    res = {}
    for eng in ("f16x3", "bf16x3"):
        RQ.conditioner_engine = eng
        with torch.no_grad():
            bits()
            z, lad = flow._transform(x); b1 = bits()
            xr, ladr = flow._transform.inverse(z if eng == "f16x3" else res["f16x3"][0]); b3 = bits()
        res[eng] = (z, lad, xr, ladr)
        print(features, rows, layers, eng, "status fwd/inv", hex(b1 & 0xffffffff), hex(b3 & 0xffffffff), ops.last_layer_kernel()[:60], flush=True)
    a, b = res["f16x3"], res["bf16x3"]
    print("   max diffs z %.2e lad %.2e xr %.2e ladr %.2e" % tuple(float((u - v).abs().max()) for u, v in zip(a, b)), flush=True)
