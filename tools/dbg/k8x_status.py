import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nflows_amd import configs, ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
def bits():
    w = ops._status_word(torch.device(DEV)); b = int(w.item()); w.zero_(); return b
for features, hidden, rows in ((22, 64, 1000), (100, 128, 4100), (64, 128, 128), (64, 128, 1000), (22, 128, 1024), (24, 128, 1024), (100, 128, 4096)):
    flow_cpu = configs.rq_nsf_flow(num_layers=4, features=features, num_bins=8, hidden_features=hidden, seed=5).eval()
    for steep in (1.0, 20.0):
        for t in flow_cpu._transform._transforms:
            if hasattr(t, "transform_net"):
                with torch.no_grad():
                    t.transform_net.final_layer.weight.mul_(steep)
        flow = copy.deepcopy(flow_cpu).to(DEV).eval()
        x = torch.randn(rows, features, generator=torch.Generator().manual_seed(features)).to(DEV)
        for eng in ("f16x3", "bf16x3", "f16x2"):
            RQ.conditioner_engine = eng
            with torch.no_grad():
                bits()
                z, lad = flow._transform(x); b1 = bits(); k1 = ops.last_layer_kernel()[:40]; r1 = ops.last_redo_blocks() if eng != "bf16x3" else -1
                lp = flow.log_prob(x); b2 = bits()
                xr, ladr = flow._transform.inverse(z); b3 = bits(); r3 = ops.last_redo_blocks() if eng != "bf16x3" else -1
            print(features, hidden, rows, steep, eng, "status fwd/lp/inv", b1, b2, b3, "redo", r1, r3, k1, flush=True)
