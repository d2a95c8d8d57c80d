import copy, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nflows_amd
from nflows_amd import ops, configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
for features, hidden, rows, layers in ((22, 64, 1000, 4), (100, 128, 4100, 4), (64, 128, 128, 4), (100, 128, 4096, 2), (72, 128, 4096, 2), (64, 128, 4096, 2), (22, 128, 1024, 2), (24, 128, 1024, 2)):
    flow_cpu = configs.rq_nsf_flow(num_layers=layers, features=features, num_bins=8, hidden_features=hidden, seed=5).eval()
    for t in flow_cpu._transform._transforms:
        if hasattr(t, "transform_net"):
            with torch.no_grad():
                t.transform_net.final_layer.weight.mul_(20.0)
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    x = torch.randn(rows, features, generator=torch.Generator().manual_seed(features)).to(DEV)
    out = {}
    for eng in ("f16x3", "bf16x3"):
        RQ.conditioner_engine = eng
        with torch.no_grad():
            z, lad = flow._transform(x)
            kf = ops.last_layer_kernel()
            xr, ladr = flow._transform.inverse(z)
            ki = ops.last_layer_kernel()
            noise = torch.randn(rows, features, generator=torch.Generator().manual_seed(1)).to(DEV)
            xi, ladi = flow._transform.inverse(noise)
        out[eng] = (z, lad, xr, xi, ladi)
        print("D=%d H=%d rows=%d L=%d %-6s roundtrip max %.2e mean %.2e | redo %s | %s" % (features, hidden, rows, layers, eng, (xr - x).abs().max().item(), (xr - x).abs().mean().item(), ops.last_redo_blocks() if eng == "f16x3" else None, ki[:60]))
    a, b = out["f16x3"], out["bf16x3"]
    print("    fwd z diff max %.2e | inverse(noise) diff max %.2e mean %.2e, ladi diff max %.2e; bad rows %s" % ((a[0] - b[0]).abs().max().item(), (a[3] - b[3]).abs().max().item(), (a[3] - b[3]).abs().mean().item(), (a[4] - b[4]).abs().max().item(),
          ((a[3] - b[3]).abs().max(1).values > 1e-2).nonzero().flatten()[:10].tolist()))
