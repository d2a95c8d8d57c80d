import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs, ops
for features in (63, 43, 64):
    flow = configs.rq_nsf_flow(num_layers=4, features=features, num_bins=8, hidden_features=64, seed=features)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if "final_layer" in name:
                p.mul_(4.0)
            elif "linear_layers.1" in name:
                p.mul_(30.0)
    flow = flow.cuda().eval()
    x = torch.randn(300, features, generator=torch.Generator().manual_seed(7)).cuda()
    for k8s in (True, False):
        ops.K8S_ENABLED = k8s
        with torch.no_grad():
            z, lad = flow._transform(x)
            xr, lad_inv = flow._transform.inverse(z)
        d = (lad + lad_inv).abs()
        print("D=%d k8s=%s: max|lad+lad_inv| %.3e (row %d), max|xr-x| %.3e, redo %s" % (features, k8s, float(d.max()), int(d.argmax()), float((xr - x).abs().max()), ops.last_redo_blocks()))
