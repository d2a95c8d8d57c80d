"""Times Flow.log_prob of the bench flow (32 layers, D = 64, K = 8) for one library build:
    NFLOWS_AMD_LIB=build_variants/<v>.so NFA_K8H_NOREDO=1 python tools/k8h_time.py [rows ...]
One line per batch size: ms per step (torch events around 20 steps).  Used by tools/k8h_ablation.sh."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs

rows = [int(a) for a in sys.argv[1:]] or [262144]
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval().cuda()
tag = os.path.basename(os.environ.get("NFLOWS_AMD_LIB", "product"))
for B in rows:
    x = torch.randn(B, 64, generator=torch.Generator().manual_seed(1234)).cuda()
    with torch.no_grad():
        for _ in range(5):
            lp = flow.log_prob(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lp = flow.log_prob(x)
        e1.record()
        torch.cuda.synchronize()
    print("%s rows %d: %.4f ms/step  (mean log_prob %.4f)" % (tag, B, e0.elapsed_time(e1) / 20, lp.double().mean().item()), flush=True)
