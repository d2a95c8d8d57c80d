import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import golden_realnvp_flow
from nflows_amd import ops
from nflows_amd.transforms import AffineCouplingTransform
flow_cpu, g, cfg = golden_realnvp_flow(os.path.join(ROOT, "tests", "golden"), sys.argv[1] if len(sys.argv) > 1 else "realnvp_affine")
flow = copy.deepcopy(flow_cpu).to("cuda:0")
x = torch.from_numpy(g[(sys.argv[1] if len(sys.argv) > 1 else "realnvp_affine") + "/x"]).to("cuda:0")
with torch.no_grad():
    z, lad = flow._transform(x)
    print(ops.last_layer_kernel())
    AffineCouplingTransform.fuse_conditioner = False
    z2, lad2 = flow._transform(x)
    per_layer = []
    h = x
    for t in flow._transform._transforms:
        h, l = t(h)
        per_layer.append(l)
    AffineCouplingTransform.fuse_conditioner = True
d = (lad - lad2).abs()
bad = (d > 1e-3).nonzero().flatten().tolist()
print("rows wrong:", len(bad), bad[:40], "...", bad[-10:])
pl = torch.stack(per_layer, 1)
for r in bad[:4] + bad[-2:]:
    print(r, "fused", float(lad[r]), "unfused", float(lad2[r]), "per layer", [round(float(v), 4) for v in pl[r]])
