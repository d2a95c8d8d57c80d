#!/usr/bin/env python3
"""(measurement script, not collected by pytest; lives under tests/ because it uses the oracle)
Tail of |inv(fwd(x)) - x| for the 32-layer flow: HIP path vs the reference's own fp32 path (eager
port) on the same rows -- counts above thresholds, the worst elements, and where the HIP path's worst
element stands in the reference (and vice versa).  Usage: python tests/probes/fwd_inv_tail_probe.py [rows]"""
import os, sys, copy
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nflows_amd import configs
from oracle import eager
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
flow_cpu = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
x = torch.randn(65536, 64, generator=torch.Generator().manual_seed(1234))[:rows]
with torch.no_grad():
    z32, _ = eager.flow_transform(flow_cpu, x)
    xr32, _ = eager.flow_transform(flow_cpu, z32, inverse=True)
    f64 = copy.deepcopy(flow_cpu).double()
    z64, _ = eager.flow_transform(f64, x.double())
flow = copy.deepcopy(flow_cpu).cuda().eval()
with torch.no_grad():
    z, _ = flow._transform(x.cuda())
    xr, _ = flow._transform.inverse(z)
    xr_from_ref_z, _ = flow._transform.inverse(z32.cuda())
e_hip = (xr.cpu() - x).abs().numpy()
e_ref = (xr32 - x).abs().numpy()
e_hip_on_refz = (xr_from_ref_z.cpu() - x).abs().numpy()
for name, e in (("hip fwd+inv", e_hip), ("reference fp32 fwd+inv", e_ref), ("hip inverse of the reference's z", e_hip_on_refz)):
    print("%-34s max %.3e  mean %.3e  q999 %.3e  #>1e-3 %5d  #>3e-3 %4d  #>1e-2 %3d" % (
        name, e.max(), e.mean(), np.quantile(e, 0.999), (e > 1e-3).sum(), (e > 3e-3).sum(), (e > 1e-2).sum()))
i = np.unravel_index(e_hip.argmax(), e_hip.shape)
j = np.unravel_index(e_ref.argmax(), e_ref.shape)
print("hip worst element %s: hip %.3e, reference there %.3e; forward error there: hip %.3e ref %.3e" % (
    i, e_hip[i], e_ref[i], abs(z.cpu().numpy()[i[0]] - z64.numpy()[i[0]]).max(), abs(z32.numpy()[i[0]] - z64.numpy()[i[0]]).max()))
print("ref worst element %s: ref %.3e, hip there %.3e" % (j, e_ref[j], e_hip[j]))
# row-level: max error per row, top 5 rows each
print("top rows hip:", np.argsort(-e_hip.max(axis=1))[:6], np.sort(e_hip.max(axis=1))[::-1][:6])
print("top rows ref:", np.argsort(-e_ref.max(axis=1))[:6], np.sort(e_ref.max(axis=1))[::-1][:6])
