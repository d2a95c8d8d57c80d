"""Round 4: where does the inverse log-determinant of the K12 + K13 path differ on the steep autoregressive fixture?"""
import os, sys, copy
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import steep_flow
from oracle import eager
from nflows_amd.transforms import MaskedPiecewiseRationalQuadraticAutoregressiveTransform as AR
case = "steep_ar_rq"
flow_cpu, g, cfg = steep_flow(os.path.join(ROOT, "tests", "golden"), case)
noise = torch.from_numpy(g[case + "/noise"])
flow = copy.deepcopy(flow_cpu).to("cuda:0").eval()
res = {}
for tag, k13, k12 in (("k13_k12", True, True), ("k13_only", True, False), ("k12_only", False, True), ("plain", False, False)):
    AR.fuse_output_layer, AR.fuse_sequential_inverse = k13, k12
    with torch.no_grad():
        xi, ladi = flow._transform.inverse(noise.to("cuda:0"))
    res[tag] = (xi.cpu(), ladi.cpu())
t64 = (torch.from_numpy(g[case + "/inv_x64"]), torch.from_numpy(g[case + "/inv_lad64"]))
r32 = (torch.from_numpy(g[case + "/inv_x"]), torch.from_numpy(g[case + "/inv_lad"]))
print("reference fp32: x max %.3e lad max %.3e (row %d)" % ((r32[0].double() - t64[0]).abs().max(), (r32[1].double() - t64[1]).abs().max(), int((r32[1].double() - t64[1]).abs().argmax())))
f64 = copy.deepcopy(flow_cpu).double()
for tag, (xi, ladi) in res.items():
    ex = (xi.double() - t64[0]).abs()
    el = (ladi.double() - t64[1]).abs()
    row = int(el.argmax())
    with torch.no_grad():
        _, lad_fwd = eager.flow_transform(f64, xi.double())     # float64 forward log-det AT the kernel's own x
    incons = (ladi.double() + lad_fwd).abs()
    print("%-9s x max %.3e mean %.3e | lad max %.3e mean %.3e worst row %d (x err in that row %.3e at col %d) | lad + lad64_fwd(x): max %.3e row %d"
          % (tag, ex.max(), ex.mean(), el.max(), el.mean(), row, ex[row].max(), int(ex[row].argmax()), incons.max(), int(incons.argmax())))
# per-feature decomposition on the worst row of the fused path
xi, ladi = res["k13_k12"]
row = int((ladi.double() - t64[1]).abs().argmax())
net64 = f64._transform._transforms[0]
with torch.no_grad():
    params = net64.autoregressive_net(t64[0][row:row + 1]).view(1, cfg["D"], -1)
    K = cfg["K"]
    from oracle.eager import rqs_unconstrained
    y, l = rqs_unconstrained(t64[0][row:row + 1], params[..., :K], params[..., K:2 * K], params[..., 2 * K:], inverse=False, tail_bound=3.0)
    print("row", row, "noise", noise[row].numpy()[:8], "...")
    print("per-feature fp64 forward lad at true x:", np.round(l.numpy()[0], 3))
    print("x err per feature (fused):", (xi[row].double() - t64[0][row]).abs().numpy())

print("--- larger samples: 4096 rows x 3 seeds, float64 eager port as truth, fp32 eager port as reference")
def stats(e):
    e = e.numpy().reshape(-1)
    return "max %.2e mean %.2e q999 %.2e" % (e.max(), e.mean(), np.quantile(e, 0.999))
for seed in (1, 2, 3):
    nz = torch.randn(4096, cfg["D"], generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        x32, l32 = eager.flow_transform(flow_cpu, nz, inverse=True)
        x64, l64 = eager.flow_transform(f64, nz.double(), inverse=True)
    print("seed", seed, "reference fp32: x", stats((x32.double() - x64).abs()), "| lad", stats((l32.double() - l64).abs()))
    for tag, k13, k12 in (("k13_k12", True, True), ("plain", False, False)):
        AR.fuse_output_layer, AR.fuse_sequential_inverse = k13, k12
        with torch.no_grad():
            xi, ladi = flow._transform.inverse(nz.to("cuda:0"))
        print("   %-8s       x" % tag, stats((xi.cpu().double() - x64).abs()), "| lad", stats((ladi.cpu().double() - l64).abs()))
