import sys, torch
sys.path.insert(0, '.')
from nflows_amd.transforms import MaskedPiecewiseRationalQuadraticAutoregressiveTransform as AR
torch.manual_seed(64 + 48)
t = AR(features=64, hidden_features=48, num_bins=8, tails="linear", tail_bound=3.0, num_blocks=2, use_residual_blocks=True).cuda().eval()
with torch.no_grad():
    for p in t.parameters(): p.mul_(1.5)
z = (2.0 * torch.randn(100, 64, generator=torch.Generator().manual_seed(1))).cuda()
with torch.no_grad():
    outs = []
    for i in range(6):
        x, lad = t.inverse(z); outs.append((x.clone(), lad.clone()))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs)
    AR.fuse_sequential_inverse = False
    xr, ladr = t.inverse(z)
    AR.fuse_sequential_inverse = True
    x, lad = outs[0]
    zz, lf = t(x); zzr, lfr = t(xr)
    print("deterministic", same, "x diff %.2e lad diff %.2e | fwd(inv) err fused %.2e loop %.2e | lad roundtrip fused %.2e loop %.2e" % (
        (x - xr).abs().max().item(), (lad - ladr).abs().max().item(), (zz - z).abs().max().item(), (zzr - z).abs().max().item(),
        (lad + lf).abs().max().item(), (ladr + lfr).abs().max().item()))
    i = (lad + lf).abs().argmax().item()
    print("worst row", i, "lad", lad[i].item(), "ladr", ladr[i].item(), "fwd lad of x", lf[i].item(), "fwd lad of xr", lfr[i].item(), "max|x-xr| row", (x[i]-xr[i]).abs().max().item())
