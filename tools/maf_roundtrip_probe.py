import sys, torch
sys.path.insert(0, '.')
from nflows_amd.transforms import autoregressive, CompositeTransform, InverseTransform
for trial in range(6):
    for urb, rm in [(False, False), (False, True), (True, False)]:
        t = autoregressive.MaskedAffineAutoregressiveTransform(features=20, hidden_features=30, num_blocks=5, use_residual_blocks=urb, random_mask=rm).cuda()
        x = torch.randn(10, 20, device='cuda')
        with torch.no_grad():
            for mode in ("tail", "loop"):
                t.__dict__["_sequential_steps_cache"] = None if mode == "tail" else 20
                xi, li = t.inverse(x); y, l = t(xi)
                print(trial, urb, rm, mode, "seq", t._sequential_steps(), "err %.2e lad %.2e" % ((y - x).abs().max().item(), (l + li).abs().max().item()))
