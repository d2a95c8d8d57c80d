import copy, math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import nflows_amd
from nflows_amd import ops
from nflows_amd.nn.nets import ResidualNet
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
torch.manual_seed(5)
for H, nb in ((128, 2), (128, 0)):
    mask = torch.ones(64); mask[::2] = -1
    layer = RQ(mask, lambda i, o: ResidualNet(i, o, hidden_features=H, num_blocks=nb), num_bins=8, tails="linear", tail_bound=3.0).eval()
    with torch.no_grad():
        layer.transform_net.final_layer.weight.mul_(30.0)
    layer = layer.to(DEV)
    x = torch.randn(4096, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    xi = x[:, layer.identity_features]
    with torch.no_grad():
        t64 = copy.deepcopy(layer.transform_net).double()(xi.double()).view(4096, 32, 23)
        t64[..., :16] /= math.sqrt(H)
        hid64 = copy.deepcopy(layer.transform_net).double().hidden(xi.double()) if hasattr(layer.transform_net, "hidden") else None
    for eng in ("f16x3",):
        RQ.conditioner_engine = eng
        ops.K8S_ENABLED = False
        with torch.no_grad():
            z0, l0 = layer(x)
            with ops.capture_last_layer_logits() as cap:
                z1, l1 = layer(x)
            z2, l2 = layer(x)
        lg = cap.logits[:, :32].double()
        err = (lg - t64).abs()
        print("H=%d nb=%d %-7s %s | logits err mean %.3e max %.3e | twin==plain %s rerun==plain %s redo %s" % (
            H, nb, eng, ops.last_layer_kernel()[:40], err.mean().item(), err.max().item(), torch.equal(z0, z1), torch.equal(z0, z2),
            None if cap.redo is None else int((cap.redo != 0).sum())))
        if eng == "f16x3":
            # by feature and by logit index
            print("   per-logit-index mean err:", " ".join("%.1e" % v for v in err.mean((0, 1)).tolist()))
            print("   per-feature mean err    :", " ".join("%.1e" % v for v in err.mean((0, 2)).tolist()))
            print("   per-row-in-block (mod 32) mean err:", " ".join("%.1e" % v for v in err.mean((1, 2)).view(-1, 32).mean(0).tolist()))
