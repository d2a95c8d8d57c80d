import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nflows_amd import configs, ops
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
def bits():
    w = ops._status_word(torch.device(DEV)); b = int(w.item()); w.zero_(); return b
features, hidden, rows = 100, 128, 4096
flow_cpu = configs.rq_nsf_flow(num_layers=4, features=features, num_bins=8, hidden_features=hidden, seed=5).eval()
flow = copy.deepcopy(flow_cpu).to(DEV).eval()
x = torch.randn(rows, features, generator=torch.Generator().manual_seed(features)).to(DEV)
RQ.conditioner_engine = "f16x3"
with torch.no_grad():
    z, lad = flow._transform(x); print("fwd status", bits())
    for trial in range(3):
        cap = ops.capture_last_layer_logits()
        try:
            with cap:
                xr, ladr = flow._transform.inverse(z)
        except Exception as e:
            print("capture finish failed (expected in the debug build):", type(e).__name__, e)
        print("inv status", bits(), ops.last_layer_kernel(), "redo", ops.last_redo_blocks(), "launches", cap.launches, "lib", os.environ.get("NFLOWS_AMD_LIB"))
        d = cap._packed.view(torch.int32).view(-1)[: (rows // 128) * 256 * 4].view(-1, 256, 4).cpu()
        print("sample words", [hex(v & 0xffffffff) for v in d[0, :3].reshape(-1).tolist()], "isnan frac", float(torch.isnan(cap._packed).float().mean()))
        ok = (d[..., 3] == 0x600D)
        print("blocks written", int(ok.all(dim=1).sum()), "of", d.shape[0])
        bad = (d[..., 1] & ~7) != 0
        badm = (d[..., 2] & ~7) != 0
        print("threads with garbage quad_status", int(bad.sum()), "my_status", int(badm.sum()))
        if bad.any():
            idx = bad.nonzero()
            print("first 10 (block, tid):", idx[:10].tolist())
            print("layers of first garbage:", torch.unique(d[..., 0][bad]).tolist()[:16])
            print("values:", [hex(v & 0xffffffff) for v in d[..., 1][bad][:10].tolist()])
            print("tids:", torch.unique(idx[:, 1]).tolist()[:64])
            print("blocks:", torch.unique(idx[:, 0]).tolist()[:64])
