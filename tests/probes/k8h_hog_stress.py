"""Round 5: is the weight stream of the whole-layer kernels sensitive to memory latency?  A second stream keeps HBM busy
(read-modify-write sweeps over 2 GB) WHILE the kernel under test runs on the main stream; every launch is compared bit for
bit with the result of the same launch on a quiet device.  A hole in the LDS-DMA ring protocol (a fragment read before its
stage has landed) would show under stretched latencies on any box.  Usage:
    python tests/probes/k8h_hog_stress.py [reps] [case] [rows per launch]"""
import copy, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import steep_flow
from test_gpu_steep import _batch
from nflows_amd import ops
DEV = "cuda:0"
golden = os.path.join(ROOT, "tests", "golden")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
case = sys.argv[2] if len(sys.argv) > 2 else "act_tanh_k10"
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
k8s = os.environ.get("PROBE_K8S", "0") == "1"
fixture = {"act": "flows_acts.npz", "ste": "flows_steep.npz", "bin": "flows_bins.npz"}[case[:3]]
ops.K8S_ENABLED = k8s
flow_cpu, g, cfg = steep_flow(golden, case, fixture)
x = _batch(g, case, "x", 65536, cfg["D"]).to(DEV)
noise = _batch(g, case, "noise", 65536, cfg["D"]).to(DEV)
flow = copy.deepcopy(flow_cpu).to(DEV).eval()
if os.environ.get("PROBE_LAYERS"):       # only the first n (Permutation, coupling) pairs of the fixture's flow
    from nflows_amd.transforms import CompositeTransform
    n = int(os.environ["PROBE_LAYERS"])
    layers = list(flow._transform._transforms)[:2 * n]
    flow._transform = CompositeTransform(layers)
    for i in range(n):
        perm = layers[2 * i]._permutation.tolist()
        tf = layers[2 * i + 1].transform_features.tolist()
        print("   layer %d: transformed features (layer order -> input column through the permutation): %s" % (i, [(j, perm[j]) for j in tf]))
quiet = {}
with torch.no_grad():
    for lo in range(0, 65536, rows):
        for _ in range(3):                      # (a cold first launch can itself be a deviating one)
            flow._transform(x[lo:lo + rows]); flow._transform.inverse(noise[lo:lo + rows])
        torch.cuda.synchronize()
        quiet[(lo, "fwd")] = tuple(t.clone() for t in flow._transform(x[lo:lo + rows]))
        kern = ops.last_layer_kernel()
        redo_f = ops.last_redo_blocks()
        quiet[(lo, "inv")] = tuple(t.clone() for t in flow._transform.inverse(noise[lo:lo + rows]))
        print("   quiet launch at row %d: blocks handed to the exact kernel: forward %s, inverse %s" % (lo, redo_f, ops.last_redo_blocks()))
torch.cuda.synchronize()
side = torch.cuda.Stream()
hog = torch.zeros(1 << 29, device=DEV)       # 2 GB
HOG = os.environ.get("PROBE_HOG", "add")
hog_i = hog.view(torch.int32)
hog2 = torch.empty_like(hog) if HOG == "copy" else None
mm_a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16) if HOG == "mm" else None
bad = launches = 0
t0 = time.time()
with torch.no_grad():
    for it in range(reps):
        mode = os.environ.get("PROBE_MODE", "concurrent")
        if mode == "concurrent":
            with torch.cuda.stream(side):
                for _ in range(6):
                    if HOG == "add":
                        hog.add_(1.0)              # 4 GB of traffic per sweep, ~1 ms each: the launches below overlap them
                    elif HOG == "iadd":            # the same traffic, integer arithmetic
                        hog_i.add_(1)
                    elif HOG == "copy":            # the same traffic, no arithmetic
                        hog2.copy_(hog)
                    elif HOG == "fill":            # writes only
                        hog.fill_(1.0)
                    elif HOG == "mm":              # matrix cores and LDS, little memory traffic
                        torch.mm(mm_a, mm_a)
                    elif HOG == "sin":             # arithmetic on a cache-resident buffer: VALU / transcendental units, no HBM traffic
                        for _ in range(8):
                            torch.sin_(hog[:1 << 22])
        for lo in range(0, 65536, rows):
            for direction, src in (("fwd", x), ("inv", noise)):
                fn = flow._transform if direction == "fwd" else flow._transform.inverse
                if mode == "before_add":         # a foreign kernel right before, SAME stream: no overlap, only what it leaves behind
                    hog[:1 << 26].add_(1.0)
                elif mode == "before_nan":
                    hog[:1 << 26].fill_(float("nan"))
                elif mode == "before_softmax":
                    torch.softmax(hog[:1 << 24].view(-1, 64), dim=1)
                elif mode == "before_sort":
                    torch.sort(hog[:1 << 22].view(-1, 256), dim=1)
                z, lad = fn(src[lo:lo + rows])
                launches += 1
                qz, ql = quiet[(lo, direction)]
                if not (torch.equal(torch.nan_to_num(z), torch.nan_to_num(qz)) and torch.equal(torch.nan_to_num(lad), torch.nan_to_num(ql))):
                    bad += 1
                    d = (torch.nan_to_num(z) != torch.nan_to_num(qz)).any(1).nonzero().flatten()
                    if bad <= 8:
                        wg_rows = 256 if "waves=8" in kern else 128
                        for wg in sorted(set((d // wg_rows).tolist()))[:6]:
                            rw = d[(d // wg_rows) == wg] - wg * wg_rows
                            dz = (torch.nan_to_num(z) - torch.nan_to_num(qz)).abs()[wg * wg_rows:(wg + 1) * wg_rows]
                            cols = (dz > 0).any(0).nonzero().flatten().tolist()
                            print("      workgroup %d (XCD %d): changed rows per wave %s, columns changed %s, max |diff| %.2e, lad rows changed %d"
                                  % (wg, wg % 8, [int(((rw // 32) == w).sum()) for w in range(wg_rows // 32)], cols[:20], float(dz.max()),
                                     int((torch.nan_to_num(lad) != torch.nan_to_num(ql))[wg * wg_rows:(wg + 1) * wg_rows].sum())))
                    if bad <= 5:
                        flags = ops._last_redo
                        print("      blocks flagged in this launch:", (flags != 0).nonzero().flatten().tolist()[:12])
                        print("   DEVIATION it %d rows [%d, %d) %s: %d rows changed, blocks %s, max |diff| %.3e"
                              % (it, lo, lo + rows, direction, d.numel(), sorted(set((d // 128).tolist()))[:10], float((torch.nan_to_num(z) - torch.nan_to_num(qz)).abs().max())))
        side.synchronize()
print("%s rows/launch %d  %s: %d of %d launches beside hog=%s deviate from the quiet result (%.1f s)" % (case, rows, kern.split("<")[0] + "<" + kern.split("<")[1][:58], bad, launches, HOG, time.time() - t0))
