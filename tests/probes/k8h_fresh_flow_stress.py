"""Round 5: like k8h_determinism_stress.py, but every iteration builds a FRESH flow object on the device (new parameter
tensors, new packed weight stream, optionally after torch.cuda.empty_cache(): memory the GPU has not touched) and looks at
its FIRST launches -- the situation of a parity test, and of a user's first call.  Usage:
    python tests/probes/k8h_fresh_flow_stress.py [reps] [case] [rows per launch] [empty_cache 0|1]"""
import copy, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import steep_flow
from test_gpu_steep import _batch
from nflows_amd import ops
DEV = "cuda:0"
golden = os.path.join(ROOT, "tests", "golden")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
case = sys.argv[2] if len(sys.argv) > 2 else "act_tanh_k10"
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
empty = (sys.argv[4] == "1") if len(sys.argv) > 4 else True
fixture = {"act": "flows_acts.npz", "ste": "flows_steep.npz", "bin": "flows_bins.npz"}[case[:3]]
ops.K8S_ENABLED = False
flow_cpu, g, cfg = steep_flow(golden, case, fixture)
x_cpu = _batch(g, case, "x", 65536, cfg["D"])
ref = None
bad = 0
t0 = time.time()
for it in range(reps):
    if empty:
        torch.cuda.empty_cache()
    if os.environ.get("PROBE_POISON"):
        # memory the caching allocator hands out next holds garbage (what a process that ran other work looks like):
        # blocks of many sizes filled with NaN bit patterns / large values, then released to the allocator
        junk = [torch.full((n,), float(os.environ["PROBE_POISON"]), device=DEV) for n in (1 << 28, 1 << 24, 1 << 20, 1 << 16, 1 << 12)] + \
               [torch.full((n,), float(os.environ["PROBE_POISON"]), device=DEV) for n in [1 << 22] * 64 + [1 << 18] * 256 + [1 << 14] * 1024]
        del junk
    flow = copy.deepcopy(flow_cpu).to(DEV).eval()
    outs = []
    redo_flags = []
    with torch.no_grad():
        for lo in range(0, 65536, rows):
            z, lad = flow._transform(x_cpu[lo:lo + rows].to(DEV))
            outs.append((z, lad))
            redo_flags.append((lo, ops._last_redo))      # (looked at after the iteration: no synchronisation between launches)
    z = torch.cat([o[0] for o in outs]); lad = torch.cat([o[1] for o in outs])
    for lo, flags in redo_flags:
        flagged = (flags != 0).nonzero().flatten()
        if flagged.numel():
            print("   iteration %d launch at row %d: blocks handed to the exact kernel: %s" % (it, lo, (flagged + lo // 128).tolist()[:16]))
    if ref is None:
        ref = (z.clone(), lad.clone())
        kern = ops.last_layer_kernel()
        continue
    if not torch.equal(torch.nan_to_num(z), torch.nan_to_num(ref[0])):
        bad += 1
        d = (torch.nan_to_num(z) != torch.nan_to_num(ref[0])).any(1).nonzero().flatten()
        print("   DEVIATION iteration %d: %d rows changed, first %s, blocks of 128: %s, max |diff| %.3e"
              % (it, d.numel(), d[:8].tolist(), sorted(set((d // 128).tolist()))[:12], float((torch.nan_to_num(z) - torch.nan_to_num(ref[0])).abs().max())))
        diff = (torch.nan_to_num(z) != torch.nan_to_num(ref[0]))
        for blk in sorted(set((d // 128).tolist()))[:6]:
            rows_b = d[(d // 128) == blk] - blk * 128
            per_wave = [int(((rows_b // 32) == w).sum()) for w in range(4)]
            cols = diff[blk * 128:(blk + 1) * 128].any(0).nonzero().flatten().tolist()
            flagged = any(bool(f[blk - lo // 128] != 0) for lo, f in redo_flags if lo // 128 <= blk < lo // 128 + f.numel())
            print("      block %d: rows per wave %s, columns changed %d of %d %s, handed to the exact kernel: %s, lad rows changed %d"
                  % (blk, per_wave, len(cols), z.shape[1], cols[:10], flagged,
                     int((torch.nan_to_num(lad[blk * 128:(blk + 1) * 128]) != torch.nan_to_num(ref[1][blk * 128:(blk + 1) * 128])).sum())))
    del flow
print("%s rows/launch %d empty_cache %d  %s: %d of %d fresh flows deviate from the first (%.1f s)" % (case, rows, empty, kern.split("<")[1][:60], bad, reps - 1, time.time() - t0))
