"""Clock / power / launch time of K1 (the unfused spline kernel) running back to back for a few seconds:
    [NFA_K1_WAVETILE=0] [NFLOWS_AMD_LIB=...] python tools/k1_power.py [seconds] [rows]
Prints the launch-time trajectory (HIP events around every launch) and rocm-smi's sclk / socket power during the run."""
import os, sys, time, subprocess, threading, re, statistics
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
samples, stop = [], False


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    mclk = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
    fclk = re.search(r"fclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None,
            int(mclk.group(1)) if mclk else None, int(fclk.group(1)) if fclk else None)


def sampler():
    while not stop:
        r = smi()
        if r:
            samples.append(r)
        time.sleep(0.15)


dev = "cuda:0"
D, K = 64, 8
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, D, device=dev, generator=g)
tidx = torch.arange(0, D, 2, device=dev)
params = [torch.randn(B, 32 * 23, device=dev, generator=g) for _ in range(4)]
spec = ops.make_rqs_spec(K, "linear", tail_bound=3.0, wh_divisor=float(np.sqrt(128)))
idle = smi()
for i in range(3):
    ops.rqs_coupling(x, params[i % 4], tidx, spec)
torch.cuda.synchronize()
time.sleep(1.0)
th = threading.Thread(target=sampler)
th.start()
evs = []
t0 = time.perf_counter()
i = 0
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.rqs_coupling(x, params[i % 4], tidx, spec)
        e.record()
        evs.append((s, e))
        i += 1
    torch.cuda.synchronize()
stop = True
th.join()
us = [s.elapsed_time(e) * 1e3 for s, e in evs]
n = len(us)
print("%s: %d launches; first 10: %s | launches 50-60: %s | last 10: %s" % (
    ops.last_layer_kernel(), n, " ".join("%.0f" % v for v in us[:10]), " ".join("%.0f" % v for v in us[50:60]),
    " ".join("%.0f" % v for v in us[-10:])))
print("median first 20 %.1f us; median of the second half %.1f us (%.0f GB/s)" % (
    statistics.median(us[:20]), statistics.median(us[n // 2:]), 4 * (2 * B * D + B * 736 + B) / statistics.median(us[n // 2:]) / 1e3))
print("idle sclk/power/mclk/fclk", idle, "| during: sclk median", statistics.median([s[0] for s in samples if s[0]]),
      "min", min(s[0] for s in samples if s[0]), "power median", statistics.median([s[1] for s in samples if s[1]]),
      "max", max(s[1] for s in samples if s[1]), "mclk", sorted(set(s[2] for s in samples)), "fclk", sorted(set(s[3] for s in samples)))
