"""Clock and power while the bench kernel runs back to back (is the launch power-limited?):
    [NFLOWS_AMD_LIB=...] python tools/k8h_power.py [seconds]
Runs Flow.log_prob on the bench workload for `seconds` in a loop and samples `rocm-smi` (sclk, socket
power) from a thread; prints ms/step, the median clock / power during the loop and at idle."""
import os, sys, time, subprocess, threading, re, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nflows_amd import configs

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples = []
stop = False


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:
        return None
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None, out if not (sclk and pw) else "")


def sampler():
    while not stop:
        r = smi()
        if r:
            samples.append((time.perf_counter(),) + r[:2])
        time.sleep(0.2)


idle = smi()
flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval().cuda()
x = torch.randn(262144, 64, generator=torch.Generator().manual_seed(1234)).cuda()
with torch.no_grad():
    for _ in range(5):
        flow.log_prob(x)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            flow.log_prob(x)
        torch.cuda.synchronize()
        steps += 20
    dt = time.perf_counter() - t0
    stop = True
    th.join()
tag = os.path.basename(os.environ.get("NFLOWS_AMD_LIB", "product"))
clk = [s[1] for s in samples if s[1]]
pw = [s[2] for s in samples if s[2]]
print("%s: %.3f ms/step; sclk median %s MHz (min %s, max %s, %d samples); power median %s W (max %s); idle %s"
      % (tag, dt / steps * 1e3, statistics.median(clk) if clk else None, min(clk) if clk else None, max(clk) if clk else None,
         len(clk), statistics.median(pw) if pw else None, max(pw) if pw else None, idle[:2] if idle else None), flush=True)
if idle and idle[2]:
    print(idle[2][:1500])
