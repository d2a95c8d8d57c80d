"""Runs tools/bin/mfma_toggle_probe and samples rocm-smi (sclk, socket power) while it runs."""
import os, re, subprocess, sys, threading, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            p = re.search(r"Graphics Package Power \(W\): ([\d.]+)", out)
            samples.append((time.perf_counter(), int(c.group(1)) if c else None, float(p.group(1)) if p else None))
        except Exception:
            pass
        time.sleep(0.2)


th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
p = subprocess.Popen([os.path.join(ROOT, "tools/bin/mfma_toggle_probe"), sys.argv[1] if len(sys.argv) > 1 else "4"], stdout=subprocess.PIPE, text=True)
marks = []
for line in p.stdout:
    marks.append((time.perf_counter(), line.strip()))
p.wait()
stop = True
th.join()
prev = t0
for t, line in marks:
    seg = [s for s in samples if prev + 1.0 < s[0] < t and s[1]]
    clk = [s[1] for s in seg]
    pw = [s[2] for s in seg if s[2]]
    print("%s | sclk median %s MHz, power median %s W (%d samples)" % (line, statistics.median(clk) if clk else None, statistics.median(pw) if pw else None, len(seg)))
    prev = t
