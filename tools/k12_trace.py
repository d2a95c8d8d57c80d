#!/usr/bin/env python3
"""Phase timeline of K12 (debug): cycle stamps of workgroup 0 / wave 0 over the first steps."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import _native, configs
dev = "cuda:0"
flow = configs.ar_rq_flow(784, 256, 8, 3.0, 2).to(dev).eval()
t = flow._transform._transforms[0]
z = torch.randn(4096, 784, device=dev)
with torch.no_grad():
    for _ in range(2): t.inverse(z)
    buf = torch.zeros(256, dtype=torch.int64, device=dev)
    lib = _native.load(); lib.nfa_debug_k7_trace.argtypes = [ctypes.c_void_p]
    lib.nfa_debug_k7_trace(ctypes.c_void_p(buf.data_ptr()))
    t.inverse(z)
    torch.cuda.synchronize(); lib.nfa_debug_k7_trace(None)
s = buf.cpu().numpy()
s = s[s > 0]
d = np.diff(s)
# per step 5 stamps: loop top | after top wait+barrier | after the unit chain | after the output rows | after the spline
n = (len(s) - 1) // 5
steps = d[:n * 5].reshape(n, 5)
print("per step: [top wait + barrier, request + unit chain, output rows, spline inverse, loop end -> next top]")
for i in list(range(0, min(n, 6))) + list(range(max(6, n - 4), n)):
    print("step %3d:" % i, steps[i].tolist(), "total", int(steps[i].sum()))
print("mean over steps 5..%d:" % n, steps[5:].mean(axis=0).round().tolist(), "total", steps[5:].sum(axis=1).mean().round())
