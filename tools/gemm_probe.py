#!/usr/bin/env python3
"""Times the conditioner GEMM shapes (B=65536) under PyTorch-ROCm: default backend, the other
BLAS library, and TunableOp.  Informational (the conditioner stays PyTorch's)."""
import os, sys, time, torch
B = 65536
shapes = [(32, 128), (128, 128), (128, 736)]
dev = "cuda:0"

def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

def run(tag):
    for (k, n) in shapes:
        x = torch.randn(B, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
        us = bench(lambda: torch.addmm(b, x, w.t()))
        us2 = bench(lambda: torch._addmm_activation(b, x, w.t()))
        print("%-22s K=%4d N=%4d  addmm %7.1f us (%5.1f TF)   addmm+relu %7.1f us" % (tag, k, n, us, 2*B*k*n/us/1e6, us2), flush=True)

print("preferred blas:", torch.backends.cuda.preferred_blas_library())
run("default")
for lib in ("hipblaslt", "rocblas" if hasattr(torch.backends.cuda, "preferred_blas_library") else None):
    if lib is None: continue
    try:
        torch.backends.cuda.preferred_blas_library(lib if lib != "rocblas" else "cublas")
        run("prefer " + lib)
    except Exception as ex:
        print("prefer", lib, "failed:", ex)
torch.backends.cuda.preferred_blas_library("default")
try:
    import torch.cuda.tunable as tn
    tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(30)
    tn.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tunableop_results.csv"))
    t0 = time.time(); run("tunableop"); print("tuning+run took %.1f s" % (time.time() - t0))
    tn.write_file()
except Exception as ex:
    print("tunableop failed:", repr(ex))
