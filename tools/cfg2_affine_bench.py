import sys, time, json, torch
sys.path.insert(0, '.')
from nflows_amd import configs
from nflows_amd.transforms import AffineCouplingTransform
flow = configs.affine_coupling_flow(8, 32, (128, 128)).cuda().eval()
x = torch.randn(16384, 32, device='cuda')
def timed(fn, reps=500, warm=500):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
with torch.no_grad():
    for fused in (True, False):
        AffineCouplingTransform.fuse_conditioner = fused
        z, _ = flow._transform(x)
        print(json.dumps({"config": "cfg2 affine x8 D=32 B=16384", "whole_run_kernel": fused,
                          "log_prob_ms": timed(lambda: flow.log_prob(x)) * 1e3,
                          "inverse_ms": timed(lambda: flow._transform.inverse(z)) * 1e3}))
    AffineCouplingTransform.fuse_conditioner = True
    g = torch.cuda.CUDAGraph()
    lp = flow.log_prob(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        flow.log_prob(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        lp = flow.log_prob(x)
    print(json.dumps({"config": "cfg2 log_prob replayed from a HIP graph", "ms": timed(lambda: g.replay()) * 1e3}))
