import sys, time, torch
sys.path.insert(0, '/root/repo')
import nflows_amd
from nflows_amd import configs
from nflows_amd.graphs import GraphedLogProb
dev='cuda:0'
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
for name, flow, B, D in [("affine x8 D=32 B=16384", configs.affine_coupling_flow(8,32,(128,128)), 16384, 32),
                         ("RQ-NSF x32 B=65536", configs.rq_nsf_flow(32,64,8,128), 65536, 64),
                         ("RQ-NSF x32 B=32768", configs.rq_nsf_flow(32,64,8,128), 32768, 64),
                         ("RQ-NSF x32 B=16384", configs.rq_nsf_flow(32,64,8,128), 16384, 64),
                         ("RQ-NSF x32 B=8192", configs.rq_nsf_flow(32,64,8,128), 8192, 64),
                         ("RQ-NSF x32 B=4096", configs.rq_nsf_flow(32,64,8,128), 4096, 64)]:
    flow=flow.to(dev).eval(); x=torch.randn(B,D,device=dev)
    with torch.no_grad():
        eager=timed(lambda: flow.log_prob(x), 30)
        ref=flow.log_prob(x).clone()
    g=GraphedLogProb(flow, x)
    x2=torch.randn(B,D,device=dev)
    out=g(x2).clone()
    with torch.no_grad(): want=flow.log_prob(x2)
    gt=timed(lambda: g(x), 30)
    print("%-28s eager %.3f ms   graph %.3f ms   equal: %s" % (name, eager, gt, torch.equal(out, want)), flush=True)
nflows_amd.check_status()
