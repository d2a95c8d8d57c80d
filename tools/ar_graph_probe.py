import sys, time, torch
sys.path.insert(0, '/root/repo')
import nflows_amd
from nflows_amd import configs
from nflows_amd.graphs import GraphedInverse
dev='cuda:0'
with torch.no_grad():
    flow = configs.ar_rq_flow(784, 256, 8, 3.0, 2).to(dev).eval()
    z = torch.randn(4096, 784, device=dev)
    t = flow._transform._transforms[0]
    x0,_ = t.inverse(z)
    torch.cuda.synchronize()
    t0=time.perf_counter(); x0,_=t.inverse(z); torch.cuda.synchronize(); print('eager', (time.perf_counter()-t0)*1e3,'ms')
    t0=time.perf_counter(); g = GraphedInverse(flow, z, warmup=1); torch.cuda.synchronize(); print('capture', time.perf_counter()-t0,'s')
    out = g(z); torch.cuda.synchronize()
    t0=time.perf_counter(); out = g(z); torch.cuda.synchronize(); print('replay', (time.perf_counter()-t0)*1e3,'ms')
    print('equal', torch.equal(out[0], x0))
