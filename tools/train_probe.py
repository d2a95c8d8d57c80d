#!/usr/bin/env python3
"""Training-step timing (forward + backward + Adam) of the 32-layer RQ-NSF flow, launched eagerly
and replayed from one HIP graph (nflows_amd.graphs.GraphedTrainStep); informational.
    python tools/train_probe.py [batch ...]      ->  one JSON line per batch size"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nflows_amd
from nflows_amd import configs
from nflows_amd.graphs import GraphedTrainStep
dev = "cuda:0"
FUSED = os.environ.get("NFA_TRAIN_FUSED_ADAM", "1") == "1"


def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for B in [int(a) for a in sys.argv[1:]] or [16384, 65536]:
    torch.manual_seed(0)
    flow = configs.rq_nsf_flow(32, 64, 8, 128).to(dev).train()
    opt = torch.optim.Adam(flow.parameters(), lr=1e-4, capturable=True, fused=FUSED)
    x = torch.randn(B, 64, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    eager_ms, l = wall(step, 10)
    l = l.item()  # keep no autograd graph of an eager step alive: its AccumulateGrad nodes belong to the default stream
    graphed = GraphedTrainStep(flow, opt, x, warmup=1)
    graph_ms, lg = wall(lambda: graphed(x), 10)
    nflows_amd.check_status()
    print(json.dumps({"config": "training step 32-layer RQ-NSF (tools/train_probe.py): forward + backward + Adam",
                      "batch": B, "adam": "fused" if FUSED else "foreach", "eager_ms": round(eager_ms, 2), "graph_replay_ms": round(graph_ms, 2),
                      "samples_per_s_eager": round(B / eager_ms * 1e3), "samples_per_s_graph": round(B / graph_ms * 1e3),
                      "loss_eager": round(l, 4), "loss_graph": round(lg.item(), 4)}))
