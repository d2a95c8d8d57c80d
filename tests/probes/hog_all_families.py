"""Round 5: every kernel family beside a busy second stream (the condition that exposed the K8h fragment-read defect).

For each family: the same call three times on a quiet device (is it bit-deterministic at all?), then `reps` times while a
side stream sweeps 2 GB; every output tensor compared bit for bit with the quiet result.  Prints one line per family.
    python tests/probes/hog_all_families.py [reps]"""
import copy, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import steep_flow, trained_flow
from test_gpu_steep import _batch, _nsf_engines
from nflows_amd import ops, configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
DEV = "cuda:0"
golden = os.path.join(ROOT, "tests", "golden")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hog = torch.zeros(1 << 29, device=DEV)
side = torch.cuda.Stream()


def flat(out):
    if isinstance(out, torch.Tensor):
        return [out]
    res = []
    for o in out:
        res += flat(o)
    return res


def same(a, b):
    return all(torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) for x, y in zip(a, b))


def run(name, fn, note=lambda: ""):
    try:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        quiet = [t.clone() for t in flat(fn())]
        det = all(same(quiet, flat(fn())) for _ in range(3))
        bad, worst = 0, 0.0
        for _ in range(reps):
            with torch.cuda.stream(side):
                for _ in range(4):
                    hog.add_(1.0)
            for _ in range(4):
                got = flat(fn())
                if not same(quiet, got):
                    bad += 1
                    worst = max(worst, max(float((torch.nan_to_num(a.double()) - torch.nan_to_num(b.double())).abs().max()) for a, b in zip(quiet, got)))
            side.synchronize()
        print("%-46s quiet-deterministic %-5s beside the hog: %3d of %3d calls deviate (max |diff| %.2e)  %s"
              % (name, det, bad, reps * 4, worst, note()), flush=True)
    except Exception as e:   # noqa: BLE001
        print("%-46s ERROR %s: %s" % (name, type(e).__name__, str(e)[:160]), flush=True)


saved = (RQ.fuse_conditioner, RQ.fuse_final_linear, RQ.final_linear_engine, RQ.conditioner_engine, ops.K8S_ENABLED)
with torch.no_grad():
    # 1. the coupling layer on every engine
    flow_cpu, g, cfg = steep_flow(golden, "steep_nsf_k8")
    x = _batch(g, "steep_nsf_k8", "x", 65536, cfg["D"]).to(DEV)
    noise = _batch(g, "steep_nsf_k8", "noise", 65536, cfg["D"]).to(DEV)
    for engine, (sw, rows, k8s, expect) in _nsf_engines(8).items():
        flow = copy.deepcopy(flow_cpu).to(DEV).eval()
        RQ.fuse_conditioner = sw["path"] == "k8"
        RQ.fuse_final_linear = sw["path"] != "none"
        RQ.final_linear_engine = "f32" if sw["path"] == "k7" else "bf16x3"
        RQ.conditioner_engine = sw["engine"]
        ops.K8S_ENABLED = k8s
        for k, v in sw.get("env", {}).items():
            os.environ[k] = v
        run("steep_nsf_k8 / %s fwd" % engine, lambda: flow._transform(x[:rows]), ops.last_layer_kernel)
        run("steep_nsf_k8 / %s inv" % engine, lambda: flow._transform.inverse(noise[:rows]), ops.last_layer_kernel)
        for k in sw.get("env", {}):
            os.environ.pop(k, None)
    RQ.fuse_conditioner, RQ.fuse_final_linear, RQ.final_linear_engine, RQ.conditioner_engine, ops.K8S_ENABLED = saved
    # 2. affine couplings, 3. the autoregressive spline flow
    for case, rows in (("steep_affine", 65536), ("steep_ar_rq", 4096)):
        flow_cpu, g, cfg = steep_flow(golden, case)
        xx = _batch(g, case, "x", rows, cfg["D"]).to(DEV)
        nn_ = _batch(g, case, "noise", rows, cfg["D"]).to(DEV)
        flow = copy.deepcopy(flow_cpu).to(DEV).eval()
        run("%s fwd" % case, lambda: flow._transform(xx), ops.last_layer_kernel)
        run("%s inv" % case, lambda: flow._transform.inverse(nn_), ops.last_layer_kernel)
    # 4. log_prob / sample of the bench flow at a small size, 5. elementwise float64
    flow = configs.rq_nsf_flow(8, 64, 8, 128, 2, 3.0, seed=0).to(DEV).eval()
    xb = torch.randn(65536, 64, device=DEV)
    run("bench flow (8 layers) log_prob 65536", lambda: flow.log_prob(xb), ops.last_layer_kernel)
    flow64 = copy.deepcopy(flow).double()
    xb64 = xb[:8192].double()
    run("bench flow float64 log_prob 8192", lambda: flow64.log_prob(xb64))

# 6. a training step's gradients (forward + backward of the fused training kernels)
flow = configs.rq_nsf_flow(8, 64, 8, 128, 2, 3.0, seed=0).to(DEV).train()
xt = torch.randn(65536, 64, device=DEV)


def grads():
    for p in flow.parameters():
        p.grad = None
    loss = -flow.log_prob(xt).mean()
    loss.backward()
    return [loss.detach()] + [p.grad for p in flow.parameters() if p.grad is not None]


run("training step (8 layers, 65536 rows): loss + grads", grads)
flow_cpu, g, cfg = steep_flow(golden, "steep_affine")
flow = copy.deepcopy(flow_cpu).to(DEV).train()
xa = _batch(g, "steep_affine", "x", 65536, cfg["D"]).to(DEV)


def grads_affine():
    for p in flow.parameters():
        p.grad = None
    loss = -flow.log_prob(xa).mean()
    loss.backward()
    return [loss.detach()] + [p.grad for p in flow.parameters() if p.grad is not None]


run("affine flow training step: loss + grads", grads_affine)
