#!/usr/bin/env python3
"""Headline benchmark: Flow.log_prob throughput of the 32-layer RQ-NSF coupling flow
(dim=64, K=8, batch 65536 per GPU) on MI355X, with the spline kernel's HBM roofline fraction
and the reference-CPU-path baseline (timed through its bit-identical port, oracle/eager.py).

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N > 1 the driver launches one process per GPU with torch.distributed.run; ranks shard the
samples (weak scaling: 65536 rows per GPU) and exchange one 16-byte all-reduce per step.
A "step" is one full log_prob pass over the rank's batch (inputs already resident in HBM) plus
the log-likelihood reduction.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PFLOP/s headline includes sparsity)


class EventHook:
    """Brackets every K1 launch with HIP events recorded on the stream it is launched on
    (torch's current stream: ops.py passes exactly that handle to the library)."""

    def __init__(self):
        self.pairs = []
        self.bytes = 0
        self.enabled = False

    def begin(self, name):
        if not self.enabled:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, start, nbytes):
        if start is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.pairs.append((start, e))
        self.bytes = nbytes

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        return sum(ms) / len(ms), len(ms)


def dispatch_durations_ms(capacity):
    """Per-launch K1 durations from the HIP events the library attached to each dispatch
    (hipExtLaunchKernelGGL start/stop events on the launch stream): the kernel's own begin/end
    timestamps, i.e. the quantity rocprofv3's kernel trace reports."""
    import ctypes
    from nflows_amd import _native
    buf = (ctypes.c_float * capacity)()
    n = ctypes.c_int32(0)
    _native.check(_native.load().nfa_profile_collect(buf, capacity, ctypes.byref(n)))
    return [buf[i] for i in range(n.value)]


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(flow_cpu, features, sample_rows, budget_s=20.0):
    """The reference's CPU path, via its bit-identical PyTorch-eager port, on the host cores this
    process may use.  Bounded: rows are halved until one pass fits the time budget."""
    from oracle import eager
    threads = usable_cores()
    torch.set_num_threads(threads)
    log("cpu baseline: %d threads (os.cpu_count=%s)" % (threads, os.cpu_count()))
    rows = sample_rows
    with torch.no_grad():
        while True:
            x = torch.randn(rows, features, generator=torch.Generator().manual_seed(1234))
            t0 = time.perf_counter()
            eager.flow_log_prob(flow_cpu, x)  # warm-up pass, also the probe
            probe = time.perf_counter() - t0
            log("cpu baseline: %d rows, probe pass %.2f s" % (rows, probe))
            if probe <= budget_s / 4 or rows <= 512:
                break
            rows //= 4
        reps = max(1, min(3, int(budget_s / max(probe, 1e-3)) - 1))
        t0 = time.perf_counter()
        for _ in range(reps):
            eager.flow_log_prob(flow_cpu, x)
        dt = (time.perf_counter() - t0) / reps
    return {"value": rows / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "same 32-layer flow and weights, %d rows x %d timed passes of "
                      "oracle/eager.py (bit-identical to the reference CPU path), %.2f s/pass"
                      % (rows, reps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=16384)
    ap.add_argument("--no-fuse", action="store_true", help="run permutations as separate kernels")
    ap.add_argument("--path", choices=["k8", "k7b", "k7", "k1"], default="k8",
                    help="layer kernel: k8 = whole ResidualNet conditioner + spline in one kernel "
                         "(default), k7b / k7 = only the final Linear fused (split-bf16 / fp32 MFMA), "
                         "k1 = PyTorch conditioner + spline kernel")
    ap.add_argument("--no-fuse-linear", action="store_true",
                    help="leave the conditioner's final Linear to hipBLASLt (GEMM + K1 instead of K7)")
    ap.add_argument("--skip-k1-roofline", action="store_true")
    ap.add_argument("--skip-graph", action="store_true",
                    help="do not add the HIP-graph replay timing of the same step (extra field)")
    ap.add_argument("--bracket-events", action="store_true",
                    help="additionally bracket every K1 launch with torch events (includes launch gaps)")
    ap.add_argument("--skip-consistency", action="store_true",
                    help="skip the fwd/inv check (profiling runs: only full-batch launches in the trace)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    # NFA_BENCH_BACKEND=gloo: dry run of the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks share devices; numbers are meaningless) -- the real runs use RCCL
    backend = os.environ.get("NFA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import nflows_amd
    from nflows_amd import configs, ops, parallel
    log("imports done; building flow")

    D, K, H = 64, 8, 128
    flow_cpu = configs.rq_nsf_flow(num_layers=args.layers, features=D, num_bins=K, hidden_features=H,
                                   num_blocks=2, tail_bound=3.0, seed=0).eval()  # same seed on every rank
    import copy
    flow = copy.deepcopy(flow_cpu).to(dev)
    flow._transform.fuse_permutations = not args.no_fuse
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ

    def select_path(path):
        RQ.fuse_conditioner = path == "k8"
        RQ.fuse_final_linear = path != "k1"
        RQ.final_linear_engine = "f32" if path == "k7" else "bf16x3"

    if args.no_fuse_linear:
        args.path = "k1"
    select_path(args.path)
    B = args.batch_per_gpu
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)

    log("flow on device; warm-up")
    hook = EventHook()
    ops.set_launch_hook(hook)

    def step():
        with torch.no_grad():
            lp = flow.log_prob(x)
        return parallel.reduce_log_likelihood(lp)

    for _ in range(args.warmup):
        acc = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    log("warm-up done; timing %d steps" % args.steps)
    hook.enabled = args.bracket_events
    from nflows_amd import _native
    max_launches = args.layers * args.steps
    _native.check(_native.load().nfa_profile_enable(max_launches))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    hook.enabled = False
    k1_ms = dispatch_durations_ms(max_launches)
    _native.check(_native.load().nfa_profile_enable(0))
    log("timed region done: %.1f ms/step" % (elapsed / args.steps * 1e3))
    nflows_amd.check_status()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    mean_ll = (acc[0] / acc[1]).item()

    # the same step replayed from a HIP graph (host out of the loop): reported beside the headline
    # number, which keeps per-dispatch events and therefore launches from the host
    graph_ms = None
    if world == 1 and not args.skip_graph:
        try:
            from nflows_amd.graphs import GraphedLogProb
            g = GraphedLogProb(flow, x)
            for _ in range(3):
                parallel.reduce_log_likelihood(g(x))
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(args.steps):
                acc_g = parallel.reduce_log_likelihood(g(x))
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - tg) / args.steps * 1e3
            assert abs((acc_g[0] / acc_g[1]).item() - mean_ll) < 1e-6 * abs(mean_ll)
            del g
        except Exception as e:  # measurement extra only
            log("HIP-graph timing skipped: %r" % (e,))

    # forward∘inverse consistency (second half of the metric), outside the timed region
    err_composite = err_layer = None
    xs = x[:8192]
    with torch.no_grad():
        if not args.skip_consistency:
            z, lad = flow._transform(xs)
            xr, lad_inv = flow._transform.inverse(z)
            err_composite = (xr - xs).abs().max().item()
            layer = flow._transform._transforms[1]
            y1, _ = layer(xs)
            x1, _ = layer.inverse(y1)
            err_layer = (x1 - xs).abs().max().item()

    if rank == 0:
        total_rows = B * world
        H_ = 128
        P_ = 3 * K - 1
        dt_ = D // 2
        nb_ = 2
        k1_bytes = 4 * (B * D + B * dt_ * P_ + B * D + B)  # 3460 B/sample/layer
        io_bytes = 4 * (B * D + B * D + B)                 # inputs + outputs + logabsdet

        def load_traffic(name):
            try:
                return json.load(open(os.path.join(ROOT, "profiles", name))).get("hbm_bytes_per_launch")
            except Exception:
                return None

        timing_note = ("HIP start/stop events attached to each layer-kernel dispatch on its launch "
                       "stream (hipExtLaunchKernelGGL), all launches of the timed region")

        def roofline_of(path, ms, launches_per_step):
            """`launches_per_step` layer-kernel dispatches make one step of `args.layers` layers: 32
            (one layer per launch) or 1 (the run of K8 layers in a single launch)."""
            avg_ms, launches = sum(ms) / len(ms), len(ms)
            layers_per_launch = args.layers / launches_per_step
            common = {"avg_launch_ms": avg_ms, "launches_timed": launches, "layers_per_launch": layers_per_launch,
                      "timing": timing_note}
            if path in ("k8", "k7b"):
                # GEMMs on the bf16 matrix pipe with split-bf16 operands: 6 bf16 products per fp32
                # multiply-add; flops of the unpadded layers
                macs = dt_ * P_ * H_ + ((D - dt_) * H_ + nb_ * 2 * H_ * H_ if path == "k8" else 0)
                flops = 6 * 2.0 * B * macs * layers_per_launch
                ach = flops / (avg_ms * 1e-3) / 1e12
                # HBM: a run of layers reads its rows once and writes them once, and streams every
                # layer's packed weights (bf16 triples in 12 KB stages) once
                k8_weights = layers_per_launch * (2 + 16 * nb_ + 2 * (dt_ * 24 // 32)) * 12288
                bytes_ = io_bytes + k8_weights if path == "k8" else (io_bytes + 4 * B * H_) * layers_per_launch
                r = {"bound": "mfma",
                     "kernel": ("nfa::rqs_resnet_kernel<false, 1, 2, %s, 8>" % os.environ.get("NFA_K8_PIPE", "2")) if path == "k8" else "nfa::rqs_fused_linear_bf16_kernel<false>",
                     "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_PEAK_TFLOPS,
                     "traffic": load_traffic("k8_pmc_traffic.json" if path == "k8" else "k7b_pmc_traffic.json"),
                     "algorithmic_flops_per_launch": flops,
                     "algorithmic_bytes_per_launch": bytes_,
                     "fp32_equivalent_tflops": ach / 6,
                     "note": "achieved = 6 x (fp32 multiply-adds of the layers' GEMMs) x 2 / time: every fp32 "
                             "operand is three bf16 pieces and six cross products run on the bf16 pipe "
                             "(fp32-accurate); peak = dense bf16 MFMA peak.  As fp32 GEMM work this is "
                             "%.1f TFLOP/s (fp32 matrix peak: 157.3)" % (ach / 6)}
            elif path == "k7":
                flops = 2.0 * B * H_ * dt_ * P_
                ach = flops / (avg_ms * 1e-3) / 1e12
                r = {"bound": "mfma", "kernel": "nfa::rqs_fused_linear_kernel<false>", "achieved": ach,
                     "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3,
                     "traffic": load_traffic("k7_pmc_traffic.json"), "algorithmic_flops_per_launch": flops,
                     "algorithmic_bytes_per_launch": io_bytes + 4 * B * H_}
            else:
                ach = k1_bytes / (avg_ms * 1e-3) / 1e9
                r = {"bound": "hbm", "kernel": "nfa::rqs_coupling_pipelined<8, false, true>", "achieved": ach,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": load_traffic("k1_pmc_traffic.json"), "algorithmic_bytes_per_launch": k1_bytes}
            r.update(common)
            return r

        roofline = roofline_of(args.path, k1_ms, len(k1_ms) / args.steps) if k1_ms else None
        roofline_k1 = None
        if args.path != "k1" and not args.skip_k1_roofline and world == 1:  # (N = 1 only: the other ranks are done)
            # the HBM-bound spline kernel K1 (what the fused kernels replace on this shape), measured
            # the same way in a short separate run with the conditioner left to PyTorch / hipBLASLt
            select_path("k1")
            try:
                with torch.no_grad():
                    flow.log_prob(x)
                    torch.cuda.synchronize()
                    _native.check(_native.load().nfa_profile_enable(args.layers * 3))
                    for _ in range(3):
                        flow.log_prob(x)
                    torch.cuda.synchronize()
                    ms = dispatch_durations_ms(args.layers * 3)
                    _native.check(_native.load().nfa_profile_enable(0))
            finally:
                select_path(args.path)
            if ms:
                roofline_k1 = roofline_of("k1", ms, args.layers)
                roofline_k1["note"] = "not in the timed region: PyTorch conditioner + K1 path (--path k1)"
        result = {
            "metric": "log_prob samples/sec (dim=64, K=8, 32-layer RQ-NSF) + max |fwd∘inv − x|",
            "value": total_rows * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic standard-Gaussian inputs, random-init weights (seed 0)",
            "config": {"workload": "%d-layer RQ-NSF coupling flow (RandomPermutation + RQ coupling, "
                                   "ResidualNet H=128 x2 blocks), dim=64, K=8, tail_bound=3, "
                                   "batch=%d per GPU, Flow.log_prob + scalar all-reduce" % (args.layers, B),
                       "global_batch": total_rows, "features": D, "num_bins": K, "layers": args.layers,
                       "parallelism": "sample-sharded x%d" % world,
                       "fused_permutations": not args.no_fuse,
                       "layer_kernel": {"k8": "K8: ResidualNet conditioner + spline layer in one kernel, the run of "
                                              "layers in one launch",
                                        "k7b": "K7b: final Linear (split-bf16 MFMA) + spline layer",
                                        "k7": "K7: final Linear (fp32 MFMA) + spline layer",
                                        "k1": "PyTorch conditioner + K1 spline layer"}[args.path]},
            "fwd_inv_max_err": {"composite_%d_layers" % args.layers: err_composite, "single_layer": err_layer,
                                "rows": 8192},
            "mean_log_likelihood": mean_ll,
            "roofline": roofline,
            "roofline_k1_unfused": roofline_k1,
        }
        if graph_ms is not None:
            result["hip_graph_replay"] = {"ms_per_step": graph_ms, "value": total_rows / (graph_ms * 1e-3),
                                          "note": "same step (incl. copying the batch into the graph's input "
                                                  "buffer) replayed from one captured HIP graph; not the headline"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(flow_cpu, D, args.cpu_rows)
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
