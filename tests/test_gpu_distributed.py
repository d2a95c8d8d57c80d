"""The multi-GPU path on the one GPU a test box has: RCCL initialised once (world size 1), and
bench.py's multi-rank control flow as a two-rank dry run over gloo (both ranks on cuda:0).  The
sharding / reduction arithmetic itself is covered on CPU (tests/test_distributed_cpu.py, world 2)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_world_size_one():
    """backend "nccl" (= RCCL on ROCm) comes up, all-reduces and gathers on the device."""
    code = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from nflows_amd import configs, parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
flow = configs.rq_nsf_flow(num_layers=2, features=64, num_bins=8, hidden_features=128, seed=0).cuda().eval()
parallel.broadcast_model(flow)
x = torch.randn(640, 64, device="cuda")
total, mean = parallel.sharded_log_likelihood(flow, x)
with torch.no_grad():
    lp = flow.log_prob(x)
t = torch.ones(2, dtype=torch.float64, device="cuda")
dist.all_reduce(t)                       # a collective really runs through RCCL
allp = parallel.gather_log_prob(lp)
assert allp.shape == (640,) and torch.equal(allp, lp)
assert abs(total.item() - lp.double().sum().item()) <= 1e-9 * abs(total.item())
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), PYTHONPATH=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bench_two_rank_dry_run():
    """bench.py --gpus 2 through torch.distributed.run with the gloo stand-in backend: BASELINE
    configs[3] as named (262 144 global rows, 131 072 per rank), one JSON line from rank 0."""
    env = dict(os.environ, NFA_BENCH_BACKEND="gloo", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--steady-seconds", "0", "--skip-graph"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 262144 and r["scaling"] == "strong"
    assert r["value"] > 0 and r["rows_65536_per_gpu_extra"]["rows_per_gpu"] == 65536
    assert abs(r["mean_log_likelihood"]) < 1e3
