"""world_size-2 gloo test of the sample-sharded log-likelihood (nflows_amd/parallel.py) on CPU.
The per-rank log_prob is stood in by the CPU eager port (test infrastructure); what is tested is
the sharding + reduction logic: it must reproduce the single-process result on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nflows_amd import parallel


def test_row_block_partitions():
    for n in (0, 1, 7, 8, 65536, 262144):
        for world in (1, 2, 3, 8):
            blocks = [parallel.row_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.row_block(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _EagerFlow:
    """Adapter: .log_prob through the CPU eager port."""

    def __init__(self, flow):
        self.flow = flow

    def log_prob(self, x):
        from oracle import eager
        return eager.flow_log_prob(self.flow, x)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from nflows_amd import configs
        flow = configs.rq_nsf_flow(num_layers=2, features=8, num_bins=4, hidden_features=16, seed=0).eval()
        parallel.broadcast_model(flow)
        x = torch.randn(101, 8, generator=torch.Generator().manual_seed(1234))
        lo, hi = parallel.row_block(x.shape[0], rank, world)
        total, mean = parallel.sharded_log_likelihood(_EagerFlow(flow), x[lo:hi])
        lp_local = _EagerFlow(flow).log_prob(x[lo:hi]).detach()
        acc = parallel.reduce_log_likelihood(lp_local)
        assert acc[1].item() == 101
        # per-sample values on every rank: the blocks differ by one row (101 = 51 + 50)
        everything = parallel.gather_log_prob(lp_local)
        assert everything.shape == (101,)
        assert torch.equal(everything[lo:hi], lp_local)
        # a replica that started from other weights is overwritten by rank 0's
        other = configs.rq_nsf_flow(num_layers=2, features=8, num_bins=4, hidden_features=16, seed=rank).eval()
        parallel.broadcast_model(other)
        for a, b in zip(other.state_dict().values(), flow.state_dict().values()):
            assert torch.equal(a, b)
        np.save(out_path % rank, np.array([total.item(), mean.item(), acc[0].item(),
                                           everything.double().sum().item()]))
    finally:
        dist.destroy_process_group()


def test_sharded_log_likelihood_world2(tmp_path):
    port = _free_port()
    out = str(tmp_path / "r%d.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    from nflows_amd import configs
    from oracle import eager
    torch.set_num_threads(1)
    flow = configs.rq_nsf_flow(num_layers=2, features=8, num_bins=4, hidden_features=16, seed=0).eval()
    x = torch.randn(101, 8, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        want = eager.flow_log_prob(flow, x).double().sum().item()
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.array_equal(r0, r1)                      # every rank holds the same answer
    assert abs(r0[0] - want) <= 1e-9 * abs(want) + 1e-9  # fp64-accumulated sum of the same fp32 values
    assert abs(r0[1] - want / 101) <= 1e-9
    assert abs(r0[3] - want) <= 1e-9 * abs(want) + 1e-9  # the gathered per-sample values are the whole batch
