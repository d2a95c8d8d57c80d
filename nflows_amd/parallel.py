"""Sample-sharded density evaluation across the GPUs of one node.

Every sample's forward / inverse / logabsdet is independent, so the path shards over rows with
no data-path collective: each rank (one process per GPU) evaluates `log_prob` on its own
contiguous block of rows with a replicated model.  The only exchange is ONE all-reduce(SUM) of a
2-element float64 vector [sum_i log p(x_i), count] per evaluation -- 16 bytes over RCCL/xGMI,
latency-bound.  Per-sample values, when a caller wants them, are an all_gather of [rows] fp32.
The reference has no distributed code; the single-process result on the concatenated batch is
the oracle for this module (tests/test_distributed_cpu.py, gloo, world_size 2).
"""
import torch
import torch.distributed as dist


def row_block(num_rows, rank, world_size):
    """Contiguous, balanced row range [start, stop) owned by `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(num_rows, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def reduce_log_likelihood(local_log_prob, group=None):
    """[rows_local] per-sample log-densities -> (global sum, global count) as a float64 tensor of
    shape [2] on the same device; one all-reduce when a process group is initialised."""
    if local_log_prob.is_cuda and local_log_prob.dtype == torch.float32:
        from . import ops
        acc = ops.sum_count(local_log_prob)     # one launch
    else:
        acc = torch.full((2,), float(local_log_prob.numel()), dtype=torch.float64, device=local_log_prob.device)
        torch.sum(local_log_prob.reshape(-1), dim=0, keepdim=True, dtype=torch.float64, out=acc[0:1])
    if _world(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc


def sharded_log_likelihood(flow, local_inputs, context=None, group=None):
    """Total and mean log-likelihood of the global batch whose rows are spread over the ranks.
    Returns (total, mean) as 0-dim float64 tensors (identical on every rank)."""
    with torch.no_grad():
        lp = flow.log_prob(local_inputs, context) if context is not None else flow.log_prob(local_inputs)
    acc = reduce_log_likelihood(lp, group)
    return acc[0], acc[0] / acc[1]


def gather_log_prob(local_log_prob, group=None):
    """Per-sample log-densities of the whole batch on every rank, in rank order.  Row blocks may
    differ in size (`row_block` hands out blocks that differ by one row when the batch does not
    divide evenly): the counts are exchanged first and the blocks padded to the largest, so every
    rank posts receive buffers of the same size (a mismatch hangs or fails in RCCL)."""
    world = _world(group)
    if world == 1:
        return local_log_prob
    local = local_log_prob.contiguous().reshape(-1)
    count = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    counts = [torch.empty_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    padded = local if local.numel() == longest else torch.cat((local, local.new_zeros(longest - local.numel())))
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, counts)], dim=0)


def broadcast_model(module, src=0, group=None):
    """Replicates parameters and buffers from `src` (once, at start-up; ~17 MB for the 32-layer
    flow).  Building every replica from the same seed makes this unnecessary.  The tensors are
    written in place under no_grad (their version counters advance) and the layers' packed-weight
    caches are dropped explicitly, so the fused kernels never see weights from before the
    broadcast."""
    if _world(group) == 1:
        return module
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)
    from . import invalidate_packed_weights
    invalidate_packed_weights()
    return module
