"""HIP-graph replay of a whole `Flow.log_prob` / `Flow.sample` pass.

The per-layer Python loop of `CompositeTransform` (the reference's `_cascade`,
transforms/base.py:45-52) enqueues ~10 kernels per layer; for small batches the pass is bound by
launch latency, not by the GPU.  Capturing the pass once into a HIP graph (through PyTorch's
`torch.cuda.CUDAGraph`, which on ROCm is hipGraph) and replaying it removes the host from the
loop.  The kernels from libnflows_amd.so are captured like any other launch because they are
enqueued on PyTorch's current stream; no host synchronisation happens inside a linear-tail flow
(data-dependent errors are recorded in the device status word, see `nflows_amd.check_status`).

Usage:
    g = GraphedLogProb(flow, example_inputs)      # captures once
    lp = g(inputs)                                # same shape/dtype/device as example_inputs
"""
import torch


class GraphedLogProb:
    """Replays `flow.log_prob(inputs)` (no grad) for a fixed input shape."""

    def __init__(self, flow, example_inputs, warmup=3):
        if not example_inputs.is_cuda:
            raise NotImplementedError("nflows_amd: HIP graphs need inputs on a HIP device")
        self.flow = flow
        self._static_in = example_inputs.detach().clone()
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=example_inputs.device)
        side.wait_stream(torch.cuda.current_stream(example_inputs.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):  # allocator pools, status word, index caches, BLAS workspaces
                flow.log_prob(self._static_in)
        torch.cuda.current_stream(example_inputs.device).wait_stream(side)
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._static_out = flow.log_prob(self._static_in)

    def __call__(self, inputs):
        if inputs.shape != self._static_in.shape or inputs.dtype != self._static_in.dtype:
            raise ValueError("GraphedLogProb was captured for %s %s" % (tuple(self._static_in.shape), self._static_in.dtype))
        self._static_in.copy_(inputs)
        self._graph.replay()
        return self._static_out


class GraphedInverse:
    """Replays `flow._transform.inverse(noise)` (the sampling path) for a fixed noise shape."""

    def __init__(self, flow, example_noise, warmup=3):
        if not example_noise.is_cuda:
            raise NotImplementedError("nflows_amd: HIP graphs need inputs on a HIP device")
        self.flow = flow
        self._static_in = example_noise.detach().clone()
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=example_noise.device)
        side.wait_stream(torch.cuda.current_stream(example_noise.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):
                flow._transform.inverse(self._static_in)
        torch.cuda.current_stream(example_noise.device).wait_stream(side)
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._static_out = flow._transform.inverse(self._static_in)

    def __call__(self, noise):
        self._static_in.copy_(noise)
        self._graph.replay()
        return self._static_out
