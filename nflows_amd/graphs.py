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
    """Replays `flow.log_prob(inputs)` (no grad) for a fixed input shape.

    The captured kernels read the weights of capture time: parameters updated in place are seen
    by the library GEMMs, but the whole-layer kernels read packed copies made before the capture.
    Re-capture (build a new GraphedLogProb) after changing weights."""

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


class GraphedTrainStep:
    """Replays one whole optimisation step -- `loss = -flow.log_prob(x).mean()`, `loss.backward()`,
    `optimizer.step()`: the reference's training loop, examples/moons.ipynb cell 3 -- from ONE HIP
    graph.  A 32-layer flow enqueues ~1 500 kernels per step (conditioner GEMMs and their
    gradients, the layer kernels and their backward kernels, the optimizer); eagerly the step is
    bound by the host, replayed it is bound by the GPU.

    The optimizer must be capture-safe: its state may not live on the host
    (`torch.optim.Adam(..., capturable=True)`; plain SGD qualifies as is).

    Usage:
        opt = torch.optim.Adam(flow.parameters(), lr=1e-4, capturable=True)
        step = GraphedTrainStep(flow, opt, example_inputs)     # warms up on a side stream, captures
        for x in batches:                                       # same shape as example_inputs
            loss = step(x)                                      # 0-dim tensor, valid until the next call

    Drop every reference to losses / outputs of earlier EAGER steps first: a live autograd graph
    keeps its AccumulateGrad nodes bound to the stream they were created on, and the engine would
    then synchronise the capturing stream with that one in the middle of the capture.

    The warm-up runs `warmup` REAL optimisation steps on `example_inputs` (PyTorch's whole-network
    capture recipe needs the optimizer state and the gradient buffers to exist before capture).
    `loss_fn(flow, inputs) -> 0-dim tensor` replaces the default negative mean log-likelihood.
    """

    def __init__(self, flow, optimizer, example_inputs, loss_fn=None, warmup=3):
        if not example_inputs.is_cuda:
            raise NotImplementedError("nflows_amd: HIP graphs need inputs on a HIP device")
        if optimizer.defaults.get("capturable") is False:
            raise ValueError("GraphedTrainStep needs a capture-safe optimizer: construct it with capturable=True")
        self.flow = flow
        self.optimizer = optimizer
        self._loss_fn = loss_fn if loss_fn is not None else (lambda f, x: -f.log_prob(x).mean())
        self._static_in = example_inputs.detach().clone()
        device = example_inputs.device
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream(device).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)  # gradients are (re)allocated inside the graph's pool
        with torch.cuda.graph(self._graph):
            self._static_loss = self._loss_fn(flow, self._static_in)
            self._static_loss.backward()
            optimizer.step()

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self._loss_fn(self.flow, self._static_in)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, inputs):
        if inputs.shape != self._static_in.shape or inputs.dtype != self._static_in.dtype:
            raise ValueError("GraphedTrainStep was captured for %s %s" % (tuple(self._static_in.shape), self._static_in.dtype))
        self._static_in.copy_(inputs)
        self._graph.replay()
        return self._static_loss
