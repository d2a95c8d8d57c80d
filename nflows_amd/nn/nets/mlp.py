"""Multi-layer perceptron conditioner (reference: nflows/nn/nets/mlp.py), PyTorch-ROCm GEMMs.

Same parameter names (`_input_layer`, `_hidden_layers.{i}`, `_output_layer`).  The reference
MLP's forward takes no `context`, so it cannot be handed to a coupling layer directly
(SURVEY a12); this one accepts and ignores an optional `context`, which makes it usable as
`transform_net_create_fn=lambda i, o: MLP([i], [o], hidden)` without a wrapper.
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


class MLP(nn.Module):
    def __init__(self, in_shape, out_shape, hidden_sizes, activation=F.relu, activate_output=False):
        super().__init__()
        self._in_shape = torch.Size(in_shape)
        self._out_shape = torch.Size(out_shape)
        self._hidden_sizes = hidden_sizes
        self._activation = activation
        self._activate_output = activate_output
        if len(hidden_sizes) == 0:
            raise ValueError("List of hidden sizes can't be empty.")
        self._input_layer = nn.Linear(int(np.prod(in_shape)), hidden_sizes[0])
        self._hidden_layers = nn.ModuleList(
            nn.Linear(a, b) for a, b in zip(hidden_sizes[:-1], hidden_sizes[1:]))
        self._output_layer = nn.Linear(hidden_sizes[-1], int(np.prod(out_shape)))

    def forward(self, inputs, context=None):
        if inputs.shape[1:] != self._in_shape:
            raise ValueError("Expected inputs of shape {}, got {}.".format(self._in_shape, inputs.shape[1:]))
        flat = inputs.reshape(-1, int(np.prod(self._in_shape)))
        if self._activation is F.relu and flat.is_cuda and not torch.is_grad_enabled():
            # inference on the GPU: bias + ReLU in the GEMM epilogue (hipBLASLt)
            def act_linear(layer, v):
                return torch._addmm_activation(layer.bias, v, layer.weight.t())
        else:
            def act_linear(layer, v):
                return self._activation(layer(v))
        h = act_linear(self._input_layer, flat)
        for layer in self._hidden_layers:
            h = act_linear(layer, h)
        h = self._output_layer(h)
        if self._activate_output:
            h = self._activation(h)
        return h.reshape(-1, *self._out_shape)
