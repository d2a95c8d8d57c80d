"""Multi-layer perceptron conditioner; PyTorch-ROCm GEMMs (MFMA through hipBLASLt).

Constructor and parameter names of nflows/nn/nets/mlp.py (`_input_layer`, `_hidden_layers.{i}`,
`_output_layer`).  Unlike the reference's, `forward` accepts (and ignores) an optional `context`,
so the module can be handed to a coupling layer directly:
`transform_net_create_fn=lambda i, o: MLP([i], [o], [128, 128])`.
On the GPU without grad, bias + ReLU run in the GEMM epilogue.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ..functional import apply_layer


class MLP(nn.Module):
    def __init__(self, in_shape, out_shape, hidden_sizes, activation=F.relu, activate_output=False):
        super().__init__()
        if len(hidden_sizes) == 0:
            raise ValueError("List of hidden sizes can't be empty.")
        self._in_shape, self._out_shape = torch.Size(in_shape), torch.Size(out_shape)
        self._hidden_sizes = hidden_sizes
        self._activation = activation
        self._activate_output = activate_output
        widths = [math.prod(in_shape)] + list(hidden_sizes) + [math.prod(out_shape)]
        self._input_layer = nn.Linear(widths[0], widths[1])
        self._hidden_layers = nn.ModuleList(
            nn.Linear(widths[i], widths[i + 1]) for i in range(1, len(widths) - 2))
        self._output_layer = nn.Linear(widths[-2], widths[-1])

    def _activated(self, layer, v, fused):
        if fused:
            return torch._addmm_activation(layer.bias, v, layer.weight.t())
        return self._activation(apply_layer(layer, v))

    def forward(self, inputs, context=None):
        if inputs.shape[1:] != self._in_shape:
            raise ValueError("Expected inputs of shape {}, got {}.".format(self._in_shape, inputs.shape[1:]))
        h = inputs.reshape(inputs.shape[0], -1)
        fused = self._activation is F.relu and h.is_cuda and not torch.is_grad_enabled()
        for layer in [self._input_layer, *self._hidden_layers]:
            h = self._activated(layer, h, fused)
        h = apply_layer(self._output_layer, h)
        if self._activate_output:
            h = self._activation(h)
        return h.reshape(-1, *self._out_shape)
