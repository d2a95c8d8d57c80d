"""Layer primitives of the conditioner networks."""
import os

import torch
from torch.nn import functional as F

from .. import autograd as AG

# below this many rows the library's weight-gradient GEMM is no slower than K10's two launches
_WGRAD_MIN_ROWS = int(os.environ.get("NFA_WGRAD_MIN_ROWS", "2048"))


def linear(inputs, weight, bias=None):
    """`F.linear`; on a HIP device under autograd the weight / bias gradients of large batches are
    computed by K10 (nflows_amd.autograd.Linear) instead of the library's result-tiled GEMM."""
    if (inputs.is_cuda and inputs.dim() == 2 and inputs.dtype == torch.float32 and torch.is_grad_enabled()
            and inputs.shape[0] >= _WGRAD_MIN_ROWS and weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0
            and (weight.requires_grad or (bias is not None and bias.requires_grad))):
        return AG.Linear.apply(inputs, weight, bias)
    return F.linear(inputs, weight, bias)


def apply_layer(layer, inputs):
    """`layer(inputs)`, routed through `linear` for plain `nn.Linear` layers."""
    if type(layer) is torch.nn.Linear:
        return linear(inputs, layer.weight, layer.bias)
    return layer(inputs)
