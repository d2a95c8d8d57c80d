"""Standard-normal base distribution (reference: nflows/distributions/normal.py:11-50).

`_log_prob` on a HIP float32 tensor is one fused kernel (square, per-sample sum, -0.5*, -log_z).
`_log_z` is a non-persistent float64 0-dim buffer exactly like the reference, so state_dicts match.
"""
import numpy as np
import torch

from .. import ops
from ..utils import torchutils
from .base import Distribution


class StandardNormal(Distribution):
    """Zero-mean, unit-covariance Gaussian over tensors of shape `shape`."""

    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)
        self.register_buffer(
            "_log_z", torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64),
            persistent=False)

    def _log_prob(self, inputs, context):
        if inputs.shape[1:] != self._shape:
            raise ValueError("Expected input of shape {}, got {}".format(self._shape, inputs.shape[1:]))
        return ops.standard_normal_log_prob(inputs)

    def _sample(self, num_samples, context):
        if context is None:
            return torch.randn(num_samples, *self._shape, device=self._log_z.device)
        rows = context.shape[0]
        draws = torch.randn(rows * num_samples, *self._shape, device=context.device)
        return torchutils.split_leading_dim(draws, [rows, num_samples])

    def _mean(self, context):
        if context is None:
            return self._log_z.new_zeros(self._shape)
        return context.new_zeros(context.shape[0], *self._shape)
