"""Batch-shared spline CDF transforms (reference: nflows/transforms/nonlinearities.py:230-319,
:386-467), the members of that module on the hot path: the spline coupling layers apply them to
their identity half when `apply_unconditional_transform=True` (coupling.py:318-330, :524-535).

Parameters have shape [*shape, K] and are shared by every sample.  Without grad the K6 kernel
builds each feature's knots once per workgroup in LDS; with grad the logits are broadcast over
the batch and the differentiable elementwise functional is used (autograd then reduces the
gradient over the batch, exactly like the reference's `_share_across_batch`).
"""
import numpy as np
import torch
from torch import nn

from .. import autograd as AG
from .. import ops
from ..utils import torchutils
from .base import Transform
from .splines import rational_quadratic
from . import splines


class PiecewiseRationalQuadraticCDF(Transform):
    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0, identity_init=False,
                 min_bin_width=rational_quadratic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=rational_quadratic.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=rational_quadratic.DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tail_bound = tail_bound
        self.tails = tails
        if isinstance(shape, int):
            shape = (shape,)
        num_derivatives = (num_bins - 1) if self.tails == "linear" else (num_bins + 1)
        if identity_init:
            self.unnormalized_widths = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.zeros(*shape, num_bins))
            constant = np.log(np.exp(1 - min_derivative) - 1)
            self.unnormalized_derivatives = nn.Parameter(constant * torch.ones(*shape, num_derivatives))
        else:  # same RNG consumption order as the reference
            self.unnormalized_widths = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.rand(*shape, num_derivatives))

    def _spline(self, inputs, inverse=False):
        if self.tails is not None and self.tails != "linear":
            raise RuntimeError("{} tails are not implemented.".format(self.tails))
        uw, uh, ud = self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives
        if inputs.shape[1:] != uw.shape[:-1]:
            raise ValueError("Expected inputs of shape [batch, {}], got {}".format(
                tuple(uw.shape[:-1]), tuple(inputs.shape)))
        if AG.needs_grad(inputs, uw, uh, ud):
            batch = inputs.shape[0]

            def share(p):
                return p[None, ...].expand(batch, *p.shape)
            if self.tails is None:
                y, lad = rational_quadratic.rational_quadratic_spline(
                    inputs, share(uw), share(uh), share(ud), inverse=inverse,
                    min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                    min_derivative=self.min_derivative)
            else:
                y, lad = rational_quadratic.unconstrained_rational_quadratic_spline(
                    inputs, share(uw), share(uh), share(ud), inverse=inverse, tails=self.tails,
                    tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                    min_bin_height=self.min_bin_height, min_derivative=self.min_derivative)
            return y, torchutils.sum_except_batch(lad)
        spec = ops.make_rqs_spec(uw.shape[-1], self.tails, tail_bound=self.tail_bound,
                                 min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                 min_derivative=self.min_derivative)
        return ops.rqs_shared(inputs, uw, uh, ud, spec, inverse)

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)


def _share_across_batch(params, batch_size):
    return params[None, ...].expand(batch_size, *params.shape)


class PiecewiseLinearCDF(Transform):
    """Piecewise-linear CDF with one parameter set [*shape, K] shared by every sample
    (nonlinearities.py:230-263)."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0):
        super().__init__()
        self.tail_bound = tail_bound
        self.tails = tails
        self.unnormalized_pdf = nn.Parameter(torch.randn(*shape, num_bins))

    def _spline(self, inputs, inverse=False):
        pdf = _share_across_batch(self.unnormalized_pdf, inputs.shape[0])
        if self.tails is None:
            outputs, logabsdet = splines.linear_spline(inputs, pdf, inverse=inverse)
        else:
            outputs, logabsdet = splines.unconstrained_linear_spline(inputs, pdf, inverse=inverse, tails=self.tails,
                                                                     tail_bound=self.tail_bound)
        return outputs, torchutils.sum_except_batch(logabsdet)

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)


class PiecewiseQuadraticCDF(Transform):
    """Piecewise-quadratic CDF, parameters shared by every sample (nonlinearities.py:266-319):
    K width logits and K+1 (tails=None) or K-1 (linear tails) height logits per element."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0,
                 min_bin_width=splines.quadratic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=splines.quadratic.DEFAULT_MIN_BIN_HEIGHT):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tail_bound = tail_bound
        self.tails = tails
        self.unnormalized_widths = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnormalized_heights = nn.Parameter(torch.randn(*shape, num_bins + 1 if tails is None else num_bins - 1))

    def _spline(self, inputs, inverse=False):
        uw = _share_across_batch(self.unnormalized_widths, inputs.shape[0])
        uh = _share_across_batch(self.unnormalized_heights, inputs.shape[0])
        if self.tails is None:
            outputs, logabsdet = splines.quadratic_spline(inputs, uw, uh, inverse=inverse,
                                                          min_bin_width=self.min_bin_width,
                                                          min_bin_height=self.min_bin_height)
        else:
            outputs, logabsdet = splines.unconstrained_quadratic_spline(
                inputs, uw, uh, inverse=inverse, tails=self.tails, tail_bound=self.tail_bound,
                min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height)
        return outputs, torchutils.sum_except_batch(logabsdet)

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)


class PiecewiseCubicCDF(Transform):
    """Piecewise-cubic CDF, parameters shared by every sample (nonlinearities.py:322-383)."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0,
                 min_bin_width=splines.cubic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=splines.cubic.DEFAULT_MIN_BIN_HEIGHT):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tail_bound = tail_bound
        self.tails = tails
        self.unnormalized_widths = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnormalized_heights = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnorm_derivatives_left = nn.Parameter(torch.randn(*shape, 1))
        self.unnorm_derivatives_right = nn.Parameter(torch.randn(*shape, 1))

    def _spline(self, inputs, inverse=False):
        batch = inputs.shape[0]
        params = [_share_across_batch(p, batch) for p in (self.unnormalized_widths, self.unnormalized_heights,
                                                          self.unnorm_derivatives_left,
                                                          self.unnorm_derivatives_right)]
        if self.tails is None:
            outputs, logabsdet = splines.cubic_spline(inputs, *params, inverse=inverse,
                                                      min_bin_width=self.min_bin_width,
                                                      min_bin_height=self.min_bin_height)
        else:
            outputs, logabsdet = splines.unconstrained_cubic_spline(
                inputs, *params, inverse=inverse, tails=self.tails, tail_bound=self.tail_bound,
                min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height)
        return outputs, torchutils.sum_except_batch(logabsdet)

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)
