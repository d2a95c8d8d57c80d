"""Shift / scale transforms the reference's base tests are written with.

Mirrors nflows/transforms/standard.py: `IdentityTransform` (:12-21), `PointwiseAffineTransform`
(:24-66) and its deprecated aliases `AffineTransform` / `AffineScalarTransform` (:69-88).  These are
not on the hot path (one multiply-add per element, a constant log-determinant): they are plain
tensor expressions on whatever device the inputs live on, here so that flows and tests written
against the reference's API run unchanged.
"""
import warnings

import torch

from .base import Transform


class IdentityTransform(Transform):
    """outputs = inputs, logabsdet = 0 (standard.py:12-21)."""

    def forward(self, inputs, context=None):
        return inputs, inputs.new_zeros(inputs.shape[0])

    def inverse(self, inputs, context=None):
        return self.forward(inputs, context)


class PointwiseAffineTransform(Transform):
    """outputs = inputs * scale + shift, element by element (standard.py:24-66).  `shift` and `scale`
    are numbers or tensors that broadcast against one sample; a zero scale is rejected."""

    def __init__(self, shift=0.0, scale=1.0):
        super().__init__()
        shift = torch.as_tensor(shift)
        scale = torch.as_tensor(scale)
        if bool((scale == 0.0).any()):
            raise ValueError("Scale must be non-zero.")
        self.register_buffer("_shift", shift)
        self.register_buffer("_scale", scale)

    @property
    def _log_abs_scale(self):
        return self._scale.abs().log()

    def _sample_logabsdet(self, sample_shape):
        """log |det| of one sample of shape `sample_shape`: the sum of log|scale| over its elements
        -- for a scalar scale the element count times log|scale| (one rounding instead of n)."""
        log_scale = self._log_abs_scale
        if log_scale.numel() > 1:
            return log_scale.expand(sample_shape).sum()
        return log_scale * torch.Size(sample_shape).numel()

    def forward(self, inputs, context=None):
        outputs = inputs * self._scale + self._shift
        return outputs, self._sample_logabsdet(inputs.shape[1:]).expand(inputs.shape[0])

    def inverse(self, inputs, context=None):
        outputs = (inputs - self._shift) / self._scale
        return outputs, (-self._sample_logabsdet(inputs.shape[1:])).expand(inputs.shape[0])


class AffineTransform(PointwiseAffineTransform):
    """Deprecated spelling of PointwiseAffineTransform (standard.py:69-84); `None` means the default."""

    def __init__(self, shift=0.0, scale=1.0):
        warnings.warn("Use PointwiseAffineTransform", DeprecationWarning)
        if shift is None:
            shift = 0.0
            warnings.warn("`shift=None` deprecated; default is 0.0")
        if scale is None:
            scale = 1.0
            warnings.warn("`scale=None` deprecated; default is 1.0.")
        super().__init__(shift, scale)


AffineScalarTransform = AffineTransform  # (standard.py:87-88)
