"""Column permutations of the hot path.

API of nflows/transforms/permutations.py: `Permutation(permutation, dim=1)`,
`RandomPermutation(features, dim=1)`, `ReversePermutation(features, dim=1)`, buffer
`_permutation`; forward = index_select with the permutation (:27-39), inverse = index_select
with its argsort (:22-24, :44-45), logabsdet = zeros.

On a HIP device a 2-D tensor of 4-byte elements permuted along dim 1 is a bit-exact K4 kernel;
next to a coupling layer inside `CompositeTransform` the permutation is not launched at all
(folded into that layer's gather / scatter).
"""
import torch

from .. import ops
from ..utils import typechecks as check
from .base import Transform


def _require_feature_count(features):
    if not check.is_positive_int(features):
        raise ValueError("Number of features must be a positive integer.")


class Permutation(Transform):
    def __init__(self, permutation, dim=1):
        if permutation.ndimension() != 1:
            raise ValueError("Permutation must be a 1D tensor.")
        if not check.is_positive_int(dim):
            raise ValueError("dim must be a positive integer.")
        super().__init__()
        self.register_buffer("_permutation", permutation)
        self._dim = dim
        self._argsort_cache = None  # (key, tensor): argsort is recomputed only if the buffer changes

    @property
    def _inverse_permutation(self):
        buf = self._permutation
        key = (buf.data_ptr(), buf._version, buf.device)
        if self._argsort_cache is None or self._argsort_cache[0] != key:
            self._argsort_cache = (key, torch.argsort(buf))
        return self._argsort_cache[1]

    def _check(self, inputs):
        dim, size = self._dim, self._buffers["_permutation"].shape[0]
        if inputs.ndimension() <= dim:
            raise ValueError("No dimension {} in inputs.".format(dim))
        if inputs.shape[dim] != size:
            raise ValueError("Dimension {} in inputs must be of size {}.".format(dim, size))

    def _select(self, inputs, index):
        self._check(inputs)
        if not inputs.is_cuda:
            raise NotImplementedError(
                "nflows_amd: inputs on %s; the MI355X path has no CPU fallback" % inputs.device)
        wants_grad = torch.is_grad_enabled() and inputs.requires_grad
        if self._dim == 1 and inputs.dim() == 2 and inputs.element_size() == 4 and not wants_grad:
            permuted = ops.permute_cols(inputs, index)
        else:  # other ranks / dtypes / autograd: the device's index_select
            permuted = torch.index_select(inputs, self._dim, index)
        return permuted, inputs.new_zeros(inputs.shape[0])

    def forward(self, inputs, context=None):
        return self._select(inputs, self._permutation)

    def inverse(self, inputs, context=None):
        return self._select(inputs, self._inverse_permutation)


class RandomPermutation(Permutation):
    """torch.randperm drawn once at construction (consumes the global RNG like the reference)."""

    def __init__(self, features, dim=1):
        _require_feature_count(features)
        super().__init__(torch.randperm(features), dim)


class ReversePermutation(Permutation):
    """features-1, ..., 1, 0."""

    def __init__(self, features, dim=1):
        _require_feature_count(features)
        super().__init__(torch.arange(features - 1, -1, -1), dim)
