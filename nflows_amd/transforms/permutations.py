"""Column permutations (reference: nflows/transforms/permutations.py).

`forward` is `index_select(inputs, dim, permutation)` with a zero logabsdet, `inverse` selects
with the inverse permutation.  2-D HIP tensors of 4-byte elements with dim=1 go through the K4
kernel (bit-exact copy); when the permutation sits next to a coupling layer inside a
`CompositeTransform` it is not launched at all but folded into that layer's kernel.
"""
import torch

from .. import ops
from ..utils import typechecks as check
from .base import Transform


class Permutation(Transform):
    """Permutes `dim` of the inputs with a fixed permutation (permutations.py:9-45)."""

    def __init__(self, permutation, dim=1):
        if permutation.ndimension() != 1:
            raise ValueError("Permutation must be a 1D tensor.")
        if not check.is_positive_int(dim):
            raise ValueError("dim must be a positive integer.")
        super().__init__()
        self._dim = dim
        self.register_buffer("_permutation", permutation)
        self._inv_cache = None  # (version, data_ptr, tensor)

    @property
    def _inverse_permutation(self):
        p = self._permutation
        key = (p._version, p.data_ptr(), p.device)
        if self._inv_cache is None or self._inv_cache[0] != key:
            self._inv_cache = (key, torch.argsort(p))
        return self._inv_cache[1]

    def _check(self, inputs):
        if self._dim >= inputs.ndimension():
            raise ValueError("No dimension {} in inputs.".format(self._dim))
        if inputs.shape[self._dim] != len(self._permutation):
            raise ValueError("Dimension {} in inputs must be of size {}.".format(
                self._dim, len(self._permutation)))

    def _permute(self, inputs, permutation):
        self._check(inputs)
        if self._dim == 1 and inputs.dim() == 2 and inputs.is_cuda and inputs.element_size() == 4 \
                and not (torch.is_grad_enabled() and inputs.requires_grad):
            outputs = ops.permute_cols(inputs, permutation)
        elif inputs.is_cuda:
            outputs = torch.index_select(inputs, self._dim, permutation)  # other ranks / dtypes
        else:
            raise NotImplementedError(
                "nflows_amd: inputs on %s; the MI355X path has no CPU fallback" % inputs.device)
        return outputs, inputs.new_zeros(inputs.shape[0])

    def forward(self, inputs, context=None):
        return self._permute(inputs, self._permutation)

    def inverse(self, inputs, context=None):
        return self._permute(inputs, self._inverse_permutation)


class RandomPermutation(Permutation):
    """A random permutation fixed at construction (permutations.py:48-54)."""

    def __init__(self, features, dim=1):
        if not check.is_positive_int(features):
            raise ValueError("Number of features must be a positive integer.")
        super().__init__(torch.randperm(features), dim)


class ReversePermutation(Permutation):
    """Reverses the order of the features (permutations.py:57-63)."""

    def __init__(self, features, dim=1):
        if not check.is_positive_int(features):
            raise ValueError("Number of features must be a positive integer.")
        super().__init__(torch.arange(features - 1, -1, -1), dim)
