"""The Transform API -- the drop-in boundary of this package.

Mirrors nflows/transforms/base.py: `Transform.forward/inverse(inputs, context=None) ->
(outputs, logabsdet)` (:22-29), `CompositeTransform` (:32-60) and `InverseTransform` (:215-231).

MI355X-specific addition: `CompositeTransform` recognises a column `Permutation` that is
adjacent to a coupling layer and hands the permutation to the coupling kernel (gather on the way
in for `forward`, scatter on the way out for `inverse`), which removes one full read+write pass
over the [batch, features] activations per layer.  Results are bit-identical to running the two
transforms one after the other.
"""
import torch
from torch import nn

from ..errors import InputOutsideDomain, InverseNotAvailable  # noqa: F401  (re-exported)


class Transform(nn.Module):
    """Base class of all transforms."""

    def forward(self, inputs, context=None):
        raise NotImplementedError()

    def inverse(self, inputs, context=None):
        raise InverseNotAvailable()


def _is_column_permutation(t):
    from .permutations import Permutation
    return isinstance(t, Permutation) and t._dim == 1


def _accepts_fused_permutation(t, inputs):
    return getattr(t, "supports_fused_permutation", False) and inputs.dim() == 2


class CompositeTransform(Transform):
    """Applies transforms in the given order; log-determinants add up (base.py:45-52)."""

    def __init__(self, transforms, fuse_permutations=True):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)
        self.fuse_permutations = fuse_permutations

    @staticmethod
    def _cascade(inputs, funcs, context):
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        for func in funcs:
            outputs, logabsdet = func(outputs, context)
            total += logabsdet
        return outputs, total

    def forward(self, inputs, context=None):
        layers = list(self._transforms)
        if not self.fuse_permutations:
            return self._cascade(inputs, layers, context)
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        i = 0
        while i < len(layers):
            t = layers[i]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if nxt is not None and _is_column_permutation(t) and _accepts_fused_permutation(nxt, outputs):
                t._check(outputs)
                # permutation folded into the layer's gather, `total += logabsdet` into its kernel
                outputs, _ = nxt.forward(outputs, context, in_perm=t._permutation,
                                         logabsdet_accumulator=total)
                i += 2
            elif _accepts_fused_permutation(t, outputs):
                outputs, _ = t.forward(outputs, context, logabsdet_accumulator=total)
                i += 1
            else:
                outputs, logabsdet = t(outputs, context)
                total += logabsdet
                i += 1
        return outputs, total

    def inverse(self, inputs, context=None):
        layers = list(self._transforms)[::-1]
        if not self.fuse_permutations:
            return self._cascade(inputs, (t.inverse for t in layers), context)
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        i = 0
        while i < len(layers):
            t = layers[i]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if nxt is not None and _is_column_permutation(nxt) and _accepts_fused_permutation(t, outputs):
                nxt._check(outputs)
                # Permutation.inverse after the layer == scatter through the forward permutation
                outputs, _ = t.inverse(outputs, context, out_scatter=nxt._permutation,
                                       logabsdet_accumulator=total)
                i += 2
            elif _accepts_fused_permutation(t, outputs):
                outputs, _ = t.inverse(outputs, context, logabsdet_accumulator=total)
                i += 1
            else:
                outputs, logabsdet = t.inverse(outputs, context)
                total += logabsdet
                i += 1
        return outputs, total


class InverseTransform(Transform):
    """Swaps forward and inverse of a transform (base.py:215-231)."""

    def __init__(self, transform):
        super().__init__()
        self._transform = transform

    def forward(self, inputs, context=None):
        return self._transform.inverse(inputs, context)

    def inverse(self, inputs, context=None):
        return self._transform(inputs, context)
