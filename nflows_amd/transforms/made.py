"""MADE conditioner of the autoregressive transforms (configs 1 and 5); PyTorch-ROCm GEMMs.

Checkpoint-compatible with nflows/transforms/made.py: modules `initial_layer`,
`blocks.{i}.linear_layers.{0,1}` (residual) or `blocks.{i}.linear` (feed-forward),
`context_layer`, `final_layer`; every masked layer carries `mask` and `degrees` buffers built by
the same rules (made.py:42-69).  MADE has no `hidden_features` attribute on purpose: the
autoregressive spline transform then applies no 1/sqrt(hidden) scaling (autoregressive.py:464-466).
"""
import torch
from torch import nn
from torch.nn import functional as F

from .. import _cache
from ..nn.functional import linear


# ---- degrees and masks ------------------------------------------------------------------------
# Input feature i has degree i+1.  A hidden unit of degree m may look at inputs of degree <= m,
# an output unit of degree m only at degrees < m; hidden degrees cycle through 1..D-1.

def input_degrees(features):
    return torch.arange(1, features + 1)


def hidden_degrees(units, features, in_degrees, random_mask):
    if random_mask:
        lowest = min(int(in_degrees.min()), features - 1)
        return torch.randint(low=lowest, high=features, size=[units], dtype=torch.long)
    top, bottom = max(1, features - 1), min(1, features - 1)
    return torch.arange(units) % top + bottom


def output_degrees(units, features):
    return torch.repeat_interleave(input_degrees(features), units // features)


class MaskedLinear(nn.Linear):
    """y = x (W * mask)^T + b.

    The mask is a constant 0/1 buffer, so the product W * mask is formed on the fly in front of
    a plain GEMM (hipBLASLt on the MI355X).  The column-wise autoregressive inverse
    (autoregressive.py in this package) slices `weight * mask` of the OUTPUT layer by feature:
    rows [f * multiplier, (f + 1) * multiplier) hold feature f's parameters because output
    degrees are laid out feature-major (`output_degrees`)."""

    def __init__(self, in_degrees, out_features, autoregressive_features, random_mask, is_output, bias=True):
        super().__init__(len(in_degrees), out_features, bias=bias)
        if is_output:
            degrees = output_degrees(out_features, autoregressive_features)
            mask = degrees[:, None] > in_degrees
        else:
            degrees = hidden_degrees(out_features, autoregressive_features, in_degrees, random_mask)
            mask = degrees[:, None] >= in_degrees
        self.register_buffer("mask", mask.float())
        self.register_buffer("degrees", degrees)

    # `weight * mask` kept across calls while the weights are known not to change (the sampling
    # loop evaluates the net once per feature); set / cleared by MADE.frozen_masks
    _frozen_weight = None

    def masked_weight(self):
        if self._frozen_weight is not None:
            return self._frozen_weight
        if torch.is_grad_enabled():
            return self.mask * self.weight           # (training: autograd needs the product)
        # no-grad passes: the product is kept until the weight changes (one elementwise launch per layer
        # and call otherwise -- 18 MB of traffic for the output layer of the D = 784 model); keyed like the
        # packed-weight caches (_cache.py: writes through `.data` need invalidate_packed_weights())
        key = (_cache.epoch(), self.weight.data_ptr(), self.weight._version, self.mask.data_ptr())
        hit = self.__dict__.get("_masked_weight_cache")
        if hit is None or hit[0] != key:
            hit = (key, self.mask * self.weight)
            self.__dict__["_masked_weight_cache"] = hit
        return hit[1]

    def forward(self, x):
        return linear(x, self.masked_weight(), self.bias)


def _norm(features):
    return nn.BatchNorm1d(features, eps=1e-3)


class MaskedFeedforwardBlock(nn.Module):
    """dropout(act(masked_linear(norm(x)))), same width in and out."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        width = len(in_degrees)
        self.batch_norm = _norm(width) if use_batch_norm else None
        self.linear = MaskedLinear(in_degrees, width, autoregressive_features, random_mask, is_output=False)
        self.degrees = self.linear.degrees
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)

    def forward(self, inputs, context=None):
        normed = inputs if self.batch_norm is None else self.batch_norm(inputs)
        return self.dropout(self.activation(self.linear(normed)))


class MaskedResidualBlock(nn.Module):
    """x + L1(dropout(act(norm(L0(act(norm(x))) + context_layer(context)))))."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 zero_initialization=True):
        if random_mask:
            raise ValueError("Masked residual block can't be used with random masks.")
        super().__init__()
        width = len(in_degrees)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, width)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([_norm(width), _norm(width)])
        lower = MaskedLinear(in_degrees, width, autoregressive_features, False, is_output=False)
        upper = MaskedLinear(lower.degrees, width, autoregressive_features, False, is_output=False)
        self.linear_layers = nn.ModuleList([lower, upper])
        self.degrees = upper.degrees
        if bool((self.degrees < in_degrees).any()):
            raise RuntimeError("In a masked residual block, the output degrees can't be"
                               " less than the corresponding input degrees.")
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            for tensor in (upper.weight, upper.bias):
                nn.init.uniform_(tensor, a=-1e-3, b=1e-3)

    def forward(self, inputs, context=None):
        lower, upper = self.linear_layers
        h = self.batch_norm_layers[0](inputs) if self.use_batch_norm else inputs
        h = lower(self.activation(h))
        if context is not None:
            h = h + self.context_layer(context)
        if self.use_batch_norm:
            h = self.batch_norm_layers[1](h)
        return inputs + upper(self.dropout(self.activation(h)))


class MADE(nn.Module):
    """Masked autoencoder: the `output_multiplier` outputs of feature d depend on features < d.

    Layout of the result: [batch, features * output_multiplier], feature-major, which is exactly
    the [batch, d_t * P] parameter layout the fused spline kernel consumes with d_t = features
    (and the interleaved [batch, features, 2] layout of the affine autoregressive kernel).
    `hidden()` exposes the activations in front of the output layer so that the sampling path can
    evaluate one feature's parameters at a time."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2,
                 output_multiplier=1, use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False):
        if use_residual_blocks and random_mask:
            raise ValueError("Residual blocks can't be used with random masks.")
        super().__init__()
        self.use_residual_blocks = use_residual_blocks
        self.activation = activation
        self.initial_layer = MaskedLinear(input_degrees(features), hidden_features, features, random_mask,
                                          is_output=False)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, hidden_features)
        block_type = MaskedResidualBlock if use_residual_blocks else MaskedFeedforwardBlock
        self.blocks = nn.ModuleList()
        degrees = self.initial_layer.degrees
        for _ in range(num_blocks):
            block = block_type(in_degrees=degrees, autoregressive_features=features,
                               context_features=context_features, random_mask=random_mask,
                               activation=activation, dropout_probability=dropout_probability,
                               use_batch_norm=use_batch_norm)
            self.blocks.append(block)
            degrees = block.degrees
        self.final_layer = MaskedLinear(degrees, features * output_multiplier, features, random_mask,
                                        is_output=True)

    def frozen_masks(self):
        """Context manager: every masked layer forms `weight * mask` once instead of per call.
        Only for no-grad evaluation with fixed weights (the autoregressive sampling loop)."""
        import contextlib
        net = self

        @contextlib.contextmanager
        def scope():
            layers = [m for m in net.modules() if isinstance(m, MaskedLinear)]
            try:
                with torch.no_grad():
                    for m in layers:
                        m._frozen_weight = m.mask * m.weight
                yield
            finally:
                for m in layers:
                    m._frozen_weight = None
        return scope()

    def hidden(self, inputs, context=None):
        """Activations fed to the final masked layer."""
        return self.hidden_from_initial(self.initial_layer(inputs), context)

    def hidden_from_initial(self, h, context=None):
        """`hidden` given the output of `initial_layer` (the sampling loop updates that output by one
        rank-1 term per step instead of re-running the [B, D] x [D, H] product)."""
        if context is not None:
            h = h + self.activation(self.context_layer(context))
        if not self.use_residual_blocks:
            h = self.activation(h)
        for block in self.blocks:
            h = block(h, context)
        return h

    def forward(self, inputs, context=None):
        return self.final_layer(self.hidden(inputs, context))

    def is_deterministic(self):
        """No active dropout and no batch-statistics batch norm: two evaluations on the same inputs
        agree (precondition of the column-wise autoregressive inverse)."""
        for m in self.modules():
            if isinstance(m, nn.Dropout) and m.p > 0 and m.training:
                return False
            if isinstance(m, nn.BatchNorm1d) and m.training:
                return False
        return True
