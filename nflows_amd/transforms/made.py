"""MADE conditioner for the autoregressive transforms (configs 1 and 5); PyTorch-ROCm GEMMs.

Mask / degree rules and parameter names follow nflows/transforms/made.py (`initial_layer`,
`blocks.{i}.linear_layers.{0,1}` or `blocks.{i}.linear`, `context_layer`, `final_layer`, each
masked layer with `mask` and `degrees` buffers), so reference checkpoints load unchanged.
MADE deliberately has no `hidden_features` attribute: the autoregressive spline transform
therefore applies no 1/sqrt(hidden) scaling (autoregressive.py:464-466).
"""
import torch
from torch import nn
from torch.nn import functional as F


def _input_degrees(features):
    return torch.arange(1, features + 1)


class MaskedLinear(nn.Linear):
    """Linear layer whose weight is multiplied by a fixed 0/1 mask (made.py:16-72).

    Hidden units get degrees 1..D-1 cyclically (or at random); unit j may see input i iff
    degree_j >= degree_i (hidden) or degree_j > degree_i (output layer)."""

    def __init__(self, in_degrees, out_features, autoregressive_features, random_mask, is_output,
                 bias=True):
        super().__init__(in_features=len(in_degrees), out_features=out_features, bias=bias)
        mask, degrees = self._get_mask_and_degrees(in_degrees, out_features, autoregressive_features,
                                                   random_mask, is_output)
        self.register_buffer("mask", mask)
        self.register_buffer("degrees", degrees)

    @classmethod
    def _get_mask_and_degrees(cls, in_degrees, out_features, autoregressive_features, random_mask,
                              is_output):
        D = autoregressive_features
        if is_output:
            out_degrees = torch.repeat_interleave(_input_degrees(D), out_features // D)
            mask = (out_degrees[:, None] > in_degrees).float()
        else:
            if random_mask:
                low = min(int(torch.min(in_degrees).item()), D - 1)
                out_degrees = torch.randint(low=low, high=D, size=[out_features], dtype=torch.long)
            else:
                out_degrees = torch.arange(out_features) % max(1, D - 1) + min(1, D - 1)
            mask = (out_degrees[:, None] >= in_degrees).float()
        return mask, out_degrees

    def forward(self, x):
        return F.linear(x, self.weight * self.mask, self.bias)


class MaskedFeedforwardBlock(nn.Module):
    """(batch norm) -> masked linear -> activation -> dropout (made.py:75-123)."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        features = len(in_degrees)
        self.batch_norm = nn.BatchNorm1d(features, eps=1e-3) if use_batch_norm else None
        self.linear = MaskedLinear(in_degrees, features, autoregressive_features, random_mask, False)
        self.degrees = self.linear.degrees
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)

    def forward(self, inputs, context=None):
        h = self.batch_norm(inputs) if self.batch_norm else inputs
        return self.dropout(self.activation(self.linear(h)))


class MaskedResidualBlock(nn.Module):
    """x + L1(act(L0(act(x)) + context)) with masked layers (made.py:126-202)."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 zero_initialization=True):
        if random_mask:
            raise ValueError("Masked residual block can't be used with random masks.")
        super().__init__()
        features = len(in_degrees)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(nn.BatchNorm1d(features, eps=1e-3) for _ in range(2))
        first = MaskedLinear(in_degrees, features, autoregressive_features, False, False)
        second = MaskedLinear(first.degrees, features, autoregressive_features, False, False)
        self.linear_layers = nn.ModuleList([first, second])
        self.degrees = second.degrees
        if not bool(torch.all(self.degrees >= in_degrees)):
            raise RuntimeError("In a masked residual block, the output degrees can't be"
                               " less than the corresponding input degrees.")
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            nn.init.uniform_(second.weight, a=-1e-3, b=1e-3)
            nn.init.uniform_(second.bias, a=-1e-3, b=1e-3)

    def forward(self, inputs, context=None):
        h = inputs
        if self.use_batch_norm:
            h = self.batch_norm_layers[0](h)
        h = self.linear_layers[0](self.activation(h))
        if context is not None:
            h = h + self.context_layer(context)
        if self.use_batch_norm:
            h = self.batch_norm_layers[1](h)
        h = self.linear_layers[1](self.dropout(self.activation(h)))
        return inputs + h


class MADE(nn.Module):
    """Masked autoencoder: output block d depends only on inputs < d (made.py:205-283)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2,
                 output_multiplier=1, use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False):
        if use_residual_blocks and random_mask:
            raise ValueError("Residual blocks can't be used with random masks.")
        super().__init__()
        self.initial_layer = MaskedLinear(_input_degrees(features), hidden_features, features,
                                          random_mask, False)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, hidden_features)
        self.use_residual_blocks = use_residual_blocks
        self.activation = activation
        block_cls = MaskedResidualBlock if use_residual_blocks else MaskedFeedforwardBlock
        blocks = []
        degrees = self.initial_layer.degrees
        for _ in range(num_blocks):
            blocks.append(block_cls(in_degrees=degrees, autoregressive_features=features,
                                    context_features=context_features, random_mask=random_mask,
                                    activation=activation, dropout_probability=dropout_probability,
                                    use_batch_norm=use_batch_norm))
            degrees = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = MaskedLinear(degrees, features * output_multiplier, features, random_mask, True)

    def hidden(self, inputs, context=None):
        """Activations fed to the final masked layer."""
        h = self.initial_layer(inputs)
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
        """True when two evaluations on the same inputs give the same outputs (no active dropout,
        no batch-statistics batch norm): the precondition of the column-wise inverse."""
        for m in self.modules():
            if isinstance(m, nn.Dropout) and m.p > 0 and m.training:
                return False
            if isinstance(m, nn.BatchNorm1d) and m.training:
                return False
        return True
