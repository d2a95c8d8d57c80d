"""Residual conditioner networks for feature vectors and for image maps.

The whole-layer kernel (K8) absorbs a plain `ResidualNet`; every other configuration runs here as
PyTorch-ROCm modules (GEMMs / convolutions on the matrix cores through hipBLASLt / MIOpen).

Parameter names and initialisation order match nflows/nn/nets/resnet.py (`initial_layer`,
`blocks.{i}.linear_layers.{0,1}` / `conv_layers.{0,1}`, `blocks.{i}.context_layer`,
`blocks.{i}.batch_norm_layers`, `final_layer`), so reference checkpoints load unchanged and the
same seed gives the same weights.  The nets expose `.hidden_features` / `.hidden_channels`, which
the spline coupling layers read (coupling.py:554-559).

Vector and image variants share one implementation: they differ only in the layer type (Linear /
1x1 and 3x3 Conv2d), the batch-norm type and the attribute name of the two main layers.
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from ..functional import apply_layer


def _dense(n_in, n_out, kernel_size=None):
    return nn.Linear(n_in, n_out)


def _conv(n_in, n_out, kernel_size=1):
    return nn.Conv2d(n_in, n_out, kernel_size=kernel_size, padding=kernel_size // 2)


class _Block(nn.Module):
    """x + L2(drop(act(bn(L1(act(bn(x))))))), optionally gated by a context through a GLU
    (resnet.py:9-52 for vectors, :103-149 for maps)."""

    _layers_name = None   # "linear_layers" / "conv_layers": the reference's attribute names

    def __init__(self, width, context_width, activation, dropout_probability, use_batch_norm,
                 zero_initialization):
        super().__init__()
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        # registration order = the reference's, so that a seed reproduces its weights
        self._register_in_reference_order(width, context_width, use_batch_norm)
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            last = self._main()[-1]
            nn.init.uniform_(last.weight, -1e-3, 1e-3)
            nn.init.uniform_(last.bias, -1e-3, 1e-3)

    def _register_in_reference_order(self, width, context_width, use_batch_norm):
        raise NotImplementedError()

    def _main(self):
        return getattr(self, self._layers_name)

    def _generic_forward(self, inputs, context):
        first, second = self._main()
        h = inputs
        if self.use_batch_norm:
            h = self.batch_norm_layers[0](h)
        h = apply_layer(first, self.activation(h))
        if self.use_batch_norm:
            h = self.batch_norm_layers[1](h)
        h = apply_layer(second, self.dropout(self.activation(h)))
        if context is not None:
            h = F.glu(torch.cat((h, self.context_layer(context)), dim=1), dim=1)
        return inputs + h

    def forward(self, inputs, context=None):
        return self._generic_forward(inputs, context)


class ResidualBlock(_Block):
    _layers_name = "linear_layers"

    def __init__(self, features, context_features, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False, zero_initialization=True):
        super().__init__(features, context_features, activation, dropout_probability, use_batch_norm,
                         zero_initialization)

    def _register_in_reference_order(self, width, context_width, use_batch_norm):
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(nn.BatchNorm1d(width, eps=1e-3) for _ in range(2))
        if context_width is not None:
            self.context_layer = _dense(context_width, width)
        self.linear_layers = nn.ModuleList(_dense(width, width) for _ in range(2))

    def _fused_inference(self, inputs, context):
        """No-grad HIP path: bias + ReLU of the first linear layer run in the GEMM epilogue
        (hipBLASLt via torch._addmm_activation) instead of a separate elementwise kernel."""
        return (context is None and self.activation is F.relu and not self.use_batch_norm
                and inputs.is_cuda and not torch.is_grad_enabled()
                and (not self.training or self.dropout.p == 0.0))

    def forward(self, inputs, context=None):
        if self._fused_inference(inputs, context):
            first, second = self.linear_layers
            h = torch._addmm_activation(first.bias, F.relu(inputs), first.weight.t())
            return inputs + torch.addmm(second.bias, h, second.weight.t())
        return self._generic_forward(inputs, context)


class ConvResidualBlock(_Block):
    _layers_name = "conv_layers"

    def __init__(self, channels, context_channels=None, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False, zero_initialization=True):
        super().__init__(channels, context_channels, activation, dropout_probability, use_batch_norm,
                         zero_initialization)

    def _register_in_reference_order(self, width, context_width, use_batch_norm):
        if context_width is not None:
            self.context_layer = _conv(context_width, width, kernel_size=1)
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(nn.BatchNorm2d(width, eps=1e-3) for _ in range(2))
        self.conv_layers = nn.ModuleList(_conv(width, width, kernel_size=3) for _ in range(2))


def _has_inner_hooks(net):
    """forward / backward hooks on any module INSIDE the net (the net's own hooks run either way)"""
    for m in net.modules():
        if m is net:
            continue
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return True
    return False


class fused_training:
    """`with fused_training(False): ...` -- the eager conditioner modules under autograd inside the block (autocast
    experiments, hooks, double backward THROUGH THE CONDITIONER); restores the previous class-level setting on exit.
    The spline / affine layer kernels behind the conditioner (autograd.RqsCoupling, AffineCoupling, ...) stay
    once-differentiable either way: a gradient penalty that differentiates twice through the layer's own map needs
    the float64 functional path or the reference's eager ops."""

    def __init__(self, enabled):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.saved = _Net.fuse_training
        _Net.fuse_training = self.enabled
        return self

    def __exit__(self, *exc):
        _Net.fuse_training = self.saved
        return False


class _Net(nn.Module):
    """first layer (context concatenated to the input if given) -> blocks -> last layer."""

    _block = None
    _make = None
    # K14 for the hidden part under autograd.  Class-level default from NFA_K14 (0 = off); `net.fuse_training = False`
    # on an instance, or the context manager `nflows_amd.nn.nets.resnet.fused_training(False)`, switch it at run time.
    # The fused path is once-differentiable (no double backward: gradient penalties / Jacobian regularisers with
    # create_graph=True need it OFF), computes in fp32 whatever torch.autocast says, and does not run forward hooks of
    # the inner layers -- so it steps aside by itself while autocast is active or any inner module carries a hook.
    fuse_training = os.environ.get("NFA_K14", "1") != "0"

    def _build(self, n_in, n_out, width, context_width, num_blocks, activation, dropout_probability,
               use_batch_norm):
        make = type(self)._make
        self.initial_layer = make(n_in if context_width is None else n_in + context_width, width)
        self.blocks = nn.ModuleList(
            type(self)._block(width, context_width, activation=activation,
                              dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
            for _ in range(num_blocks))
        self.final_layer = make(width, n_out)

    def _fused_training(self, inputs, context):
        """K14 (one kernel for the hidden part's forward pass, one for its input gradients) applies: plain
        ResidualNet under autograd on the device, ReLU, no context / batch norm / active dropout, the shapes
        `ops.resnet_hidden_train_supported` lists."""
        if not (self.fuse_training and type(self) is ResidualNet and context is None and torch.is_grad_enabled()
                and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 2):
            return False
        if torch.is_autocast_enabled() or _has_inner_hooks(self):
            return False
        from ... import ops
        if not ops.resnet_hidden_train_supported(inputs.shape[0], inputs.shape[1], self.hidden_features, len(self.blocks)):
            return False
        w = self.initial_layer.weight
        if inputs.shape[1] != self.initial_layer.in_features or w.dtype != torch.float32 or w.device != inputs.device:
            return False   # (half-precision or misplaced parameters: the eager modules report it their own way)
        for b in self.blocks:
            if (b.activation is not F.relu or b.use_batch_norm or (b.training and b.dropout.p != 0.0)
                    or getattr(b, "context_layer", None) is not None):
                return False
        return any(p.requires_grad for p in self.parameters()) or inputs.requires_grad

    def hidden(self, inputs, context=None):
        """Activations in front of `final_layer` (the fused spline kernels K7 / K7b consume these
        and apply `final_layer` themselves)."""
        if self._fused_training(inputs, context):
            from ... import autograd as AG
            return AG.ResidualNetHidden.apply(inputs, False, *self._hidden_parameters())
        h = inputs if context is None else torch.cat((inputs, context), dim=1)
        h = apply_layer(self.initial_layer, h)
        for block in self.blocks:
            h = block(h, context=context)
        return h

    def _hidden_parameters(self):
        params = [self.initial_layer.weight, self.initial_layer.bias]
        for b in self.blocks:
            params += [b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight,
                       b.linear_layers[1].bias]
        return params

    def forward(self, inputs, context=None):
        final = self.final_layer
        if (self._fused_training(inputs, context) and type(final) is nn.Linear and final.bias is not None
                and final.out_features % 4 == 0 and final.out_features <= 32768   # (the packer's / kernel's limit)
                and final.in_features == self.hidden_features):
            # the whole conditioner's forward pass in one kernel (K14 with the final Linear appended)
            from ... import autograd as AG
            return AG.ResidualNetHidden.apply(inputs, True, *self._hidden_parameters(), final.weight, final.bias)
        return apply_layer(final, self.hidden(inputs, context))


class ResidualNet(_Net):
    """Linear -> num_blocks residual blocks -> Linear, on feature vectors (resnet.py:55-100)."""

    _block = ResidualBlock
    _make = staticmethod(_dense)

    def __init__(self, in_features, out_features, hidden_features, context_features=None,
                 num_blocks=2, activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        self._build(in_features, out_features, hidden_features, context_features, num_blocks, activation,
                    dropout_probability, use_batch_norm)


class ConvResidualNet(_Net):
    """1x1 conv -> num_blocks blocks of 3x3 convs -> 1x1 conv, on [B, C, H, W] maps
    (resnet.py:152-205); the conditioner of the image coupling layers."""

    _block = ConvResidualBlock
    _make = staticmethod(_conv)

    def __init__(self, in_channels, out_channels, hidden_channels, context_channels=None, num_blocks=2,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        self.context_channels = context_channels
        self.hidden_channels = hidden_channels
        self._build(in_channels, out_channels, hidden_channels, context_channels, num_blocks, activation,
                    dropout_probability, use_batch_norm)
