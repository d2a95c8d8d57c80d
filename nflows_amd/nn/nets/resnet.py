"""Residual conditioner network (stays PyTorch-ROCm: its GEMMs run on MFMA through hipBLASLt).

Parameter names and initialisation order match nflows/nn/nets/resnet.py (`initial_layer`,
`blocks.{i}.linear_layers.{0,1}`, `blocks.{i}.context_layer`, `blocks.{i}.batch_norm_layers`,
`final_layer`), so reference checkpoints load unchanged and the same seed gives the same weights.
Exposes `.hidden_features`, which the spline coupling layer reads (coupling.py:554-556).
"""
import torch
from torch import nn
from torch.nn import functional as F


class ResidualBlock(nn.Module):
    """x + W2 * act(W1 * act(x)), optional batch norm, dropout and GLU context gate
    (resnet.py:9-52)."""

    def __init__(self, features, context_features, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False, zero_initialization=True):
        super().__init__()
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(nn.BatchNorm1d(features, eps=1e-3) for _ in range(2))
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList(nn.Linear(features, features) for _ in range(2))
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            last = self.linear_layers[-1]
            nn.init.uniform_(last.weight, -1e-3, 1e-3)
            nn.init.uniform_(last.bias, -1e-3, 1e-3)

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
        h = inputs
        for step in range(2):
            if self.use_batch_norm:
                h = self.batch_norm_layers[step](h)
            h = self.activation(h)
            if step == 1:
                h = self.dropout(h)
            h = self.linear_layers[step](h)
        if context is not None:
            h = F.glu(torch.cat((h, self.context_layer(context)), dim=1), dim=1)
        return inputs + h


class ResidualNet(nn.Module):
    """Linear -> num_blocks residual blocks -> Linear, on 1-D feature vectors (resnet.py:55-100)."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None,
                 num_blocks=2, activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        first_in = in_features if context_features is None else in_features + context_features
        self.initial_layer = nn.Linear(first_in, hidden_features)
        self.blocks = nn.ModuleList(
            ResidualBlock(features=hidden_features, context_features=context_features,
                          activation=activation, dropout_probability=dropout_probability,
                          use_batch_norm=use_batch_norm)
            for _ in range(num_blocks))
        self.final_layer = nn.Linear(hidden_features, out_features)

    def hidden(self, inputs, context=None):
        """Activations in front of `final_layer` (the fused spline kernel consumes these and
        applies `final_layer` itself)."""
        h = inputs if context is None else torch.cat((inputs, context), dim=1)
        h = self.initial_layer(h)
        for block in self.blocks:
            h = block(h, context=context)
        return h

    def forward(self, inputs, context=None):
        return self.final_layer(self.hidden(inputs, context))


class ConvResidualBlock(nn.Module):
    """ResidualBlock on [B, C, H, W] maps: 3x3 convolutions, optional batch norm, dropout and a
    1x1-convolved GLU context gate (resnet.py:103-149).  Parameter names match the reference
    (`conv_layers.{0,1}`, `context_layer`, `batch_norm_layers`)."""

    def __init__(self, channels, context_channels=None, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False, zero_initialization=True):
        super().__init__()
        self.activation = activation
        if context_channels is not None:
            self.context_layer = nn.Conv2d(context_channels, channels, kernel_size=1, padding=0)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(nn.BatchNorm2d(channels, eps=1e-3) for _ in range(2))
        self.conv_layers = nn.ModuleList(nn.Conv2d(channels, channels, kernel_size=3, padding=1) for _ in range(2))
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            last = self.conv_layers[-1]
            nn.init.uniform_(last.weight, -1e-3, 1e-3)
            nn.init.uniform_(last.bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        h = inputs
        for step in range(2):
            if self.use_batch_norm:
                h = self.batch_norm_layers[step](h)
            h = self.activation(h)
            if step == 1:
                h = self.dropout(h)
            h = self.conv_layers[step](h)
        if context is not None:
            h = F.glu(torch.cat((h, self.context_layer(context)), dim=1), dim=1)
        return inputs + h


class ConvResidualNet(nn.Module):
    """1x1 conv -> num_blocks ConvResidualBlocks -> 1x1 conv (resnet.py:152-205); the conditioner of
    the image coupling layers.  Exposes `.hidden_channels` (coupling.py:557-559)."""

    def __init__(self, in_channels, out_channels, hidden_channels, context_channels=None, num_blocks=2,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        self.context_channels = context_channels
        self.hidden_channels = hidden_channels
        first_in = in_channels if context_channels is None else in_channels + context_channels
        self.initial_layer = nn.Conv2d(first_in, hidden_channels, kernel_size=1, padding=0)
        self.blocks = nn.ModuleList(
            ConvResidualBlock(channels=hidden_channels, context_channels=context_channels,
                              activation=activation, dropout_probability=dropout_probability,
                              use_batch_norm=use_batch_norm)
            for _ in range(num_blocks))
        self.final_layer = nn.Conv2d(hidden_channels, out_channels, kernel_size=1, padding=0)

    def forward(self, inputs, context=None):
        h = inputs if context is None else torch.cat((inputs, context), dim=1)
        h = self.initial_layer(h)
        for block in self.blocks:
            h = block(h, context)
        return self.final_layer(h)
