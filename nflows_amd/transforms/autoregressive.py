"""Autoregressive transforms of configs 1 and 5 on the MI355X kernels
(reference: nflows/transforms/autoregressive.py:25-128 and :404-495).

forward : one MADE pass + ONE fused kernel (elementwise map + per-sample logabsdet sum).
inverse : the reference's D-step fixed-point loop (autoregressive.py:43-52): every step re-runs
          MADE on the current outputs and the elementwise inverse over all D features; after
          step t the first t features are final.  Same semantics, same number of passes.
"""
import os

import numpy as np
import torch
from torch.nn import functional as F

from .. import _cache
from .. import _native as N
from .. import ops
from . import made as made_module
from .base import Transform
from .splines import rational_quadratic
from . import splines
from ..utils import torchutils


class _LazyRows:
    """`make()[index]`, with `make` called on the first access only."""

    def __init__(self, make):
        self._make, self._value = make, None

    def __getitem__(self, index):
        if self._value is None:
            self._value = self._make()
        return self._value[index]


class AutoregressiveTransform(Transform):
    def __init__(self, autoregressive_net):
        super().__init__()
        self.autoregressive_net = autoregressive_net

    def forward(self, inputs, context=None):
        params = self.autoregressive_net(inputs, context)
        return self._elementwise_forward(inputs, params)

    # Set to False to run the reference's loop verbatim (D full passes, autoregressive.py:43-52).
    columnwise_inverse = True

    def inverse(self, inputs, context=None):
        net = self.autoregressive_net
        if (self.columnwise_inverse and inputs.dim() == 2 and isinstance(net, made_module.MADE)
                and net.is_deterministic() and hasattr(self, "_inverse_column")):
            return self._inverse_columnwise(inputs, context)
        return self._inverse_reference_loop(inputs, context)

    def _inverse_reference_loop(self, inputs, context=None):
        num_inputs = int(np.prod(inputs.shape[1:]))
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for _ in range(num_inputs):
            params = self.autoregressive_net(outputs, context)
            outputs, logabsdet = self._elementwise_inverse(inputs, params)
        return outputs, logabsdet

    def _inverse_columnwise(self, inputs, context=None):
        """Same result as the reference loop with O(D) instead of O(D^2) work in the output layer
        and the elementwise transform.

        In the reference loop, iteration t+1 recomputes ALL D*P conditioner outputs and ALL D
        elementwise inverses, although (by the autoregressive masks) only feature t changes: its
        parameters depend on features < t, which became final in earlier iterations, and the
        features it has not reached yet are multiplied by exactly-zero masked weights.  Here step
        t evaluates the hidden layers on the current outputs (unreached features still zero),
        takes only feature t's P rows of the final masked layer, inverts that one column and adds
        its log-derivative.  Identical up to summation order."""
        net = self.autoregressive_net
        batch, features = inputs.shape
        mult = self._output_dim_multiplier()
        final = net.final_layer
        bias = final.bias.view(features, mult)
        if torch.is_grad_enabled():
            weight = (final.weight * final.mask).view(features, mult, -1)
        else:   # (formed on first use: the kernels below pack their own copy of the output layer)
            weight = _LazyRows(lambda: final.masked_weight().view(features, mult, -1))
        outputs = torch.zeros_like(inputs)
        logabsdet = inputs.new_zeros(batch)
        if torch.is_grad_enabled():
            # differentiable form (training an inverse autoregressive flow, reparameterised
            # sampling): the tensor handed to the conditioner is saved by its first layer for the
            # weight gradient, so it is never written again -- every step writes its column into a
            # fresh copy, like the reference's out-of-place loop (autoregressive.py:43-52)
            for t in range(features):
                h = net.hidden(outputs, context)
                params_t = torch.addmm(bias[t], h, weight[t].t())
                column, lad_t = self._inverse_column(inputs[:, t], params_t)
                outputs = outputs.clone()
                outputs[:, t] = column
                logabsdet = logabsdet + lad_t
            return outputs, logabsdet
        # No-grad (sampling): `weight * mask` of every layer is formed once, and the first layer's
        # output -- bias + sum over the features found so far of column (x) masked weight column --
        # is carried along and grows by one rank-1 term per step (all still-zero features contribute
        # exact zeros to the reference's full product).
        #
        # The sequential part ends early: a hidden unit of degree d reads features < d only, so once
        # feature `last` = (largest hidden degree) - 1 is found every hidden activation is final, and
        # the remaining features -- each a function of that one hidden vector and its own P output
        # rows -- are inverted together: one GEMM for their parameters, one elementwise launch.
        # (MADE with H < D - 1 sequential degrees: D = 784, H = 256 leaves 256 sequential steps.)
        first = net.initial_layer
        sequential = min(features, self._sequential_steps())
        with net.frozen_masks():
            # the sequential features in one persistent kernel where there is one (K12), else step by step
            fast = self._sequential_kernel(inputs, context, sequential)
            if fast is not None:
                outputs, logabsdet, h = fast
            else:
                first_weight = first.masked_weight().t().contiguous()  # [D, H]: row t = feature t's weights
                pre = first.bias.detach().expand(batch, -1).contiguous() if first.bias is not None \
                    else inputs.new_zeros(batch, first_weight.shape[1])
                for t in range(sequential):
                    h = net.hidden_from_initial(pre, context)
                    params_t = torch.addmm(bias[t], h, weight[t].t())
                    column, lad_t = self._inverse_column(inputs[:, t], params_t)
                    outputs[:, t] = column
                    logabsdet += lad_t
                    if t + 1 < features:
                        pre.addr_(column, first_weight[t])
                if sequential < features:
                    h = net.hidden_from_initial(pre, context)
            if sequential < features:
                tail = self._output_layer_kernel(inputs, h, sequential, inverse=True, out=outputs) \
                    if hasattr(self, "_output_layer_kernel") and outputs.is_contiguous() else None
                if tail is not None:   # (K13: the remaining features' rows of the output layer inside the spline kernel)
                    logabsdet += tail[1]
                else:
                    rest = features - sequential
                    params = torch.addmm(bias[sequential:].reshape(-1), h, weight[sequential:].reshape(rest * mult, -1).t())
                    columns, lad_rest = self._elementwise_inverse(inputs[:, sequential:].contiguous(),
                                                                  params.view(batch, rest, mult))
                    outputs[:, sequential:] = columns
                    logabsdet += lad_rest
        return outputs, logabsdet

    def _sequential_kernel(self, inputs, context, sequential):
        """Hook: (outputs with the first `sequential` columns found, their logabsdet, the final hidden
        vector) from a kernel that runs all sequential steps itself, or None."""
        return None

    def _sequential_steps(self):
        """Number of leading features whose inversion changes a hidden activation of the MADE: the
        largest degree of any hidden unit (made.py: a unit of degree d is connected to inputs of
        degree <= d, i.e. features 0 .. d - 1)."""
        # (one device read per cache epoch: `degrees` / `mask` are registered buffers, a load_state_dict from a
        #  checkpoint with other random masks replaces them and advances the epoch)
        cached = self.__dict__.get("_sequential_steps_cache")
        if cached is None or cached[0] != _cache.epoch():
            net = self.autoregressive_net
            degrees = [net.initial_layer.degrees] + [b.degrees for b in net.blocks]
            cached = (_cache.epoch(), int(max(int(d.max()) for d in degrees)))
            self.__dict__["_sequential_steps_cache"] = cached
        return cached[1]

    def _output_dim_multiplier(self):
        raise NotImplementedError()

    def _elementwise_forward(self, inputs, autoregressive_params):
        raise NotImplementedError()

    def _elementwise_inverse(self, inputs, autoregressive_params):
        raise NotImplementedError()


class MaskedAffineAutoregressiveTransform(AutoregressiveTransform):
    """MAF layer: y_d = softplus(u_d)+1e-3) * x_d + shift_d with (u, shift) = MADE(x)
    (autoregressive.py:64-128); parameters interleaved [B, D, 2]."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False):
        self.features = features
        net = made_module.MADE(features=features, hidden_features=hidden_features,
                               context_features=context_features, num_blocks=num_blocks,
                               output_multiplier=self._output_dim_multiplier(),
                               use_residual_blocks=use_residual_blocks, random_mask=random_mask,
                               activation=activation, dropout_probability=dropout_probability,
                               use_batch_norm=use_batch_norm)
        self._epsilon = 1e-3
        super().__init__(net)

    def _output_dim_multiplier(self):
        return 2

    def _elementwise_forward(self, inputs, autoregressive_params):
        return ops.affine_autoregressive(inputs, autoregressive_params, inverse=False)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return ops.affine_autoregressive(inputs, autoregressive_params, inverse=True)

    def _inverse_column(self, column, params):
        """One feature: params [B, 2] = (scale logit, shift); K2b with a single feature."""
        out, lad = ops.affine_autoregressive(column.reshape(-1, 1), params, inverse=True)
        return out.reshape(-1), lad


class MaskedPiecewiseRationalQuadraticAutoregressiveTransform(AutoregressiveTransform):
    """Autoregressive neural spline layer (autoregressive.py:404-495): per feature 3K-1 / 3K+1
    logits from MADE, the RQ functional elementwise, logabsdet summed per sample.  Runs as the
    fused coupling kernel with every feature transformed (d_t = D)."""

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, tails=None,
                 tail_bound=1.0, num_blocks=2, use_residual_blocks=True, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 min_bin_width=rational_quadratic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=rational_quadratic.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=rational_quadratic.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        net = made_module.MADE(features=features, hidden_features=hidden_features,
                               context_features=context_features, num_blocks=num_blocks,
                               output_multiplier=self._output_dim_multiplier(),
                               use_residual_blocks=use_residual_blocks, random_mask=random_mask,
                               activation=activation, dropout_probability=dropout_probability,
                               use_batch_norm=use_batch_norm)
        super().__init__(net)
        self._all_features = None

    def _output_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        if self.tails is None:
            return self.num_bins * 3 + 1
        raise ValueError

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        N.require_device_f32("inputs", inputs, 2)
        batch, features = inputs.shape
        divisor = 0.0
        if hasattr(self.autoregressive_net, "hidden_features"):
            divisor = float(np.sqrt(self.autoregressive_net.hidden_features))
        if self.tails not in (None, "linear"):
            raise ValueError
        spec = ops.make_rqs_spec(self.num_bins, self.tails, tail_bound=self.tail_bound,
                                 min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                 min_derivative=self.min_derivative, wh_divisor=divisor)
        cache = self._all_features if isinstance(self._all_features, dict) else {}
        cols = cache.get((features, inputs.device))
        if cols is None:
            cols = torch.arange(features, device=inputs.device)
            cache[(features, inputs.device)] = cols
            self._all_features = cache
        params = autoregressive_params.reshape(batch, features * self._output_dim_multiplier())
        return ops.rqs_coupling(inputs, params, cols, spec, inverse=inverse)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)

    # the MADE's output layer inside the spline kernel (K13, csrc/made_output.hip): the forward pass and the
    # last pass of the inverse never form the [batch, features * multiplier] parameter tensor
    fuse_output_layer = os.environ.get("NFA_K13", "1") != "0"

    def _wh_divisor(self):
        net = self.autoregressive_net
        return float(np.sqrt(net.hidden_features)) if hasattr(net, "hidden_features") else 0.0

    def _output_layer_kernel(self, inputs, hidden, first_feature, inverse, out=None):
        """(outputs with the columns from `first_feature` on written, their logabsdet) from K13, or None."""
        net = self.autoregressive_net
        if not (self.fuse_output_layer and isinstance(net, made_module.MADE) and self.tails == "linear"
                and self.num_bins == 8 and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and not torch.is_grad_enabled() and hidden.shape[1] <= 256 and hidden.shape[1] % 4 == 0
                and inputs.shape[0] >= 1):
            return None
        final = net.final_layer
        key = (_cache.epoch(), first_feature) + tuple((t.data_ptr(), t._version) for t in (final.weight, final.bias, final.mask))
        cache = self.__dict__.setdefault("_made_output_cache", {})
        packed = cache.get(first_feature)
        if packed is None or packed[0] != key:
            if len(cache) > 4:
                cache.clear()
            packed = (key, ops.pack_made_output(net, self._output_dim_multiplier(), first_feature))
            cache[first_feature] = packed
        spec = ops.make_rqs_spec(self.num_bins, self.tails, tail_bound=self.tail_bound,
                                 min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                 min_derivative=self.min_derivative, wh_divisor=self._wh_divisor())
        return ops.made_output_spline(inputs, hidden, packed[1], spec, first_column=first_feature, inverse=inverse,
                                      out=out)

    def forward(self, inputs, context=None):
        net = self.autoregressive_net
        if (self.fuse_output_layer and isinstance(net, made_module.MADE) and inputs.dim() == 2 and inputs.is_cuda
                and inputs.dtype == torch.float32 and not torch.is_grad_enabled() and self.tails == "linear"
                and self.num_bins == 8):
            hidden = net.hidden(inputs, context)
            fused = self._output_layer_kernel(inputs, hidden, 0, inverse=False)
            if fused is not None:
                return fused
            return self._elementwise_forward(inputs, net.final_layer(hidden))
        return super().forward(inputs, context)

    # persistent kernel for the sequential features of the inverse (K12, csrc/made_inverse.hip)
    fuse_sequential_inverse = os.environ.get("NFA_K12", "1") != "0"

    def _sequential_kernel(self, inputs, context, sequential):
        net = self.autoregressive_net
        if not (self.fuse_sequential_inverse and context is None and sequential >= 1
                and isinstance(net, made_module.MADE) and net.activation is F.relu
                and not hasattr(net, "context_layer") and self.tails == "linear" and self.num_bins in (8, 10)
                and not any(isinstance(m, torch.nn.BatchNorm1d) for m in net.modules())):
            return None
        key = (_cache.epoch(), sequential) + tuple((p.data_ptr(), p._version) for p in net.parameters())
        cached = self.__dict__.get("_made_schedule_cache")
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_made_schedule(net, sequential, self._output_dim_multiplier()))
            self.__dict__["_made_schedule_cache"] = cached
        spec = ops.make_rqs_spec(self.num_bins, self.tails, tail_bound=self.tail_bound,
                                 min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                 min_derivative=self.min_derivative,
                                 wh_divisor=float(np.sqrt(net.hidden_features)) if hasattr(net, "hidden_features") else 0.0)
        hidden_features = net.initial_layer.weight.shape[0]
        return ops.made_rqs_inverse(inputs, cached[1], hidden_features, sequential, spec)

    def _inverse_column(self, column, params):
        """One feature: params [B, P]; the spline layer kernel with a single (transformed) feature."""
        out, lad = self._elementwise(column.reshape(-1, 1), params, inverse=True)
        return out.reshape(-1), lad


class _SiblingSplineAutoregressiveTransform(AutoregressiveTransform):
    """Shared part of the linear / quadratic / cubic autoregressive spline layers
    (autoregressive.py:196-401): MADE emits `_output_dim_multiplier()` logits per feature, the
    spline functional (K9) runs elementwise, logabsdet is summed per sample."""

    def _make_net(self, features, hidden_features, context_features, num_blocks, use_residual_blocks,
                  random_mask, activation, dropout_probability, use_batch_norm):
        return made_module.MADE(features=features, hidden_features=hidden_features,
                                context_features=context_features, num_blocks=num_blocks,
                                output_multiplier=self._output_dim_multiplier(),
                                use_residual_blocks=use_residual_blocks, random_mask=random_mask,
                                activation=activation, dropout_probability=dropout_probability,
                                use_batch_norm=use_batch_norm)

    def _wh_divisor(self):
        net = self.autoregressive_net
        return float(np.sqrt(net.hidden_features)) if hasattr(net, "hidden_features") else 0.0

    def _functional(self, inputs, params, inverse):
        """inputs [...], params [..., multiplier] -> (outputs, logabsdet) elementwise"""
        raise NotImplementedError()

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        # (inputs may be a subset of the features: the independent tail of the column-wise inverse)
        params = autoregressive_params.reshape(inputs.shape[0], inputs.shape[1], self._output_dim_multiplier())
        outputs, logabsdet = self._functional(inputs, params, inverse)
        return outputs, torchutils.sum_except_batch(logabsdet)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)

    def _inverse_column(self, column, params):
        return self._functional(column, params, True)


class MaskedPiecewiseLinearAutoregressiveTransform(_SiblingSplineAutoregressiveTransform):
    """autoregressive.py:196-246: piecewise-linear CDF on [0, 1] per feature."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False):
        self.num_bins = num_bins
        self.features = features
        super().__init__(self._make_net(features, hidden_features, context_features, num_blocks,
                                        use_residual_blocks, random_mask, activation, dropout_probability,
                                        use_batch_norm))

    def _output_dim_multiplier(self):
        return self.num_bins

    def _functional(self, inputs, params, inverse):
        return splines.linear_spline(inputs, params, inverse=inverse)


class MaskedPiecewiseQuadraticAutoregressiveTransform(_SiblingSplineAutoregressiveTransform):
    """autoregressive.py:249-334.  Only the width logits are divided by sqrt(hidden_features) when
    the net exposes it (:304-306; the height scaling is commented out in the reference)."""

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, num_blocks=2,
                 tails=None, tail_bound=1.0, use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False,
                 min_bin_width=rational_quadratic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=rational_quadratic.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=rational_quadratic.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        self.features = features
        super().__init__(self._make_net(features, hidden_features, context_features, num_blocks,
                                        use_residual_blocks, random_mask, activation, dropout_probability,
                                        use_batch_norm))

    def _output_dim_multiplier(self):
        return self.num_bins * 2 - 1 if self.tails == "linear" else self.num_bins * 2 + 1

    def _functional(self, inputs, params, inverse):
        K = self.num_bins
        uw, uh = params[..., :K], params[..., K:]
        if self._wh_divisor():
            uw = uw / self._wh_divisor()
        kw = dict(inverse=inverse, min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height)
        if self.tails is None:
            return splines.quadratic_spline(inputs, uw, uh, **kw)
        if self.tails == "linear":
            return splines.unconstrained_quadratic_spline(inputs, uw, uh, tails=self.tails,
                                                          tail_bound=self.tail_bound, **kw)
        raise ValueError


class MaskedPiecewiseCubicAutoregressiveTransform(_SiblingSplineAutoregressiveTransform):
    """autoregressive.py:337-401: cubic spline on [0, 1] per feature."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False):
        self.num_bins = num_bins
        self.features = features
        super().__init__(self._make_net(features, hidden_features, context_features, num_blocks,
                                        use_residual_blocks, random_mask, activation, dropout_probability,
                                        use_batch_norm))

    def _output_dim_multiplier(self):
        return self.num_bins * 2 + 2

    def _functional(self, inputs, params, inverse):
        K = self.num_bins
        uw, uh = params[..., :K], params[..., K:2 * K]
        if self._wh_divisor():
            uw, uh = uw / self._wh_divisor(), uh / self._wh_divisor()
        return splines.cubic_spline(inputs, uw, uh, params[..., 2 * K:2 * K + 1], params[..., 2 * K + 1:2 * K + 2],
                                    inverse=inverse)
