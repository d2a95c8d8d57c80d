"""Coupling layers on the MI355X kernels.

Drop-in for the ★ classes of nflows/transforms/coupling.py: same constructor signatures, buffer
names (`identity_features`, `transform_features`) and attributes, so a reference `state_dict`
loads unchanged.  Per layer the host does exactly three things:

  1. gather the identity half (one small index_select -- the conditioner's input),
  2. run the conditioner (any nn.Module taking (identity_split, context); PyTorch-ROCm GEMMs),
  3. launch ONE fused HIP kernel (ops.rqs_coupling / ops.affine_coupling) that does the split,
     the elementwise transform, the per-sample logabsdet sum and the scatter, optionally with
     the neighbouring column permutation folded in.

Only 2-D float32 inputs on a HIP device are implemented; 4-D image inputs
(coupling.py:280-285) and the other piecewise families are out of scope and raise.
"""
import os
import warnings

import numpy as np
import torch
from torch.nn.functional import softplus

from .. import _native as N
from .. import ops
from .base import Transform
from .splines import rational_quadratic


class CouplingTransform(Transform):
    """Base class: mask bookkeeping, conditioner call and the fused-kernel hand-off.

    mask[i] > 0: feature i is transformed; mask[i] <= 0: passed through (coupling.py:25-63)."""

    supports_fused_permutation = True

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None):
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError("Mask must be a 1-dim tensor.")
        if mask.numel() <= 0:
            raise ValueError("Mask can't be empty.")
        super().__init__()
        self.features = len(mask)
        columns = torch.arange(self.features)
        self.register_buffer("identity_features", columns.masked_select(mask <= 0))
        self.register_buffer("transform_features", columns.masked_select(mask > 0))
        assert self.num_identity_features + self.num_transform_features == self.features

        self.transform_net = transform_net_create_fn(
            self.num_identity_features,
            self.num_transform_features * self._transform_dim_multiplier(),
        )
        if unconditional_transform is None:
            self.unconditional_transform = None
        else:
            self.unconditional_transform = unconditional_transform(features=self.num_identity_features)

    @property
    def num_identity_features(self):
        return len(self.identity_features)

    @property
    def num_transform_features(self):
        return len(self.transform_features)

    # ------------------------------------------------------------------ shared plumbing
    def _check_inputs(self, inputs):
        if inputs.dim() not in [2, 4]:
            raise ValueError("Inputs must be a 2D or a 4D tensor.")
        if inputs.shape[1] != self.features:
            raise ValueError("Expected features = {}, got {}.".format(self.features, inputs.shape[1]))
        if inputs.dim() == 4:
            raise NotImplementedError(
                "nflows_amd: 4-D (image) coupling inputs are outside the MI355X hot path")
        N.require_device_f32("inputs", inputs, 2)

    def _identity_columns(self, perm):
        """identity_features seen through a fused permutation, cached per permutation tensor."""
        if perm is None:
            return self.identity_features
        key = (perm.data_ptr(), perm._version, self.identity_features.data_ptr(),
               self.identity_features._version)
        cached = getattr(self, "_id_cols_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, perm[self.identity_features])
            self._id_cols_cache = cached
        return cached[1]

    def forward(self, inputs, context=None, in_perm=None, logabsdet_accumulator=None):
        """outputs[:, identity] = inputs[:, identity]; outputs[:, transform] = f(inputs[:, transform];
        net(inputs[:, identity])) (coupling.py:73-100).
        `in_perm`: treat inputs[:, in_perm] as the layer input (a preceding Permutation, fused).
        `logabsdet_accumulator`: a [batch] running total the layer's logabsdet is added to in the
        kernel (CompositeTransform's `total_logabsdet +=`); it is then also the returned tensor."""
        self._check_inputs(inputs)
        if self.unconditional_transform is None:
            whole = self._whole_layer(inputs, context, False, in_perm, None, logabsdet_accumulator)
            if whole is not None:
                return whole
        identity_split = inputs.index_select(1, self._identity_columns(in_perm))
        outputs, logabsdet = self._condition_and_transform(
            inputs, identity_split, context, inverse=False, in_perm=in_perm,
            accumulate_into=logabsdet_accumulator)
        if self.unconditional_transform is not None:
            identity_split, logabsdet_identity = self.unconditional_transform(identity_split, context)
            if logabsdet_accumulator is not None:
                logabsdet += logabsdet_identity
            else:
                logabsdet = logabsdet + logabsdet_identity
            outputs.index_copy_(1, self.identity_features, identity_split)
        return outputs, logabsdet

    def inverse(self, inputs, context=None, out_scatter=None, logabsdet_accumulator=None):
        """Inverse pass (coupling.py:102-130).  `out_scatter`: store layer column c at
        outputs[:, out_scatter[c]] (a following Permutation.inverse, fused)."""
        self._check_inputs(inputs)
        if self.unconditional_transform is None:
            whole = self._whole_layer(inputs, context, True, None, out_scatter, logabsdet_accumulator)
            if whole is not None:
                return whole
        identity_split = inputs.index_select(1, self.identity_features)
        logabsdet_identity = None
        if self.unconditional_transform is not None:
            identity_split, logabsdet_identity = self.unconditional_transform.inverse(identity_split, context)
        outputs, logabsdet = self._condition_and_transform(
            inputs, identity_split, context, inverse=True, out_scatter=out_scatter,
            accumulate_into=logabsdet_accumulator)
        if self.unconditional_transform is not None:
            if logabsdet_accumulator is not None:
                logabsdet += logabsdet_identity
            else:
                logabsdet = logabsdet + logabsdet_identity
            outputs.index_copy_(1, self._identity_columns(out_scatter), identity_split)
        return outputs, logabsdet

    def _whole_layer(self, inputs, context, inverse, in_perm, out_scatter, accumulate_into):
        """Hook for a kernel that contains the conditioner as well; None = not applicable."""
        return None

    def _condition_and_transform(self, inputs, identity_split, context, inverse, in_perm=None,
                                 out_scatter=None, accumulate_into=None):
        """Conditioner call + fused layer kernel.  Subclasses may replace the pair by a kernel
        that also contains the conditioner's last layer."""
        transform_params = self.transform_net(identity_split, context)
        return self._fused_layer(inputs, transform_params, inverse, in_perm=in_perm,
                                 out_scatter=out_scatter, accumulate_into=accumulate_into)

    def _transform_dim_multiplier(self):
        """Number of conditioner outputs per transformed feature."""
        raise NotImplementedError()

    def _fused_layer(self, inputs, transform_params, inverse, in_perm=None, out_scatter=None,
                     accumulate_into=None):
        raise NotImplementedError()


class AffineCouplingTransform(CouplingTransform):
    """y = x * scale + shift on the transformed half (RealNVP); coupling.py:212-252.

    `scale_activation` may be any callable.  The two predefined ones are evaluated inside the
    kernel; any other callable is evaluated with PyTorch and the kernel receives the scale."""

    DEFAULT_SCALE_ACTIVATION = lambda x: torch.sigmoid(x + 2) + 1e-3  # noqa: E731
    GENERAL_SCALE_ACTIVATION = lambda x: (softplus(x) + 1e-3).clamp(0, 3)  # noqa: E731

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None,
                 scale_activation=DEFAULT_SCALE_ACTIVATION):
        self.scale_activation = scale_activation
        super().__init__(mask, transform_net_create_fn, unconditional_transform)

    def _transform_dim_multiplier(self):
        return 2

    def _activation_code(self):
        if self.scale_activation is AffineCouplingTransform.DEFAULT_SCALE_ACTIVATION:
            return N.SCALE_DEFAULT
        if self.scale_activation is AffineCouplingTransform.GENERAL_SCALE_ACTIVATION:
            return N.SCALE_GENERAL
        return N.SCALE_GIVEN

    def _fused_layer(self, inputs, transform_params, inverse, in_perm=None, out_scatter=None,
                     accumulate_into=None):
        code = self._activation_code()
        scale = None
        if code == N.SCALE_GIVEN:
            scale = self.scale_activation(transform_params[:, self.num_transform_features:])
        return ops.affine_coupling(inputs, transform_params, self.transform_features, code,
                                   inverse=inverse, scale=scale, in_perm=in_perm,
                                   out_scatter=out_scatter, accumulate_into=accumulate_into)


class AdditiveCouplingTransform(AffineCouplingTransform):
    """y = x + shift (NICE); logabsdet is exactly zero; coupling.py:255-269."""

    def _transform_dim_multiplier(self):
        return 1

    def _fused_layer(self, inputs, transform_params, inverse, in_perm=None, out_scatter=None,
                     accumulate_into=None):
        return ops.affine_coupling(inputs, transform_params, self.transform_features,
                                   N.SCALE_ADDITIVE, inverse=inverse, in_perm=in_perm,
                                   out_scatter=out_scatter, accumulate_into=accumulate_into)


class PiecewiseRationalQuadraticCouplingTransform(CouplingTransform):
    """Neural-spline-flow coupling layer; coupling.py:502-582.

    Conditioner output per transformed feature: K width logits, K height logits and K-1
    (tails="linear") or K+1 (tails=None) derivative logits.  Width/height logits are divided by
    sqrt(hidden_features) of the conditioner when it exposes `hidden_features` /
    `hidden_channels` (coupling.py:554-559) -- done on the fly inside the kernel, the conditioner
    output tensor itself is left untouched."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None,
                 min_bin_width=rational_quadratic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=rational_quadratic.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=rational_quadratic.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        if apply_unconditional_transform:
            from .nonlinearities import PiecewiseRationalQuadraticCDF

            def unconditional_transform(features):
                return PiecewiseRationalQuadraticCDF(
                    shape=[features] + (img_shape if img_shape else []), num_bins=num_bins,
                    tails=tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                    min_bin_height=min_bin_height, min_derivative=min_derivative)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        return self.num_bins * 3 + 1

    def _spec(self):
        if hasattr(self.transform_net, "hidden_features"):
            divisor = float(np.sqrt(self.transform_net.hidden_features))
        elif hasattr(self.transform_net, "hidden_channels"):
            divisor = float(np.sqrt(self.transform_net.hidden_channels))
        else:
            warnings.warn("Inputs to the softmax are not scaled down: initialization might be bad.")
            divisor = 0.0
        if self.tails is not None and self.tails != "linear":
            raise RuntimeError("{} tails are not implemented.".format(self.tails))
        return ops.make_rqs_spec(self.num_bins, self.tails, tail_bound=self.tail_bound,
                                 min_bin_width=self.min_bin_width,
                                 min_bin_height=self.min_bin_height,
                                 min_derivative=self.min_derivative, wh_divisor=divisor)

    # K7: fold the conditioner's final Linear into the spline kernel (no [B, d_t*P] round trip
    # through HBM).  Class-level switch for A/B measurements.
    fuse_final_linear = True
    # GEMM engine of K7: "bf16x3" = split-bf16 operands on the bf16 matrix pipe (fp32-accurate,
    # default), "f32" = v_mfma_f32_32x32x2_f32
    final_linear_engine = os.environ.get("NFA_K7_ENGINE", "bf16x3")

    # K8: the whole ResidualNet conditioner inside the spline kernel (class-level switch)
    fuse_conditioner = os.environ.get("NFA_K8", "1") != "0"

    def _resnet_eligible(self, context):
        net = self.transform_net
        from ..nn.nets.resnet import ResidualNet
        return (self.fuse_conditioner and self.fuse_final_linear and not torch.is_grad_enabled()
                and context is None and type(net) is ResidualNet and net.context_features is None
                and net.hidden_features == 128 and self.tails == "linear" and self.num_bins == 8
                and self.num_identity_features <= 32 and self.num_transform_features % 4 == 0
                and self.num_transform_features <= 64 and self.features <= 128
                and all(b.activation is torch.nn.functional.relu and not b.use_batch_norm
                        and (not b.training or b.dropout.p == 0.0) for b in net.blocks))

    # experiment switch: fold log2(e) into the width / height logits as well (one v_exp_f32 per
    # softmax numerator)
    resnet_log2e = os.environ.get("NFA_K8_LOG2E", "0") != "0"

    def _packed_resnet(self):
        net = self.transform_net
        key = tuple((p.data_ptr(), p._version) for p in net.parameters()) + (self.resnet_log2e,)
        cached = getattr(self, "_packed_resnet_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_resnet_conditioner(net, self.num_transform_features,
                                                       self._transform_dim_multiplier(),
                                                       log2e=self.resnet_log2e))
            self._packed_resnet_cache = cached
        return cached[1]

    def _layer_tables(self, in_perm, out_scatter):
        key = tuple(None if t is None else (t.data_ptr(), t._version) for t in
                    (in_perm, out_scatter, self.transform_features, self.identity_features))
        cache = self.__dict__.setdefault("_layer_tables_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) > 8:
                cache.clear()
            hit = ops.coupling_layer_tables(self.features, self.transform_features, self.identity_features,
                                            in_perm, out_scatter)
            cache[key] = hit
        return hit

    def _whole_layer(self, inputs, context, inverse, in_perm, out_scatter, accumulate_into):
        B = inputs.shape[0]
        if B < 128 or self.features % 4 != 0 or not self._resnet_eligible(context):
            return None
        wp, bp = self._packed_resnet()
        tables = self._layer_tables(in_perm, out_scatter)
        nb = len(self.transform_net.blocks)
        dt, di = self.num_transform_features, self.num_identity_features
        spec = self._spec()
        full = (B // 128) * 128
        if full == B:
            return ops.rqs_coupling_resnet(inputs, wp, bp, tables, dt, di, nb, spec, inverse, accumulate_into,
                                           log2e=self.resnet_log2e)
        # ragged batch: full 128-row blocks here, the tail through the PyTorch conditioner + K1
        acc_head = None if accumulate_into is None else accumulate_into[:full]
        acc_tail = None if accumulate_into is None else accumulate_into[full:]
        head = ops.rqs_coupling_resnet(inputs[:full], wp, bp, tables, dt, di, nb, spec, inverse, acc_head,
                                       log2e=self.resnet_log2e)
        if head is None:
            return None
        tail_in = inputs[full:]
        cols = self._identity_columns(in_perm) if not inverse else self.identity_features
        params = self.transform_net(tail_in.index_select(1, cols), None)
        tail = self._fused_layer(tail_in, params, inverse, in_perm=in_perm, out_scatter=out_scatter,
                                 accumulate_into=acc_tail)
        outputs = torch.cat((head[0], tail[0]), dim=0)
        logabsdet = accumulate_into if accumulate_into is not None else torch.cat((head[1], tail[1]), dim=0)
        return outputs, logabsdet

    def _packed_final_linear(self, layer):
        split = self.final_linear_engine == "bf16x3"
        key = (layer.weight.data_ptr(), layer.weight._version, layer.bias.data_ptr(), layer.bias._version,
               split)
        cached = getattr(self, "_packed_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_final_linear(layer.weight, layer.bias, self.num_transform_features,
                                                 self._transform_dim_multiplier(), split_bf16=split))
            self._packed_cache = cached
        return cached[1]

    def _condition_and_transform(self, inputs, identity_split, context, inverse, in_perm=None,
                                 out_scatter=None, accumulate_into=None):
        net = self.transform_net
        final = getattr(net, "final_layer", None)
        dt = self.num_transform_features
        eligible = (self.fuse_final_linear and not torch.is_grad_enabled() and self.tails == "linear"
                    and self.num_bins == 8 and hasattr(net, "hidden") and isinstance(final, torch.nn.Linear)
                    and final.bias is not None and final.in_features == 128
                    and getattr(net, "hidden_features", None) == 128 and dt % 4 == 0 and dt <= 64
                    and self.features <= 128 and inputs.shape[0] >= 128)
        if not eligible:
            return super()._condition_and_transform(inputs, identity_split, context, inverse, in_perm,
                                                    out_scatter, accumulate_into)
        hidden = net.hidden(identity_split, context)
        wp, bp = self._packed_final_linear(final)
        B = inputs.shape[0]
        rows = 128 if wp.dtype == torch.bfloat16 else 32  # rows per workgroup / per wave
        full = (B // rows) * rows
        spec = self._spec()
        if full == B:
            res = ops.rqs_coupling_fused_linear(inputs, hidden, wp, bp, self.transform_features, spec,
                                                inverse, in_perm, out_scatter, accumulate_into)
            if res is not None:
                return res
            return self._fused_layer(inputs, final(hidden), inverse, in_perm=in_perm,
                                     out_scatter=out_scatter, accumulate_into=accumulate_into)
        # ragged batch: the full row blocks through K7, the tail through GEMM + K1
        acc_head = None if accumulate_into is None else accumulate_into[:full]
        acc_tail = None if accumulate_into is None else accumulate_into[full:]
        head = ops.rqs_coupling_fused_linear(inputs[:full], hidden[:full], wp, bp, self.transform_features,
                                             spec, inverse, in_perm, out_scatter, acc_head)
        if head is None:
            return self._fused_layer(inputs, final(hidden), inverse, in_perm=in_perm,
                                     out_scatter=out_scatter, accumulate_into=accumulate_into)
        tail = self._fused_layer(inputs[full:], final(hidden[full:]), inverse, in_perm=in_perm,
                                 out_scatter=out_scatter, accumulate_into=acc_tail)
        outputs = torch.cat((head[0], tail[0]), dim=0)
        logabsdet = accumulate_into if accumulate_into is not None else torch.cat((head[1], tail[1]), dim=0)
        return outputs, logabsdet

    def _fused_layer(self, inputs, transform_params, inverse, in_perm=None, out_scatter=None,
                     accumulate_into=None):
        return ops.rqs_coupling(inputs, transform_params, self.transform_features, self._spec(),
                                inverse=inverse, in_perm=in_perm, out_scatter=out_scatter,
                                accumulate_into=accumulate_into)
