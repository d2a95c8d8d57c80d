"""Coupling layers on the MI355X kernels.

Drop-in for the ★ classes of nflows/transforms/coupling.py: same constructor signatures, buffer
names (`identity_features`, `transform_features`) and attributes, so a reference `state_dict`
loads unchanged.  Per layer the host does exactly three things:

  1. gather the identity half (one small index_select -- the conditioner's input),
  2. run the conditioner (any nn.Module taking (identity_split, context); PyTorch-ROCm GEMMs),
  3. launch ONE fused HIP kernel (ops.rqs_coupling / ops.affine_coupling) that does the split,
     the elementwise transform, the per-sample logabsdet sum and the scatter, optionally with
     the neighbouring column permutation folded in.

The fused kernels serve 2-D float32 inputs.  4-D image inputs (coupling.py:280-285) and the
linear / quadratic piecewise families take the generic path of `CouplingTransform._generic`: the
reference's own sequence (split, conditioner, elementwise spline functional, sum, merge) with the
functional running as one HIP kernel (K5 / K9) on strided views.
"""
import os
import warnings

import numpy as np
import torch
from torch.nn.functional import softplus

from .. import _cache
from .. import autograd as _autograd
from .. import _native as N
from .. import ops
from ..utils import torchutils
from .base import Transform
from . import splines
from .splines import rational_quadratic


def _held_parameters(net):
    """[(the `_parameters` / `_buffers` dict of a leaf module, name, tensor)] of every parameter and buffer of `net`
    (buffers: the running statistics of batch-norm layers, folded into the packed weights).  The parameters are re-classed
    as _cache.WatchedParameter on the way: their `.data` then reports its use (writes through it reach no version counter)."""
    params = [(m._parameters, name, p) for m in net.modules() for name, p in m._parameters.items() if p is not None]
    for _, _, p in params:
        _cache.watch(p)
    return params + [(m._buffers, name, b) for m in net.modules() for name, b in m._buffers.items() if b is not None]


class StalePackedWeights(RuntimeError):
    """NFA_VERIFY_WEIGHTS: a conditioner's weights changed without any of the signs the packed-weight caches look at --
    version counters, storage pointers, registrations, and (round 6) the `.data` of its Parameters (_cache.WatchedParameter):
    a write into raw storage, through a view's `.data`, by a foreign kernel.  The fused kernels have been using the OLD
    weights; `nflows_amd.invalidate_packed_weights()` after such writes is the remedy."""


# How the packed weights follow the parameters (round 6).
#   visible changes   in-place updates (version counters), rebound storage (data_ptr), (re)registered objects,
#                     load_state_dict / .to() (cache epoch): part of every weight key, the packs are rebuilt.
#   `.data`           reading or assigning the `.data` of a conditioner Parameter advances _cache.data_epoch(); the next
#                     use of a weight key compares the parameters' CONTENTS with the checksum recorded when the key was new
#                     (one synchronising comparison per layer, or one for a whole run) and, on a difference, advances the
#                     layer's `_data_salt` -- part of the key: repacked from the new values, nothing raised, and no call
#                     without such an event pays anything.
#   everything else   NFA_VERIFY_WEIGHTS=N (default 256; 0 = off, 1 = every use): every N-th use of a key repeats the
#                     comparison, staggered over the layers, and raises StalePackedWeights; also on every plan-cache miss of
#                     a run (_Run.verify_before_packing).  Skipped while a stream is being captured into a HIP graph.
VERIFY_WEIGHTS_EVERY = int(os.environ.get("NFA_VERIFY_WEIGHTS", "256") or 0)

_CHECKSUM_LAYOUTS = {}


def _checksum(params):
    """int64 [len(params)] on the parameters' device: per parameter the sum over its elements of (bit pattern as int32) x (an
    odd multiplier that depends on the position), modulo 2^64 -- exact, sensitive to the sign and the position of every
    element (round 5's fp32 norms were blind to a sign flip of a small entry), NaNs included (a bit pattern like any
    other).  fp32 parameters on one device: one concatenation, one multiply, one segmented sum; no synchronisation."""
    with torch.no_grad():
        params = [p.detach() for p in params]
        if not params:
            return torch.zeros(0, dtype=torch.int64)
        if not (all(p.dtype == torch.float32 for p in params) and len({p.device for p in params}) == 1):
            return torch.stack([(p.double().reshape(-1).view(torch.int64) >> 7).sum().cpu() + p.numel() for p in params])
        dev = params[0].device
        sizes = tuple(p.numel() for p in params)
        layout = _CHECKSUM_LAYOUTS.get((dev, sizes))
        if layout is None:
            if len(_CHECKSUM_LAYOUTS) > 64:
                _CHECKSUM_LAYOUTS.clear()
            sz = torch.tensor(sizes, device=dev)
            seg = torch.repeat_interleave(torch.arange(len(sizes), device=dev), sz)
            # (the position INSIDE the parameter: a parameter's checksum does not depend on what it is concatenated with)
            at = torch.arange(sum(sizes), dtype=torch.int64, device=dev) - (torch.cumsum(sz, 0) - sz)[seg]
            mult = ((at * 2654435761) % 2147483647) * 2 + 1
            layout = _CHECKSUM_LAYOUTS[(dev, sizes)] = (mult, seg)
        mult, seg = layout
        bits = torch.cat([p.reshape(-1) for p in params]).view(torch.int32).to(torch.int64)
        return torch.zeros(len(sizes), dtype=torch.int64, device=dev).index_add_(0, seg, bits * mult)


def _same_checksum(a, b):
    return a.shape == b.shape and bool(torch.equal(a, b))    # (synchronises)


def _capturing(params):
    return bool(params) and params[0].is_cuda and torch.cuda.is_current_stream_capturing()


def _stale(owner):
    return StalePackedWeights(
        "the parameters of %s changed through a write the packed-weight caches cannot see (raw storage, a view's `.data`, a "
        "foreign kernel): call nflows_amd.invalidate_packed_weights() after such writes" % type(owner).__name__)


def _track_contents(owner, slot, key, params, periodic=True, compare_now=False):
    """The record `owner.__dict__[slot]` = [visible key, checksum of the contents when the key was new, use count, data epoch]
    and what is done with it: a new key records (no synchronisation); a `.data` event since the record compares and, on a
    difference, advances `_data_salt` (repack, no exception); `periodic` counts uses and compares every
    NFA_VERIFY_WEIGHTS-th (`compare_now`: at once) -- a difference there is a write nothing announced: StalePackedWeights."""
    de = _cache.data_epoch()
    state = owner.__dict__.get(slot)
    if state is None or state[0] != key:
        # (the count starts at a per-layer offset: the layers of a flow are used once per call each, and 32 synchronising
        #  comparisons landing in ONE call were a 3 ms spike every 256th step)
        owner.__dict__[slot] = [key, _checksum(params), (id(owner) >> 6) % max(1, VERIFY_WEIGHTS_EVERY), de]
        return
    if state[3] != de:
        if _capturing(params):
            return
        state[3] = de
        now = _checksum(params)
        if not _same_checksum(now, state[1]):
            state[1] = now
            owner.__dict__["_data_salt"] = owner.__dict__.get("_data_salt", 0) + 1
        return
    if not VERIFY_WEIGHTS_EVERY or _capturing(params):
        return
    if periodic:
        state[2] += 1
    if (compare_now or (periodic and state[2] % VERIFY_WEIGHTS_EVERY == 0)) and not _same_checksum(_checksum(params), state[1]):
        owner.__dict__.pop(slot, None)
        raise _stale(owner)


def _weights_key(owner, net):
    """Fingerprint of a conditioner's weights for the packed-weight caches: storage pointers and version counters of its
    parameters, read from a list made once per cache epoch (walking `net.parameters()` costs ~10 us per call and layer: a
    millisecond per `log_prob` of a 32-layer flow), and the layer's `_data_salt` (see above: writes through `.data`).
    In-place updates advance the counters; moves, `load_state_dict` and (re)registered Parameter objects advance the epoch
    (_cache.py).  Swaps that bypass the registration hooks -- `torch.func.functional_call`,
    `stateless._reparametrize_module`, a direct `module._parameters[name] = other` -- are caught by checking on every
    call that each held object still IS the entry of its module's `_parameters` dict."""
    epoch = _cache.epoch()
    held = owner.__dict__.get("_weights_list")
    if held is None or held[0] != epoch or held[1] is not net or not _cache.HOOKED:
        held = (epoch, net, _held_parameters(net))
        owner.__dict__["_weights_list"] = held
    else:
        for d, name, p in held[2]:
            if d.get(name) is not p:
                held = (epoch, net, _held_parameters(net))
                owner.__dict__["_weights_list"] = held
                break
    # (data_ptr as well: `p.data = other` rebinds the storage without touching the counter)
    key = (epoch,) + tuple([(p.data_ptr(), p._version) for _, _, p in held[2]])
    _track_contents(owner, "_contents", key, [p for _, _, p in held[2]])
    return key + (owner.__dict__.get("_data_salt", 0),)


class CouplingTransform(Transform):
    """Base class: mask bookkeeping, conditioner call and the fused-kernel hand-off.

    mask[i] > 0: feature i is transformed; mask[i] <= 0: passed through (coupling.py:25-63)."""

    supports_fused_permutation = True   # 2-D inputs go through one fused kernel per layer
    supports_image_inputs = False       # 4-D inputs allowed (generic path)

    # The reference's extension points (coupling.py:132-136, :234-252, :263-269, :279-296): a subclass DEFINED OUTSIDE this
    # module that overrides one of them means "run the reference's sequence with MY function".  The fused kernels would
    # silently bypass such an override (they contain the library's arithmetic), so such a class -- and its subclasses --
    # takes `_reference_sequence`: the reference's own order of calls (coupling.py:73-130) on device tensors, its hooks
    # called where the reference calls them; it joins no fused run and takes no fused permutation.
    _REFERENCE_HOOKS = ("_coupling_transform_forward", "_coupling_transform_inverse", "_piecewise_cdf", "_scale_and_shift")
    _HOOK_VERDICTS = {}

    @property
    def _user_hooks(self):
        """True when one of the reference's extension points resolves -- through the MRO: a mixin counts, and so does a
        function assigned to the class or the instance after its creation (round 5 looked at the new class's own
        `__dict__` at class-creation time only) -- to a function defined outside this module.  Four attribute lookups per
        call; the verdict is kept per class and per set of resolved functions."""
        cls = type(self)
        hooks = CouplingTransform._REFERENCE_HOOKS
        if any(h in self.__dict__ for h in hooks):
            return True
        fns = tuple([getattr(cls, h, None) for h in hooks])
        held = CouplingTransform._HOOK_VERDICTS.get(cls)
        if held is None or any(a is not b for a, b in zip(held[0], fns)):
            mine = any(f is not None and getattr(f, "__module__", __name__) != __name__ for f in fns)
            held = CouplingTransform._HOOK_VERDICTS[cls] = (fns, mine)
        return held[1]

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
        if inputs.dim() == 4 and not self.supports_image_inputs:
            raise NotImplementedError(
                "nflows_amd: 4-D (image) inputs are implemented for the piecewise spline couplings only")
        # float64: the generic sequence on the device (torch gathers, the float64 functional kernel)
        N.require_device_real("inputs", inputs, inputs.dtype, inputs.dim())

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
        if self._user_hooks:
            return self._reference_sequence(inputs, context, False, logabsdet_accumulator)
        if inputs.dim() == 4 or inputs.dtype == torch.float64 or not self.supports_fused_permutation:
            return self._generic(inputs, context, False, logabsdet_accumulator)
        # (with an unconditional transform -- coupling.py:90-94 -- the whole-layer kernel still does the conditioned half: the
        #  conditioner sees the identity features as they come in, they are transformed afterwards)
        whole = self._whole_layer(inputs, context, False, in_perm, None, logabsdet_accumulator)
        if whole is not None and self.unconditional_transform is None:
            return whole
        identity_split = _autograd.select_columns(inputs, self._identity_columns(in_perm))
        if whole is not None:
            outputs, logabsdet = whole
        else:
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
        if self._user_hooks:
            return self._reference_sequence(inputs, context, True, logabsdet_accumulator)
        if inputs.dim() == 4 or inputs.dtype == torch.float64 or not self.supports_fused_permutation:
            return self._generic(inputs, context, True, logabsdet_accumulator)
        if self.unconditional_transform is None:
            whole = self._whole_layer(inputs, context, True, None, out_scatter, logabsdet_accumulator)
            if whole is not None:
                return whole
        identity_split = _autograd.select_columns(inputs, self.identity_features)
        logabsdet_identity = None
        if self.unconditional_transform is not None:
            identity_split, logabsdet_identity = self.unconditional_transform.inverse(identity_split, context)
            if not torch.is_grad_enabled():
                # (coupling.py:114-118: the conditioner sees the identity features AFTER the unconditional inverse -- the
                #  whole-layer kernel on the rows with those columns replaced; they pass through it unchanged)
                moved = inputs.clone()
                moved.index_copy_(1, self.identity_features, identity_split)
                whole = self._whole_layer(moved, context, True, None, out_scatter, logabsdet_accumulator)
                if whole is not None:
                    outputs, logabsdet = whole
                    if logabsdet_accumulator is not None:
                        logabsdet += logabsdet_identity
                    else:
                        logabsdet = logabsdet + logabsdet_identity
                    return outputs, logabsdet
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

    def _generic(self, inputs, context, inverse, logabsdet_accumulator):
        """The reference's sequence (coupling.py:73-130) for inputs [B, C] or [B, C, H, W]: split on
        dim 1, conditioner on the identity part, `_elementwise` on the other, merge."""
        identity_split = inputs.index_select(1, self.identity_features)
        transform_split = inputs.index_select(1, self.transform_features)
        logabsdet = None
        if inverse and self.unconditional_transform is not None:
            identity_split, logabsdet = self.unconditional_transform.inverse(identity_split, context)
        transform_params = self.transform_net(identity_split, context)
        if inputs.dim() == 4:
            b, c, h, w = transform_split.shape
            transform_params = transform_params.reshape(b, c, -1, h, w).permute(0, 1, 3, 4, 2)
        else:
            b, d = transform_split.shape
            transform_params = transform_params.reshape(b, d, -1)
        transform_split, lad_elementwise = self._elementwise(transform_split, transform_params, inverse)
        lad_split = torchutils.sum_except_batch(lad_elementwise)
        logabsdet = lad_split if logabsdet is None else logabsdet + lad_split
        if not inverse and self.unconditional_transform is not None:
            identity_split, lad_identity = self.unconditional_transform(identity_split, context)
            logabsdet = logabsdet + lad_identity
        outputs = torch.empty_like(inputs)
        outputs.index_copy_(1, self.identity_features, identity_split)
        outputs.index_copy_(1, self.transform_features, transform_split)
        if logabsdet_accumulator is not None:
            logabsdet_accumulator += logabsdet
            logabsdet = logabsdet_accumulator
        return outputs, logabsdet

    def _reference_sequence(self, inputs, context, inverse, logabsdet_accumulator):
        """coupling.py:73-130 call for call, for classes whose hooks a user overrode (`_user_hooks`): split, (inverse:
        unconditional transform first), conditioner, `_coupling_transform_forward / _inverse(transform_split,
        transform_params)` -- the user's, or the library's default, which in turn calls the user's `_piecewise_cdf` /
        `_scale_and_shift` --, (forward: unconditional transform), merge."""
        identity_split = inputs.index_select(1, self.identity_features)
        transform_split = inputs.index_select(1, self.transform_features)
        logabsdet = None
        if inverse and self.unconditional_transform is not None:
            identity_split, logabsdet = self.unconditional_transform.inverse(identity_split, context)
        transform_params = self.transform_net(identity_split, context)
        hook = self._coupling_transform_inverse if inverse else self._coupling_transform_forward
        transform_split, lad_split = hook(transform_split, transform_params)
        logabsdet = lad_split if logabsdet is None else logabsdet + lad_split
        if not inverse and self.unconditional_transform is not None:
            identity_split, lad_identity = self.unconditional_transform(identity_split, context)
            logabsdet = logabsdet + lad_identity
        outputs = torch.empty_like(inputs)
        outputs.index_copy_(1, self.identity_features, identity_split)
        outputs.index_copy_(1, self.transform_features, transform_split)
        if logabsdet_accumulator is not None:
            logabsdet_accumulator += logabsdet
            logabsdet = logabsdet_accumulator
        return outputs, logabsdet

    def _coupling_transform_forward(self, inputs, transform_params):
        """coupling.py:132-134 (abstract in the reference's base class)"""
        raise NotImplementedError()

    def _coupling_transform_inverse(self, inputs, transform_params):
        """coupling.py:135-136"""
        raise NotImplementedError()

    def _elementwise(self, inputs, transform_params, inverse):
        """Elementwise transform of the transformed part given per-element parameters
        [..., params]; returns (outputs, logabsdet) of the inputs' shape (generic path only)."""
        raise NotImplementedError("nflows_amd: {} has no generic path".format(type(self).__name__))

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

    supports_image_inputs = True

    def _transform_dim_multiplier(self):
        return 2

    # The reference's hooks (coupling.py:234-252), for subclasses that override one of them (`_user_hooks`: the layer then
    # runs `_reference_sequence`, tensor operations on the device); the library's own layers go through K2 / K11.
    def _scale_and_shift(self, transform_params):
        dt = self.num_transform_features
        return self.scale_activation(transform_params[:, dt:, ...]), transform_params[:, :dt, ...]

    def _coupling_transform_forward(self, inputs, transform_params):
        scale, shift = self._scale_and_shift(transform_params)
        return inputs * scale + shift, torchutils.sum_except_batch(torch.log(scale), num_batch_dims=1)

    def _coupling_transform_inverse(self, inputs, transform_params):
        scale, shift = self._scale_and_shift(transform_params)
        return (inputs - shift) / scale, -torchutils.sum_except_batch(torch.log(scale), num_batch_dims=1)

    def _generic(self, inputs, context, inverse, logabsdet_accumulator):
        """Image inputs [B, C, H, W] (coupling.py:212-252 applies the same expressions to any rank;
        the conditioner's channels are [shift block | scale block]): every pixel is a row of C
        features for the layer kernel, the log-determinant is summed over the pixels afterwards."""
        if inputs.dim() != 4:
            return self._generic_float64(inputs, context, inverse, logabsdet_accumulator)
        b, c, h, w = inputs.shape
        identity_split = inputs.index_select(1, self.identity_features)
        logabsdet = None
        rows = inputs
        if self.unconditional_transform is not None:
            if inverse:
                identity_split, logabsdet = self.unconditional_transform.inverse(identity_split, context)
            rows = inputs.clone()
        transform_params = self.transform_net(identity_split, context)
        if not inverse and self.unconditional_transform is not None:
            moved, logabsdet = self.unconditional_transform(identity_split, context)
            rows.index_copy_(1, self.identity_features, moved)
        elif self.unconditional_transform is not None:
            rows.index_copy_(1, self.identity_features, identity_split)
        pixel_rows = rows.permute(0, 2, 3, 1).reshape(b * h * w, c)
        pixel_params = transform_params.permute(0, 2, 3, 1).reshape(b * h * w, -1)
        out_rows, lad_rows = self._fused_layer(pixel_rows, pixel_params, inverse)
        outputs = out_rows.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
        lad = lad_rows.reshape(b, h * w).sum(dim=1)
        logabsdet = lad if logabsdet is None else logabsdet + lad
        if logabsdet_accumulator is not None:
            logabsdet_accumulator += logabsdet
            logabsdet = logabsdet_accumulator
        return outputs, logabsdet

    # whole-layer kernel K11 (MLP conditioner inside the layer kernel, runs of layers in one launch)
    fuse_conditioner = os.environ.get("NFA_K11", "1") != "0"

    def _run_kind(self, context):
        net = self.transform_net
        ok = (self.fuse_conditioner and not torch.is_grad_enabled() and context is None
              and self._conditioner_shape() is not None and self._activation_code() != N.SCALE_GIVEN
              and 1 <= self.num_identity_features <= 64 and 1 <= self.num_transform_features <= 64
              and self._fused_geometry()[0] <= 128
              and (type(net).__name__ == "MLP"
                   or all(not b.training or b.dropout.p == 0.0 for b in net.blocks)))   # (an active dropout: layer by layer)
        return "k11" if ok else None

    def _conditioner_shape(self):
        """(hidden Linears, residual blocks?) of a conditioner K11 runs inside the layer kernel -- an MLP of 128-wide ReLU
        layers (nn/nets/mlp.py:47-68), or (round 5) a ResidualNet of width <= 128 with ReLU blocks, no context and no batch
        norm (nn/nets/resnet.py:55-100: the conditioner of the reference's own SimpleRealNVP, flows/realnvp.py:44-71) --,
        or None.  Read once per cache epoch (a swapped conditioner registers Parameters, which advances it)."""
        held = self.__dict__.get("_conditioner_shape_held")
        if held is None or held[0] != _cache.epoch():
            from ..nn.nets.mlp import MLP
            from ..nn.nets.resnet import ResidualNet
            net, shape = self.transform_net, None
            if type(net) is MLP:
                if (net._activation is torch.nn.functional.relu and not net._activate_output
                        and all(h <= 128 for h in net._hidden_sizes)):
                    shape = (len(net._hidden_layers), False)
            elif type(net) is ResidualNet:
                if (net.context_features is None and net.hidden_features <= 128
                        and not any(b.use_batch_norm for b in net.blocks)):
                    shape = (2 * len(net.blocks), True)
            held = (_cache.epoch(), shape)
            self.__dict__["_conditioner_shape_held"] = held
        shape = held[1]
        if shape is not None and shape[1]:   # (a block's activation is a plain attribute: read on every call)
            relu = (torch.nn.functional.relu, torch.relu)
            if not all(b.__dict__.get("activation") in relu for b in self.transform_net.blocks):
                return None
        return shape

    def _fused_geometry(self, others=()):
        """(padded features, transformed features, identity features, pad value): K11 wants the row length in
        multiples of four; the pad columns pass through (the layers of a run share their split)."""
        return (self.features + 3) // 4 * 4, self.num_transform_features, self.num_identity_features, 0.0

    def _run_signature(self):
        return ("k11", self.features, self.num_transform_features, self.num_identity_features,
                self._conditioner_shape(), self._activation_code())

    def _packed_mlp(self):
        net = self.transform_net
        key = _weights_key(self, net)
        cached = getattr(self, "_packed_mlp_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_mlp_conditioner(net, self.num_transform_features,
                                                    additive=self._activation_code() == N.SCALE_ADDITIVE))
            self._packed_mlp_cache = cached
        return cached[1]

    def _generic_float64(self, inputs, context, inverse, logabsdet_accumulator):
        """float64 vectors: the reference's expressions (coupling.py:73-130, :228-252) as tensor operations on
        the device (conditioner outputs [shift block | unconstrained scale block])."""
        identity_split = inputs.index_select(1, self.identity_features)
        transform_split = inputs.index_select(1, self.transform_features)
        logabsdet = None
        if inverse and self.unconditional_transform is not None:
            identity_split, logabsdet = self.unconditional_transform.inverse(identity_split, context)
        params = self.transform_net(identity_split, context)
        dt = self.num_transform_features
        shift = params[:, :dt]
        if self._transform_dim_multiplier() == 1:
            transform_split = transform_split - shift if inverse else transform_split + shift
            lad = inputs.new_zeros(inputs.shape[0])
        else:
            scale = self.scale_activation(params[:, dt:])
            log_scale = torch.log(scale).sum(dim=1)
            if inverse:
                transform_split, lad = (transform_split - shift) / scale, -log_scale
            else:
                transform_split, lad = transform_split * scale + shift, log_scale
        logabsdet = lad if logabsdet is None else logabsdet + lad
        if not inverse and self.unconditional_transform is not None:
            identity_split, lad_identity = self.unconditional_transform(identity_split, context)
            logabsdet = logabsdet + lad_identity
        outputs = torch.empty_like(inputs)
        outputs.index_copy_(1, self.identity_features, identity_split)
        outputs.index_copy_(1, self.transform_features, transform_split)
        if logabsdet_accumulator is not None:
            logabsdet_accumulator += logabsdet
            logabsdet = logabsdet_accumulator
        return outputs, logabsdet

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

    def _scale_and_shift(self, transform_params):   # coupling.py:266-269
        return torch.ones_like(transform_params), transform_params

    def _activation_code(self):
        return N.SCALE_ADDITIVE

    def _fused_layer(self, inputs, transform_params, inverse, in_perm=None, out_scatter=None,
                     accumulate_into=None):
        return ops.affine_coupling(inputs, transform_params, self.transform_features,
                                   N.SCALE_ADDITIVE, inverse=inverse, in_perm=in_perm,
                                   out_scatter=out_scatter, accumulate_into=accumulate_into)


class PiecewiseCouplingTransform(CouplingTransform):
    """The base of the piecewise (spline) couplings; coupling.py:272-296.  The reference's protocol, for code that
    subclasses it or asks `isinstance(t, PiecewiseCouplingTransform)`: `_coupling_transform_forward / _inverse(inputs,
    transform_params)` reshape the conditioner's output to [batch, features, params] (2-D) or [batch, channels, height,
    width, params] (4-D) (:279-289), call `_piecewise_cdf` and row-sum the log-derivatives (:291-293).  The four spline
    couplings below are its subclasses -- their `_piecewise_cdf` is the HIP functional of their spline --, and a user
    subclass that defines only `_piecewise_cdf` and `_transform_dim_multiplier` runs the reference's sequence with its own
    function (tensor operations on the device)."""

    supports_fused_permutation = False   # (a subclass with a fused layer kernel says so: the rational-quadratic one)
    supports_image_inputs = True

    def _coupling_transform_forward(self, inputs, transform_params):
        return self._coupling_transform(inputs, transform_params, inverse=False)

    def _coupling_transform_inverse(self, inputs, transform_params):
        return self._coupling_transform(inputs, transform_params, inverse=True)

    def _coupling_transform(self, inputs, transform_params, inverse=False):
        if inputs.dim() == 4:
            b, c, h, w = inputs.shape
            transform_params = transform_params.reshape(b, c, -1, h, w).permute(0, 1, 3, 4, 2)
        elif inputs.dim() == 2:
            b, d = inputs.shape
            transform_params = transform_params.reshape(b, d, -1)
        outputs, logabsdet = self._piecewise_cdf(inputs, transform_params, inverse)
        return outputs, torchutils.sum_except_batch(logabsdet)

    def _piecewise_cdf(self, inputs, transform_params, inverse=False):
        # (the spline couplings of this module define `_elementwise`, their functional's kernel: that IS their cdf)
        if type(self)._elementwise is not PiecewiseCouplingTransform._elementwise:
            return self._elementwise(inputs, transform_params, inverse)
        raise NotImplementedError()

    def _elementwise(self, inputs, transform_params, inverse):
        return self._piecewise_cdf(inputs, transform_params, inverse)


_F16_PACK_SLOTS = ("_packed_resnet_f16_cache", "_packed_resnet_f16s_cache", "_packed_resnet_f16c_cache")   # K8h, K8s, K8c


class PiecewiseRationalQuadraticCouplingTransform(PiecewiseCouplingTransform):
    """Neural-spline-flow coupling layer; coupling.py:502-582.

    Conditioner output per transformed feature: K width logits, K height logits and K-1
    (tails="linear") or K+1 (tails=None) derivative logits.  Width/height logits are divided by
    sqrt(hidden_features) of the conditioner when it exposes `hidden_features` /
    `hidden_channels` (coupling.py:554-559) -- done on the fly inside the kernel, the conditioner
    output tensor itself is left untouched."""

    supports_fused_permutation = True    # K1 / the whole-layer kernels
    supports_image_inputs = True

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

    def _elementwise(self, inputs, transform_params, inverse):
        K = self.num_bins
        return ops.rqs_elementwise(inputs, transform_params[..., :K], transform_params[..., K:2 * K],
                                   transform_params[..., 2 * K:], self._spec(), inverse)

    # K7: fold the conditioner's final Linear into the spline kernel (no [B, d_t*P] round trip
    # through HBM).  Class-level switch for A/B measurements.
    fuse_final_linear = True
    # GEMM engine of K7: "bf16x3" = split-bf16 operands on the bf16 matrix pipe (fp32-accurate,
    # default), "f32" = v_mfma_f32_32x32x2_f32
    final_linear_engine = os.environ.get("NFA_K7_ENGINE", "bf16x3")

    # K8: the whole ResidualNet conditioner inside the spline kernel (class-level switch)
    fuse_conditioner = os.environ.get("NFA_K8", "1") != "0"

    def _fused_geometry(self, others=()):
        """(padded features, transformed features, identity features, pad value) the whole-layer kernels are
        given for this layer -- or for the run of this layer and `others` (ops.fused_geometry)."""
        if not others:   # (asked several times per call and layer: kept per tail bound)
            held = self.__dict__.get("_own_geometry")
            if held is None or held[0] != self.tail_bound:
                held = (self.tail_bound, ops.fused_geometry(
                    self.features, [(self.num_transform_features, self.num_identity_features)], self.tail_bound))
                self.__dict__["_own_geometry"] = held
            return held[1]
        layers = [(c.num_transform_features, c.num_identity_features) for c in (self,) + tuple(others)]
        return ops.fused_geometry(self.features, layers, self.tail_bound)

    def _run_kind(self, context):
        return "k8" if self._resnet_eligible(context) else None

    def _run_signature(self):
        # (layers of one run may differ in their feature split -- odd feature counts under alternating masks --:
        # the run is given one padded geometry)
        features, num_blocks, ce = self._static_signature()
        # (tails=None: no spare columns -- the layers of a run share their exact split)
        split = None if self.tails == "linear" else (self.num_transform_features, self.num_identity_features)
        return ("k8", features, num_blocks, self.num_bins, self.tail_bound, self.tails, split,
                self.min_bin_width, self.min_bin_height, self.min_derivative,
                self._log2e(), self._use_f16(), self.conditioner_act_scale, ce, self.conditioner_engine,
                self._block_activation())

    def _block_activation(self):
        """The whole-layer kernels' code of the conditioner blocks' activation (one for all blocks), or None"""
        # (read on every call -- an activation is a plain attribute, swapping it advances no cache epoch --, so kept to
        #  dictionary lookups: ~1 us per layer)
        blocks = self._modules["transform_net"]._modules.get("blocks")
        if blocks is None:
            return None
        first = None
        for b in blocks._modules.values():
            act = b.__dict__.get("activation")
            if first is None:
                first = act
                if first is None:
                    return None
            elif act is not first:
                return None
        return N.ACTIVATION_RELU if first is None else ops.activation_code(first)

    def _static_signature(self):
        """(features, residual blocks, context features) of the layer's conditioner: read once per cache epoch
        (a conditioner swapped for another one registers its Parameters, which advances the epoch)."""
        held = self.__dict__.get("_static_sig")
        if held is None or held[0] != _cache.epoch():
            net = self.transform_net
            held = (_cache.epoch(), (self.features, len(getattr(net, "blocks", ())), getattr(net, "context_features", None)))
            self.__dict__["_static_sig"] = held
        return held[1]

    def _resnet_eligible(self, context):
        net = self.transform_net
        from ..nn.nets.resnet import ResidualNet
        if context is None:
            context_ok = net.context_features is None if type(net) is ResidualNet else False
        else:   # (K8 with a context: identity features + context within the initial layer's 64 columns)
            context_ok = (type(net) is ResidualNet and net.context_features is not None and context.dim() == 2
                          and context.shape[1] == net.context_features and context.is_cuda
                          and context.dtype == torch.float32 and not self._log2e()
                          and self.num_identity_features + net.context_features <= 64)
        return (self.fuse_conditioner and self.fuse_final_linear and not torch.is_grad_enabled()
                and context_ok and type(net) is ResidualNet
                and net.hidden_features <= 128
                # tails=None (round 6: the constrained spline, K8's plain loop -- csrc/rqs_resnet_tails.hip): ReLU, no
                # context, and a geometry without spare columns (a pad feature would be INSIDE the box: transformed)
                and (self.tails == "linear" or (self.tails is None and context is None
                                                and self.num_transform_features % 4 == 0
                                                and self._block_activation() == N.ACTIVATION_RELU))
                and ops.whole_layer_bins(self.num_bins)
                and 1 <= self.num_identity_features <= 64 and 1 <= self.num_transform_features
                and self._fused_geometry()[1] <= 64 and self._fused_geometry()[0] <= 128
                and self._activation_ok(context)
                and all(not b.training or (b.dropout.p == 0.0 and not b.use_batch_norm) for b in net.blocks)
                and (not any(b.use_batch_norm for b in net.blocks) or self._folded_net() is not None))

    def _activation_ok(self, context):
        """ReLU everywhere; F.leaky_relu / F.elu / tanh (round 4; with a context: round 5) at 8 or 10 bins without batch
        norm (its fold moves a positive scale through the ReLU) or the log2(e) fold."""
        act = self._block_activation()
        if act is None:
            return False
        if act == N.ACTIVATION_RELU:
            return True
        # (ELU / tanh are applied to the value at the pieces' scale: the experiment switch NFA_K8_ACT_SCALE must be 1)
        return (self.num_bins in (8, 10) and not self._log2e()
                and (act == N.ACTIVATION_LEAKY_RELU or self.conditioner_act_scale == 1.0)
                and not any(b.use_batch_norm for b in self.transform_net.blocks))

    def _folded_net(self):
        """The conditioner with its eval-mode batch norms folded into weights and biases (ops.fold_batch_norm),
        the net itself without batch norm, None when no fold exists; per weight key."""
        net = self.transform_net
        if not any(b.use_batch_norm for b in net.blocks):
            return net
        key = (tuple(b.training for b in net.blocks),) + _weights_key(self, net)
        cached = self.__dict__.get("_folded_net_cache")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, ops.fold_batch_norm(net))
            self.__dict__["_folded_net_cache"] = cached
        return cached[1]

    # experiment switch: fold log2(e) into the width / height logits as well (one v_exp_f32 per
    # softmax numerator)
    resnet_log2e = os.environ.get("NFA_K8_LOG2E", "0") != "0"

    def _log2e(self):
        """The fold is implemented for the 8-bin evaluation only."""
        return self.resnet_log2e and self.num_bins == 8 and self.tails == "linear"

    # GEMM engine of the whole-layer kernel: "f16x2" = two f16 pieces per operand on the f16 matrix
    # pipe (K8h / K8s, three products: 22-bit operand significands, the fp32 fma chain's error class; row blocks that
    # leave the f16 range are redone by the bf16x3 kernel), "bf16x3" = three bf16 pieces (K8, six products: 24-bit
    # operands, full fp32 range), "f16x3" (round 6) = three f16 pieces (K8x, five products: operands carried at the
    # reference's fp32 width on the f16 pipe; ReLU, no context -- other shapes take K8; the same redo pass)
    conditioner_engine = os.environ.get("NFA_K8_ENGINE", "f16x2")
    # scale of the hidden activations' f16 pieces (a power of two; K8h)
    conditioner_act_scale = float(os.environ.get("NFA_K8_ACT_SCALE", "1"))

    def _use_f16(self, geometry=None):
        """K8h serves 8 and 10 bins; with a context up to 32 context features beside up to 32 identity features
        (of the run's geometry) -- otherwise the bf16x3 kernel (K8) runs."""
        ce = self._static_signature()[2]
        return (self.conditioner_engine == "f16x2" and self.tails == "linear" and ops.whole_layer_bins(self.num_bins) and not self._log2e()
                and (ce is None or (ce <= 32 and (geometry or self._fused_geometry())[2] <= 32)))

    def _use_f16x3(self, geometry=None):
        """K8x serves the whole-layer bin counts (ops.whole_layer_bins) with ReLU blocks and no context -- otherwise engine
        "f16x3" means the bf16x3 kernel (K8)."""
        return (self.conditioner_engine == "f16x3" and self.tails == "linear" and ops.whole_layer_bins(self.num_bins) and not self._log2e()
                and self._static_signature()[2] is None and self._block_activation() == N.ACTIVATION_RELU)

    def _packed_resnet_f16x3(self, geometry=None):
        """(weights, biases, scales) for K8x (ops.pack_resnet_conditioner_f16x3), per weight key"""
        net = self.transform_net
        _, dt4, di_u, _ = geometry or self._fused_geometry()
        key = (ops.K8X_ACT_SCALE, dt4, di_u) + _weights_key(self, net)
        cached = self.__dict__.get("_packed_resnet_f16x3_cache")
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_resnet_conditioner_f16x3(self._folded_net(), self.num_transform_features,
                                                             self._transform_dim_multiplier(),
                                                             pad_transform_to=dt4, pad_identity_to=di_u))
            self.__dict__["_packed_resnet_f16x3_cache"] = cached
        return cached[1]

    def _packed_resnet_f16(self, geometry=None, tile16=False):
        """(weights, parameter words) for K8h, or -- `tile16` = 1 / 2 (ops.use_tile16) -- for K8s / K8c (the 16-sample-tile
        kernels of small batches)."""
        net = self.transform_net
        _, dt4, di_u, _ = geometry or self._fused_geometry()
        key = (self.conditioner_act_scale, dt4, di_u, tile16) + _weights_key(self, net)
        slot = _F16_PACK_SLOTS[int(tile16)]
        cached = self.__dict__.get(slot)
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_resnet_conditioner_f16(self._folded_net(), self.num_transform_features,
                                                           self._transform_dim_multiplier(),
                                                           act_scale=self.conditioner_act_scale,
                                                           pad_transform_to=dt4, pad_identity_to=di_u, tile16=bool(tile16),
                                                           colsplit=int(tile16) == 2))
            self.__dict__[slot] = cached
        return cached[1]

    def _f16_stream(self, tables, tile16=False):
        """K8h / K8s stream of this layer alone (its parameter stage carries `tables`), cached per table set."""
        pack = self._packed_resnet_f16(tile16=tile16)
        key = (self.__dict__[_F16_PACK_SLOTS[int(tile16)]][0],
               tables.data_ptr(), tables._version)
        cache = self.__dict__.setdefault("_f16_stream_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) > 8:
                cache.clear()
            hit = ops.build_f16_stream([pack], tables) or False   # (False: non-finite weights, no stream)
            cache[key] = hit
        return hit or None

    def _packed_resnet(self, geometry=None):
        net = self.transform_net
        _, dt4, di_u, _ = geometry or self._fused_geometry()
        key = (dt4, di_u, self._log2e()) + _weights_key(self, net)
        cached = getattr(self, "_packed_resnet_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.pack_resnet_conditioner(self._folded_net(), self.num_transform_features,
                                                       self._transform_dim_multiplier(),
                                                       log2e=self._log2e(),
                                                       pad_transform_to=dt4, pad_identity_to=di_u))
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
            Dp, dt4, di_u, _ = self._fused_geometry()
            hit = ops.coupling_layer_tables(self.features, self.transform_features, self.identity_features,
                                            in_perm, out_scatter, padded_features=Dp, padded_transform=dt4,
                                            padded_identity=di_u)
            cache[key] = hit
        return hit

    def _whole_layer(self, inputs, context, inverse, in_perm, out_scatter, accumulate_into):
        if not self._resnet_eligible(context):
            return None
        wp, bp = self._packed_resnet()
        tables = self._layer_tables(in_perm, out_scatter)
        nb = len(self.transform_net.blocks)
        Dp, dt4, di, pad_value = self._fused_geometry()
        spec = self._spec()
        # (ragged batches are padded to full 128-row blocks, odd shapes to multiples of four columns, in `ops`)
        act = self._block_activation()
        if self._use_f16x3():
            res = ops.rqs_coupling_resnet_f16x3(inputs, self._packed_resnet_f16x3(), (wp, bp), tables, dt4, di, nb, spec,
                                                inverse, accumulate_into, pad=(Dp, pad_value))
            if res is not None:
                return res
        tile16 = self._use_f16() and inputs.is_cuda and ops.use_tile16(inputs.shape[0], self.num_bins, context, inputs.device, act)
        stream = self._f16_stream(tables, tile16) if self._use_f16() else None   # (None: non-finite weights -> exact kernel)
        if stream is not None:
            res = ops.rqs_coupling_resnet_f16(inputs, stream, (wp, bp), tables, dt4, di, nb, spec,
                                              inverse, accumulate_into, pad=(Dp, pad_value), context=context,
                                              tile16=tile16, activation=act)
            if res is None and tile16:   # (K8s over its LDS budget: K8h with its own stream)
                stream = self._f16_stream(tables, False)
                if stream is not None:
                    res = ops.rqs_coupling_resnet_f16(inputs, stream, (wp, bp), tables, dt4, di, nb, spec,
                                                      inverse, accumulate_into, pad=(Dp, pad_value), context=context,
                                                      tile16=False, activation=act)
            return res
        return ops.rqs_coupling_resnet(inputs, wp, bp, tables, dt4, di, nb, spec, inverse, accumulate_into,
                                       log2e=self._log2e(), context=context, pad=(Dp, pad_value), activation=act)

    def _packed_final_linear(self, layer):
        split = self.final_linear_engine == "bf16x3"
        key = (_cache.epoch(), layer.weight.data_ptr(), layer.weight._version, layer.bias.data_ptr(),
               layer.bias._version, split)
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


class PiecewiseLinearCouplingTransform(PiecewiseCouplingTransform):
    """Piecewise-linear coupling layer (Mueller et al. 2018); coupling.py:297-350.  The conditioner
    emits K pdf logits per transformed element."""

    supports_fused_permutation = False
    supports_image_inputs = True

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None):
        self.num_bins = num_bins
        self.tails = tails
        self.tail_bound = tail_bound
        if apply_unconditional_transform:
            from .nonlinearities import PiecewiseLinearCDF

            def unconditional_transform(features):
                return PiecewiseLinearCDF(shape=[features] + (img_shape if img_shape else []), num_bins=num_bins,
                                          tails=tails, tail_bound=tail_bound)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return self.num_bins

    def _elementwise(self, inputs, transform_params, inverse):
        if self.tails is None:
            return splines.linear_spline(inputs, transform_params, inverse=inverse)
        return splines.unconstrained_linear_spline(inputs, transform_params, inverse=inverse, tails=self.tails,
                                                   tail_bound=self.tail_bound)


class PiecewiseQuadraticCouplingTransform(PiecewiseCouplingTransform):
    """Piecewise-quadratic coupling layer (Mueller et al. 2018); coupling.py:353-429.  K width
    logits and K+1 (tails=None) / K-1 (linear tails) height logits per transformed element, both
    divided by sqrt(hidden_features) when the conditioner exposes it (coupling.py:408-410)."""

    supports_fused_permutation = False
    supports_image_inputs = True

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None,
                 min_bin_width=splines.quadratic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=splines.quadratic.DEFAULT_MIN_BIN_HEIGHT):
        self.num_bins = num_bins
        self.tails = tails
        self.tail_bound = tail_bound
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        if apply_unconditional_transform:
            from .nonlinearities import PiecewiseQuadraticCDF

            def unconditional_transform(features):
                return PiecewiseQuadraticCDF(shape=[features] + (img_shape if img_shape else []),
                                             num_bins=num_bins, tails=tails, tail_bound=tail_bound,
                                             min_bin_width=min_bin_width, min_bin_height=min_bin_height)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return self.num_bins * 2 - 1 if self.tails == "linear" else self.num_bins * 2 + 1

    def _elementwise(self, inputs, transform_params, inverse):
        if self.tails is not None and self.tails != "linear":
            raise RuntimeError("{} tails are not implemented.".format(self.tails))
        K = self.num_bins
        divisor = 0.0
        if hasattr(self.transform_net, "hidden_features"):
            divisor = float(np.sqrt(self.transform_net.hidden_features))
        spec = ops.make_rqs_spec(K, self.tails, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                                 min_bin_height=self.min_bin_height, wh_divisor=divisor)
        if self.tails == "linear":
            assert transform_params.shape[-1] == 2 * K - 1  # quadratic.py:34
        return ops.quadratic_spline(inputs, transform_params[..., :K], transform_params[..., K:], spec, inverse)


class PiecewiseCubicCouplingTransform(PiecewiseCouplingTransform):
    """Piecewise-cubic coupling layer (Durkan et al. 2019, "Cubic-spline flows"); coupling.py:429-499.
    Per transformed element K width logits, K height logits (both divided by sqrt(hidden_features)
    when the conditioner exposes it) and the two boundary-derivative logits."""

    supports_fused_permutation = False
    supports_image_inputs = True

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None,
                 min_bin_width=splines.cubic.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=splines.cubic.DEFAULT_MIN_BIN_HEIGHT):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tails = tails
        self.tail_bound = tail_bound
        if apply_unconditional_transform:
            from .nonlinearities import PiecewiseCubicCDF

            def unconditional_transform(features):
                return PiecewiseCubicCDF(shape=[features] + (img_shape if img_shape else []), num_bins=num_bins,
                                         tails=tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                                         min_bin_height=min_bin_height)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return self.num_bins * 2 + 2

    def _elementwise(self, inputs, transform_params, inverse):
        if self.tails is not None and self.tails != "linear":
            raise RuntimeError("{} tails are not implemented.".format(self.tails))
        K = self.num_bins
        divisor = 0.0
        if hasattr(self.transform_net, "hidden_features"):
            divisor = float(np.sqrt(self.transform_net.hidden_features))
        spec = ops.make_rqs_spec(K, self.tails, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                                 min_bin_height=self.min_bin_height, wh_divisor=divisor)
        return ops.cubic_spline(inputs, transform_params[..., :K], transform_params[..., K:2 * K],
                                transform_params[..., 2 * K:2 * K + 1], transform_params[..., 2 * K + 1:2 * K + 2],
                                spec, inverse)
