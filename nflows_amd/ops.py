"""Tensor-level entry points of the MI355X hot path.

Each function validates its arguments the way the reference does (same exception types),
allocates the outputs with PyTorch's caching allocator, and enqueues ONE HIP kernel from
libnflows_amd.so on the current stream.  There is no eager / CPU implementation behind these
functions.

Data-dependent errors
---------------------
The reference raises `InputOutsideDomain` (rational_quadratic.py:81-82) and asserts a
non-negative discriminant (:142) by synchronising with the device.  The kernels record the same
conditions in a per-device status word instead:
  * constrained splines (tails=None) check it immediately (a sync, exactly as the reference);
  * linear-tail splines cannot leave their domain; their inverse records a negative
    discriminant lazily -- call `check_status()` (or set `set_error_mode("immediate")`).
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _native as N
from . import autograd as AG
from .errors import InputOutsideDomain

_status_words = {}
_error_mode = "deferred"
_launch_hook = None


_last_redo = None   # the `redo_blocks` tensor of the most recent K8h launch (flags per 128-row block)


def last_redo_blocks():
    """Number of 128-row blocks the most recent K8h (f16x2) launch handed to the exact kernel because a
    value left the f16 range or the inputs were non-finite (reads the device flags: synchronises).
    None if no such launch happened yet.  bench.py reports it; a trained flow that makes this non-zero on
    ordinary data runs those blocks at the bf16x3 kernel's speed."""
    return None if _last_redo is None else int((_last_redo != 0).sum().item())


def set_launch_hook(hook):
    """Measurement aid for bench.py: `hook.begin(name)` / `hook.end(token, algorithmic_bytes)` are
    called right before / after the K1 kernel is enqueued (on the current stream), so a caller
    can bracket it with HIP events.  None (default) disables it; nothing else changes."""
    global _launch_hook
    _launch_hook = hook


def set_error_mode(mode):
    """"deferred" (default): only conditions the reference would raise on *every* call site sync
    immediately; "immediate": every spline call reads the status word back (one sync per call)."""
    global _error_mode
    if mode not in ("deferred", "immediate"):
        raise ValueError(mode)
    _error_mode = mode


def _status_word(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    w = _status_words.get(key)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=device)
        _status_words[key] = w
    return w


def check_status(device=None):
    """Reads (and clears) the device status word; raises what the reference would have raised."""
    devices = [device] if device is not None else None
    words = []
    if devices is None:
        words = list(_status_words.values())
    else:
        words = [_status_word(torch.device(device))]
    for w in words:
        bits = int(w.item())
        if bits:
            w.zero_()
            if bits & N.STATUS_BAD_INDEX:
                raise IndexError("nflows_amd: feature index / permutation entry out of range")
            if bits & N.STATUS_OUTSIDE_DOMAIN:
                raise InputOutsideDomain()
            if bits & N.STATUS_NEG_DISCRIMINANT:
                raise AssertionError("negative discriminant in rational-quadratic inverse")


def _no_grad_guard(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "nflows_amd: the HIP kernels have no backward yet; evaluate under torch.no_grad() "
            "(density evaluation / sampling), or detach the inputs")


def make_rqs_spec(num_bins, tails, tail_bound=1.0, left=0.0, right=1.0, bottom=0.0, top=1.0,
                  min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
                  enable_identity_init=False, wh_divisor=0.0):
    """Builds struct nfa_rqs_spec from the reference functional's keyword arguments
    (rational_quadratic.py:13-25 / :66-80)."""
    if tails == "linear":
        left, right, bottom, top = -tail_bound, tail_bound, -tail_bound, tail_bound
        t = N.TAILS_LINEAR
    elif tails is None:
        t = N.TAILS_NONE
    else:
        raise RuntimeError("{} tails are not implemented.".format(tails))
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    beta = float(np.log(2) / (1 - min_derivative)) if enable_identity_init else 1.0
    tail_logit = float(np.log(np.exp(1 - min_derivative) - 1))
    return N.RqsSpec(int(num_bins), t, float(left), float(right), float(bottom), float(top),
                     float(min_bin_width), float(min_bin_height), float(min_derivative), beta,
                     tail_logit, float(wh_divisor))


def _idx(name, t, device, n=None):
    if t is None:
        return None
    if not torch.is_tensor(t) or t.dtype != torch.int64 or t.dim() != 1:
        raise TypeError("%s must be a 1-D int64 tensor" % name)
    if t.device != device:
        raise ValueError("%s is on %s, inputs on %s" % (name, t.device, device))
    if n is not None and t.numel() != n:
        raise ValueError("%s must have %d entries, got %d" % (name, n, t.numel()))
    return t.contiguous()


def _lad_buffer(accumulate_into, batch, device, inverse):
    """Output buffer and `flags` for the coupling kernels: a fresh [batch] tensor, or the
    caller's running total (CompositeTransform's `total_logabsdet`) updated in place."""
    flags = N.FLAG_INVERSE if inverse else 0
    if accumulate_into is None:
        return torch.empty(batch, dtype=torch.float32, device=device), flags
    t = accumulate_into
    if (t.dtype != torch.float32 or t.device != device or tuple(t.shape) != (batch,)
            or not t.is_contiguous()):
        raise ValueError("accumulate_into must be a contiguous float32 [batch] tensor on the inputs' device")
    return t, flags | N.FLAG_ACCUMULATE_LOGABSDET


def _after_spline(spec, inverse, device):
    if spec.tails == N.TAILS_NONE or _error_mode == "immediate":
        check_status(device)


def rqs_coupling(inputs, params, transform_idx, spec, inverse=False, in_perm=None, out_scatter=None,
                 accumulate_into=None, return_bin_idx=False):
    """K1 -- fused rational-quadratic coupling layer.  inputs [B, D], params [B, d_t*P].
    Returns (outputs [B, D], logabsdet [B]).  Differentiable (K1-backward kernel) when an input
    requires grad.  return_bin_idx=True (no-grad passes): a third result, int32 [B, d_t], the bin the
    kernel's search chose for every spline (`bin_idx` of rational_quadratic.py:115-118; -1 for elements
    in the tails, see include/nflows_amd.h)."""
    N.require_device_f32("inputs", inputs, 2)
    N.require_device_f32("transform_params", params, 2)
    dev = inputs.device
    B, D = inputs.shape
    tidx = _idx("transform_features", transform_idx, dev)
    perm = _idx("in_perm", in_perm, dev, D)
    scat = _idx("out_scatter", out_scatter, dev, D)
    dt = tidx.numel()
    P = 3 * spec.num_bins - 1 if spec.tails == N.TAILS_LINEAR else 3 * spec.num_bins + 1
    if params.shape[0] != B or params.shape[1] != dt * P:
        raise ValueError("transform_params must be [%d, %d], got %s" % (B, dt * P, tuple(params.shape)))
    if return_bin_idx:
        if AG.needs_grad(inputs, params):
            raise ValueError("return_bin_idx: a diagnostic of no-grad passes")
        return _rqs_coupling_launch(inputs, params, tidx, spec, inverse, perm, scat, accumulate_into, True)
    if AG.needs_grad(inputs, params):
        out, lad = AG.RqsCoupling.apply(inputs.contiguous(), params.contiguous(), tidx, spec, bool(inverse),
                                        perm, scat)
        if accumulate_into is not None:
            accumulate_into += lad
            lad = accumulate_into
        return out, lad
    return _rqs_coupling_launch(inputs, params, tidx, spec, inverse, perm, scat, accumulate_into)


def _rqs_coupling_launch(inputs, params, tidx, spec, inverse, perm, scat, accumulate_into, want_bins=False):
    dev = inputs.device
    B, D = inputs.shape
    dt = tidx.numel()
    P = 3 * spec.num_bins - 1 if spec.tails == N.TAILS_LINEAR else 3 * spec.num_bins + 1
    x = inputs.detach().contiguous()
    p = params.detach().contiguous()
    out = torch.empty_like(x)
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    bins = torch.empty(B, dt, dtype=torch.int32, device=dev) if want_bins else None
    hook = _launch_hook
    with torch.cuda.device(dev):
        token = hook.begin("rqs_coupling") if hook is not None else None
        rc = N.load().nfa_rqs_coupling_f32(N.ptr(x), N.ptr(p), N.ptr(tidx), N.ptr(perm), N.ptr(scat),
                                           N.ptr(out), N.ptr(lad), N.ptr(bins), N.ptr(_status_word(dev)), B, D, dt,
                                           ctypes.byref(spec), flags, N.stream_handle(dev))
        if hook is not None:
            # algorithmic bytes: inputs + conditioner output + outputs + logabsdet
            hook.end(token, 4 * (B * D + B * dt * P + B * D + B))
    if rc == N.ERR_UNSUPPORTED:
        if torch.is_grad_enabled() and (inputs.requires_grad or params.requires_grad):
            raise NotImplementedError("nflows_amd: this layer shape has no backward kernel yet")
        res = _rqs_coupling_unfused(x, p, tidx, perm, scat, spec, inverse, want_bins)
        out, l = res[0], res[1]
        if accumulate_into is not None:
            accumulate_into += l
            l = accumulate_into
        return (out, l, res[2]) if want_bins else (out, l)
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return (out, lad, bins) if want_bins else (out, lad)


def _rqs_coupling_unfused(x, p, tidx, perm, scat, spec, inverse, want_bins=False):
    """Layers whose sample does not fit the fused kernel's LDS tile (d_t*P > ~12k floats): the same
    result from the elementwise spline kernel + row-sum kernel + index copies (all on device)."""
    if perm is not None:
        x = permute_cols(x, perm)
    B, D = x.shape
    dt = tidx.numel()
    K = spec.num_bins
    pr = p.view(B * dt, -1)
    # the divisor is part of the spec, so the packed [N, P] fast path still applies
    xt = x.index_select(1, tidx).contiguous()
    res = rqs_elementwise(xt.view(-1), pr[:, :K], pr[:, K:2 * K], pr[:, 2 * K:], spec, inverse,
                          return_bin_idx=want_bins)
    y, l = res[0], res[1]
    out = x.clone()
    out[:, tidx] = y.view(B, dt)
    if scat is not None:
        out = permute_cols(out, torch.argsort(scat))
    if want_bins:
        return out, rowsum(l.view(B, dt)), res[2].view(B, dt)
    return out, rowsum(l.view(B, dt))


def rqs_elementwise(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
                    spec, inverse=False, return_bin_idx=False):
    """K5 -- elementwise functional on tensors of any leading shape S; logits S+[K], S+[K],
    S+[K-1 | K+1].  Returns (outputs S, logabsdet S); with return_bin_idx=True (no-grad passes) also the
    searched bin of every element, int32 S (`bin_idx` of rational_quadratic.py:115-118; -1 in the tails)."""
    dtype = inputs.dtype if torch.is_tensor(inputs) else torch.float32   # float64: the plain K5d kernel
    N.require_device_real("inputs", inputs, dtype)
    for nm, t in (("unnormalized_widths", unnormalized_widths),
                  ("unnormalized_heights", unnormalized_heights),
                  ("unnormalized_derivatives", unnormalized_derivatives)):
        N.require_device_real(nm, t, dtype)
    if return_bin_idx:
        if AG.needs_grad(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives):
            raise ValueError("return_bin_idx: a diagnostic of no-grad passes")
        return _rqs_elementwise_launch(inputs, unnormalized_widths, unnormalized_heights,
                                       unnormalized_derivatives, spec, inverse, True)
    if dtype == torch.float64 and AG.needs_grad(inputs, unnormalized_widths, unnormalized_heights,
                                                unnormalized_derivatives):
        return AG.RqsElementwise64.apply(inputs, unnormalized_widths, unnormalized_heights,
                                         unnormalized_derivatives, spec, bool(inverse))
    if AG.needs_grad(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives):
        return AG.RqsElementwise.apply(inputs, unnormalized_widths, unnormalized_heights,
                                       unnormalized_derivatives, spec, bool(inverse))
    return _rqs_elementwise_launch(inputs, unnormalized_widths, unnormalized_heights,
                                   unnormalized_derivatives, spec, inverse)


def _rqs_elementwise_launch(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
                            spec, inverse, want_bins=False):
    inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives = (
        t.detach() for t in (inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives))
    dev = inputs.device
    K = spec.num_bins
    nd_min = K - 1 if spec.tails == N.TAILS_LINEAR else K + 1
    nd = unnormalized_derivatives.shape[-1] if unnormalized_derivatives.dim() else -1
    shape = inputs.shape
    if (unnormalized_widths.shape != shape + (K,) or unnormalized_heights.shape != shape + (K,)
            or unnormalized_derivatives.shape[:-1] != shape or nd < nd_min):
        raise ValueError("spline logits must have shapes %s+[%d], +[%d], +[>=%d]"
                         % (tuple(shape), K, K, nd_min))
    n = inputs.numel()
    x = inputs.contiguous().view(-1)

    def rows(t, width):
        """[n, width] view with unit inner stride and a uniform row stride, copying only if needed."""
        if width == 0:
            return t.reshape(n, 0), 1
        try:
            v = t.view(n, width) if n else t.reshape(n, width)
        except RuntimeError:
            v = t.reshape(n, width)
        if v.stride(1) != 1 or (n > 1 and v.stride(0) < width):
            v = v.contiguous()
        return v, (v.stride(0) if n > 1 else width)

    uw, sw = rows(unnormalized_widths, K)
    uh, sh = rows(unnormalized_heights, K)
    ud, sd = rows(unnormalized_derivatives, nd)
    y = torch.empty_like(x)
    lad = torch.empty_like(x)
    bins = torch.empty(n, dtype=torch.int32, device=dev) if want_bins else None
    launch = N.load().nfa_rqs_elementwise_f64 if x.dtype == torch.float64 else N.load().nfa_rqs_elementwise_f32
    with torch.cuda.device(dev):
        rc = launch(N.ptr(x), N.ptr(uw), sw, N.ptr(uh), sh, N.ptr(ud) if nd else N.ptr(uw), sd, nd, N.ptr(y),
                    N.ptr(lad), N.ptr(bins), N.ptr(_status_word(dev)), n, ctypes.byref(spec),
                    int(bool(inverse)), N.stream_handle(dev))
    N.check(rc)
    _after_spline(spec, inverse, dev)
    if want_bins:
        return y.view(shape), lad.view(shape), bins.view(shape)
    return y.view(shape), lad.view(shape)


def _logit_rows(t, n, width):
    """[n, width] view with unit inner stride and a uniform row stride, copying only if needed."""
    try:
        v = t.view(n, width) if n else t.reshape(n, width)
    except RuntimeError:
        v = t.reshape(n, width)
    if v.stride(1) != 1 or (n > 1 and v.stride(0) < width):
        v = v.contiguous()
    return v, (v.stride(0) if n > 1 else width)


def linear_spline(inputs, unnormalized_pdf, spec, inverse=False):
    """K9 -- piecewise-linear spline functional (splines/linear.py) on tensors of any leading shape
    S; unnormalized_pdf S+[K].  Returns (outputs S, logabsdet S)."""
    N.require_device_f32("inputs", inputs)
    N.require_device_f32("unnormalized_pdf", unnormalized_pdf)
    if AG.needs_grad(inputs, unnormalized_pdf):
        return AG.LinearSpline.apply(inputs, unnormalized_pdf, spec, bool(inverse))
    return _linear_spline_launch(inputs, unnormalized_pdf, spec, inverse)


def _linear_spline_launch(inputs, unnormalized_pdf, spec, inverse):
    K = spec.num_bins
    shape = inputs.shape
    if unnormalized_pdf.shape != shape + (K,):
        raise ValueError("unnormalized_pdf must have shape %s+[%d]" % (tuple(shape), K))
    dev = inputs.device
    n = inputs.numel()
    x = inputs.detach().contiguous().view(-1)
    pdf, stride = _logit_rows(unnormalized_pdf.detach(), n, K)
    y, lad = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = N.load().nfa_linear_spline_f32(N.ptr(x), N.ptr(pdf), stride, N.ptr(y), N.ptr(lad),
                                            N.ptr(_status_word(dev)), n, ctypes.byref(spec),
                                            int(bool(inverse)), N.stream_handle(dev))
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return y.view(shape), lad.view(shape)


def quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, spec, inverse=False):
    """K9 -- piecewise-quadratic spline functional (splines/quadratic.py); widths S+[K], heights
    S+[K-1] (boundary heights derived) or S+[K+1]."""
    N.require_device_f32("inputs", inputs)
    N.require_device_f32("unnormalized_widths", unnormalized_widths)
    N.require_device_f32("unnormalized_heights", unnormalized_heights)
    if AG.needs_grad(inputs, unnormalized_widths, unnormalized_heights):
        return AG.QuadraticSpline.apply(inputs, unnormalized_widths, unnormalized_heights, spec, bool(inverse))
    return _quadratic_spline_launch(inputs, unnormalized_widths, unnormalized_heights, spec, inverse)


def _quadratic_spline_launch(inputs, unnormalized_widths, unnormalized_heights, spec, inverse):
    K = spec.num_bins
    shape = inputs.shape
    nh = unnormalized_heights.shape[-1] if unnormalized_heights.dim() else -1
    if (unnormalized_widths.shape != shape + (K,) or unnormalized_heights.shape[:-1] != shape
            or nh not in (K - 1, K + 1)):
        raise ValueError("spline logits must have shapes %s+[%d], +[%d or %d]" % (tuple(shape), K, K - 1, K + 1))
    dev = inputs.device
    n = inputs.numel()
    x = inputs.detach().contiguous().view(-1)
    uw, sw = _logit_rows(unnormalized_widths.detach(), n, K)
    uh, sh = _logit_rows(unnormalized_heights.detach(), n, nh)
    y, lad = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = N.load().nfa_quadratic_spline_f32(N.ptr(x), N.ptr(uw), sw, N.ptr(uh), sh, nh, N.ptr(y), N.ptr(lad),
                                               N.ptr(_status_word(dev)), n, ctypes.byref(spec),
                                               int(bool(inverse)), N.stream_handle(dev))
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return y.view(shape), lad.view(shape)


def cubic_spline(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                 unnorm_derivatives_right, spec, inverse=False):
    """K9 -- piecewise-cubic spline functional (splines/cubic.py); widths / heights S+[K], the two
    boundary-derivative logits S+[1]."""
    tensors = (inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left, unnorm_derivatives_right)
    for nm, t in zip(("inputs", "unnormalized_widths", "unnormalized_heights", "unnorm_derivatives_left",
                      "unnorm_derivatives_right"), tensors):
        N.require_device_f32(nm, t)
    if AG.needs_grad(*tensors):
        return AG.CubicSpline.apply(*tensors, spec, bool(inverse))
    return _cubic_spline_launch(*tensors, spec, inverse)


def _cubic_spline_launch(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                         unnorm_derivatives_right, spec, inverse):
    K = spec.num_bins
    shape = inputs.shape
    if (unnormalized_widths.shape != shape + (K,) or unnormalized_heights.shape != shape + (K,)
            or unnorm_derivatives_left.shape != shape + (1,) or unnorm_derivatives_right.shape != shape + (1,)):
        raise ValueError("spline logits must have shapes %s+[%d], +[%d], +[1], +[1]" % (tuple(shape), K, K))
    dev = inputs.device
    n = inputs.numel()
    x = inputs.detach().contiguous().view(-1)
    uw, sw = _logit_rows(unnormalized_widths.detach(), n, K)
    uh, sh = _logit_rows(unnormalized_heights.detach(), n, K)
    dl, sl = _logit_rows(unnorm_derivatives_left.detach(), n, 1)
    dr, sr = _logit_rows(unnorm_derivatives_right.detach(), n, 1)
    y, lad = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = N.load().nfa_cubic_spline_f32(N.ptr(x), N.ptr(uw), sw, N.ptr(uh), sh, N.ptr(dl), sl, N.ptr(dr), sr,
                                           N.ptr(y), N.ptr(lad), N.ptr(_status_word(dev)), n, ctypes.byref(spec),
                                           int(bool(inverse)), N.stream_handle(dev))
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return y.view(shape), lad.view(shape)


def affine_coupling(inputs, params, transform_idx, activation, inverse=False, scale=None,
                    in_perm=None, out_scatter=None, accumulate_into=None):
    """K2 -- fused affine/additive coupling.  params [B, 2*d_t] = [shift | scale logits]
    (additive: [B, d_t])."""
    N.require_device_f32("inputs", inputs, 2)
    N.require_device_f32("transform_params", params, 2)
    dev = inputs.device
    B, D = inputs.shape
    tidx = _idx("transform_features", transform_idx, dev)
    perm = _idx("in_perm", in_perm, dev, D)
    scat = _idx("out_scatter", out_scatter, dev, D)
    dt = tidx.numel()
    pcols = dt if activation == N.SCALE_ADDITIVE else 2 * dt
    if params.shape[0] != B or params.shape[1] != pcols:
        raise ValueError("transform_params must be [%d, %d], got %s" % (B, pcols, tuple(params.shape)))
    if activation == N.SCALE_GIVEN:
        N.require_device_f32("scale", scale, 2)
        if tuple(scale.shape) != (B, dt):
            raise ValueError("scale must be [%d, %d]" % (B, dt))
        scale = scale.contiguous()
    if AG.needs_grad(inputs, params, scale):
        out, lad = AG.AffineCoupling.apply(inputs.contiguous(), params.contiguous(), scale, tidx,
                                           int(activation), bool(inverse), perm, scat)
        if accumulate_into is not None:
            accumulate_into += lad
            lad = accumulate_into
        return out, lad
    return _affine_coupling_launch(inputs, params, scale, tidx, activation, inverse, perm, scat,
                                   accumulate_into)


def _affine_coupling_launch(inputs, params, scale, tidx, activation, inverse, perm, scat, accumulate_into):
    dev = inputs.device
    B, D = inputs.shape
    dt = tidx.numel()
    x = inputs.detach().contiguous()
    p = params.detach().contiguous()
    scale = None if scale is None else scale.detach().contiguous()
    out = torch.empty_like(x)
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    with torch.cuda.device(dev):
        rc = N.load().nfa_affine_coupling_f32(N.ptr(x), N.ptr(p), N.ptr(scale), N.ptr(tidx), N.ptr(perm),
                                              N.ptr(scat), N.ptr(out), N.ptr(lad),
                                              N.ptr(_status_word(dev)), B, D, dt, int(activation),
                                              flags, N.stream_handle(dev))
    N.check(rc)
    return out, lad


def affine_autoregressive(inputs, params, inverse=False):
    """K2b -- elementwise affine with interleaved [B, D, 2] parameters (scale logit, shift)."""
    N.require_device_f32("inputs", inputs, 2)
    N.require_device_f32("autoregressive_params", params)
    B, D = inputs.shape
    if params.numel() != B * D * 2:
        raise ValueError("autoregressive_params must hold %d values" % (B * D * 2))
    if AG.needs_grad(inputs, params):
        return AG.AffineAutoregressive.apply(inputs.contiguous(), params.contiguous(), bool(inverse))
    return _affine_autoregressive_launch(inputs, params, inverse)


def _affine_autoregressive_launch(inputs, params, inverse):
    B, D = inputs.shape
    dev = inputs.device
    x = inputs.detach().contiguous()
    p = params.detach().contiguous()
    out = torch.empty_like(x)
    lad = torch.empty(B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = N.load().nfa_affine_autoregressive_f32(N.ptr(x), N.ptr(p), N.ptr(out), N.ptr(lad), B, D,
                                                    int(bool(inverse)), N.stream_handle(dev))
    N.check(rc)
    return out, lad


def permute_cols(inputs, permutation):
    """K4 -- torch.index_select(inputs, 1, permutation) for 2-D tensors of 4-byte elements."""
    if not torch.is_tensor(inputs) or not inputs.is_cuda:
        raise NotImplementedError("nflows_amd: inputs must live on a HIP device (no CPU fallback)")
    if inputs.dim() != 2 or inputs.element_size() != 4:
        raise NotImplementedError("nflows_amd.permute_cols: 2-D tensors of 4-byte elements only")
    dev = inputs.device
    B, D = inputs.shape
    perm = _idx("permutation", permutation, dev, D)
    x = inputs.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = N.load().nfa_permute_cols_b32(N.ptr(x), N.ptr(perm), N.ptr(out), N.ptr(_status_word(dev)),
                                           B, D, N.stream_handle(dev))
    N.check(rc)
    return out


def rowsum(x):
    """K3 -- torch.sum over everything but the batch dimension."""
    N.require_device_f32("x", x)
    _no_grad_guard(x)
    B = x.shape[0]
    cols = x.numel() // B if B else 0
    v = x.contiguous().view(B, cols)
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = N.load().nfa_rowsum_f32(N.ptr(v), N.ptr(out), B, cols, N.stream_handle(x.device))
    N.check(rc)
    return out


def searchsorted(bin_locations, inputs, eps=1e-6):
    """torchutils.searchsorted (utils/torchutils.py:134-136) on device tensors: knots S_b+[n_knots] float32,
    inputs S_i float32 (S_b broadcasts against S_i like `inputs[..., None] >= bin_locations` does), int64
    result of the broadcast shape.  A single row of knots (the reference's `bin_locations[None, :]`) is
    staged once per workgroup; per-input rows are read through LDS tiles.  `bin_locations` is not modified
    (the reference leaves `+= eps` behind in the caller's tensor; nobody reads it)."""
    N.require_device_f32("bin_locations", bin_locations)
    N.require_device_f32("inputs", inputs)
    if bin_locations.dim() < 1:
        raise ValueError("bin_locations needs a last dimension of knots")
    nk = bin_locations.shape[-1]
    dev = inputs.device
    if bin_locations.device != dev:
        raise ValueError("bin_locations is on %s, inputs on %s" % (bin_locations.device, dev))
    shape = torch.broadcast_shapes(tuple(inputs.shape), tuple(bin_locations.shape[:-1]))
    if nk == 0:
        return torch.full(shape, -1, dtype=torch.int64, device=dev)
    x = inputs.detach().expand(shape).contiguous().view(-1)
    n = x.numel()
    if bin_locations.numel() == nk:
        knots, stride = bin_locations.detach().reshape(nk).contiguous(), 0
    else:
        knots = bin_locations.detach().expand(tuple(shape) + (nk,)).contiguous().view(n, nk)
        stride = nk
    out = torch.empty(n, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = N.load().nfa_searchsorted_f32(N.ptr(knots), stride, nk, N.ptr(x), N.ptr(out), n, float(eps),
                                           N.stream_handle(dev))
    N.check(rc)
    return out.view(shape)


def standard_normal_log_prob(z, logabsdet=None):
    """-0.5*sum(z^2) - 0.5*D*log(2*pi) (+ logabsdet), one kernel (float64: the same expression as device
    tensor operations, distributions/normal.py:27-33)."""
    if torch.is_tensor(z) and z.is_cuda and z.dtype == torch.float64:
        flat = z.reshape(z.shape[0], -1)
        out = -0.5 * torch.sum(flat ** 2, dim=1) - 0.5 * flat.shape[1] * math.log(2 * math.pi)
        return out if logabsdet is None else out + logabsdet
    N.require_device_f32("inputs", z)
    if logabsdet is not None:
        N.require_device_f32("logabsdet", logabsdet, 1)
    if AG.needs_grad(z, logabsdet):
        return AG.StandardNormalLogProb.apply(z, logabsdet)
    return _standard_normal_log_prob_launch(z, logabsdet)


def _standard_normal_log_prob_launch(z, logabsdet):
    z = z.detach()
    B = z.shape[0]
    cols = z.numel() // B if B else 1
    v = z.contiguous().view(B, cols)
    if logabsdet is not None:
        logabsdet = logabsdet.detach().contiguous()
    out = torch.empty(B, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        rc = N.load().nfa_standard_normal_log_prob_f32(N.ptr(v), N.ptr(logabsdet), N.ptr(out), B, cols,
                                                       N.stream_handle(z.device))
    N.check(rc)
    return out


def pack_made_schedule(net, sequential_steps, params_per_feature):
    """Packs a MADE (transforms/made.py) for K12 (csrc/made_inverse.hip): one contiguous block per step
    t = 0 .. T holding everything the step reads -- header (16 ints: number of units of degree t, offsets of
    the unit rows / the output rows / the output biases), the unit table (8 words per unit in layer order:
    bias, unit index, padded columns, source vector (-1 = the features), destination vector (-1 = none),
    add_stream, set_stream, 0), the units' masked weight rows (zero-padded to multiples of 16 columns),
    feature t's P rows of the output layer and its P biases (t < T) -- padded to 1 KB.
    Returns (blocks fp32, block starts int32 [T + 2] in 1 KB grains, layout list)."""
    T, P = int(sequential_steps), int(params_per_feature)
    dev = net.final_layer.weight.device
    H = net.initial_layer.weight.shape[0]
    Hp = (H + 15) // 16 * 16
    Xp = max(16, (T + 15) // 16 * 16)
    residual = bool(net.use_residual_blocks)
    linears = [net.initial_layer]
    for block in net.blocks:
        linears += list(block.linear_layers) if residual else [block.linear]
    n = len(linears)
    HDR, GRAIN = 16, 256
    per_layer, sorted_rows, sorted_bias, orders, starts = [], [], [], [], []
    for l, lin in enumerate(linears):
        w = (lin.weight.detach() * lin.mask).float().cpu()
        deg = lin.degrees.to(torch.int64).cpu()
        order = torch.argsort(deg, stable=True)
        counts = torch.bincount(deg, minlength=T + 2)[:T + 2]
        start = torch.cat((counts.new_zeros(1), torch.cumsum(counts, 0)))   # start[d] = #units of degree < d
        kp = Xp if l == 0 else Hp
        if l == 0:
            w = w[:, :min(Xp, w.shape[1])]      # features >= T meet zero masks in every hidden unit
        w = torch.cat((w, w.new_zeros(w.shape[0], kp - w.shape[1])), dim=1)
        bias = lin.bias.detach().float().cpu() if lin.bias is not None else w.new_zeros(w.shape[0])
        sorted_rows.append(w.index_select(0, order))
        sorted_bias.append(bias.index_select(0, order))
        orders.append(order.to(torch.int32))
        starts.append(start.tolist())
        is_upper = residual and l > 0 and l % 2 == 0          # second Linear of a residual block
        last = l == n - 1
        dst = -1 if (residual and last) else l                # (the residual stream itself feeds the output layer)
        per_layer.append([kp, l - 1, dst, 1 if is_upper else 0, 1 if residual and (l == 0 or is_upper) else 0])
    final = net.final_layer
    D = final.weight.shape[0] // P
    wf = (final.weight.detach() * final.mask).float().cpu().view(D, P, H)[:T]
    wf = torch.cat((wf, wf.new_zeros(T, P, Hp - H)), dim=2)
    bf = final.bias.detach().float().cpu().view(D, P)[:T]
    blocks, block_at, at = [], [], 0
    UNIT = 8
    for t in range(T + 1):
        header = torch.zeros(HDR, dtype=torch.int32)
        table, rows = [], []
        for l in range(n):
            s0, s1 = starts[l][t], starts[l][t + 1]
            kp, src_v, dst_v, add_stream, set_stream = per_layer[l]
            for u in range(s0, s1):
                entry = torch.zeros(UNIT, dtype=torch.int32)
                entry[0] = sorted_bias[l][u:u + 1].view(torch.int32)[0]
                entry[1] = orders[l][u]
                entry[2], entry[3], entry[4], entry[5], entry[6] = kp, src_v, dst_v, add_stream, set_stream
                table.append(entry.view(torch.float32))
                rows.append(sorted_rows[l][u])
        units = len(table)
        header[0] = units
        header[1] = HDR + UNIT * units                      # unit rows
        rows_len = sum(r.numel() for r in rows)
        header[2] = header[1] + rows_len                    # output rows
        parts = table + rows
        if t < T:
            parts.append(wf[t].reshape(-1))
            header[3] = header[2] + wf[t].numel()           # output biases
            parts.append(bf[t])
        block = torch.cat([header.view(torch.float32)] + parts)
        pad = (-block.numel()) % GRAIN
        # (every lane reads the header and four unit entries unconditionally: a block is at least that long)
        if block.numel() + pad < HDR + UNIT * 4:
            pad += GRAIN
        if pad:
            block = torch.cat((block, block.new_zeros(pad)))
        block_at.append(at)
        at += block.numel() // GRAIN
        blocks.append(block)
    block_at += [at, at]
    max_block = max(b.numel() for b in blocks)
    num_vectors = n                                         # residual: n - 1 ReLU'd vectors + the stream
    final_src = n - 1                                       # residual: the stream lives in the last vector's place
    layout = [n, int(residual), final_src, n - 1 if residual else -1, num_vectors, Hp, Xp, max_block]
    for e in per_layer:
        layout += e
    return (torch.cat(blocks).contiguous().to(dev), torch.tensor(block_at, dtype=torch.int32, device=dev), layout)


def made_rqs_inverse(inputs, schedule, hidden_features, sequential_steps, spec):
    """K12 -- the sequential features of the autoregressive spline inverse in one persistent kernel.
    Returns (outputs [B, D] with columns < sequential_steps filled, their logabsdet [B], hidden [B, H]),
    or None outside the kernel's shape family."""
    N.require_device_f32("inputs", inputs, 2)
    dev = inputs.device
    B, D = inputs.shape
    z = inputs.detach().contiguous()
    out = torch.empty_like(z)
    lad = torch.empty(B, dtype=torch.float32, device=dev)
    hidden = torch.empty(B, hidden_features, dtype=torch.float32, device=dev)
    floats, ints, layout = schedule
    arr = (ctypes.c_int32 * len(layout))(*layout)
    with torch.cuda.device(dev):
        rc = N.load().nfa_made_rqs_inverse_f32(
            N.ptr(z), N.ptr(floats), N.ptr(ints), arr, len(layout), N.ptr(out), N.ptr(lad), N.ptr(hidden),
            N.ptr(_status_word(dev)), B, D, hidden_features, sequential_steps, ctypes.byref(spec),
            N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    _after_spline(spec, True, dev)
    return out, lad, hidden


_sum_workspaces = {}


def pack_made_output(net, params_per_feature, first_feature=0):
    """The output layer of a MADE (made.py:261-268: weight [D * P, H] times its mask, feature-major rows) for K13
    (layout in include/nflows_amd.h): the rows of features `first_feature` .. D - 1, per feature padded 23 -> 24,
    features padded to a multiple of eight (zero rows: an even number of four-feature groups), the hidden width zero-padded to 256, rows in K7's order
    (`_k7_row_order`), split into three bf16 pieces per weight.  Returns (weights, biases, number of features)."""
    P = params_per_feature
    final = net.final_layer
    with torch.no_grad():
        W = final.masked_weight().detach().float()
        H = W.shape[1]
        D = W.shape[0] // P
        nf = D - first_feature
        nf4 = (nf + 7) // 8 * 8   # an even number of four-feature groups
        w = W.view(D, P, H)[first_feature:]
        b = final.bias.detach().float().view(D, P)[first_feature:]
        w = torch.cat((w, w.new_zeros(nf, 24 - P, H)), dim=1)
        b = torch.cat((b, b.new_zeros(nf, 24 - P)), dim=1)
        if nf4 > nf:
            w = torch.cat((w, w.new_zeros(nf4 - nf, 24, H)), dim=0)
            b = torch.cat((b, b.new_zeros(nf4 - nf, 24)), dim=0)
        if H < 256:
            w = torch.cat((w, w.new_zeros(nf4, 24, 256 - H)), dim=2)
        order = _k7_row_order(nf4).to(W.device)
        w = w.reshape(nf4 * 24, 256).index_select(0, order)
        b = b.reshape(nf4 * 24).index_select(0, order)
        tiles = nf4 * 24 // 32
        # (piece, tile, r, half, ks, j) -> (tile, piece, ks, half, r, j); column = half*128 + ks*8 + j
        wp = torch.stack(split_bf16x3(w)).view(3, tiles, 32, 2, 16, 8).permute(1, 0, 4, 3, 2, 5).contiguous()
        # row i of a tile sits in accumulator register q = 4*(i//8) + i%4 of lane-half (i//4)%2
        bp = b.view(tiles, 4, 2, 4).permute(0, 2, 1, 3).contiguous()
    return wp, bp, nf


MADE_OUTPUT_GROUPS_PER_CHUNK = 26   # NFA_MADE_OUTPUT_GROUPS_PER_CHUNK (include/nflows_amd.h)


def made_output_spline(inputs, hidden, packed, spec, first_column=0, inverse=False, out=None):
    """K13 -- MADE's output layer + the spline of every feature it parameterises + the per-sample logabsdet sum in
    one kernel.  inputs [B, D] (the columns from `first_column` on are the spline's inputs), hidden [B, H] (the
    output layer's input), packed = pack_made_output(net, P, first_feature=first_column).  Returns
    (outputs [B, D] -- `out` if given, only those columns written --, logabsdet [B] of those columns), or None
    when the shape is outside the kernel (callers run the GEMM + the spline kernel)."""
    N.require_device_f32("inputs", inputs, 2)
    N.require_device_f32("hidden", hidden, 2)
    wp, bp, nf = packed
    dev = inputs.device
    B, D = inputs.shape
    H = hidden.shape[1]
    if first_column + nf != D or hidden.shape[0] != B or H > 256 or H % 4 or spec.num_bins != 8 or B == 0:
        return None
    x = inputs.detach().contiguous()
    h = hidden.detach().contiguous()
    target = out
    pad = (-B) % 128
    if pad:   # whole 128-row blocks on padded copies
        x = torch.cat((x, x.new_zeros(pad, D)), dim=0)
        h = torch.cat((h, h.new_zeros(pad, H)), dim=0)
        out = None
    if out is None:
        out = torch.empty_like(x)
    elif (out.dtype != torch.float32 or out.device != dev or tuple(out.shape) != (B, D) or not out.is_contiguous()):
        raise ValueError("out must be a contiguous float32 [batch, features] tensor on the inputs' device")
    groups = (nf + 7) // 8 * 2
    chunks = (groups + MADE_OUTPUT_GROUPS_PER_CHUNK - 1) // MADE_OUTPUT_GROUPS_PER_CHUNK
    part = torch.empty(chunks, B + pad, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = N.load().nfa_rqs_made_output_f32(
            N.ptr(x), D, first_column, N.ptr(h), H, N.ptr(wp), N.ptr(bp), N.ptr(out), N.ptr(part),
            N.ptr(_status_word(dev)), B + pad, nf, ctypes.byref(spec), N.FLAG_INVERSE if inverse else 0,
            N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    lad = part[0] if chunks == 1 else part.sum(dim=0)   # chunk by chunk, the same order on every run
    if pad:
        lad = lad[:B]
        if target is None:
            out = out[:B]
        else:
            target[:, first_column:] = out[:B, first_column:]
            out = target
    _after_spline(spec, inverse, dev)
    return out, lad


def sum_count(values):
    """float64 [2] = (sum of `values` accumulated in float64 in a fixed order, number of values), one
    launch."""
    N.require_device_f32("values", values)
    v = values.detach().contiguous().view(-1)
    dev = v.device
    out = torch.empty(2, dtype=torch.float64, device=dev)
    lib = N.load()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream)
        ws = _sum_workspaces.get(key)
        if ws is None:
            ws = torch.zeros(lib.nfa_sum_count_workspace_bytes(), dtype=torch.uint8, device=dev)
            _sum_workspaces[key] = ws
        N.check(lib.nfa_sum_count_f64(N.ptr(v), v.numel(), N.ptr(out), N.ptr(ws), N.stream_handle(dev)))
    return out


def linear_wgrad(inputs, grad_outputs, need_bias=True):
    """K10 -- gradients of y = x W^T + b with respect to W and b for x [B, I], dL/dy [B, O]:
    (grad_weight [O, I], grad_bias [O] or None).  The reduction over the batch is split over the
    chip and summed in a fixed order (deterministic).  Returns None when the layer shape has no
    kernel (widths not multiples of 4): callers then use the library GEMM."""
    N.require_device_f32("inputs", inputs, 2)
    N.require_device_f32("grad_outputs", grad_outputs, 2)
    if inputs.shape[0] != grad_outputs.shape[0]:
        raise ValueError("inputs and grad_outputs must have the same number of rows")
    B, I = inputs.shape
    O = grad_outputs.shape[1]
    if I % 4 or O % 4:
        return None
    x, gy = inputs.contiguous(), grad_outputs.contiguous()
    if x.data_ptr() % 16:
        x = x.clone()
    if gy.data_ptr() % 16:
        gy = gy.clone()
    dev = x.device
    lib = N.load()
    gw = torch.empty(O, I, dtype=torch.float32, device=dev)
    gb = torch.empty(O, dtype=torch.float32, device=dev) if need_bias else None
    ws = torch.empty(max(1, lib.nfa_linear_wgrad_workspace_bytes(B, I, O) // 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nfa_linear_wgrad_f32(N.ptr(x), N.ptr(gy), N.ptr(gw), N.ptr(gb), N.ptr(ws), B, I, O, 0,
                                      N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    return gw, gb


def linear_wgrad_batched(problems, need_bias=True):
    """K10 for several Linear layers of ONE shape in one launch pair (`nfa_linear_wgrad_batched_f32`: the four
    128 x 128 layers of a ResidualNet conditioner).  `problems`: [(inputs [B, I], grad_outputs [B, O]), ...], at most 8.
    Returns [(grad_weight, grad_bias or None), ...], or None when the shape has no kernel."""
    import ctypes
    if not 1 <= len(problems) <= 8:
        raise ValueError("1 .. 8 problems per launch")
    B, I = problems[0][0].shape
    O = problems[0][1].shape[1]
    if I % 4 or O % 4:
        return None
    xs, gys = [], []
    for x, gy in problems:
        N.require_device_f32("inputs", x, 2)
        N.require_device_f32("grad_outputs", gy, 2)
        if tuple(x.shape) != (B, I) or tuple(gy.shape) != (B, O):
            raise ValueError("the problems of one launch share one shape")
        x, gy = x.contiguous(), gy.contiguous()
        xs.append(x.clone() if x.data_ptr() % 16 else x)
        gys.append(gy.clone() if gy.data_ptr() % 16 else gy)
    dev = xs[0].device
    lib = N.load()
    n = len(problems)
    gws = torch.empty(n, O, I, dtype=torch.float32, device=dev)
    gbs = torch.empty(n, O, dtype=torch.float32, device=dev) if need_bias else None
    ws = torch.empty(max(1, lib.nfa_linear_wgrad_batched_workspace_bytes(n, B, I, O) // 4), dtype=torch.float32, device=dev)
    arr = ctypes.c_void_p * n
    a_x = arr(*[t.data_ptr() for t in xs])
    a_gy = arr(*[t.data_ptr() for t in gys])
    a_gw = arr(*[gws[q].data_ptr() for q in range(n)])
    a_gb = arr(*[gbs[q].data_ptr() for q in range(n)]) if need_bias else None
    with torch.cuda.device(dev):
        rc = lib.nfa_linear_wgrad_batched_f32(n, a_x, a_gy, a_gw, a_gb, N.ptr(ws), B, I, O, 0, N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    return [(gws[q], gbs[q] if need_bias else None) for q in range(n)]


def rqs_shared(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, spec,
               inverse=False):
    """K6 -- rational-quadratic CDF transform with batch-shared logits [*shape, K]; inputs
    [B, *shape].  Returns (outputs, logabsdet [B])."""
    N.require_device_f32("inputs", inputs)
    K = spec.num_bins
    nd = K - 1 if spec.tails == N.TAILS_LINEAR else K + 1
    B = inputs.shape[0]
    F = inputs.numel() // B if B else int(np.prod(inputs.shape[1:]))
    for nm, t, w in (("unnormalized_widths", unnormalized_widths, K),
                     ("unnormalized_heights", unnormalized_heights, K),
                     ("unnormalized_derivatives", unnormalized_derivatives, nd)):
        N.require_device_f32(nm, t)
        if t.numel() != F * w:
            raise ValueError("%s must hold %d x %d values" % (nm, F, w))
    dev = inputs.device
    x = inputs.detach().contiguous().view(B, F)
    uw, uh, ud = (t.detach().contiguous() for t in (unnormalized_widths, unnormalized_heights,
                                                    unnormalized_derivatives))
    y = torch.empty_like(x)
    lad = torch.empty(B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = N.load().nfa_rqs_shared_f32(N.ptr(x), N.ptr(uw), N.ptr(uh), N.ptr(ud), N.ptr(y), N.ptr(lad),
                                         N.ptr(_status_word(dev)), B, F, ctypes.byref(spec),
                                         N.FLAG_INVERSE if inverse else 0, N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:  # tables do not fit in LDS: broadcast the logits, K5 + K3
        def share(p, w):
            return p.view(1, F, w).expand(B, F, w)
        yy, ll = _rqs_elementwise_launch(x, share(uw, K), share(uh, K), share(ud, nd), spec, inverse)
        return yy.view(inputs.shape), rowsum(ll)
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return y.view(inputs.shape), lad


def _k7_row_order(num_transform):
    """For every row of the packed (tile-major) weight: which row of the feature-major padded
    matrix [d_t, 24] it is.  The kernel computes params^T = W_tile @ hidden^T, so lane (sample,
    half) of a wave receives rows 8*(q//4) + 4*half + q%4 (q = 0..15) of each 32-row tile; the
    order below makes the 48 values a lane gets from the three tiles of a group exactly the 24 + 24
    logits of features 4g + 2*half and 4g + 2*half + 1."""
    i = torch.arange(32)
    half = (i >> 2) & 1
    q = ((i >> 3) << 2) | (i & 3)
    tiles = num_transform * 24 // 32
    t = torch.arange(tiles)[:, None]
    idx = (t % 3) * 16 + q[None, :]
    feat = 4 * (t // 3) + 2 * half[None, :] + idx // 24
    return (feat * 24 + idx % 24).reshape(-1)  # [tiles * 32]


def _k8_row_order_32(num_transform, tiles_per_group=2):
    """The same for features padded to 32 rows (10 bins: 29 logits): two tiles per group, the 32
    values a lane-half gets from them are the logits of feature 2g + half.  In general (`tiles_per_group` = T, round 4:
    any bin count from 2 to 16) a feature's 3 K - 1 logits are padded to 16 T rows and group g's T tiles give lane-half
    h the 16 T logits of feature 2g + h, tile t of the group holding logits 16 t .. 16 t + 15."""
    T = tiles_per_group
    i = torch.arange(32)
    half = (i >> 2) & 1
    q = ((i >> 3) << 2) | (i & 3)
    t = torch.arange(num_transform * T // 2)[:, None]  # T / 2 tiles per feature
    feat = 2 * (t // T) + half[None, :]
    return (feat * (16 * T) + (t % T) * 16 + q[None, :]).reshape(-1)  # [tiles * 32]


def whole_layer_bins(num_bins):
    """Bin counts the whole-layer kernels K8h / K8 are built for (linear tails): 2 .. 16, and 20, 24, 32"""
    return 2 <= num_bins <= 16 or num_bins in (20, 24, 32)


def final_rows_per_feature(params_per_feature):
    """Rows of the packed final layer per transformed feature in the whole-layer kernels: 8 bins (23 logits) -> 24, two
    features sharing three 32-row tiles; any other bin count -> 3 K - 1 padded to whole lane-half shares of 16."""
    P = params_per_feature
    return 24 if P == 23 else 16 * ((P + 15) // 16)


def split_bf16x3(w):
    """fp32 -> three bf16 tensors with w == hi + mid + lo up to 2^-25 |w| (round to nearest even)."""
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def pack_final_linear(weight, bias, num_transform, params_per_feature, split_bf16=False):
    """Re-tiles a Linear(H=128 -> d_t*23) for K7's MFMA A operand (layout in include/nflows_amd.h):
    each feature's 23 rows padded to 24, rows reordered per lane-half (`_k7_row_order`), then
    fp32 [tiles][16][64 lanes][4], or with split_bf16 bf16 [tiles][3][8][64 lanes][8]; bias in
    accumulator order [tiles][2 halves][16]."""
    dt, P = num_transform, params_per_feature
    H = weight.shape[1]
    order = _k7_row_order(dt).to(weight.device)
    w = weight.detach().view(dt, P, H)
    w = torch.cat((w, w.new_zeros(dt, 24 - P, H)), dim=1).reshape(dt * 24, H).index_select(0, order)
    b = torch.cat((bias.detach().view(dt, P), bias.new_zeros(dt, 24 - P)), dim=1).reshape(dt * 24)
    b = b.index_select(0, order)
    tiles = dt * 24 // 32
    # (tile, r, half, j4, q) -> (tile, j4, half, r, q); lane = half*32 + r
    if split_bf16:
        # (piece, tile, r, half, ks, j) -> (tile, piece, ks, half, r, j)
        wp = torch.stack(split_bf16x3(w)).view(3, tiles, 32, 2, 8, 8).permute(1, 0, 4, 3, 2, 5).contiguous()
    else:
        wp = w.view(tiles, 32, 2, 16, 4).permute(0, 3, 2, 1, 4).contiguous()
    # row i of a tile sits in accumulator register q = 4*(i//8) + i%4 of lane-half (i//4)%2
    bp = b.view(tiles, 4, 2, 4).permute(0, 2, 1, 3).contiguous()
    return wp, bp


def _k8_column_order():
    """Input feature consumed at (k-step ks, lane-half hf, element j) of a GEMM whose input is the
    previous layer's accumulator tiles: tile ks//2, register 8*(ks%2) + j of lane-half hf holds
    feature 32*(ks//2) + 16*(ks%2) + 8*(j//4) + 4*hf + j%4 (MFMA 32x32 C/D layout)."""
    ks = torch.arange(8)[:, None, None]
    hf = torch.arange(2)[None, :, None]
    j = torch.arange(8)[None, None, :]
    return (32 * (ks // 2) + 16 * (ks % 2) + 8 * (j // 4) + 4 * hf + j % 4).reshape(-1)  # [ks][hf][j]


def _k8s_column_order():
    """K8s (16-sample tiles, MFMA 16x16x32): input feature at MFMA k position 32 S + 8 g + j of a GEMM whose input
    is the previous layer's accumulator tiles -- lane group g holds features 4 g + i (i = 0..3) of every 16-feature
    tile, k-step S is made of tiles 2 S (j < 4) and 2 S + 1 (j >= 4)."""
    S = torch.arange(4)[:, None, None]
    g = torch.arange(4)[None, :, None]
    j = torch.arange(8)[None, None, :]
    return (32 * S + 16 * (j // 4) + 4 * g + j % 4).reshape(-1)  # [S][g][j]


def _k8s_row_order(num_transform):
    """K8s: for every row of the packed (tile-major, 16-row tiles) final weight, which row of the feature-major
    padded matrix [d_t, 24] it is: the six tiles of a group G of four features give lane group g (rows 4 g + i
    of every tile) the 24 logits of feature 4 G + g, tile tau holding logits 4 tau + i."""
    tiles = num_transform * 24 // 16
    t = torch.arange(tiles)[:, None]
    m = torch.arange(16)[None, :]
    feat = 4 * (t // 6) + m // 4
    return (feat * 24 + 4 * (t % 6) + m % 4).reshape(-1)  # [tiles * 16]


def _bias_accumulator_order(b):
    """[tiles*32] -> [tiles][2 lane-halves][16]: row i of a tile sits in accumulator register
    q = 4*(i//8) + i%4 of lane-half (i//4)%2."""
    return b.view(-1, 4, 2, 4).permute(0, 2, 1, 3).reshape(-1)


def _pad_to(t, rows=None, cols=None):
    """Zero-pads a 2-D weight (or a 1-D bias with `rows`) up to the kernels' fixed hidden width: hidden
    units beyond the network's own have zero weights and biases on both sides, relu(0) = 0, so they
    change nothing -- a conditioner narrower than 128 runs in the 128-wide kernels as it is."""
    if t.dim() == 1:
        return t if rows is None or t.shape[0] == rows else torch.cat((t, t.new_zeros(rows - t.shape[0])))
    if rows is not None and t.shape[0] < rows:
        t = torch.cat((t, t.new_zeros(rows - t.shape[0], t.shape[1])), dim=0)
    if cols is not None and t.shape[1] < cols:
        t = torch.cat((t, t.new_zeros(t.shape[0], cols - t.shape[1])), dim=1)
    return t


def _batch_norm_affine(bn):
    """Eval-mode BatchNorm1d as y = a * x + c (float64), or None when it normalises with batch statistics."""
    if bn.training or not bn.track_running_stats or bn.running_mean is None or bn.running_var is None:
        return None
    a = (bn.running_var.detach().double() + bn.eps).rsqrt()
    if bn.weight is not None:
        a = a * bn.weight.detach().double()
    c = -bn.running_mean.detach().double() * a
    if bn.bias is not None:
        c = c + bn.bias.detach().double()
    return a, c


class _FoldedLayer:
    def __init__(self, weight, bias):
        self.weight, self.bias = weight.float(), bias.float()


class _FoldedBlock:
    context_layer = None

    def __init__(self, linear_layers):
        self.linear_layers = linear_layers


class _FoldedNet:
    """What the packers read of a ResidualNet (initial_layer, blocks[*].linear_layers, final_layer, widths)."""
    context_features = None

    def __init__(self, initial_layer, blocks, final_layer, hidden_features):
        self.initial_layer, self.blocks, self.final_layer = initial_layer, blocks, final_layer
        self.hidden_features = hidden_features


def fold_batch_norm(net):
    """A ResidualNet whose blocks use batch norm (resnet.py:24-27, :41-47), in eval mode, as an equivalent net
    WITHOUT it -- weights and biases only, nothing for the kernels to do.  Per block, with BN_i(x) = a_i x + c_i
    (a = gamma / sqrt(running_var + eps), c = beta - running_mean a):
      * BN_1 follows linear_0 directly:            W0' = diag(a_1) W0 diag(a_0),  b0' = a_1 b0 + c_1;
      * BN_0 sits between the residual stream and the ReLU: for a_0 > 0, relu(a_0 x + c_0) = a_0 relu(x + d),
        d = c_0 / a_0 -- a_0 goes into W0' (above) and the shift into the STREAM: the kernels carry s_k = x_k + d_k
        instead of x_k, which costs the bias of the layer that produces x_k + d_k and returns - d_k in the bias of
        the block's second Linear (whose sum x_k + ... would otherwise inherit it): b1' = b1 - d_k + d_{k+1}, the
        initial layer's b' = b + d_0, d past the last block = 0 (the final layer reads the stream as it is,
        resnet.py:99).
    Returns `net` itself when no block uses batch norm, None when the fold does not exist: a block in training
    mode (batch statistics), a_0 <= 0 or non-finite somewhere, or a context (the GLU gate multiplies b1,
    resnet.py:49-50).  Sums in float64, one rounding to float32 at the end.  Reads the device (one
    synchronising check): callers cache the result per weight key."""
    blocks = list(net.blocks)
    if not any(getattr(b, "use_batch_norm", False) for b in blocks):
        return net
    if getattr(net, "context_features", None) is not None:
        return None
    H = net.initial_layer.weight.shape[0]
    dev = net.initial_layer.weight.device
    zeros = torch.zeros(H, dtype=torch.float64, device=dev)
    folded, shifts, ok = [], [], []
    for b in blocks:
        w0, b0 = b.linear_layers[0].weight.detach().double(), b.linear_layers[0].bias.detach().double()
        w1, b1 = b.linear_layers[1].weight.detach().double(), b.linear_layers[1].bias.detach().double()
        d = zeros
        if getattr(b, "use_batch_norm", False):
            bn0, bn1 = _batch_norm_affine(b.batch_norm_layers[0]), _batch_norm_affine(b.batch_norm_layers[1])
            if bn0 is None or bn1 is None:
                return None
            (a0, c0), (a1, c1) = bn0, bn1
            d = c0 / a0
            ok.append((a0 > 0).all() & torch.isfinite(d).all())
            w0 = a1[:, None] * w0 * a0[None, :]
            b0 = a1 * b0 + c1
        shifts.append(d)
        folded.append([w0, b0, w1, b1])
    if ok and not bool(torch.stack(ok).all()):
        return None
    shifts.append(zeros)
    for k, f in enumerate(folded):
        f[3] = f[3] - shifts[k] + shifts[k + 1]
    initial = _FoldedLayer(net.initial_layer.weight.detach(), net.initial_layer.bias.detach().double() + shifts[0])
    return _FoldedNet(initial, [_FoldedBlock([_FoldedLayer(w0, b0), _FoldedLayer(w1, b1)]) for w0, b0, w1, b1 in folded],
                      net.final_layer, net.hidden_features)


def _initial_weight(net, pad_identity_to):
    """The initial layer's weight [128, identity features (+ context features)], hidden rows zero-padded;
    with `pad_identity_to` zero columns stand in for the run's surplus identity features (fused_geometry),
    in front of the context columns (resnet.py:93-94: the context follows the inputs)."""
    wi = _pad_to(net.initial_layer.weight.detach().float(), rows=128)
    ce = getattr(net, "context_features", None) or 0
    di = wi.shape[1] - ce
    if pad_identity_to is not None and pad_identity_to > di:
        wi = torch.cat((wi[:, :di], wi.new_zeros(128, pad_identity_to - di), wi[:, di:]), dim=1)
    return wi


def pack_resnet_conditioner(net, num_transform, params_per_feature, log2e=False, pad_transform_to=None,
                            pad_identity_to=None):
    """Packs a ResidualNet (initial_layer, blocks[*].linear_layers[0,1], final_layer) for K8
    (layout in include/nflows_amd.h): every weight as split-bf16 triples in 12 KB stages, in the
    order the kernel consumes them, all biases in accumulator order, and the 1/sqrt(hidden) scale
    of the width / height logits (coupling.py:554-556) folded into the final layer's rows
    (times log2(e) with `log2e`).  Returns (weights [stages, 768*8] bf16, biases fp32)."""
    dt, P = num_transform, params_per_feature
    K = (P + 1) // 3
    dev = net.final_layer.weight.device
    order_k = _k8_column_order().to(dev)

    def pieces(w):
        return torch.stack(split_bf16x3(w))  # [3, ...]

    stages, biases = [], []
    H = net.initial_layer.weight.shape[0]                      # <= 128: narrower nets are zero-padded
    wi = _initial_weight(net, pad_identity_to)
    di = wi.shape[1]                                           # identity features (+ context features)
    init_ks = 4 if di > 32 else 2                              # k-steps of the initial layer (the run's count)
    wi = torch.cat((wi, wi.new_zeros(128, 16 * init_ks - di)), dim=1)  # k = ks*16 + hf*8 + j
    # (p, t, i, ks, hf, j) -> (ks, t, p, hf, i, j)
    stages.append(pieces(wi).view(3, 4, 32, init_ks, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(init_ks, -1))
    biases.append(_bias_accumulator_order(_pad_to(net.initial_layer.bias.detach().float(), rows=128)))
    for block in net.blocks:
        for lin in block.linear_layers:
            w = _pad_to(lin.weight.detach().float(), rows=128, cols=128).index_select(1, order_k)  # columns in (ks, hf, j) order
            # k-major: (p, t, i, ks, hf, j) -> (ks, t, p, hf, i, j), one stage per k-step
            stages.append(pieces(w).view(3, 4, 32, 8, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(8, -1))
            biases.append(_bias_accumulator_order(_pad_to(lin.bias.detach().float(), rows=128)))
        if getattr(block, "context_layer", None) is not None:
            # the GLU gate's Linear.  Up to 16 context features: ONE k-major stage like a k-step of the other
            # Linears, (p, t, i, hf, j) -> (t, p, hf, i, j); more: tile-major, one stage per 32-row tile,
            # (p, t, i, k4, hf, j) -> (t, p, k4, hf, i, j)
            if block.context_layer.weight.shape[1] <= 16:
                wc = _pad_to(block.context_layer.weight.detach().float(), rows=128, cols=16)
                stages.append(pieces(wc).view(3, 4, 32, 2, 8).permute(1, 0, 3, 2, 4).reshape(1, -1))
            else:
                wc = _pad_to(block.context_layer.weight.detach().float(), rows=128, cols=64)
                stages.append(pieces(wc).view(3, 4, 32, 4, 2, 8).permute(1, 0, 3, 4, 2, 5).reshape(4, -1))
            biases.append(_bias_accumulator_order(_pad_to(block.context_layer.bias.detach().float(), rows=128)))
    scale = torch.ones(P, dtype=torch.float64, device=dev)
    scale[:2 * K] = (math.log2(math.e) if log2e else 1.0) / math.sqrt(net.hidden_features)
    wf = net.final_layer.weight.detach().double().view(dt, P, H)
    wf = torch.cat((wf, wf.new_zeros(dt, P, 128 - H)), dim=2) if H < 128 else wf
    wf = (wf * scale[None, :, None]).float()
    bf = (net.final_layer.bias.detach().double().view(dt, P) * scale[None, :]).float()
    if pad_transform_to is not None and pad_transform_to > dt:   # surplus features: zero rows (fused_geometry)
        wf = torch.cat((wf, wf.new_zeros(pad_transform_to - dt, P, 128)), dim=0)
        bf = torch.cat((bf, bf.new_zeros(pad_transform_to - dt, P)), dim=0)
        dt = pad_transform_to
    R = final_rows_per_feature(P)  # rows per feature after padding (8 bins: 23 -> 24; 10 bins: 29 -> 32; ...)
    order_r = (_k7_row_order(dt) if P == 23 else _k8_row_order_32(dt, R // 16)).to(dev)
    wf = torch.cat((wf, wf.new_zeros(dt, R - P, 128)), dim=1).reshape(dt * R, 128)
    wf = wf.index_select(0, order_r).index_select(1, order_k)
    bf = torch.cat((bf, bf.new_zeros(dt, R - P)), dim=1).reshape(dt * R).index_select(0, order_r)
    tiles = dt * R // 32
    # (p, tile, i, hs, k4, hf, j) -> (tile, hs, p, k4, hf, i, j): two stages per tile
    stages.append(pieces(wf).view(3, tiles, 32, 2, 4, 2, 8).permute(1, 3, 0, 4, 5, 2, 6).reshape(tiles * 2, -1))
    biases.append(_bias_accumulator_order(bf))
    return torch.cat(stages, dim=0).contiguous(), torch.cat(biases).contiguous()


_TRAIN_ORDER_K = {}


def pack_resnet_hidden_train(w_in, b_in, block_params, final=None):
    """K14's packer as ONE launch (nfa_pack_resnet_hidden_train_f32): fp32 parameters on the device -> (forward stages,
    forward biases, backward stages, final-layer bias or None), the bytes of `pack_resnet_hidden_train_reference`.
    `final` = (weight [out, 128], bias [out]) appends the net's final Linear to the forward stream.  Runs inside
    every training step (the weights change), including captured ones."""
    dev = w_in.device
    if not (w_in.is_cuda and w_in.dtype == torch.float32):
        return pack_resnet_hidden_train_reference(w_in, b_in, block_params, final)
    di, nb = w_in.shape[1], len(block_params)
    init_ks = 4 if di > 32 else 2
    tiles = (di + 31) // 32
    flat = [w_in.detach().contiguous(), b_in.detach().contiguous()]
    for group in block_params:
        flat += [t.detach().contiguous() for t in group]
    fin = [t.detach().contiguous() for t in final] if final is not None else []
    for t in flat + fin:
        if t.dtype != torch.float32 or not t.is_cuda:
            raise TypeError("nflows_amd: the conditioner's parameters must be float32 tensors on the device")
    out = fin[0].shape[0] if fin else 0
    ft = (out + 31) // 32
    fwd = torch.empty(init_ks + 16 * nb + 2 * ft, 6144, dtype=torch.bfloat16, device=dev)
    bias = torch.empty(128 * (1 + 2 * nb), dtype=torch.float32, device=dev)
    fbias = torch.empty(32 * ft, dtype=torch.float32, device=dev) if fin else None
    # (with the final Linear: W_f^T's ceil(out / 16) k-major stages behind W_in^T's -- K14's backward kernel starts with them)
    bwd = torch.empty(16 * nb + 2 * tiles + (out + 15) // 16, 6144, dtype=torch.bfloat16, device=dev)
    ptrs = (ctypes.c_void_p * max(1, 4 * nb))(*[t.data_ptr() for t in flat[2:]])
    with torch.cuda.device(dev):
        rc = N.load().nfa_pack_resnet_hidden_train_f32(
            N.ptr(flat[0]), N.ptr(flat[1]), ptrs, N.ptr(fin[0]) if fin else None, N.ptr(fin[1]) if fin else None, out,
            di, w_in.shape[0], nb, N.ptr(fwd), N.ptr(bias), N.ptr(fbias), N.ptr(bwd), N.stream_handle(dev))
    N.check(rc)
    return fwd, bias, bwd, fbias


_TRAIN_ORDER_K = {}


def pack_resnet_hidden_train_reference(w_in, b_in, block_params, final=None):
    """Packs the hidden part of a ResidualNet (and optionally its final Linear) for K14
    (nfa_resnet_hidden_forward_f32 / _backward_f32; layout in include/nflows_amd.h) with tensor operations.
    w_in [128, d_i], b_in [128], block_params = [(W_0, b_0, W_1, b_1), ...] (all 128 wide), final = (W_f [out, 128],
    b_f [out]).  Returns (forward stages, forward biases, backward stages, final bias or None): the forward stream
    is the initial layer + W_0, W_1 per block (pack_resnet_conditioner's hidden layers) + W_f tile-major, the backward
    stream W_1^T, W_0^T per block from the last to the first, then W_in^T tile-major (rows padded to 32) and -- with
    `final` -- W_f^T k-major over its out_features columns in their natural order (ceil(out / 16) stages: the final
    Linear's input gradient, which the backward kernel computes first)."""
    dev = w_in.device
    order_k = _TRAIN_ORDER_K.get(dev)
    if order_k is None:
        order_k = _TRAIN_ORDER_K[dev] = _k8_column_order().to(dev)

    def pieces(w):
        return torch.stack(split_bf16x3(w))  # [3, ...]

    def kmajor(w):   # [128, 128], columns in accumulator order: (p, t, i, ks, hf, j) -> (ks, t, p, hf, i, j)
        return pieces(w.index_select(1, order_k)).view(3, 4, 32, 8, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(8, -1)

    def tilemajor(w, rows):   # [rows, 128] -> rows padded to 32 per tile, two stages per tile
        tiles = (rows + 31) // 32
        wp = torch.cat((w, w.new_zeros(tiles * 32 - rows, 128)), dim=0).index_select(1, order_k)
        # (p, tile, i, hs, k4, hf, j) -> (tile, hs, p, k4, hf, i, j)
        return pieces(wp).view(3, tiles, 32, 2, 4, 2, 8).permute(1, 3, 0, 4, 5, 2, 6).reshape(tiles * 2, -1)

    di = w_in.shape[1]
    init_ks = 4 if di > 32 else 2
    wi = _pad_to(w_in.detach().float(), rows=128)          # (hidden widths below 128: zero rows / columns)
    wi_padded = torch.cat((wi, wi.new_zeros(128, 16 * init_ks - di)), dim=1)   # k = ks*16 + hf*8 + j
    fwd = [pieces(wi_padded).view(3, 4, 32, init_ks, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(init_ks, -1)]
    bias = [_bias_accumulator_order(_pad_to(b_in.detach().float(), rows=128))]
    bwd = []
    for w0, b0, w1, b1 in block_params:
        w0, w1 = (_pad_to(w.detach().float(), rows=128, cols=128) for w in (w0, w1))
        fwd += [kmajor(w0), kmajor(w1)]
        bias += [_bias_accumulator_order(_pad_to(b.detach().float(), rows=128)) for b in (b0, b1)]
        bwd = [kmajor(w1.t()), kmajor(w0.t())] + bwd        # last block first
    bwd.append(tilemajor(wi.t(), di))
    fbias = None
    if final is not None:
        wf, bf = _pad_to(final[0].detach().float(), cols=128), final[1].detach().float()
        out = wf.shape[0]
        fwd.append(tilemajor(wf, out))
        fbias = _bias_accumulator_order(torch.cat((bf, bf.new_zeros((out + 31) // 32 * 32 - out)))).contiguous()
        kf = (out + 15) // 16
        wft = torch.cat((wf.t(), wf.new_zeros(128, 16 * kf - out)), dim=1)   # [128 units, k = ks*16 + hf*8 + j]
        bwd.append(pieces(wft).view(3, 4, 32, kf, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(kf, -1))
    return torch.cat(fwd, dim=0).contiguous(), torch.cat(bias).contiguous(), torch.cat(bwd, dim=0).contiguous(), fbias


def resnet_hidden_train_supported(batch, num_identity, hidden_features, num_blocks):
    """The shapes K14 takes (everything else keeps the eager path); hidden widths below 128 are zero-padded."""
    return (4 <= hidden_features <= 128 and hidden_features % 4 == 0 and 1 <= num_identity <= 64
            and 0 <= num_blocks <= 3 and batch > 0 and batch % 128 == 0)   # (identity features: padded to a multiple of 4)


def _train_mask_words(num_blocks, batch):
    """floats behind K14's saved planes: one 8-byte mask word per lane, 32-row wave tile and plane"""
    return 2 * num_blocks * batch * 4


def _saved_with_masks(saved, num_blocks, batch):
    """`saved` as the backward kernels want it: the forward kernel's buffer, whose storage carries the packed ReLU
    masks behind the planes (a clone or a slice of it does not)"""
    need = (2 * num_blocks * batch * 128 + _train_mask_words(num_blocks, batch)) * 4
    if saved.untyped_storage().nbytes() - saved.storage_offset() * saved.element_size() < need:
        raise ValueError("nflows_amd: `saved` must be the tensor nfa_resnet_hidden_forward_f32 filled (its storage holds "
                         "the packed ReLU masks behind the planes)")
    return saved


def resnet_hidden_forward(x, fwd_stages, fwd_bias, num_blocks, final_bias=None, out_features=0):
    """K14 forward: identity features [B, d_i] -> (hidden [B, 128], saved [2 num_blocks, B, 128], params [B, out]
    or None -- the conditioner's output when the final Linear was packed into the stream)."""
    N.require_device_f32("inputs", x, 2)
    x = x.detach().contiguous()
    B, di = x.shape
    dev = x.device
    hidden = torch.empty(B, 128, dtype=torch.float32, device=dev)
    # the 2 nb planes and, behind them, the packed ReLU masks the backward kernel reads (16 bytes per row and plane)
    planes = 2 * num_blocks * B * 128
    saved_all = torch.empty(planes + _train_mask_words(num_blocks, B), dtype=torch.float32, device=dev)
    saved = saved_all[:planes].view(2 * num_blocks, B, 128)
    params = torch.empty(B, out_features, dtype=torch.float32, device=dev) if final_bias is not None else None
    with torch.cuda.device(dev):
        rc = N.load().nfa_resnet_hidden_forward_f32(N.ptr(x), N.ptr(fwd_stages), N.ptr(fwd_bias),
                                                    N.ptr(saved_all) if num_blocks else None, N.ptr(hidden),
                                                    N.ptr(final_bias), N.ptr(params),
                                                    out_features if final_bias is not None else 0, B, di, 128,
                                                    num_blocks, N.stream_handle(dev))
    N.check(rc)
    return hidden, saved, params


def resnet_hidden_backward(grad_hidden, bwd_stages, saved, num_identity):
    """K14 backward: d loss / d hidden [B, 128] -> (d loss / d identity features [B, d_i], grads [2 nb, B, 128])."""
    N.require_device_f32("grad_hidden", grad_hidden, 2)
    g = grad_hidden.detach().contiguous()
    B = g.shape[0]
    dev = g.device
    nb = saved.shape[0] // 2
    if nb:
        _saved_with_masks(saved, nb, B)
    grads = torch.empty(2 * nb, B, 128, dtype=torch.float32, device=dev)
    gx = torch.empty(B, num_identity, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = N.load().nfa_resnet_hidden_backward_f32(N.ptr(g), N.ptr(bwd_stages), N.ptr(saved) if nb else None,
                                                     N.ptr(grads) if nb else None, N.ptr(gx), B, num_identity, 128, nb,
                                                     N.stream_handle(dev))
    N.check(rc)
    return gx, grads


# K14's backward kernel starting from d loss / d params (the final Linear's input gradient inside it; A/B switch)
FUSED_FINAL_DGRAD = os.environ.get("NFA_K14_FINAL_DGRAD", "1") != "0"


def resnet_backward(grad_params, bwd_stages, saved, num_identity):
    """K14 backward from the conditioner's OUTPUT gradient (round 4, `nfa_resnet_backward_f32`): d loss / d params
    [B, out] -> (d loss / d identity features [B, d_i], grads [2 nb, B, 128], d loss / d hidden [B, 128]); `bwd_stages`
    packed with the final Linear (its W_f^T stages behind W_in^T's).  None when the kernel does not take the width."""
    N.require_device_f32("grad_params", grad_params, 2)
    g = grad_params.detach().contiguous()
    B, out = g.shape
    dev = g.device
    nb = saved.shape[0] // 2
    if out % 4 or out < 4:
        return None
    if nb:
        _saved_with_masks(saved, nb, B)
    grads = torch.empty(2 * nb, B, 128, dtype=torch.float32, device=dev)
    g_hidden = torch.empty(B, 128, dtype=torch.float32, device=dev)
    gx = torch.empty(B, num_identity, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = N.load().nfa_resnet_backward_f32(N.ptr(g), out, N.ptr(bwd_stages), N.ptr(saved) if nb else None,
                                              N.ptr(grads) if nb else None, N.ptr(g_hidden), N.ptr(gx), B, num_identity,
                                              128, nb, N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    return gx, grads, g_hidden


def _affine_row_order(num_transform, additive):
    """For every row of the packed output layer of K11: which conditioner output it is (-1 = zero
    padding).  Row i of tile t sits in accumulator register q = 4 (i // 8) + i % 4 of lane-half
    (i // 4) % 2; affine: q < 8 is the shift of feature 16 t + 8 half + q, q >= 8 the unconstrained
    scale of feature 16 t + 8 half + q - 8 (conditioner outputs [shift block | scale block],
    coupling.py:228-231); additive: the shift of feature 32 t + 16 half + q."""
    dt = num_transform
    per_tile = 32 if additive else 16
    tiles = (dt + per_tile - 1) // per_tile
    i = torch.arange(32)
    half = (i >> 2) & 1
    q = ((i >> 3) << 2) | (i & 3)
    t = torch.arange(tiles)[:, None]
    if additive:
        feat = 32 * t + 16 * half[None, :] + q[None, :]
        row = feat
    else:
        feat = 16 * t + 8 * half[None, :] + (q & 7)[None, :]
        row = torch.where((q >= 8)[None, :], dt + feat, feat)
    return torch.where(feat < dt, row, torch.full_like(row, -1)).reshape(-1)


def _conditioner_linears(net):
    """(input Linear, hidden Linears, output Linear) of a K11 conditioner: an MLP's layers, or (round 5) a ResidualNet's
    initial_layer, the blocks' linear_layers in order, final_layer (nn/nets/resnet.py:58-100) -- the same stream of
    stages, told apart by NFA_FLAG_RESIDUAL_BLOCKS."""
    if hasattr(net, "_input_layer"):
        return net._input_layer, list(net._hidden_layers), net._output_layer
    return net.initial_layer, [lin for block in net.blocks for lin in block.linear_layers], net.final_layer


def pack_mlp_conditioner(net, num_transform, additive=False):
    """Packs an MLP conditioner (nn/nets/mlp.py: _input_layer, _hidden_layers[*], _output_layer; all
    hidden widths 128) -- or a ResidualNet without a context (`_conditioner_linears`) -- for K11 (layout in
    include/nflows_amd.h).  Returns (weights [stages, 768*8] bf16, biases fp32)."""
    dt = num_transform
    input_layer, hidden_layers, output_layer = _conditioner_linears(net)
    dev = output_layer.weight.device
    order_k = _k8_column_order().to(dev)

    def pieces(w):
        return torch.stack(split_bf16x3(w))  # [3, ...]

    stages, biases = [], []
    wi = _pad_to(input_layer.weight.detach().float(), rows=128)   # hidden widths <= 128 are zero-padded
    di = wi.shape[1]
    init_ks = 4 if di > 32 else 2
    wi = torch.cat((wi, wi.new_zeros(128, 16 * init_ks - di)), dim=1)  # k = ks*16 + hf*8 + j
    stages.append(pieces(wi).view(3, 4, 32, init_ks, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(init_ks, -1))
    biases.append(_bias_accumulator_order(_pad_to(input_layer.bias.detach().float(), rows=128)))
    for lin in hidden_layers:
        w = _pad_to(lin.weight.detach().float(), rows=128, cols=128).index_select(1, order_k)
        stages.append(pieces(w).view(3, 4, 32, 8, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(8, -1))
        biases.append(_bias_accumulator_order(_pad_to(lin.bias.detach().float(), rows=128)))
    order_r = _affine_row_order(dt, additive).to(dev)
    wo = _pad_to(output_layer.weight.detach().float(), cols=128)
    bo = output_layer.bias.detach().float()
    wo = torch.cat((wo, wo.new_zeros(1, 128)), dim=0)     # row -1 = zero padding
    bo = torch.cat((bo, bo.new_zeros(1)))
    wf = wo.index_select(0, order_r % wo.shape[0]).index_select(1, order_k)
    bf = bo.index_select(0, order_r % bo.shape[0])
    tiles = order_r.numel() // 32
    # (p, tile, i, hs, k4, hf, j) -> (tile, hs, p, k4, hf, i, j): two stages per tile
    stages.append(pieces(wf).view(3, tiles, 32, 2, 4, 2, 8).permute(1, 3, 0, 4, 5, 2, 6).reshape(tiles * 2, -1))
    biases.append(_bias_accumulator_order(bf))
    return torch.cat(stages, dim=0).contiguous(), torch.cat(biases).contiguous()


def affine_flow_mlp(inputs, weights_packed, bias_packed, tables, num_transform, num_identity, num_hidden_layers,
                    scale_activation, inverse=False, accumulate_into=None, num_layers=1,
                    standard_normal_log_prob=False, pad=None, _pad_columns_count=0, residual_blocks=False):
    """K11 -- a run of affine / additive coupling layers with their MLP conditioners in one launch
    (weights / biases of the layers concatenated in execution order, tables from
    `flow_layer_tables`).  `residual_blocks` (round 5): the conditioners are ResidualNets, `num_hidden_layers` = twice
    their number of blocks.  Returns (outputs, logabsdet), or (None, log_prob) with
    `standard_normal_log_prob`; None when the shape is outside the fast path."""
    N.require_device_f32("inputs", inputs, 2)
    if pad is not None and inputs.shape[1] != pad[0]:   # rows padded to a multiple of four columns (tables too)
        out = affine_flow_mlp(_pad_columns(inputs, pad[0], pad[1]), weights_packed, bias_packed, tables, num_transform,
                              num_identity, num_hidden_layers, scale_activation, inverse, accumulate_into, num_layers,
                              standard_normal_log_prob, None, pad[0] - inputs.shape[1], residual_blocks)
        return _without_pad_columns(out, inputs.shape[1])
    if inputs.shape[0] % 128:
        return _on_full_blocks(
            lambda x_, acc_, ctx_: affine_flow_mlp(x_, weights_packed, bias_packed, tables, num_transform, num_identity,
                                                   num_hidden_layers, scale_activation, inverse, acc_, num_layers,
                                                   standard_normal_log_prob, None, _pad_columns_count, residual_blocks),
            inputs, accumulate_into)
    dev = inputs.device
    B, D = inputs.shape
    x = inputs.detach().contiguous()
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    flags, out = _density_epilogue(flags, standard_normal_log_prob, inverse, x, _pad_columns_count)
    if residual_blocks:
        flags |= N.FLAG_RESIDUAL_BLOCKS
    with torch.cuda.device(dev):
        rc = N.load().nfa_affine_flow_mlp_f32(
            N.ptr(x), N.ptr(weights_packed), N.ptr(bias_packed), N.ptr(tables), num_layers, N.ptr(out),
            N.ptr(lad), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128, num_hidden_layers,
            scale_activation, flags, N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    return out, lad


def split_f16x2(w):
    """fp32 -> two f16 tensors with w == hi + lo up to 2^-24 |w| while the low piece stays in the
    normal f16 range (round to nearest even both times)."""
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return hi, lo


def _f16_weight_scale(w):
    """Power of two T with max |w T| in [2^13, 2^14): the high pieces stay far from the f16
    overflow (65504), and every weight down to 2^-16 of the largest has a low piece in the normal
    f16 range, i.e. is represented to 2^-24 relative."""
    m = float(w.abs().max())
    if not math.isfinite(m) or m == 0.0:
        return 1.0
    return 2.0 ** (13 - math.floor(math.log2(m)))


def pack_resnet_conditioner_f16(net, num_transform, params_per_feature, act_scale=1.0, pad_transform_to=None,
                                pad_identity_to=None, tile16=False, colsplit=False):
    """Packs a ResidualNet for K8h (csrc/rqs_resnet_f16.hip; layout in include/nflows_amd.h).

    Weights: every GEMM's weights as TWO f16 pieces of (weight x T), T a power of two chosen per
    GEMM, in 8 KB stages of four (hi, lo) fragment pairs -- initial and hidden Linears k-major: one
    stage per k-step, pair g = output tile g; final Linear tile-major: two stages per 32-row tile,
    pair g of stage hs = k-step 4 hs + g.
    Parameters (fp32 words behind the 128 table words of a layer's parameter stage): per GEMM a
    header {out_scale, skip_scale, 0, 0} followed by the biases in accumulator order, pre-multiplied
    by the scale their accumulators carry.  Hidden activations' pieces live at scale S = `act_scale`
    (a power of two); the residual stream stays in fp32 accumulators at the scale of the GEMM that
    wrote it last, and a block's second Linear multiplies it by skip_scale = (its own scale) /
    (that scale) when it prepares its accumulators.  8 bins only.
    `tile16`: the same network for K8s (csrc/rqs_resnet_f16s.hip: 16-sample tiles on the 16x16x32 MFMA) -- the same
    stages and parameter words with the fragments, column / row orders and bias order of that tile shape (no
    context, 8 bins).  `colsplit` (with `tile16`): for K8c (csrc/rqs_resnet_f16c.hip, round 6: the four waves of a 64-row
    workgroup split every GEMM by columns) -- K8s's stages and words, except the final layer: per ROUND of four groups of
    four features twelve stages, stage 2 i + s holding for every wave w (pairs 2 w, 2 w + 1) k-steps 2 s, 2 s + 1 of tile
    i of group 4 r + w; groups beyond d_t / 4 are zero fragments.
    Returns (weights [stages, 4096] f16, parameter words fp32)."""
    dt, P = num_transform, params_per_feature
    K = (P + 1) // 3
    if P % 3 != 2 or not whole_layer_bins(K):
        raise ValueError("K8h packs linear-tail layers of 2 .. 16, 20, 24 or 32 bins")
    S = float(act_scale)
    dev = net.final_layer.weight.device
    order_k = (_k8s_column_order() if tile16 else _k8_column_order()).to(dev)
    if tile16 and (P != 23 or getattr(net, "context_features", None)):
        raise ValueError("K8s packs 8-bin layers without a context")
    # accumulator order of a GEMM's biases: K8s holds rows 4 g + i of a 16-row tile in lane group g: natural order
    bias_order = (lambda b: b) if tile16 else _bias_accumulator_order

    def pieces(w):
        return torch.stack(split_f16x2(w))  # [2, ...]

    def header(a, b):
        return torch.tensor([a, b, 0.0, 0.0], dtype=torch.float32, device=dev)

    stages, blob = [], []
    H = net.initial_layer.weight.shape[0]                      # <= 128: narrower nets are zero-padded
    ce = getattr(net, "context_features", None) or 0
    if ce:
        # with a context: [identity features, zero-padded to 32 | context, zero-padded to 32] (four k-steps)
        wi = _pad_to(net.initial_layer.weight.detach().float(), rows=128)
        di = wi.shape[1] - ce
        if di > 32 or ce > 32:
            raise ValueError("K8h with a context takes up to 32 identity and up to 32 context features")
        wi = torch.cat((wi[:, :di], wi.new_zeros(128, 32 - di), wi[:, di:], wi.new_zeros(128, 32 - ce)), dim=1)
        init_ks = 4
    else:
        wi = _initial_weight(net, pad_identity_to)
        di = wi.shape[1]
        init_ks = 4 if di > 32 else 2
        wi = torch.cat((wi, wi.new_zeros(128, 16 * init_ks - di)), dim=1)  # k = ks*16 + hf*8 + j
    T = _f16_weight_scale(wi)
    if tile16:
        # k-major, one stage per 32-wide k-step: (p, T, m, S, g, j) -> (S, T, p, g, m, j): pair T of stage S = the
        # (hi, lo) fragments of tile T, lane g * 16 + m holding k = 32 S + 8 g + j
        stages.append(pieces(wi * T).view(2, 8, 16, init_ks // 2, 4, 8).permute(3, 1, 0, 4, 2, 5).reshape(init_ks // 2, -1))
    else:
        # k-major: (p, t, i, ks, hf, j) -> (ks, t, p, hf, i, j), one 16 KB stage of 2 x 4 tile pairs per two k-steps
        stages.append(pieces(wi * T).view(2, 4, 32, init_ks, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(init_ks // 2, -1))
    blob += [header(S / T, 0.0), bias_order(_pad_to(net.initial_layer.bias.detach().float(), rows=128) * T)]
    stream_scale = T          # scale of the fp32 residual stream after the initial layer (inputs at scale 1)
    for block in net.blocks:
        for which, lin in enumerate(block.linear_layers):
            w = _pad_to(lin.weight.detach().float(), rows=128, cols=128).index_select(1, order_k)  # columns in (ks, hf, j) order
            T = _f16_weight_scale(w)
            if tile16:   # (p, T, m, S, g, j) -> (S, T, p, g, m, j), one stage per 32-wide k-step
                stages.append(pieces(w * T).view(2, 8, 16, 4, 4, 8).permute(3, 1, 0, 4, 2, 5).reshape(4, -1))
            else:        # k-major: (p, t, i, ks, hf, j) -> (ks, t, p, hf, i, j), one stage per two k-steps
                stages.append(pieces(w * T).view(2, 4, 32, 8, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(4, -1))
            lin_bias = _pad_to(lin.bias.detach().float(), rows=128)
            if which == 0:   # accumulators = S T (W relu(h) + b)
                blob += [header(1.0 / T, 0.0), bias_order(lin_bias * (S * T))]
            else:            # accumulators = S T (W relu(u) + b + h), h taken from the stream at stream_scale
                blob += [header(1.0 / T, S * T / stream_scale), bias_order(lin_bias * (S * T))]
                stream_scale = S * T
        if ce:
            # the gate's Linear on the context pieces (scale 1): two k-steps like the initial layer's, one stage;
            # accumulators = T_c (W_c context + b_c), the kernel takes sigmoid(accumulator / T_c) and adds
            # (second Linear's accumulators, without h) x that to the stream x skip_scale (resnet.py:46-52)
            wc = _pad_to(block.context_layer.weight.detach().float(), rows=128, cols=32)
            Tc = _f16_weight_scale(wc)
            stages.append(pieces(wc * Tc).view(2, 4, 32, 2, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(1, -1))
            blob += [header(1.0 / Tc, 0.0),
                     _bias_accumulator_order(_pad_to(block.context_layer.bias.detach().float(), rows=128) * Tc)]
    scale = torch.ones(P, dtype=torch.float64, device=dev)
    scale[:2 * K] = 1.0 / math.sqrt(net.hidden_features)
    wf = net.final_layer.weight.detach().double().view(dt, P, H)
    wf = torch.cat((wf, wf.new_zeros(dt, P, 128 - H)), dim=2) if H < 128 else wf
    wf = (wf * scale[None, :, None]).float()
    bf = (net.final_layer.bias.detach().double().view(dt, P) * scale[None, :]).float()
    if pad_transform_to is not None and pad_transform_to > dt:   # surplus features: zero rows (fused_geometry)
        wf = torch.cat((wf, wf.new_zeros(pad_transform_to - dt, P, 128)), dim=0)
        bf = torch.cat((bf, bf.new_zeros(pad_transform_to - dt, P)), dim=0)
        dt = pad_transform_to
    R = final_rows_per_feature(P)   # rows per feature after padding
    order_r = (_k8s_row_order(dt) if tile16 else _k7_row_order(dt) if P == 23 else _k8_row_order_32(dt, R // 16)).to(dev)
    wf = torch.cat((wf, wf.new_zeros(dt, R - P, 128)), dim=1).reshape(dt * R, 128)
    wf = wf.index_select(0, order_r).index_select(1, order_k)
    bf = torch.cat((bf, bf.new_zeros(dt, R - P)), dim=1).reshape(dt * R).index_select(0, order_r)
    T = _f16_weight_scale(wf)
    if tile16 and colsplit:
        groups = dt // 4
        rounds = (groups + 3) // 4
        pc = pieces(wf * T).view(2, groups, 6, 16, 4, 4, 8)                       # (p, G, i, m, S, g, j)
        if rounds * 4 > groups:
            pc = torch.cat((pc, pc.new_zeros(2, rounds * 4 - groups, 6, 16, 4, 4, 8)), dim=1)
        # (p, r, w, i, m, s, q, g, j) -> (r, i, s, w, q, p, g, m, j): pair 2 w + q of stage (r, i, s) = k-step 2 s + q
        stages.append(pc.view(2, rounds, 4, 6, 16, 2, 2, 4, 8).permute(1, 3, 5, 2, 6, 0, 7, 4, 8).reshape(rounds * 12, -1))
    elif tile16:
        # tile-major, two 16-row tiles per stage: (p, tile, m, S, g, j) -> (tile, S, p, g, m, j)
        tiles = dt * R // 16
        stages.append(pieces(wf * T).view(2, tiles, 16, 4, 4, 8).permute(1, 3, 0, 4, 2, 5).reshape(tiles // 2, -1))
    else:
        tiles = dt * R // 32
        stages.append(pieces(wf * T).view(2, tiles, 32, 2, 4, 2, 8).permute(1, 3, 4, 0, 5, 2, 6).reshape(tiles, -1))
    # the spline evaluation reads logits = accumulators x kappa, kappa = 1 / (S T)
    blob += [header(1.0 / (S * T), S * T), bias_order(bf * (S * T))]
    return torch.cat(stages, dim=0).contiguous(), torch.cat(blob).contiguous()


K8H_PARAM_STAGE_WORDS = 2048     # parameter words per parameter stage (its first 8 KB)
K8H_STAGE_HALVES = 8192          # f16 values of a 16 KB stage


def build_f16_stream(layer_packs, tables):
    """The stream K8h consumes for a run of layers, in 16 KB stages: per layer its parameter stage(s) --
    each carries 2048 words in its first 8 KB: 128 table words (int32: slots of the identity /
    transformed features, from `flow_layer_tables`) followed by the layer's parameter words -- and then its
    weight stages.  `layer_packs`: [(weights, parameter words)] from pack_resnet_conditioner_f16 in
    execution order; `tables`: int32 [(L + 1) * 128].  Returns (stream [stages, 8192] f16, parameter stages
    per layer, final table int32 [128]), or None when a weight or bias is not finite (callers run K8)."""
    L = len(layer_packs)
    # K8h applies ReLU with v_max_f32, which returns the other operand for a NaN: non-finite WEIGHTS would
    # not propagate the way torch.relu propagates them.  Such a run is left to the exact kernel (K8 keeps
    # NaNs through its ReLUs): no stream.  (One synchronising check per weight version.)
    finite = torch.stack([torch.isfinite(w).all() & torch.isfinite(prm).all() for w, prm in layer_packs]).all()
    if not bool(finite):
        return None
    words = 128 + layer_packs[0][1].numel()
    P = (words + K8H_PARAM_STAGE_WORDS - 1) // K8H_PARAM_STAGE_WORDS
    parts = []
    for l, (w, prm) in enumerate(layer_packs):
        block = torch.zeros(P * K8H_PARAM_STAGE_WORDS, dtype=torch.float32, device=w.device)
        block[:128] = tables[l * 128:(l + 1) * 128].contiguous().view(torch.float32)
        block[128:128 + prm.numel()] = prm
        stage = torch.zeros(P, K8H_STAGE_HALVES // 2, dtype=torch.float32, device=w.device)
        stage[:, :K8H_PARAM_STAGE_WORDS] = block.view(P, K8H_PARAM_STAGE_WORDS)
        parts.append(stage.view(torch.float16).view(P, K8H_STAGE_HALVES))
        parts.append(w)
    return torch.cat(parts, dim=0).contiguous(), P, tables[L * 128:(L + 1) * 128].contiguous()


def coupling_layer_tables(features, transform_idx, identity_idx, in_perm=None, out_scatter=None,
                          padded_features=None, padded_transform=None, padded_identity=None):
    """int32 [256] column bookkeeping of ONE layer for K8 (layout in include/nflows_amd.h), built on
    the device: the row tile's slot j holds input column j; [0, 64) slots of the identity features,
    [64, 128) slots of the transformed features (their results stay there), [128, 256) the slot that
    ends up at every output position."""
    return flow_layer_tables(features, [(transform_idx, identity_idx, in_perm, out_scatter)],
                             padded_features=padded_features, padded_transform=padded_transform,
                             padded_identity=padded_identity)


def flow_layer_tables(features, layers, padded_features=None, padded_transform=None, padded_identity=None):
    """Tables of a run of coupling layers executed back to back on one row tile (K8 with
    num_layers > 1).  `layers`: (transform_idx, identity_idx, in_perm, out_scatter) per layer in
    execution order; layer column c reads logical column in_perm[c] of its input and leaves its
    output at logical position out_scatter[c].  The tile never moves: `where[j]` tracks the slot
    holding logical column j, every layer reads / overwrites the slots of its features, and the
    last 128 entries say which slot ends up at which output position.  int32 [(L + 1) * 128].

    `padded_features` / `padded_transform` / `padded_identity` (see `fused_geometry`): the rows the kernel
    sees have `padded_features` columns (the extra ones pass through every layer and permutation unmoved) and
    every layer lists `padded_transform` transformed and `padded_identity` identity features, the extra ones
    all at the first pad column -- which holds a constant outside the spline's box."""
    dev = layers[0][0].device
    Dp = features if padded_features is None else padded_features
    extra = torch.arange(features, Dp, device=dev)
    where = torch.arange(Dp, device=dev)
    rows = []
    for tidx, iidx, perm, scat in layers:
        slot_of_column = where if perm is None else where[torch.cat((perm.to(dev), extra))]
        row = torch.zeros(128, dtype=torch.int64, device=dev)
        row[:iidx.numel()] = slot_of_column[iidx]
        row[64:64 + tidx.numel()] = slot_of_column[tidx]
        if padded_transform is not None and padded_transform > tidx.numel():
            row[64 + tidx.numel():64 + padded_transform] = slot_of_column[features]   # the spare column
        if padded_identity is not None and padded_identity > iidx.numel():
            row[iidx.numel():padded_identity] = slot_of_column[features]
        rows.append(row)
        if scat is None:
            where = slot_of_column
        else:
            where = torch.zeros_like(slot_of_column).index_copy_(0, torch.cat((scat.to(dev), extra)), slot_of_column)
    final = torch.zeros(128, dtype=torch.int64, device=dev)
    final[:Dp] = where
    rows.append(final)
    return torch.cat(rows).to(torch.int32)


def fused_geometry(features, layers, tail_bound):
    """What the whole-layer spline kernels are given for a run of layers over `features` columns, `layers` =
    [(transformed features, identity features)]: one geometry for the whole run, both counts of transformed
    features in multiples of four.  Returns (padded features, transformed features, identity features, pad
    value): rows are padded with columns holding a constant outside the spline's box; a layer with fewer
    transformed features than the run's count gets surplus ones -- zero rows in its packed final layer -- that
    all read the first pad column, find it outside the box and leave it as it is with a zero
    log-derivative; a layer with fewer identity features gets surplus ones that read the same column and
    meet zero weights in its initial layer.  A run of equal layers in multiples of four comes back unchanged."""
    dt4 = max((dt + 3) // 4 * 4 for dt, _ in layers)
    di_u = max(di for _, di in layers)
    need_spare = any(dt != dt4 or di != di_u for dt, di in layers)
    Dp = (features + (1 if need_spare else 0) + 3) // 4 * 4
    return Dp, dt4, di_u, 2.0 * float(tail_bound) + 1.0


def _pad_columns(inputs, padded_features, value):
    if inputs.shape[1] == padded_features:
        return inputs
    fill = inputs.new_full((inputs.shape[0], padded_features - inputs.shape[1]), value)
    return torch.cat((inputs, fill), dim=1)


def _on_full_blocks(run, inputs, accumulate_into, context=None):
    """The whole-layer kernels work on full 128-row blocks.  A ragged batch is padded with zero rows whose
    results are dropped (rows are independent: a row's result does not depend on the others in its block);
    `run(inputs, accumulate_into, context)` is the launch on a batch of full blocks."""
    B = inputs.shape[0]
    pad = (-B) % 128
    if pad == 0:
        return run(inputs, accumulate_into, context)
    padded = torch.cat((inputs, inputs.new_zeros(pad, inputs.shape[1])), dim=0)
    padded_context = None if context is None else torch.cat((context, context.new_zeros(pad, context.shape[1])), dim=0)
    result = run(padded, None, padded_context)
    if result is None:
        return None
    out, lad = result
    out = None if out is None else out[:B]
    lad = lad[:B]
    if accumulate_into is not None:
        accumulate_into += lad
        lad = accumulate_into
    return out, lad


def _density_epilogue(flags, standard_normal_log_prob, inverse, like, pad_columns=0):
    """`flags` and the outputs buffer for the whole-layer kernels: with `standard_normal_log_prob` the
    kernel's second result is the flow's log-density and z never leaves the chip.  `pad_columns`: trailing
    columns of the rows that are the host's padding (fused_geometry), not features of the density."""
    if not standard_normal_log_prob:
        return flags, torch.empty_like(like)
    if inverse:
        raise ValueError("the standard-normal epilogue belongs to the forward pass")
    if not 0 <= pad_columns <= 7:
        raise ValueError("at most seven pad columns")
    return flags | N.FLAG_STANDARD_NORMAL_LOG_PROB | N.FLAG_SKIP_OUTPUTS | (pad_columns << N.FLAG_PAD_COLUMNS_SHIFT), None


def _without_pad_columns(result, features):
    """(outputs, logabsdet) of a launch on padded rows -> the caller's columns ((None, log_prob) stays)."""
    if result is None or result[0] is None:
        return result
    return result[0][:, :features], result[1]


def rqs_coupling_resnet(inputs, weights_packed, bias_packed, tables, num_transform, num_identity, num_blocks,
                        spec, inverse=False, accumulate_into=None, log2e=False, num_layers=1,
                        standard_normal_log_prob=False, context=None, pad=None, _pad_columns_count=0, activation=0):
    """K8 -- ResidualNet conditioner + spline coupling layer in one kernel; with num_layers > 1 a
    whole run of such layers (weights / biases concatenated in execution order, tables from
    `flow_layer_tables`).  Returns (outputs, logabsdet), or (None, log_prob) with
    `standard_normal_log_prob` (Flow.log_prob with a StandardNormal base: flows/base.py:42-49);
    None when the shape is outside the fast path."""
    N.require_device_f32("inputs", inputs, 2)
    if pad is not None and inputs.shape[1] != pad[0]:
        # `pad` = (padded features, pad value) of `fused_geometry`: `num_transform`, the blobs and the tables
        # are the padded layer's; the pad columns come off the result again
        # (the density epilogue is told how many trailing columns are padding)
        out = rqs_coupling_resnet(_pad_columns(inputs, pad[0], pad[1]), weights_packed, bias_packed, tables,
                                  num_transform, num_identity, num_blocks, spec, inverse, accumulate_into, log2e,
                                  num_layers, standard_normal_log_prob, context, None, pad[0] - inputs.shape[1], activation)
        return _without_pad_columns(out, inputs.shape[1])
    if inputs.shape[0] % 128:
        return _on_full_blocks(
            lambda x_, acc_, ctx_: rqs_coupling_resnet(x_, weights_packed, bias_packed, tables, num_transform,
                                                       num_identity, num_blocks, spec, inverse, acc_, log2e,
                                                       num_layers, standard_normal_log_prob, ctx_, None,
                                                       _pad_columns_count, activation),
            inputs, accumulate_into, context)
    dev = inputs.device
    B, D = inputs.shape
    x = inputs.detach().contiguous()
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    flags, out = _density_epilogue(flags, standard_normal_log_prob, inverse, x, _pad_columns_count)
    if log2e:
        flags |= N.FLAG_LOGITS_LOG2E
    flags |= int(activation) << N.FLAG_ACTIVATION_SHIFT
    with torch.cuda.device(dev):
        if context is not None:   # conditioners with a context: [B, context_features] rows
            N.require_device_f32("context", context, 2)
            ctx = context.detach().contiguous()
            if ctx.shape[0] != B:
                raise ValueError("context must have one row per input row")
            rc = N.load().nfa_rqs_flow_resnet_context_f32(
                N.ptr(x), N.ptr(ctx), ctx.shape[1], N.ptr(weights_packed), N.ptr(bias_packed), N.ptr(tables),
                num_layers, N.ptr(out), N.ptr(lad), N.ptr(_status_word(dev)), B, D, num_transform, num_identity,
                128, num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev))
        elif capture_last_layer_logits.active is not None:
            capture = capture_last_layer_logits.active
            rc = N.load().nfa_rqs_flow_resnet_logits_f32(
                N.ptr(x), N.ptr(weights_packed), N.ptr(bias_packed), N.ptr(tables), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128, num_blocks,
                ctypes.byref(spec), flags, N.stream_handle(dev), N.ptr(capture.buffer(B, num_transform, dev)))
            if rc == N.OK:
                capture.finish(None)
        else:
            rc = N.load().nfa_rqs_flow_resnet_f32(
                N.ptr(x), N.ptr(weights_packed), N.ptr(bias_packed), N.ptr(tables), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128, num_blocks,
                ctypes.byref(spec), flags, N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return out, lad


_ACTIVATION_CODES = {id(torch.nn.functional.relu): N.ACTIVATION_RELU, id(torch.relu): N.ACTIVATION_RELU,
                     id(torch.nn.functional.leaky_relu): N.ACTIVATION_LEAKY_RELU, id(torch.nn.functional.elu): N.ACTIVATION_ELU,
                     id(torch.tanh): N.ACTIVATION_TANH, id(torch.nn.functional.tanh): N.ACTIVATION_TANH}


def activation_code(fn):
    """The whole-layer kernels' code of a residual block's `activation` (nn/nets/resnet.py:27), or None: the
    reference's default F.relu, and (round 4) F.leaky_relu / F.elu with their default parameters and tanh --
    recognised by identity (the module-level functions live as long as torch does: their ids are stable keys)."""
    return _ACTIVATION_CODES.get(id(fn))


K8S_ENABLED = os.environ.get("NFA_K8S", "1") != "0"
K8S_ALWAYS = os.environ.get("NFA_K8S", "1") == "2"     # (measurements: the 16-sample-tile kernel at every batch size)
K8C_ENABLED = os.environ.get("NFA_K8C", "1") != "0"    # the column-split form of K8s for batches of at most 64 rows per CU
K8C_ALWAYS = os.environ.get("NFA_K8C", "1") == "2"
_cu_counts = {}


def use_tile16(batch, num_bins, context, device, activation=0):
    """K8s (16-sample tiles, csrc/rqs_resnet_f16s.hip) serves the batches that give a CU at most ONE 128-row block
    -- K8h would run them one wave per SIMD (or leave CUs idle): 8 bins, no context.  `NFA_K8S=0` switches it off.
    Returns 0 (K8h), 1 (K8s) or 2 (round 6: K8c, csrc/rqs_resnet_f16c.hip, the column-split form: batches of at most 96
    rows per CU -- measured: 24 576 rows 0.875 ms against K8s's 0.936, 32 768 rows 1.15 against 1.01 --; `NFA_K8C=0` switches it off, the diagnostic bin capture has no twin of it)."""
    if not K8S_ENABLED or num_bins != 8 or context is not None or activation != N.ACTIVATION_RELU:
        return 0
    key = device.index if device.index is not None else torch.cuda.current_device()
    cus = _cu_counts.get(key)
    if cus is None:
        cus = _cu_counts[key] = torch.cuda.get_device_properties(key).multi_processor_count
    if K8C_ENABLED and capture_last_layer_bins.active is None and (K8C_ALWAYS or (batch + 95) // 96 <= cus):
        return 2
    return 1 if (K8S_ALWAYS or (batch + 127) // 128 <= cus) else 0


class capture_last_layer_bins:
    """Diagnostic (tests/test_gpu_bin_index.py): while active, launches of the f16 whole-layer kernels (K8h / K8s, 8 bins,
    ReLU, no context) go through their diagnostic twins (nfa_rqs_flow_resnet_f16x2[_tile16]_bins_f32) and leave the bin
    every spline evaluation of the run's LAST layer chose in `.bins` (int32 [rows, transformed features of the padded
    layer], -1 outside the box) together with the launch's `.redo` flags.  Not re-entrant, not for timed runs."""
    active = None

    def __enter__(self):
        self.bins = self.redo = None
        self.launches = 0
        capture_last_layer_bins.active = self
        return self

    def __exit__(self, *exc):
        capture_last_layer_bins.active = None
        return False


def rqs_coupling_resnet_f16(inputs, stream_f16, packed_exact, tables, num_transform, num_identity, num_blocks,
                            spec, inverse=False, accumulate_into=None, num_layers=1,
                            standard_normal_log_prob=False, pad=None, context=None, _pad_columns_count=0,
                            tile16=False, activation=0):
    """K8h -- the run of whole-layer kernels on the f16 matrix pipe (two f16 pieces per operand),
    followed by the exact kernel (three bf16 pieces, full fp32 range) on the row blocks the first
    pass gave up on: blocks with a non-finite result, i.e. an activation beyond the f16 range or
    non-finite inputs.  `stream_f16`: (stream, parameter stages per layer, final table) from
    `build_f16_stream`; `packed_exact`: (weights, biases) from pack_resnet_conditioner; `tables`:
    the run's `flow_layer_tables` (for the exact kernel).  `tile16`: `stream_f16` was packed for K8s
    (pack_resnet_conditioner_f16(tile16=True)) and the launch goes to the 16-sample-tile kernel; 2: packed for K8c
    (`colsplit=True`), the column-split form.
    Results as for `rqs_coupling_resnet`."""
    N.require_device_f32("inputs", inputs, 2)
    if pad is not None and inputs.shape[1] != pad[0]:   # (see rqs_coupling_resnet)
        out = rqs_coupling_resnet_f16(_pad_columns(inputs, pad[0], pad[1]), stream_f16, packed_exact, tables,
                                      num_transform, num_identity, num_blocks, spec, inverse, accumulate_into,
                                      num_layers, standard_normal_log_prob, None, context,
                                      pad[0] - inputs.shape[1], tile16, activation)
        return _without_pad_columns(out, inputs.shape[1])
    if inputs.shape[0] % 128:
        return _on_full_blocks(
            lambda x_, acc_, ctx_: rqs_coupling_resnet_f16(x_, stream_f16, packed_exact, tables, num_transform,
                                                           num_identity, num_blocks, spec, inverse, acc_, num_layers,
                                                           standard_normal_log_prob, None, ctx_, _pad_columns_count,
                                                           tile16, activation),
            inputs, accumulate_into, context)
    dev = inputs.device
    B, D = inputs.shape
    x = inputs.detach().contiguous()
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    flags, out = _density_epilogue(flags, standard_normal_log_prob, inverse, x, _pad_columns_count)
    flags |= int(activation) << N.FLAG_ACTIVATION_SHIFT
    global _last_redo
    redo = torch.empty(max(1, B // 128), dtype=torch.int32, device=dev)
    _last_redo = redo
    lib = N.load()
    stream, param_stages, final_table = stream_f16
    ctx = None
    if context is not None:   # conditioners with a context: [B, context_features] rows
        N.require_device_f32("context", context, 2)
        ctx = context.detach().contiguous()
        if ctx.shape[0] != B:
            raise ValueError("context must have one row per input row")
    capture = capture_last_layer_bins.active
    capture_logits = capture_last_layer_logits.active
    with torch.cuda.device(dev):
        if ctx is None and capture_logits is not None and not tile16:
            bins = torch.full((B, num_transform), -2, dtype=torch.int32, device=dev)
            rc = lib.nfa_rqs_flow_resnet_f16x2_logits_f32(
                N.ptr(x), N.ptr(stream), param_stages, N.ptr(final_table), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128,
                num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev), N.ptr(bins),
                N.ptr(capture_logits.buffer(B, num_transform, dev)))
            if rc == N.OK:
                capture_logits.finish(redo)
        elif ctx is None and capture is not None:
            entry = lib.nfa_rqs_flow_resnet_f16x2_tile16_bins_f32 if tile16 else lib.nfa_rqs_flow_resnet_f16x2_bins_f32
            capture.bins = torch.full((B, num_transform), -2, dtype=torch.int32, device=dev)
            capture.redo = redo
            capture.launches += 1
            rc = entry(
                N.ptr(x), N.ptr(stream), param_stages, N.ptr(final_table), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128,
                num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev), N.ptr(capture.bins))
        elif ctx is None:
            entry = (lib.nfa_rqs_flow_resnet_f16x2_colsplit_f32 if tile16 == 2 else
                     lib.nfa_rqs_flow_resnet_f16x2_tile16_f32 if tile16 else lib.nfa_rqs_flow_resnet_f16x2_f32)
            rc = entry(
                N.ptr(x), N.ptr(stream), param_stages, N.ptr(final_table), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128,
                num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev))
        else:
            rc = lib.nfa_rqs_flow_resnet_context_f16x2_f32(
                N.ptr(x), N.ptr(ctx), ctx.shape[1], N.ptr(stream), param_stages, N.ptr(final_table), num_layers,
                N.ptr(out), N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity,
                128, num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev))
        if rc == N.ERR_UNSUPPORTED:
            return None
        N.check(rc)
        if os.environ.get("NFA_K8H_NOREDO"):
            return out, lad
        if ctx is None:
            rc = lib.nfa_rqs_flow_resnet_redo_f32(
                N.ptr(x), N.ptr(packed_exact[0]), N.ptr(packed_exact[1]), N.ptr(tables), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128,
                num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev))
        else:
            rc = lib.nfa_rqs_flow_resnet_context_redo_f32(
                N.ptr(x), N.ptr(ctx), ctx.shape[1], N.ptr(packed_exact[0]), N.ptr(packed_exact[1]), N.ptr(tables),
                num_layers, N.ptr(out), N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform,
                num_identity, 128, num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev))
        N.check(rc)
    _after_spline(spec, inverse, dev)
    return out, lad


def split_f16x3(w):
    """fp32 -> three f16 tensors with w == hi + lo + r' 2^-8 EXACTLY while the last piece -- kept at 2^8 times its value --
    stays above f16's smallest subnormal (2^-24), round to nearest even every time (csrc/f16x3_gemm.hpp)."""
    hi = w.to(torch.float16)
    r1 = w - hi.float()
    lo = r1.to(torch.float16)
    r = ((r1 - lo.float()) * 256.0).to(torch.float16)
    return hi, lo, r


def _bf8_bytes(x):
    """fp32 / f16 values -> their bf8 (OCP e5m2) encodings as uint8, round to nearest even"""
    return x.float().to(torch.float8_e5m2).view(torch.uint8)


def _f16_bytes(x):
    return x.contiguous().view(torch.uint8)


K8X_ACT_SCALE = 16.0   # scale of the activations' f16 pieces in K8x: 24 bits kept down to |v| = 2^-13, overflow at |v| >= 4094.9


def _k8x_fragments(w, rows_per_tile=32):
    """`w` [tiles * 32, k-steps, 2 lane-halves, 8] fp32 (already x T, columns in MFMA order) -> per (tile, k-step) the 1 KB
    fragments of its f16 pieces and the 16-byte halves of the bf8 operand of every PAIR of k-steps:
      H, L   uint8 [tiles, k-steps, 64 lanes x 16 B]      lane = 32 half + row, its 8 f16 values
      X      uint8 [tiles, pairs, 2, 64 x 16 B]           per lane 32 bytes [bf8(hi) ks0 | bf8(r') ks0 | bf8(hi) ks1 | bf8(r') ks1],
                                                          split into bytes 0 .. 15 and 16 .. 31"""
    tiles, nks = w.shape[0] // 32, w.shape[1]
    hi, lo, r = split_f16x3(w)

    def lanes(t):   # [tiles*32, ks, hf, 8] -> [tiles, ks, hf, i, 8]
        return t.view(tiles, 32, nks, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    H = _f16_bytes(lanes(hi)).view(tiles, nks, 1024)
    L = _f16_bytes(lanes(lo)).view(tiles, nks, 1024)
    h8 = lanes(_bf8_bytes(hi))                         # [tiles, ks, hf, i, 8] uint8
    r8 = lanes(_bf8_bytes(r))
    pair = torch.cat((h8[:, 0::2], r8[:, 0::2], h8[:, 1::2], r8[:, 1::2]), dim=-1)     # [tiles, pairs, hf, i, 32]
    X = torch.stack((pair[..., :16], pair[..., 16:]), dim=2).reshape(tiles, nks // 2, 2, 1024)
    return H, L, X


def _k8x_kmajor_stages(w):
    """[128, 16 * NKS] (x T, columns in (ks, hf, j) order) -> uint8 [NKS stages, 12 KB]: per pair of k-steps two stages
    (tiles 0, 1 and tiles 2, 3), each [2 tiles][H0, L0, H1, L1, X lo, X hi]"""
    nks = w.shape[1] // 16
    H, L, X = _k8x_fragments(w.view(128, nks, 2, 8))
    stages = []
    for pr in range(nks // 2):
        for half in range(2):
            frags = []
            for t in (2 * half, 2 * half + 1):
                frags += [H[t, 2 * pr], L[t, 2 * pr], H[t, 2 * pr + 1], L[t, 2 * pr + 1], X[t, pr, 0], X[t, pr, 1]]
            stages.append(torch.cat(frags))
    return torch.stack(stages)


def _k8x_tilemajor_stages(w):
    """[tiles * 32, 128] (x T, rows / columns in kernel order) -> uint8 [2 tiles stages, 12 KB]: per tile two stages of four
    k-steps, each [H0, L0, H1, L1][H2, L2, H3, L3][X01 lo, X01 hi, X23 lo, X23 hi]"""
    tiles = w.shape[0] // 32
    H, L, X = _k8x_fragments(w.view(tiles * 32, 8, 2, 8))
    stages = []
    for t in range(tiles):
        for hs in range(2):
            k = 4 * hs
            stages.append(torch.cat([H[t, k], L[t, k], H[t, k + 1], L[t, k + 1], H[t, k + 2], L[t, k + 2], H[t, k + 3], L[t, k + 3],
                                     X[t, 2 * hs, 0], X[t, 2 * hs, 1], X[t, 2 * hs + 1, 0], X[t, 2 * hs + 1, 1]]))
    return torch.stack(stages)


def pack_resnet_conditioner_f16x3(net, num_transform, params_per_feature, act_scale=K8X_ACT_SCALE,
                                  pad_transform_to=None, pad_identity_to=None):
    """Packs a ResidualNet for K8x (csrc/rqs_resnet_f16x3.hip; layout in include/nflows_amd.h): K8's stage order, column /
    row rules and bias order (pack_resnet_conditioner); every weight x T -- T a power of two chosen per GEMM
    (_f16_weight_scale) -- as its f16 pieces hi, lo and the bf8 operand bytes of the two 2^-22-level products
    (csrc/f16x3_gemm.hpp); the biases pre-multiplied by the scale their accumulators carry (S T: activations' pieces live
    at scale S = `act_scale`), and per GEMM the pair {1 / T, T} ({1 / (S T), S T} for the final layer) the kernel takes the
    scales out with.  Bin counts of whole_layer_bins (final rows: final_rows_per_feature), no context.  Returns (weights [stages, 768 * 8] f16 -- 12 KB stages, raw bytes --, biases
    fp32, scales fp32 [(2 + 2 blocks) * 2])."""
    dt, P = num_transform, params_per_feature
    K = (P + 1) // 3
    if P % 3 != 2 or not whole_layer_bins(K) or getattr(net, "context_features", None):
        raise ValueError("K8x packs linear-tail layers of 2 .. 16, 20, 24 or 32 bins without a context")
    S = float(act_scale)
    if S <= 0 or math.frexp(S)[0] != 0.5:
        raise ValueError("act_scale must be a power of two")
    dev = net.final_layer.weight.device
    order_k = _k8_column_order().to(dev)
    stages, biases, scales = [], [], []
    H = net.initial_layer.weight.shape[0]                      # <= 128: narrower nets are zero-padded
    wi = _initial_weight(net, pad_identity_to)
    di = wi.shape[1]
    init_ks = 4 if di > 32 else 2
    wi = torch.cat((wi, wi.new_zeros(128, 16 * init_ks - di)), dim=1)  # k = ks*16 + hf*8 + j
    T = _f16_weight_scale(wi)
    stages.append(_k8x_kmajor_stages(wi * T))
    biases.append(_bias_accumulator_order(_pad_to(net.initial_layer.bias.detach().float(), rows=128) * (S * T)))
    scales += [1.0 / T, T]
    for block in net.blocks:
        for lin in block.linear_layers:
            w = _pad_to(lin.weight.detach().float(), rows=128, cols=128).index_select(1, order_k)  # columns in (ks, hf, j) order
            T = _f16_weight_scale(w)
            stages.append(_k8x_kmajor_stages(w * T))
            biases.append(_bias_accumulator_order(_pad_to(lin.bias.detach().float(), rows=128) * (S * T)))
            scales += [1.0 / T, T]
    scale = torch.ones(P, dtype=torch.float64, device=dev)
    scale[:2 * K] = 1.0 / math.sqrt(net.hidden_features)
    wf = net.final_layer.weight.detach().double().view(dt, P, H)
    wf = torch.cat((wf, wf.new_zeros(dt, P, 128 - H)), dim=2) if H < 128 else wf
    wf = (wf * scale[None, :, None]).float()
    bf = (net.final_layer.bias.detach().double().view(dt, P) * scale[None, :]).float()
    if pad_transform_to is not None and pad_transform_to > dt:   # surplus features: zero rows (fused_geometry)
        wf = torch.cat((wf, wf.new_zeros(pad_transform_to - dt, P, 128)), dim=0)
        bf = torch.cat((bf, bf.new_zeros(pad_transform_to - dt, P)), dim=0)
        dt = pad_transform_to
    R = final_rows_per_feature(P)   # 8 bins: 23 -> 24 (two features per three tiles); otherwise whole 16-row shares
    order_r = (_k7_row_order(dt) if P == 23 else _k8_row_order_32(dt, R // 16)).to(dev)
    wf = torch.cat((wf, wf.new_zeros(dt, R - P, 128)), dim=1).reshape(dt * R, 128)
    wf = wf.index_select(0, order_r).index_select(1, order_k)
    bf = torch.cat((bf, bf.new_zeros(dt, R - P)), dim=1).reshape(dt * R).index_select(0, order_r)
    T = _f16_weight_scale(wf)
    stages.append(_k8x_tilemajor_stages(wf * T))
    biases.append(_bias_accumulator_order(bf * (S * T)))
    scales += [1.0 / (S * T), S * T]
    return (torch.cat(stages, dim=0).contiguous().view(torch.float16), torch.cat(biases).contiguous(),
            torch.tensor(scales, dtype=torch.float32, device=dev))


def unpack_last_layer_logits(logits_packed, num_transform):
    """[rows, num_transform * 24] in the diagnostic twins' packed order (include/nflows_amd.h: entry 32 tile + 16 half +
    q = accumulator register q of lane-half `half` of tile `tile`) -> [rows, num_transform, 23] in the reference's order
    (the conditioner's output reshaped as coupling.py:550-552 does); the pad row of every feature is dropped."""
    rows = logits_packed.shape[0]
    tiles = num_transform * 24 // 32
    v = logits_packed.view(rows, tiles, 2, 4, 4)                          # (tile, half, q // 4, q % 4)
    packed_rows = v.permute(0, 1, 3, 2, 4).reshape(rows, tiles * 32)      # row i = 8 (q // 4) + 4 half + q % 4 of the tile
    order = _k7_row_order(num_transform).to(logits_packed.device)         # packed row -> feature-major row
    out = torch.empty_like(packed_rows).index_copy_(1, order, packed_rows)
    return out.view(rows, num_transform, 24)[..., :23]


class capture_last_layer_logits:
    """Diagnostic (tests/test_gpu_logits.py): while active, launches of the whole-layer kernels K8h / K8x / K8 (8 bins, ReLU,
    no context) go through their diagnostic twins (nfa_rqs_flow_resnet_*_logits_f32) and leave the logits of the run's
    LAST layer -- the final Linear's output as the spline evaluation reads it -- in `.logits` ([rows, transformed
    features of the padded layer, 23], width / height entries already divided by sqrt(hidden_features)) together with
    the launch's `.redo` flags (None for K8).  Not re-entrant, not for timed runs."""
    active = None

    def __enter__(self):
        self.logits = self.redo = None
        self.launches = 0
        capture_last_layer_logits.active = self
        return self

    def __exit__(self, *exc):
        capture_last_layer_logits.active = None
        return False

    def buffer(self, rows, num_transform, device):
        self._packed = torch.full((rows, num_transform * 24), float("nan"), dtype=torch.float32, device=device)
        self._dt = num_transform
        self.launches += 1
        return self._packed

    def finish(self, redo):
        self.logits = unpack_last_layer_logits(self._packed, self._dt)
        self.redo = redo


def rqs_coupling_resnet_f16x3(inputs, packed_f16x3, packed_exact, tables, num_transform, num_identity, num_blocks,
                              spec, inverse=False, accumulate_into=None, num_layers=1,
                              standard_normal_log_prob=False, pad=None, _pad_columns_count=0,
                              act_scale=K8X_ACT_SCALE):
    """K8x -- the run of whole-layer kernels on the f16 matrix pipe with THREE f16 pieces per operand (five
    products: operands at the reference's fp32 width), followed by the exact kernel (K8: three bf16 pieces, full
    fp32 range) on the row blocks the first pass gave up on (a value beyond the f16 range at scale `act_scale`, or
    non-finite inputs).  Bin counts: whole_layer_bins.  `packed_f16x3`: (weights, biases, scales) of the run's layers concatenated, from
    pack_resnet_conditioner_f16x3; `packed_exact`: (weights, biases) from pack_resnet_conditioner; `tables`: the
    run's `flow_layer_tables`.  Results as for `rqs_coupling_resnet`; None when the shape is outside the kernel's."""
    N.require_device_f32("inputs", inputs, 2)
    if pad is not None and inputs.shape[1] != pad[0]:   # (see rqs_coupling_resnet)
        out = rqs_coupling_resnet_f16x3(_pad_columns(inputs, pad[0], pad[1]), packed_f16x3, packed_exact, tables,
                                        num_transform, num_identity, num_blocks, spec, inverse, accumulate_into,
                                        num_layers, standard_normal_log_prob, None, pad[0] - inputs.shape[1], act_scale)
        return _without_pad_columns(out, inputs.shape[1])
    if inputs.shape[0] % 128:
        return _on_full_blocks(
            lambda x_, acc_, ctx_: rqs_coupling_resnet_f16x3(x_, packed_f16x3, packed_exact, tables, num_transform,
                                                             num_identity, num_blocks, spec, inverse, acc_, num_layers,
                                                             standard_normal_log_prob, None, _pad_columns_count,
                                                             act_scale),
            inputs, accumulate_into)
    dev = inputs.device
    B, D = inputs.shape
    x = inputs.detach().contiguous()
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    flags, out = _density_epilogue(flags, standard_normal_log_prob, inverse, x, _pad_columns_count)
    global _last_redo
    redo = torch.empty(max(1, B // 128), dtype=torch.int32, device=dev)
    _last_redo = redo
    lib = N.load()
    weights, biases, scales = packed_f16x3
    capture = capture_last_layer_logits.active
    with torch.cuda.device(dev):
        args = (N.ptr(x), N.ptr(weights), N.ptr(biases), N.ptr(scales), N.ptr(tables), num_layers, N.ptr(out),
                N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128, num_blocks,
                float(act_scale), ctypes.byref(spec), flags, N.stream_handle(dev))
        if capture is not None:
            rc = lib.nfa_rqs_flow_resnet_f16x3_logits_f32(*args, N.ptr(capture.buffer(B, num_transform, dev)))
        else:
            rc = lib.nfa_rqs_flow_resnet_f16x3_f32(*args)
        if rc == N.ERR_UNSUPPORTED:
            return None
        N.check(rc)
        if capture is not None:
            capture.finish(redo)
        if os.environ.get("NFA_K8H_NOREDO"):
            return out, lad
        rc = lib.nfa_rqs_flow_resnet_redo_f32(
            N.ptr(x), N.ptr(packed_exact[0]), N.ptr(packed_exact[1]), N.ptr(tables), num_layers, N.ptr(out),
            N.ptr(lad), N.ptr(redo), N.ptr(_status_word(dev)), B, D, num_transform, num_identity, 128,
            num_blocks, ctypes.byref(spec), flags, N.stream_handle(dev))
        N.check(rc)
    _after_spline(spec, inverse, dev)
    return out, lad


def rqs_coupling_fused_linear(inputs, hidden, weight_packed, bias_padded, transform_idx, spec,
                              inverse=False, in_perm=None, out_scatter=None, accumulate_into=None):
    """K7 -- final Linear of the conditioner + spline coupling layer in one kernel.  Returns None
    when the shape is outside the fast path (callers then run the GEMM and K1)."""
    N.require_device_f32("inputs", inputs, 2)
    N.require_device_f32("hidden", hidden, 2)
    dev = inputs.device
    B, D = inputs.shape
    tidx = _idx("transform_features", transform_idx, dev)
    perm = _idx("in_perm", in_perm, dev, D)
    scat = _idx("out_scatter", out_scatter, dev, D)
    x = inputs.detach().contiguous()
    h = hidden.detach().contiguous()
    out = torch.empty_like(x)
    lad, flags = _lad_buffer(accumulate_into, B, dev, inverse)
    if weight_packed.dtype == torch.bfloat16:
        flags |= N.FLAG_WEIGHTS_BF16X3
    with torch.cuda.device(dev):
        rc = N.load().nfa_rqs_coupling_fused_linear_f32(
            N.ptr(x), N.ptr(h), N.ptr(weight_packed), N.ptr(bias_padded), N.ptr(tidx), N.ptr(perm),
            N.ptr(scat), N.ptr(out), N.ptr(lad), N.ptr(_status_word(dev)), B, D, tidx.numel(),
            h.shape[1], ctypes.byref(spec), flags, N.stream_handle(dev))
    if rc == N.ERR_UNSUPPORTED:
        return None
    N.check(rc)
    _after_spline(spec, inverse, dev)
    return out, lad


def last_layer_kernel():
    """Name of the layer kernel this thread launched last (`nfa_last_layer_kernel`): the launchers pick the instance
    from the batch, the CU count and the LDS budget."""
    import ctypes
    buf = ctypes.create_string_buffer(192)
    N.load().nfa_last_layer_kernel(buf, 192)
    return buf.value.decode()
