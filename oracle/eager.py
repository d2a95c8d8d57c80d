"""oracle/eager.py -- TEST / BASELINE INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

A PyTorch-eager CPU restatement of the reference's hot path, op for op in aten (softmax,
cumsum, softplus, gather, log ... each as its own fp32 CPU kernel over [N, K] temporaries), so
that (a) its results are bit-identical to the reference's CPU path and (b) its running time is
the reference's CPU running time.  bench.py times it as the `cpu_baseline` ("port") on the GPU
box, where /root/reference does not exist; tests/test_oracle_golden.py pins it bit-for-bit
against the reference's outputs.  nflows_amd never imports this module.

Restated reference code (paths relative to /root/reference):
  nflows/transforms/splines/rational_quadratic.py:13-181   spline functionals
  nflows/utils/torchutils.py:134-136                        searchsorted
  nflows/transforms/coupling.py:73-130, 279-293, 549-582    coupling layer around the spline
  nflows/transforms/coupling.py:234-252                     affine coupling
  nflows/transforms/permutations.py:27-45                   permutation
  nflows/transforms/base.py:45-60, flows/base.py:42-49      cascade (both directions), log_prob
  nflows/transforms/autoregressive.py:38-41, 453-489        autoregressive RQ layer, forward
"""
import numpy as np
import torch
import torch.nn.functional as F


def _knots(logits, lo, hi, min_size):
    """softmax -> floor at min_size -> cumulative knots in [lo, hi] with exact end points
    (:91-98 / :106-113).  Returns (knots [.., K+1], sizes [.., K])."""
    K = logits.shape[-1]
    p = F.softmax(logits, dim=-1)
    p = min_size + (1 - min_size * K) * p
    c = torch.cumsum(p, dim=-1)
    c = F.pad(c, pad=(1, 0), mode="constant", value=0.0)
    c = (hi - lo) * c + lo
    c[..., 0] = lo
    c[..., -1] = hi
    return c, c[..., 1:] - c[..., :-1]


def _bin_index(knots, values, eps=1e-6):
    """torchutils.searchsorted (utils/torchutils.py:134-136), including its in-place nudge of
    the last knot (the caller never gathers that entry)."""
    knots[..., -1] += eps
    return torch.sum(values[..., None] >= knots, dim=-1) - 1


def rqs_constrained(x, uw, uh, ud, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0,
                    min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
                    enable_identity_init=False):
    """rational_quadratic_spline (:66-181) without the domain exception (callers here only
    pass in-domain values)."""
    K = uw.shape[-1]
    if min_bin_width * K > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * K > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    cw, w = _knots(uw, left, right, min_bin_width)
    beta = np.log(2) / (1 - min_derivative) if enable_identity_init else 1
    d = min_derivative + F.softplus(ud, beta=beta)
    ch, h = _knots(uh, bottom, top, min_bin_height)
    k = _bin_index(ch if inverse else cw, x)[..., None]

    def pick(t):
        return t.gather(-1, k)[..., 0]

    x_cw, x_w, x_ch = pick(cw), pick(w), pick(ch)
    x_delta = pick(h / w)
    x_d0, x_d1, x_h = pick(d), pick(d[..., 1:]), pick(h)
    if inverse:
        a = (x - x_ch) * (x_d0 + x_d1 - 2 * x_delta) + x_h * (x_delta - x_d0)
        b = x_h * x_d0 - (x - x_ch) * (x_d0 + x_d1 - 2 * x_delta)
        c = -x_delta * (x - x_ch)
        disc = b.pow(2) - 4 * a * c
        root = (2 * c) / (-b - torch.sqrt(disc))
        out = root * x_w + x_cw
        t1mt = root * (1 - root)
        den = x_delta + ((x_d0 + x_d1 - 2 * x_delta) * t1mt)
        dnum = x_delta.pow(2) * (x_d1 * root.pow(2) + 2 * x_delta * t1mt + x_d0 * (1 - root).pow(2))
        return out, -(torch.log(dnum) - 2 * torch.log(den))
    theta = (x - x_cw) / x_w
    t1mt = theta * (1 - theta)
    num = x_h * (x_delta * theta.pow(2) + x_d0 * t1mt)
    den = x_delta + ((x_d0 + x_d1 - 2 * x_delta) * t1mt)
    out = x_ch + num / den
    dnum = x_delta.pow(2) * (x_d1 * theta.pow(2) + 2 * x_delta * t1mt + x_d0 * (1 - theta).pow(2))
    return out, torch.log(dnum) - 2 * torch.log(den)


def rqs_unconstrained(x, uw, uh, ud, inverse=False, tail_bound=1.0, min_bin_width=1e-3,
                      min_bin_height=1e-3, min_derivative=1e-3, enable_identity_init=False):
    """unconstrained_rational_quadratic_spline with linear tails (:13-63): boolean-mask
    compaction of the in-interval elements exactly like the reference (same memory traffic)."""
    inside = (x >= -tail_bound) & (x <= tail_bound)
    outside = ~inside
    out = torch.zeros_like(x)
    lad = torch.zeros_like(x)
    ud = F.pad(ud, pad=(1, 1))
    const = np.log(np.exp(1 - min_derivative) - 1)
    ud[..., 0] = const
    ud[..., -1] = const
    out[outside] = x[outside]
    lad[outside] = 0
    if torch.any(inside):
        out[inside], lad[inside] = rqs_constrained(
            x[inside], uw[inside, :], uh[inside, :], ud[inside, :], inverse=inverse,
            left=-tail_bound, right=tail_bound, bottom=-tail_bound, top=tail_bound,
            min_bin_width=min_bin_width, min_bin_height=min_bin_height,
            min_derivative=min_derivative, enable_identity_init=enable_identity_init)
    return out, lad


def rq_coupling_layer(x, net, identity_idx, transform_idx, num_bins, tail_bound, hidden_features,
                      inverse=False, context=None):
    """CouplingTransform.forward/inverse + PiecewiseRationalQuadraticCouplingTransform with
    tails="linear" (coupling.py:73-130, 279-293, 549-582)."""
    ident = x[:, identity_idx]
    xt = x[:, transform_idx]
    params = net(ident, context)          # (resnet.py:92-100: context concatenated, GLU gate per block)
    b, dt = xt.shape
    params = params.reshape(b, dt, -1)
    uw = params[..., :num_bins]
    uh = params[..., num_bins:2 * num_bins]
    ud = params[..., 2 * num_bins:]
    if hidden_features:
        uw /= np.sqrt(hidden_features)
        uh /= np.sqrt(hidden_features)
    yt, lad = rqs_unconstrained(xt, uw, uh, ud, inverse=inverse, tail_bound=tail_bound)
    out = torch.empty_like(x)
    out[:, identity_idx] = ident
    out[:, transform_idx] = yt
    return out, torch.sum(lad, dim=[1])


def affine_coupling_layer(x, net, identity_idx, transform_idx, inverse=False):
    """AffineCouplingTransform with the default scale activation (coupling.py:224, 234-252)."""
    ident = x[:, identity_idx]
    xt = x[:, transform_idx]
    params = net(ident, None)
    dt = xt.shape[1]
    scale = torch.sigmoid(params[:, dt:] + 2) + 1e-3
    shift = params[:, :dt]
    log_scale = torch.log(scale)
    if inverse:
        yt = (xt - shift) / scale
        lad = -torch.sum(log_scale, dim=[1])
    else:
        yt = xt * scale + shift
        lad = torch.sum(log_scale, dim=[1])
    out = torch.empty_like(x)
    out[:, identity_idx] = ident
    out[:, transform_idx] = yt
    return out, lad


def additive_coupling_layer(x, net, identity_idx, transform_idx, inverse=False):
    """AdditiveCouplingTransform (coupling.py:255-269: `shift = transform_params`, scale = ones; the inherited
    _coupling_transform_forward / _inverse, :241-252, then give x * 1 + shift, (x - shift) / 1 and sum(log(1)) = 0)."""
    ident = x[:, identity_idx]
    xt = x[:, transform_idx]
    shift = net(ident, None)
    scale = torch.ones_like(shift)
    log_scale = torch.log(scale)
    if inverse:
        yt = (xt - shift) / scale
        lad = -torch.sum(log_scale, dim=[1])
    else:
        yt = xt * scale + shift
        lad = torch.sum(log_scale, dim=[1])
    out = torch.empty_like(x)
    out[:, identity_idx] = ident
    out[:, transform_idx] = yt
    return out, lad


def standard_normal_log_prob(z):
    """StandardNormal._log_prob (distributions/normal.py:23-33)."""
    log_z = torch.tensor(0.5 * z.shape[1] * np.log(2 * np.pi), dtype=torch.float64)
    return -0.5 * torch.sum(z ** 2, dim=[1]) - log_z


def ar_rq_layer_forward(x, made, num_bins, tail_bound):
    """MaskedPiecewiseRationalQuadraticAutoregressiveTransform.forward with tails="linear"
    (autoregressive.py:38-41, 453-489): one MADE pass, params viewed [B, D, P]; no 1/sqrt(hidden)
    scaling because MADE has no `hidden_features` attribute."""
    params = made(x, None)
    b, d = x.shape
    params = params.view(b, d, -1)
    uw = params[..., :num_bins]
    uh = params[..., num_bins:2 * num_bins]
    ud = params[..., 2 * num_bins:]
    if hasattr(made, "hidden_features"):
        uw /= np.sqrt(made.hidden_features)
        uh /= np.sqrt(made.hidden_features)
    y, lad = rqs_unconstrained(x, uw, uh, ud, inverse=False, tail_bound=tail_bound)
    return y, torch.sum(lad, dim=[1])


def ar_rq_layer_inverse(z, made, num_bins, tail_bound):
    """AutoregressiveTransform.inverse (autoregressive.py:43-52): D passes of the whole MADE over the
    outputs found so far, each followed by the elementwise spline inverse of ALL features (the last
    pass's results are returned)."""
    b, d = z.shape
    out = torch.zeros_like(z)
    lad = None
    for _ in range(d):
        params = made(out, None).view(b, d, -1)
        uw = params[..., :num_bins]
        uh = params[..., num_bins:2 * num_bins]
        ud = params[..., 2 * num_bins:]
        if hasattr(made, "hidden_features"):
            uw /= np.sqrt(made.hidden_features)
            uh /= np.sqrt(made.hidden_features)
        out, lad = rqs_unconstrained(z, uw, uh, ud, inverse=True, tail_bound=tail_bound)
    return out, torch.sum(lad, dim=[1])


def _layer(t, h, inverse, context=None):
    name = type(t).__name__
    if name.endswith("Permutation"):
        perm = torch.argsort(t._permutation) if inverse else t._permutation  # permutations.py:22-45
        h = torch.index_select(h, 1, perm)
        return h, h.new_zeros(h.shape[0])
    if name == "PiecewiseRationalQuadraticCouplingTransform":
        return rq_coupling_layer(h, t.transform_net, t.identity_features, t.transform_features,
                                 t.num_bins, t.tail_bound,
                                 getattr(t.transform_net, "hidden_features", None), inverse=inverse,
                                 context=context)
    if name == "AffineCouplingTransform":
        return affine_coupling_layer(h, t.transform_net, t.identity_features, t.transform_features,
                                     inverse=inverse)
    if name == "AdditiveCouplingTransform":
        return additive_coupling_layer(h, t.transform_net, t.identity_features, t.transform_features, inverse=inverse)
    if name == "MaskedPiecewiseRationalQuadraticAutoregressiveTransform":
        fn = ar_rq_layer_inverse if inverse else ar_rq_layer_forward
        return fn(h, t.autoregressive_net, t.num_bins, t.tail_bound)
    raise NotImplementedError(name + (" inverse" if inverse else ""))


def flow_transform(flow, x, inverse=False, context=None):
    """CompositeTransform.forward / .inverse (transforms/base.py:45-60) of a flow built with
    nflows_amd classes (used only for their parameters, buffers and conditioner modules, all on
    CPU): returns (outputs, total logabsdet).  `context`: already embedded."""
    total = x.new_zeros(x.shape[0])
    h = x
    layers = list(flow._transform._transforms)
    for t in (reversed(layers) if inverse else layers):
        h, lad = _layer(t, h, inverse, context)
        total += lad
    return h, total


def flow_log_prob(flow, x, context=None):
    """Flow._log_prob (flows/base.py:42-49): the context goes through the embedding net first."""
    embedded = flow._embedding_net(context) if context is not None else None
    z, total = flow_transform(flow, x, context=embedded)
    return standard_normal_log_prob(z) + total
