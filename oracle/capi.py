"""ctypes/numpy bindings for oracle/libnfa_oracle.so.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package `nflows_amd` never does (tests/test_no_oracle_in_product.py enforces it).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnfa_oracle.so")

STATUS_OUTSIDE_DOMAIN = 1
STATUS_NEG_DISCRIMINANT = 2
STATUS_BAD_INDEX = 4


class RqsSpec(ctypes.Structure):
    _fields_ = [
        ("num_bins", ctypes.c_int32),
        ("tails", ctypes.c_int32),
        ("left", ctypes.c_double),
        ("right", ctypes.c_double),
        ("bottom", ctypes.c_double),
        ("top", ctypes.c_double),
        ("min_bin_width", ctypes.c_double),
        ("min_bin_height", ctypes.c_double),
        ("min_derivative", ctypes.c_double),
        ("softplus_beta", ctypes.c_double),
        ("tail_logit", ctypes.c_double),
        ("wh_divisor", ctypes.c_double),
    ]


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "nfa_oracle.c")
    ):
        subprocess.run(["make", "-C", _HERE, "-s", "clean"], check=True)
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def make_spec(num_bins, tails="linear", tail_bound=1.0, left=0.0, right=1.0, bottom=0.0, top=1.0,
              min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
              enable_identity_init=False, wh_divisor=0.0):
    """Mirrors the keyword arguments of the reference functionals
    (nflows/transforms/splines/rational_quadratic.py:13-25, :66-80)."""
    if tails == "linear":
        left, right, bottom, top = -tail_bound, tail_bound, -tail_bound, tail_bound
        t = 1
    elif tails is None:
        t = 0
    else:
        raise RuntimeError("{} tails are not implemented.".format(tails))
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    beta = float(np.log(2) / (1 - min_derivative)) if enable_identity_init else 1.0
    # rational_quadratic.py:34 -- numpy float64, later stored into an fp32 tensor
    tail_logit = float(np.log(np.exp(1 - min_derivative) - 1))
    return RqsSpec(int(num_bins), t, left, right, bottom, top, min_bin_width, min_bin_height,
                   min_derivative, beta, tail_logit, float(wh_divisor))


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    raise TypeError(dtype)


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def rqs_elementwise(x, uw, uh, ud, spec, inverse=False, return_bins=False):
    """x: [...]; uw, uh: [..., K]; ud: [..., K-1] (linear tails) or [..., K+1]."""
    dtype = x.dtype
    suf, ct = _dt(dtype)
    shape = x.shape
    K = spec.num_bins
    xf = np.ascontiguousarray(x.reshape(-1))
    uwf = np.ascontiguousarray(uw.reshape(-1, K), dtype=dtype)
    uhf = np.ascontiguousarray(uh.reshape(-1, K), dtype=dtype)
    nd = ud.shape[-1]
    assert nd >= (K - 1 if spec.tails == 1 else K + 1), (nd, K, spec.tails)
    udf = np.ascontiguousarray(ud.reshape(-1, nd), dtype=dtype) if nd else np.zeros((xf.size, 1), dtype)
    n = xf.size
    y = np.empty(n, dtype)
    lad = np.empty(n, dtype)
    bins = np.empty(n, np.int32)
    fn = getattr(lib(), "oracle_rqs_elementwise" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(xf, ct), _ptr(uwf, ct), ctypes.c_int64(K), _ptr(uhf, ct), ctypes.c_int64(K),
            _ptr(udf, ct), ctypes.c_int64(max(nd, 1) if nd else 1), ctypes.c_int(nd), ctypes.c_int64(n),
            ctypes.byref(spec), ctypes.c_int(int(inverse)), _ptr(y, ct), _ptr(lad, ct),
            _ptr(bins, ctypes.c_int32))
    out = (y.reshape(shape), lad.reshape(shape), st)
    if return_bins:
        out = out + (bins.reshape(shape),)
    return out


def rqs_knots(u, spec, axis=0):
    """`cumwidths` (axis 0) / `cumheights` (axis 1) of rational_quadratic.py:91-98 / :106-113 for logits u [..., K]:
    [..., K + 1] knots in the dtype of u (linear tails: the box is [-tail_bound, tail_bound]^2 as make_spec set it)."""
    dtype = u.dtype
    suf, ct = _dt(dtype)
    K = spec.num_bins
    uf = np.ascontiguousarray(u.reshape(-1, K), dtype=dtype)
    out = np.empty((uf.shape[0], K + 1), dtype)
    fn = getattr(lib(), "oracle_rqs_knots" + suf)
    fn.restype = None
    fn(_ptr(uf, ct), ctypes.c_int64(K), ctypes.c_int64(uf.shape[0]), ctypes.byref(spec), ctypes.c_int(int(axis)),
       _ptr(out, ct))
    return out.reshape(u.shape[:-1] + (K + 1,))


def linear_spline(x, unnormalized_pdf, spec, inverse=False):
    """splines/linear.py: x [...], unnormalized_pdf [..., K]; spec.tails = 1 -> unconstrained."""
    dtype = x.dtype
    suf, ct = _dt(dtype)
    K = spec.num_bins
    xf = np.ascontiguousarray(x.reshape(-1))
    pf = np.ascontiguousarray(unnormalized_pdf.reshape(-1, K), dtype=dtype)
    y = np.empty(xf.size, dtype)
    lad = np.empty(xf.size, dtype)
    fn = getattr(lib(), "oracle_linear_spline" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(xf, ct), _ptr(pf, ct), ctypes.c_int64(K), ctypes.c_int64(xf.size), ctypes.byref(spec),
            ctypes.c_int(int(inverse)), _ptr(y, ct), _ptr(lad, ct))
    return y.reshape(x.shape), lad.reshape(x.shape), st


def quadratic_spline(x, uw, uh, spec, inverse=False):
    """splines/quadratic.py: uw [..., K], uh [..., K-1] (boundary heights derived) or [..., K+1]."""
    dtype = x.dtype
    suf, ct = _dt(dtype)
    K = spec.num_bins
    nh = uh.shape[-1]
    xf = np.ascontiguousarray(x.reshape(-1))
    wf = np.ascontiguousarray(uw.reshape(-1, K), dtype=dtype)
    hf = np.ascontiguousarray(uh.reshape(-1, nh), dtype=dtype)
    y = np.empty(xf.size, dtype)
    lad = np.empty(xf.size, dtype)
    fn = getattr(lib(), "oracle_quadratic_spline" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(xf, ct), _ptr(wf, ct), ctypes.c_int64(K), _ptr(hf, ct), ctypes.c_int64(nh), ctypes.c_int(nh),
            ctypes.c_int64(xf.size), ctypes.byref(spec), ctypes.c_int(int(inverse)), _ptr(y, ct), _ptr(lad, ct))
    return y.reshape(x.shape), lad.reshape(x.shape), st


def cubic_spline(x, uw, uh, udl, udr, spec, inverse=False):
    """splines/cubic.py: uw, uh [..., K]; udl, udr [..., 1] boundary-derivative logits."""
    dtype = x.dtype
    suf, ct = _dt(dtype)
    K = spec.num_bins
    xf = np.ascontiguousarray(x.reshape(-1))
    wf = np.ascontiguousarray(uw.reshape(-1, K), dtype=dtype)
    hf = np.ascontiguousarray(uh.reshape(-1, K), dtype=dtype)
    lf = np.ascontiguousarray(udl.reshape(-1), dtype=dtype)
    rf = np.ascontiguousarray(udr.reshape(-1), dtype=dtype)
    y = np.empty(xf.size, dtype)
    lad = np.empty(xf.size, dtype)
    fn = getattr(lib(), "oracle_cubic_spline" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(xf, ct), _ptr(wf, ct), ctypes.c_int64(K), _ptr(hf, ct), ctypes.c_int64(K), _ptr(lf, ct),
            ctypes.c_int64(1), _ptr(rf, ct), ctypes.c_int64(1), ctypes.c_int64(xf.size), ctypes.byref(spec),
            ctypes.c_int(int(inverse)), _ptr(y, ct), _ptr(lad, ct))
    return y.reshape(x.shape), lad.reshape(x.shape), st


def rqs_coupling(x, params, transform_idx, spec, inverse=False, in_perm=None, out_scatter=None):
    dtype = x.dtype
    suf, ct = _dt(dtype)
    x = np.ascontiguousarray(x)
    params = np.ascontiguousarray(params, dtype=dtype)
    tidx = _i64(transform_idx)
    perm = _i64(in_perm)
    scat = _i64(out_scatter)
    B, D = x.shape
    dt = tidx.size
    out = np.empty_like(x)
    lad = np.empty(B, dtype)
    fn = getattr(lib(), "oracle_rqs_coupling" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(x, ct), _ptr(params, ct), _ptr(tidx, ctypes.c_int64),
            None if perm is None else _ptr(perm, ctypes.c_int64),
            None if scat is None else _ptr(scat, ctypes.c_int64), ctypes.c_int64(B),
            ctypes.c_int64(D), ctypes.c_int64(dt), ctypes.byref(spec), ctypes.c_int(int(inverse)),
            _ptr(out, ct), _ptr(lad, ct))
    return out, lad, st


AFFINE_DEFAULT, AFFINE_GENERAL, AFFINE_ADDITIVE, AFFINE_GIVEN_SCALE, AFFINE_SOFTPLUS = 0, 1, 2, 3, 4


def affine_coupling(x, params, transform_idx, activation=AFFINE_DEFAULT, inverse=False,
                    scale=None, in_perm=None, out_scatter=None):
    dtype = x.dtype
    suf, ct = _dt(dtype)
    x = np.ascontiguousarray(x)
    params = np.ascontiguousarray(params, dtype=dtype)
    tidx = _i64(transform_idx)
    perm = _i64(in_perm)
    scat = _i64(out_scatter)
    B, D = x.shape
    out = np.empty_like(x)
    lad = np.empty(B, dtype)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=dtype)
    fn = getattr(lib(), "oracle_affine_coupling" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(x, ct), _ptr(params, ct), None if sc is None else _ptr(sc, ct),
            _ptr(tidx, ctypes.c_int64), None if perm is None else _ptr(perm, ctypes.c_int64),
            None if scat is None else _ptr(scat, ctypes.c_int64),
            ctypes.c_int64(B), ctypes.c_int64(D), ctypes.c_int64(tidx.size),
            ctypes.c_int(activation), ctypes.c_int(int(inverse)), _ptr(out, ct), _ptr(lad, ct))
    return out, lad, st


def permute_cols(x, perm):
    suf, ct = _dt(x.dtype)
    x = np.ascontiguousarray(x)
    perm = _i64(perm)
    out = np.empty_like(x)
    fn = getattr(lib(), "oracle_permute_cols" + suf)
    fn.restype = ctypes.c_int
    st = fn(_ptr(x, ct), _ptr(perm, ctypes.c_int64), ctypes.c_int64(x.shape[0]),
            ctypes.c_int64(x.shape[1]), _ptr(out, ct))
    return out, st


def rowsum(x):
    suf, ct = _dt(x.dtype)
    x = np.ascontiguousarray(x.reshape(x.shape[0], -1))
    out = np.empty(x.shape[0], x.dtype)
    fn = getattr(lib(), "oracle_rowsum" + suf)
    fn.restype = None
    fn(_ptr(x, ct), ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]), _ptr(out, ct))
    return out


def standard_normal_log_prob(x):
    suf, ct = _dt(x.dtype)
    x = np.ascontiguousarray(x.reshape(x.shape[0], -1))
    out = np.empty(x.shape[0], x.dtype)
    fn = getattr(lib(), "oracle_standard_normal_log_prob" + suf)
    fn.restype = None
    fn(_ptr(x, ct), ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]), _ptr(out, ct))
    return out


def searchsorted(knots, x):
    suf, ct = _dt(knots.dtype)
    knots = np.ascontiguousarray(knots)
    x = np.ascontiguousarray(x, dtype=knots.dtype)
    idx = np.empty(x.size, np.int64)
    fn = getattr(lib(), "oracle_searchsorted" + suf)
    fn.restype = None
    fn(_ptr(knots, ct), ctypes.c_int64(knots.size), _ptr(x, ct), ctypes.c_int64(x.size),
       _ptr(idx, ctypes.c_int64))
    return idx.reshape(x.shape)


def affine_autoregressive(x, params, inverse=False):
    suf, ct = _dt(x.dtype)
    x = np.ascontiguousarray(x)
    params = np.ascontiguousarray(params, dtype=x.dtype)
    B, D = x.shape
    out = np.empty_like(x)
    lad = np.empty(B, x.dtype)
    fn = getattr(lib(), "oracle_affine_autoregressive" + suf)
    fn.restype = None
    fn(_ptr(x, ct), _ptr(params, ct), ctypes.c_int64(B), ctypes.c_int64(D), ctypes.c_int(int(inverse)),
       _ptr(out, ct), _ptr(lad, ct))
    return out, lad
