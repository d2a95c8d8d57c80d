# nflows/transforms/splines/_amd.py   (new file in the reference; ~60 lines)
import ctypes, os, numpy as np, torch

_lib = ctypes.CDLL(os.environ.get("NFLOWS_AMD_LIB", "libnflows_amd.so"))
_lib.nfa_strerror.restype = ctypes.c_char_p

class _Spec(ctypes.Structure):              # struct nfa_rqs_spec, include/nflows_amd.h
    _fields_ = [("num_bins", ctypes.c_int32), ("tails", ctypes.c_int32)] + [
        (n, ctypes.c_double) for n in ("left", "right", "bottom", "top", "min_bin_width",
                                       "min_bin_height", "min_derivative", "softplus_beta",
                                       "tail_logit", "wh_divisor")]

_status = {}
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())

def _spec(K, tail_bound, min_w, min_h, min_d, identity_init=False, wh_divisor=0.0):
    return _Spec(K, 1, -tail_bound, tail_bound, -tail_bound, tail_bound, min_w, min_h, min_d,
                 np.log(2) / (1 - min_d) if identity_init else 1.0,
                 np.log(np.exp(1 - min_d) - 1), wh_divisor)

def _word(device):
    return _status.setdefault(device, torch.zeros(1, dtype=torch.int32, device=device))

def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def unconstrained_rqs_hip(inputs, uw, uh, ud, inverse, tail_bound, min_w, min_h, min_d, identity_init):
    """body of unconstrained_rational_quadratic_spline (rational_quadratic.py:13-63) for HIP float32 tensors"""
    K = uw.shape[-1]
    spec = _spec(K, tail_bound, min_w, min_h, min_d, identity_init)
    x = inputs.contiguous().view(-1)
    n = x.numel()
    uw, uh, ud = (t.reshape(n, -1).contiguous() for t in (uw, uh, ud))
    y, lad = torch.empty_like(x), torch.empty_like(x)
    rc = _lib.nfa_rqs_elementwise_f32(
        _p(x), _p(uw), ctypes.c_int64(K), _p(uh), ctypes.c_int64(K), _p(ud), ctypes.c_int64(ud.shape[1]),
        ctypes.c_int32(ud.shape[1]), _p(y), _p(lad), None, _p(_word(x.device)), ctypes.c_int64(n),
        ctypes.byref(spec), ctypes.c_int32(int(inverse)), _stream())
    if rc != 0:
        raise RuntimeError(_lib.nfa_strerror(rc).decode())
    return y.view(inputs.shape), lad.view(inputs.shape)

def rqs_coupling_hip(inputs, transform_params, transform_features, num_bins, tail_bound, hidden_features,
                     inverse=False, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """coupling.py:82-98 / :111-128 after the conditioner call: split, _piecewise_cdf (with its / sqrt(hidden)),
    row-sum and scatter of a PiecewiseRationalQuadraticCouplingTransform(tails="linear") in one launch"""
    B, D = inputs.shape
    spec = _spec(num_bins, tail_bound, min_w, min_h, min_d, False, float(np.sqrt(hidden_features)))
    x, params = inputs.contiguous(), transform_params.contiguous()
    outputs, logabsdet = torch.empty_like(x), torch.empty(B, dtype=x.dtype, device=x.device)
    rc = _lib.nfa_rqs_coupling_f32(
        _p(x), _p(params), _p(transform_features), None, None, _p(outputs), _p(logabsdet), None, _p(_word(x.device)),
        ctypes.c_int64(B), ctypes.c_int32(D), ctypes.c_int32(transform_features.numel()), ctypes.byref(spec),
        ctypes.c_int32(1 if inverse else 0), _stream())
    if rc != 0:
        raise RuntimeError(_lib.nfa_strerror(rc).decode())
    return outputs, logabsdet
