"""ctypes binding of libnflows_amd.so (the C ABI declared in include/nflows_amd.h).

PyTorch is used here only as the owner of device memory and HIP streams: every call hands raw
device pointers (`tensor.data_ptr()`) and the current stream handle to the library.  There is
no CPU or eager fallback: if the shared library is missing or the tensors are not on a HIP
device, the call raises.
"""
import ctypes
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# NFLOWS_AMD_LIB lets tools/k1_micro.py A/B-test alternative builds of the same C ABI
LIB_PATH = os.environ.get("NFLOWS_AMD_LIB") or os.path.join(_PKG_DIR, "libnflows_amd.so")

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_UNSUPPORTED = 2
ERR_MIN_BIN_WIDTH = 3
ERR_MIN_BIN_HEIGHT = 4
ERR_HIP = 5

STATUS_OUTSIDE_DOMAIN = 1
STATUS_NEG_DISCRIMINANT = 2
STATUS_BAD_INDEX = 4

FLAG_INVERSE, FLAG_ACCUMULATE_LOGABSDET, FLAG_WEIGHTS_BF16X3, FLAG_LOGITS_LOG2E = 1, 2, 4, 8
FLAG_STANDARD_NORMAL_LOG_PROB, FLAG_SKIP_OUTPUTS = 16, 32
FLAG_RESIDUAL_BLOCKS = 64    # nfa_affine_flow_mlp_f32: the conditioner is a ResidualNet (ABI 11)
FLAG_PAD_COLUMNS_SHIFT = 8   # bits 8-10: trailing pad columns the density epilogue leaves out
FLAG_ACTIVATION_SHIFT = 12   # bits 12-14: activation of the conditioner's residual blocks in the whole-layer kernels
ACTIVATION_RELU, ACTIVATION_LEAKY_RELU, ACTIVATION_ELU, ACTIVATION_TANH = 0, 1, 2, 3
TAILS_NONE, TAILS_LINEAR = 0, 1
SCALE_DEFAULT, SCALE_GENERAL, SCALE_ADDITIVE, SCALE_GIVEN, SCALE_SOFTPLUS = 0, 1, 2, 3, 4

ABI_VERSION = 14

EXPORTS = (
    "nfa_abi_version",
    "nfa_build_arch",
    "nfa_strerror",
    "nfa_last_hip_error",
    "nfa_rqs_coupling_f32",
    "nfa_rqs_coupling_backward_f32",
    "nfa_rqs_coupling_fused_linear_f32",
    "nfa_rqs_coupling_resnet_f32",
    "nfa_rqs_flow_resnet_f32",
    "nfa_rqs_flow_resnet_redo_f32",
    "nfa_rqs_flow_resnet_context_f32",
    "nfa_rqs_elementwise_f64",
    "nfa_rqs_elementwise_backward_f64",
    "nfa_affine_flow_mlp_f32",
    "nfa_made_rqs_inverse_f32",
    "nfa_rqs_made_output_f32",
    "nfa_pack_resnet_hidden_train_f32",
    "nfa_resnet_hidden_forward_f32",
    "nfa_resnet_hidden_backward_f32",
    "nfa_resnet_backward_f32",
    "nfa_rqs_flow_resnet_f16x2_f32",
    "nfa_rqs_flow_resnet_f16x2_tile16_f32",
    "nfa_rqs_flow_resnet_f16x2_colsplit_f32",
    "nfa_rqs_flow_resnet_f16x2_bins_f32",
    "nfa_rqs_flow_resnet_f16x2_tile16_bins_f32",
    "nfa_rqs_flow_resnet_context_f16x2_f32",
    "nfa_rqs_flow_resnet_f16x3_f32",
    "nfa_rqs_flow_resnet_f16x3_logits_f32",
    "nfa_rqs_flow_resnet_f16x2_logits_f32",
    "nfa_rqs_flow_resnet_logits_f32",
    "nfa_rqs_flow_resnet_context_redo_f32",
    "nfa_linear_spline_f32",
    "nfa_quadratic_spline_f32",
    "nfa_cubic_spline_f32",
    "nfa_linear_spline_backward_f32",
    "nfa_quadratic_spline_backward_f32",
    "nfa_cubic_spline_backward_f32",
    "nfa_rqs_elementwise_f32",
    "nfa_searchsorted_f32",
    "nfa_rqs_shared_f32",
    "nfa_affine_coupling_f32",
    "nfa_affine_autoregressive_f32",
    "nfa_permute_cols_b32",
    "nfa_rowsum_f32",
    "nfa_standard_normal_log_prob_f32",
    "nfa_sum_count_f64",
    "nfa_sum_count_workspace_bytes",
    "nfa_linear_wgrad_workspace_bytes",
    "nfa_linear_wgrad_f32",
    "nfa_linear_wgrad_batched_workspace_bytes",
    "nfa_linear_wgrad_batched_f32",
    "nfa_profile_enable",
    "nfa_profile_collect",
    "nfa_last_layer_kernel",
    "nfa_debug_k7_trace",
)


class NativeLibraryMissing(RuntimeError):
    pass


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libnflows_amd: %s (code %d)" % (message, code))
        self.code = code


class RqsSpec(ctypes.Structure):
    """struct nfa_rqs_spec"""

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


_lib = None


def _declare(lib):
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    sp = ctypes.POINTER(RqsSpec)
    lib.nfa_abi_version.restype = ctypes.c_int
    lib.nfa_build_arch.restype = ctypes.c_char_p
    lib.nfa_strerror.restype = ctypes.c_char_p
    lib.nfa_strerror.argtypes = [ctypes.c_int]
    lib.nfa_last_hip_error.restype = ctypes.c_int
    lib.nfa_rqs_coupling_f32.restype = ctypes.c_int
    lib.nfa_rqs_coupling_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, sp, i32, vp]
    lib.nfa_rqs_coupling_backward_f32.restype = ctypes.c_int
    lib.nfa_rqs_coupling_backward_f32.argtypes = [vp] * 10 + [i64, i32, i32, sp, i32, vp]
    lib.nfa_rqs_coupling_fused_linear_f32.restype = ctypes.c_int
    lib.nfa_rqs_coupling_fused_linear_f32.argtypes = [vp] * 10 + [i64, i32, i32, i32, sp, i32, vp]
    lib.nfa_linear_spline_f32.restype = ctypes.c_int
    lib.nfa_linear_spline_f32.argtypes = [vp, vp, i64, vp, vp, vp, i64, sp, i32, vp]
    lib.nfa_quadratic_spline_f32.restype = ctypes.c_int
    lib.nfa_quadratic_spline_f32.argtypes = [vp, vp, i64, vp, i64, i32, vp, vp, vp, i64, sp, i32, vp]
    lib.nfa_cubic_spline_f32.restype = ctypes.c_int
    lib.nfa_cubic_spline_f32.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, i64, sp, i32, vp]
    lib.nfa_linear_spline_backward_f32.restype = ctypes.c_int
    lib.nfa_linear_spline_backward_f32.argtypes = [vp] * 6 + [i64, sp, i32, vp]
    lib.nfa_quadratic_spline_backward_f32.restype = ctypes.c_int
    lib.nfa_quadratic_spline_backward_f32.argtypes = [vp, vp, vp, i32] + [vp] * 5 + [i64, sp, i32, vp]
    lib.nfa_cubic_spline_backward_f32.restype = ctypes.c_int
    lib.nfa_cubic_spline_backward_f32.argtypes = [vp] * 12 + [i64, sp, i32, vp]
    lib.nfa_rqs_flow_resnet_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f32.argtypes = [vp] * 4 + [i32] + [vp] * 3 + [i64, i32, i32, i32, i32, i32, sp, i32, vp]
    lib.nfa_affine_flow_mlp_f32.restype = ctypes.c_int
    lib.nfa_affine_flow_mlp_f32.argtypes = [vp] * 4 + [i32] + [vp] * 3 + [i64] + [i32] * 7 + [vp]
    lib.nfa_rqs_flow_resnet_context_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_context_f32.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i64, i32, i32, i32, i32,
                                                    i32, sp, i32, vp]
    lib.nfa_rqs_elementwise_f64.restype = ctypes.c_int
    lib.nfa_rqs_elementwise_f64.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp, vp, vp, vp, i64, sp, i32, vp]
    lib.nfa_pack_resnet_hidden_train_f32.restype = ctypes.c_int
    lib.nfa_pack_resnet_hidden_train_f32.argtypes = [vp, vp, ctypes.POINTER(vp), vp, vp, i32, i32, i32, i32, vp, vp, vp,
                                                     vp, vp]
    lib.nfa_resnet_hidden_forward_f32.restype = ctypes.c_int
    lib.nfa_resnet_hidden_forward_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp]
    lib.nfa_resnet_hidden_backward_f32.restype = ctypes.c_int
    lib.nfa_resnet_hidden_backward_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    lib.nfa_resnet_backward_f32.restype = ctypes.c_int
    lib.nfa_resnet_backward_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    lib.nfa_rqs_elementwise_backward_f64.restype = ctypes.c_int
    lib.nfa_rqs_elementwise_backward_f64.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp, vp, vp, vp, vp, vp,
                                                     i64, sp, i32, vp]
    lib.nfa_made_rqs_inverse_f32.restype = ctypes.c_int
    lib.nfa_made_rqs_inverse_f32.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_int32), i32, vp, vp, vp, vp, i64,
                                             i32, i32, i32, sp, vp]
    lib.nfa_rqs_made_output_f32.restype = ctypes.c_int
    lib.nfa_rqs_made_output_f32.argtypes = [vp, i64, i32, vp, i32, vp, vp, vp, vp, vp, i64, i32, sp, i32, vp]
    lib.nfa_rqs_flow_resnet_redo_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_redo_f32.argtypes = [vp] * 4 + [i32] + [vp] * 4 + [i64, i32, i32, i32, i32, i32, sp, i32, vp]
    lib.nfa_rqs_flow_resnet_f16x2_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f16x2_f32.argtypes = [vp, vp, i32, vp, i32] + [vp] * 4 + [i64, i32, i32, i32, i32, i32, sp, i32, vp]
    lib.nfa_rqs_flow_resnet_f16x2_tile16_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f16x2_tile16_f32.argtypes = lib.nfa_rqs_flow_resnet_f16x2_f32.argtypes
    lib.nfa_rqs_flow_resnet_f16x2_colsplit_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f16x2_colsplit_f32.argtypes = lib.nfa_rqs_flow_resnet_f16x2_f32.argtypes
    for fn in (lib.nfa_rqs_flow_resnet_f16x2_bins_f32, lib.nfa_rqs_flow_resnet_f16x2_tile16_bins_f32):
        fn.restype = ctypes.c_int
        fn.argtypes = lib.nfa_rqs_flow_resnet_f16x2_f32.argtypes + [vp]
    lib.nfa_rqs_flow_resnet_context_f16x2_f32.restype = ctypes.c_int
    f32 = ctypes.c_float
    lib.nfa_rqs_flow_resnet_f16x3_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f16x3_f32.argtypes = [vp] * 5 + [i32] + [vp] * 4 + [i64, i32, i32, i32, i32, i32, f32, sp, i32, vp]
    lib.nfa_rqs_flow_resnet_f16x3_logits_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f16x3_logits_f32.argtypes = lib.nfa_rqs_flow_resnet_f16x3_f32.argtypes + [vp]
    lib.nfa_rqs_flow_resnet_f16x2_logits_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_f16x2_logits_f32.argtypes = lib.nfa_rqs_flow_resnet_f16x2_f32.argtypes + [vp, vp]
    lib.nfa_rqs_flow_resnet_logits_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_logits_f32.argtypes = lib.nfa_rqs_flow_resnet_f32.argtypes + [vp]
    lib.nfa_rqs_flow_resnet_context_f16x2_f32.argtypes = [vp, vp, i32, vp, i32, vp, i32] + [vp] * 4 + \
        [i64, i32, i32, i32, i32, i32, sp, i32, vp]
    lib.nfa_rqs_flow_resnet_context_redo_f32.restype = ctypes.c_int
    lib.nfa_rqs_flow_resnet_context_redo_f32.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, i64, i32, i32,
                                                         i32, i32, i32, sp, i32, vp]
    lib.nfa_rqs_coupling_resnet_f32.restype = ctypes.c_int
    lib.nfa_rqs_coupling_resnet_f32.argtypes = [vp] * 7 + [i64, i32, i32, i32, i32, i32, sp, i32, vp]
    lib.nfa_rqs_elementwise_f32.restype = ctypes.c_int
    lib.nfa_rqs_elementwise_f32.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp, vp, vp, vp, i64, sp, i32, vp]
    lib.nfa_searchsorted_f32.restype = ctypes.c_int
    lib.nfa_searchsorted_f32.argtypes = [vp, i64, i32, vp, vp, i64, ctypes.c_double, vp]
    lib.nfa_rqs_shared_f32.restype = ctypes.c_int
    lib.nfa_rqs_shared_f32.argtypes = [vp] * 7 + [i64, i32, sp, i32, vp]
    lib.nfa_affine_coupling_f32.restype = ctypes.c_int
    lib.nfa_affine_coupling_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.nfa_affine_autoregressive_f32.restype = ctypes.c_int
    lib.nfa_affine_autoregressive_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp]
    lib.nfa_permute_cols_b32.restype = ctypes.c_int
    lib.nfa_permute_cols_b32.argtypes = [vp, vp, vp, vp, i64, i32, vp]
    lib.nfa_rowsum_f32.restype = ctypes.c_int
    lib.nfa_rowsum_f32.argtypes = [vp, vp, i64, i64, vp]
    lib.nfa_standard_normal_log_prob_f32.restype = ctypes.c_int
    lib.nfa_standard_normal_log_prob_f32.argtypes = [vp, vp, vp, i64, i64, vp]
    lib.nfa_sum_count_f64.restype = ctypes.c_int
    lib.nfa_sum_count_f64.argtypes = [vp, i64, vp, vp, vp]
    lib.nfa_sum_count_workspace_bytes.restype = ctypes.c_size_t
    lib.nfa_sum_count_workspace_bytes.argtypes = []
    lib.nfa_linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.nfa_linear_wgrad_workspace_bytes.argtypes = [i64, i32, i32]
    lib.nfa_linear_wgrad_f32.restype = ctypes.c_int
    lib.nfa_linear_wgrad_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    lib.nfa_linear_wgrad_batched_workspace_bytes.restype = ctypes.c_size_t
    lib.nfa_linear_wgrad_batched_workspace_bytes.argtypes = [i32, i64, i32, i32]
    lib.nfa_linear_wgrad_batched_f32.restype = ctypes.c_int
    lib.nfa_linear_wgrad_batched_f32.argtypes = [i32, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    lib.nfa_profile_enable.restype = ctypes.c_int
    lib.nfa_profile_enable.argtypes = [i32]
    lib.nfa_profile_collect.restype = ctypes.c_int
    lib.nfa_profile_collect.argtypes = [ctypes.POINTER(ctypes.c_float), i32, ctypes.POINTER(i32)]
    lib.nfa_last_layer_kernel.restype = ctypes.c_int
    lib.nfa_last_layer_kernel.argtypes = [ctypes.c_char_p, i32]


def load():
    """Loads the shared library (once).  Raises NativeLibraryMissing if it has not been built:
    run `python -c "import __graft_entry__ as g; g.build()"` or `make -C nflows_amd/csrc`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found; nflows_amd has no fallback path. Build it with "
                "`make -C nflows_amd/csrc` (hipcc --offload-arch=gfx950)." % LIB_PATH
            )
        lib = ctypes.CDLL(LIB_PATH)
        missing = [name for name in EXPORTS if not hasattr(lib, name)]
        if missing:
            raise NativeLibraryMissing("%s lacks symbols: %s" % (LIB_PATH, ", ".join(missing)))
        _declare(lib)
        if lib.nfa_abi_version() != ABI_VERSION:
            raise NativeLibraryMissing("ABI version mismatch: library %d, binding %d"
                                       % (lib.nfa_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(code):
    if code == OK:
        return
    lib = load()
    msg = lib.nfa_strerror(code).decode()
    if code in (ERR_MIN_BIN_WIDTH, ERR_MIN_BIN_HEIGHT):
        raise ValueError(msg)  # same type and text as rational_quadratic.py:86-89
    if code == ERR_HIP:
        msg += " (hipError_t %d)" % lib.nfa_last_hip_error()
    raise NativeError(code, msg)


def require_device_f32(name, t, dim=None):
    """The product path runs on the GPU only; anything else fails loudly."""
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise NotImplementedError(
            "nflows_amd: %s is on %s; the MI355X path has no CPU fallback (move it to a HIP device)"
            % (name, t.device))
    if t.dtype != torch.float32:
        raise NotImplementedError("nflows_amd: %s has dtype %s; only float32 is implemented"
                                  % (name, t.dtype))
    if dim is not None and t.dim() != dim:
        raise ValueError("%s must be %d-D, got %d-D" % (name, dim, t.dim()))
    return t


def require_device_real(name, t, dtype, dim=None):
    """float32 or float64 on the device (the float64 functional path)."""
    if torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float64 and dtype == torch.float64:
        if dim is not None and t.dim() != dim:
            raise ValueError("%s must be %d-D, got %d-D" % (name, dim, t.dim()))
        return t
    return require_device_f32(name, t, dim)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_handle(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
