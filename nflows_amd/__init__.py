"""nflows_amd -- the bayesiains/nflows coupling-layer hot path, rebuilt for AMD MI355X (gfx950).

Scope: coupling-layer forward / inverse + per-sample log|det J|
(rational-quadratic spline, affine / additive), column permutations, the per-sample reduction,
behind the `Transform` API of the reference (`forward/inverse -> (outputs, logabsdet)`), plus the
sample-sharded multi-GPU log-likelihood.  The compute runs in hand-written HIP kernels loaded
through a C ABI (include/nflows_amd.h); there is no CPU or eager fallback.
"""
from . import _cache, _native
from .errors import InputOutsideDomain, InverseNotAvailable
from .ops import check_status, set_error_mode

__version__ = "0.1.0"
__all__ = ["InputOutsideDomain", "InverseNotAvailable", "check_status", "set_error_mode",
           "native_library_path", "invalidate_packed_weights"]


def invalidate_packed_weights():
    """Drops every cached re-tiled copy of conditioner weights (whole-layer kernels K7 / K8 and the
    multi-layer run plans).  Needed only after writing parameters through `.data` (which does not
    advance the tensors' version counters); in-place updates under `torch.no_grad()`, optimizer
    steps, `load_state_dict`, `.to()` and `parallel.broadcast_model` are detected without it.
    A captured `graphs.GraphedLogProb` holds the packed weights of capture time: re-capture after
    changing weights."""
    _cache.invalidate()


def native_library_path():
    """Path of the HIP shared library the package loads (raises if it was not built)."""
    _native.load()
    return _native.LIB_PATH
