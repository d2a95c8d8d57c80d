"""Rational-quadratic spline functionals backed by the K5 HIP kernel.

Same signatures, argument meaning and exceptions as
nflows/transforms/splines/rational_quadratic.py:13-25 and :66-80; elementwise over any leading
shape, no row-sum.  Unlike the reference the inputs are never modified in place (the reference
pads / nudges temporaries it owns; nobody can observe that).
"""
from ... import ops

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def unconstrained_rational_quadratic_spline(
    inputs,
    unnormalized_widths,
    unnormalized_heights,
    unnormalized_derivatives,
    inverse=False,
    tails="linear",
    tail_bound=1.0,
    min_bin_width=DEFAULT_MIN_BIN_WIDTH,
    min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
    min_derivative=DEFAULT_MIN_DERIVATIVE,
    enable_identity_init=False,
):
    """Identity outside [-tail_bound, tail_bound] (zero logabsdet there), monotone
    rational-quadratic spline inside; derivative logits have K-1 entries."""
    if tails != "linear":
        raise RuntimeError("{} tails are not implemented.".format(tails))
    spec = ops.make_rqs_spec(
        unnormalized_widths.shape[-1], "linear", tail_bound=tail_bound,
        min_bin_width=min_bin_width, min_bin_height=min_bin_height,
        min_derivative=min_derivative, enable_identity_init=enable_identity_init)
    return ops.rqs_elementwise(inputs, unnormalized_widths, unnormalized_heights,
                               unnormalized_derivatives, spec, inverse)


def rational_quadratic_spline(
    inputs,
    unnormalized_widths,
    unnormalized_heights,
    unnormalized_derivatives,
    inverse=False,
    left=0.0,
    right=1.0,
    bottom=0.0,
    top=1.0,
    min_bin_width=DEFAULT_MIN_BIN_WIDTH,
    min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
    min_derivative=DEFAULT_MIN_DERIVATIVE,
    enable_identity_init=False,
):
    """Spline on [left, right] -> [bottom, top]; derivative logits have K+1 entries.  Raises
    InputOutsideDomain if any input is outside [left, right] (checked on the device, read back
    here, like the reference's `torch.min(inputs) < left`)."""
    spec = ops.make_rqs_spec(
        unnormalized_widths.shape[-1], None, left=left, right=right, bottom=bottom, top=top,
        min_bin_width=min_bin_width, min_bin_height=min_bin_height,
        min_derivative=min_derivative, enable_identity_init=enable_identity_init)
    return ops.rqs_elementwise(inputs, unnormalized_widths, unnormalized_heights,
                               unnormalized_derivatives, spec, inverse)
