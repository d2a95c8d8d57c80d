"""Piecewise-quadratic spline functionals backed by the K9 HIP kernel.

Same signatures, argument meaning and exceptions as nflows/transforms/splines/quadratic.py:11-20
and :55-66; elementwise over any leading shape, no row-sum.
"""
from ... import ops

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3


def unconstrained_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, inverse=False,
                                   tail_bound=1.0, tails="linear", min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                                   min_bin_height=DEFAULT_MIN_BIN_HEIGHT):
    """Identity outside [-tail_bound, tail_bound]; the K-1 height logits are completed by two
    boundary heights that normalise to exactly 1 (quadratic.py:93-107)."""
    if tails != "linear":
        raise RuntimeError("{} tails are not implemented.".format(tails))
    num_bins = unnormalized_widths.shape[-1]
    assert unnormalized_heights.shape[-1] == num_bins - 1  # quadratic.py:34
    spec = ops.make_rqs_spec(num_bins, "linear", tail_bound=tail_bound, min_bin_width=min_bin_width,
                             min_bin_height=min_bin_height)
    return ops.quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, spec, inverse)


def quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, inverse=False, left=0.0, right=1.0,
                     bottom=0.0, top=1.0, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                     min_bin_height=DEFAULT_MIN_BIN_HEIGHT):
    """Spline on [left, right] -> [bottom, top]; K+1 height logits, or K-1 with derived boundary
    heights.  Raises InputOutsideDomain for inputs outside [left, right] and ValueError for
    minimal bin sizes that do not fit."""
    spec = ops.make_rqs_spec(unnormalized_widths.shape[-1], None, left=left, right=right, bottom=bottom,
                             top=top, min_bin_width=min_bin_width, min_bin_height=min_bin_height)
    return ops.quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, spec, inverse)
