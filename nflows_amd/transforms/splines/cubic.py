"""Piecewise-cubic spline functionals backed by the K9 HIP kernel.

Same signatures, argument meaning and exceptions as nflows/transforms/splines/cubic.py:15-28 and
:63-78; elementwise over any leading shape, no row-sum.  `eps` and `quadratic_threshold` are the
reference's defaults (1e-5, 1e-3); other values are not implemented.
"""
from ... import ops

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_EPS = 1e-5
DEFAULT_QUADRATIC_THRESHOLD = 1e-3


def _defaults_only(eps, quadratic_threshold):
    if eps != DEFAULT_EPS or quadratic_threshold != DEFAULT_QUADRATIC_THRESHOLD:
        raise NotImplementedError("nflows_amd: the cubic spline kernel has eps = 1e-5 and "
                                  "quadratic_threshold = 1e-3 built in")


def unconstrained_cubic_spline(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                               unnorm_derivatives_right, inverse=False, tail_bound=1.0, tails="linear",
                               min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                               eps=DEFAULT_EPS, quadratic_threshold=DEFAULT_QUADRATIC_THRESHOLD):
    """Identity outside [-tail_bound, tail_bound] (zero logabsdet there), monotone cubic inside."""
    if tails != "linear":
        raise RuntimeError("{} tails are not implemented.".format(tails))
    _defaults_only(eps, quadratic_threshold)
    spec = ops.make_rqs_spec(unnormalized_widths.shape[-1], "linear", tail_bound=tail_bound,
                             min_bin_width=min_bin_width, min_bin_height=min_bin_height)
    return ops.cubic_spline(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                            unnorm_derivatives_right, spec, inverse)


def cubic_spline(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                 unnorm_derivatives_right, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0,
                 min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT, eps=DEFAULT_EPS,
                 quadratic_threshold=DEFAULT_QUADRATIC_THRESHOLD):
    """Spline on [left, right] -> [bottom, top]; raises InputOutsideDomain for inputs outside
    [left, right] and ValueError for minimal bin sizes that do not fit."""
    _defaults_only(eps, quadratic_threshold)
    spec = ops.make_rqs_spec(unnormalized_widths.shape[-1], None, left=left, right=right, bottom=bottom,
                             top=top, min_bin_width=min_bin_width, min_bin_height=min_bin_height)
    return ops.cubic_spline(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                            unnorm_derivatives_right, spec, inverse)
