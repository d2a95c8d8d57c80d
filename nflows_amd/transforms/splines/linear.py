"""Piecewise-linear spline functionals backed by the K9 HIP kernel.

Same signatures, argument meaning and exceptions as nflows/transforms/splines/linear.py:9-11 and
:40-42; elementwise over any leading shape, no row-sum.
"""
from ... import ops


def unconstrained_linear_spline(inputs, unnormalized_pdf, inverse=False, tail_bound=1.0, tails="linear"):
    """Identity outside [-tail_bound, tail_bound] (zero logabsdet there), piecewise-linear CDF
    of the softmax of `unnormalized_pdf` inside."""
    if tails != "linear":
        raise RuntimeError("{} tails are not implemented.".format(tails))
    spec = ops.make_rqs_spec(unnormalized_pdf.shape[-1], "linear", tail_bound=tail_bound, min_bin_width=0.0,
                             min_bin_height=0.0)
    return ops.linear_spline(inputs, unnormalized_pdf, spec, inverse)


def linear_spline(inputs, unnormalized_pdf, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0):
    """Spline on [left, right] -> [bottom, top].  Raises InputOutsideDomain if any input is
    outside [left, right] (the reference tests against left / right in both directions)."""
    spec = ops.make_rqs_spec(unnormalized_pdf.shape[-1], None, left=left, right=right, bottom=bottom, top=top,
                             min_bin_width=0.0, min_bin_height=0.0)
    return ops.linear_spline(inputs, unnormalized_pdf, spec, inverse)
