"""Shared comparison helpers for the parity tests.

Tolerance model (stated once, used everywhere)
----------------------------------------------
fp32 results of the spline depend on 1-ulp differences in exp/log through cancellation
(x - knot) and can therefore differ between two correct fp32 implementations by more than a few
ulp on ill-conditioned elements (steep bins).  The reference's own fp32 result has that error
against its fp64 result.  Parity of an fp32 implementation `got` against fp32 `ref` with fp64
ground truth `truth` is therefore asserted as:

  (a) bulk agreement: >= 99% of finite elements satisfy |got-ref| <= 2e-6 * (1 + |ref|)
      (outputs) / 2e-5 * (1 + |ref|) (log-dets);
  (b) worst case: max|got-truth| <= 4 * max|ref-truth| + 2e-6 (outputs), + 2e-5 (log-dets);
  (c) identical NaN / inf pattern, and elements the reference passes through unchanged
      (tails) are bit-equal.
Integer / index work (permutations, untouched columns) is compared with array_equal.
"""
import numpy as np

OUT_TOL = 2e-6
LAD_TOL = 2e-5


def bulk_fraction(got, ref, tol):
    fin = np.isfinite(ref)
    if fin.sum() == 0:
        return 1.0
    d = np.abs(got[fin].astype(np.float64) - ref[fin].astype(np.float64))
    return float(np.mean(d <= tol * (1.0 + np.abs(ref[fin]))))


def assert_fp32_parity(got, ref, truth, tol, what="", bulk=0.99, factor=4.0):
    got = np.asarray(got)
    ref = np.asarray(ref)
    truth = np.asarray(truth)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NaN pattern differs"
    inf = np.isinf(ref)
    assert np.array_equal(got[inf], ref[inf]), what + ": inf pattern differs"
    frac = bulk_fraction(got, ref, tol)
    assert frac >= bulk, "%s: only %.4f of elements within %g" % (what, frac, tol)
    fin = np.isfinite(ref) & np.isfinite(truth)
    if fin.any():
        e_got = np.abs(got[fin].astype(np.float64) - truth[fin]).max()
        e_ref = np.abs(ref[fin].astype(np.float64) - truth[fin]).max()
        assert e_got <= factor * e_ref + tol * (1 + np.abs(truth[fin]).max()), (
            "%s: max err vs fp64 %.3e, reference fp32's own %.3e" % (what, e_got, e_ref))


def parse_kwargs(text):
    import ast
    return dict(ast.literal_eval(text))


def assert_sibling_spline_parity(got, ref, truth, tol, cap, what=""):
    """Linear / quadratic splines: same bulk criterion as assert_fp32_parity (>= 97 % of the elements
    within tol * (1 + |ref|) of the reference's fp32 output), identical NaN pattern, and the error
    against the reference's float64 result bounded by max(4 x the reference-fp32's own, cap).
    `cap` is there because these splines have elements whose result moves by 1e3 ulp when one
    intermediate prefix sum moves by one ulp (a bin of minimal width next to a steep one: the
    quadratic inverse divides a difference of two nearly equal cdf values by a tiny 2a); the
    reference's own fp32 error at such an element is a matter of luck, not a bound."""
    got, ref, truth = np.asarray(got), np.asarray(ref), np.asarray(truth)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NaN pattern differs"
    frac = bulk_fraction(got, ref, tol)
    assert frac >= 0.97, "%s: only %.4f of elements within %g" % (what, frac, tol)
    fin = np.isfinite(ref) & np.isfinite(truth)
    if fin.any():
        e_got = np.abs(got[fin].astype(np.float64) - truth[fin]).max()
        e_ref = np.abs(ref[fin].astype(np.float64) - truth[fin]).max()
        assert e_got <= max(4.0 * e_ref, cap), "%s: max err vs fp64 %.3e (reference fp32: %.3e)" % (what, e_got, e_ref)
