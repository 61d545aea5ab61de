"""Assertions shared by the end-to-end parity tests (device search vs the oracle pipeline)."""
import numpy as np


def assert_root_values_close(o_val, d_val, same, tol=2e-3, frac=0.95, relative=False):
    """Root values on the roots whose visit distributions are identical.  The two pipelines see network outputs that differ by
    ~1e-6 (the torch CPU side is not even bit-reproducible from process to process: its thread partitioning changes the summation
    order), so a *deeper* arg-max can flip on a root whose ROOT visit counts still coincide; the leaf evaluated then differs and the
    root value moves by O(1e-2).  Observed about once in 20 fresh processes, on one root of 64.  Hence: at least `frac` of those roots
    within `tol` (the bar the values meet when no decision flips), and none off by more than what a single different leaf can cause."""
    o_val, d_val, same = np.asarray(o_val, np.float64), np.asarray(d_val, np.float64), np.asarray(same, bool)
    d = np.abs(o_val - d_val)
    if relative:
        d = d / (1.0 + np.abs(d_val))
    d = d[same]
    assert d.size > 0
    assert (d < tol).mean() >= frac, "only %.0f %% of the roots with identical visit distributions have root values within %g" % (100 * (d < tol).mean(), tol)
    assert d.max() < 0.5, "a root value is off by %.3g" % d.max()
