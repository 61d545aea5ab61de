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


def hinv_ieee(x):
    """InverseScalarTransform's scalar part (lzero/policy/scaling_transform.py:88-91) evaluated operation by operation in IEEE binary32,
    every operation correctly rounded (numpy float32 arithmetic), in torch's order: |v| + 1, + 0.001, * 0.004, 1 + ., sqrt, - 1,
    / 0.002, t * t, - 1, * sign(v).  torch's CPU kernels compute exactly this EXCEPT for the square root: at::sqrt goes through the
    vector math library (MKL VML in high-accuracy mode: < 1 ulp, not always correctly rounded, and not the same on every CPU), see
    tests/test_hinv_gpu.py."""
    f = np.float32
    v = np.asarray(x, np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        t = np.abs(v) + f(1.0)
        t = t + f(0.001)
        t = f(0.004) * t
        t = f(1.0) + t
        t = np.sqrt(t)
        t = t - f(1.0)
        t = t / f(0.002)
        return (np.sign(v) * (t * t - f(1.0))).astype(np.float32)


def hinv_with_sqrt(x, sqrt):
    """the same evaluation with the square root supplied by the caller (torch.sqrt of this host)"""
    f = np.float32
    v = np.asarray(x, np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        t = np.abs(v) + f(1.0)
        t = t + f(0.001)
        t = f(0.004) * t
        t = f(1.0) + t
        t = np.asarray(sqrt(t), np.float32)
        t = t - f(1.0)
        t = t / f(0.002)
        return (np.sign(v) * (t * t - f(1.0))).astype(np.float32)
