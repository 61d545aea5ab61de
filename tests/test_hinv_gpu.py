"""GPU: the device's inverse scalar transform (csrc/lz_hinv.h, ONE definition for k_heads / k_heads_mm / the split heads / k_rowfinal)
against the reference formula (lzero/policy/scaling_transform.py:88-91) on identical input.

What is asserted, over > 2 x 10^6 inputs spanning the support range:
 (1) the device result is BIT-EQUAL to the formula evaluated operation by operation in IEEE binary32 with every operation correctly
     rounded, in torch's order (parity_util.hinv_ieee);
 (2) torch's own fp32 result on this host (oracle/torch_models.InverseScalarTransform, itself pinned to the reference file by
     tests/test_torch_models_vs_reference.py) is BIT-EQUAL to the same evaluation once torch's own square root is substituted
     (parity_util.hinv_with_sqrt): every operation but the square root agrees with torch bit for bit;
 (3) torch's square root is within 1 ulp of the correctly rounded one -- at::sqrt runs through the vector math library (MKL VML), which
     is not always correctly rounded and not the same on every CPU (0.7 % of these inputs differ on the build container, 11 % on the
     GPU box's host).  The reference's h^-1 is therefore machine-dependent at the level of one step of its own output quantum
     (~1.3e-4 (1 + |x|)); the device takes the correctly rounded square root.

Why it matters: with (1)-(3) every difference of a post-transform scalar between the device and the reference pipeline is either a
difference of the PRE-transform expectation (held to 1e-5 by tests/test_nn_gpu.py) that crossed one of the formula's quantisation
steps, or torch's own square-root rounding -- the 3e-4 gate on those scalars is the reference formula's quantum, not slack of this
engine.  (Round 4 found by disassembly that the compiler had contracted `1 + 0.004 t` and `t t - 1` into fused multiply-adds: this
test fails on that build -- emulated on the host, the contracted form differs on 32 % of the inputs in [-2, 2], by up to 3.6e-4.)"""
import numpy as np
import pytest
import torch

from parity_util import hinv_ieee, hinv_with_sqrt

pytestmark = pytest.mark.gpu


def _inputs():
    rng = np.random.default_rng(0)
    parts = [
        np.linspace(-300.0, 300.0, 1_000_001, dtype=np.float64).astype(np.float32),      # the support range, 6e-4 apart
        rng.uniform(-300.0, 300.0, 400_000).astype(np.float32),
        rng.uniform(-2.0, 2.0, 400_000).astype(np.float32),                               # where value / value-prefix scalars live
        (rng.standard_normal(200_000) * np.exp(rng.uniform(-40, 12, 200_000))).astype(np.float32),   # 1e-17 .. 1e5, both signs
        np.arange(-300, 301, dtype=np.float32),                                           # the support atoms themselves
        np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 3.4e38, -3.4e38, np.inf, -np.inf, np.nan], np.float32),
    ]
    return np.ascontiguousarray(np.concatenate(parts))


def _same_bits(a, b):
    nan = np.isnan(b)
    assert np.array_equal(nan, np.isnan(a))
    return np.nonzero(a[~nan].view(np.uint32) != b[~nan].view(np.uint32))[0], ~nan


@pytest.mark.parametrize("which", [0, 1])   # 0: the copy in lz_nn.hip (conv-model heads), 1: lz_dense.hip (k_rowfinal)
def test_device_inverse_scalar_transform_is_bit_equal_to_the_ieee_evaluation(which):
    from lightzero_amd import _lib as L
    x = _inputs()
    assert x.size >= 2_000_000
    out = np.zeros_like(x)
    L.check(L.lib().lz_debug_inverse_scalar_transform(L.default_engine(), which, x, x.size, out))
    ref = hinv_ieee(x)
    bad, ok = _same_bits(out, ref)
    assert bad.size == 0, "%d of %d inputs differ; first: x = %r device %r ieee %r" % (
        bad.size, ok.sum(), x[ok][bad[:5]], out[ok][bad[:5]], ref[ok][bad[:5]])


def test_torch_differs_from_the_ieee_evaluation_only_in_its_square_root():
    """(2) and (3) of the module docstring -- runs on the host (no device), on whatever CPU the GPU box has"""
    from oracle import torch_models as tm
    x = _inputs()
    ist = tm.InverseScalarTransform(categorical_distribution=False)
    with torch.no_grad():
        ref = ist(torch.from_numpy(x.copy()).reshape(-1, 1)).reshape(-1).numpy()
    tsqrt = lambda t: torch.sqrt(torch.from_numpy(np.ascontiguousarray(t))).numpy()
    bad, ok = _same_bits(hinv_with_sqrt(x, tsqrt), ref)
    assert bad.size == 0, "torch's result differs from the op-by-op evaluation beyond its square root on %d inputs, e.g. x = %r" % (bad.size, x[ok][bad[:5]])
    # torch's square root against the correctly rounded one on the values the formula feeds it
    f = np.float32
    arg = (f(1.0) + f(0.004) * ((np.abs(x[ok]) + f(1.0)) + f(0.001))).astype(np.float32)
    arg = arg[np.isfinite(arg)]
    a, b = tsqrt(arg), np.sqrt(arg)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    worse, _ = _same_bits(hinv_ieee(x), ref)
    print("torch.sqrt differs from the correctly rounded square root on %.2f %% of the inputs (<= 1 ulp); torch's h^-1 output differs "
          "from the IEEE evaluation on %.2f %%" % (100.0 * (ulp > 0).mean(), 100.0 * worse.size / ok.sum()))
    import parity_record
    parity_record.record("hinv/torch_vs_ieee", {}, extra=dict(torch_sqrt_not_correctly_rounded_fraction=float((ulp > 0).mean()),
                                                             torch_hinv_differs_from_ieee_fraction=worse.size / float(ok.sum()), inputs=int(x.size)))


def test_quantum_of_the_reference_formula():
    """the statement DESIGN.md section 6 rests on, as a test: neighbouring fp32 inputs near 0 map to outputs ~1.3e-4 apart (or equal)"""
    x = np.linspace(0.05, 0.06, 20001).astype(np.float32)
    y = hinv_ieee(x).astype(np.float64)
    steps = np.unique(np.round(np.diff(np.unique(y)), 7))
    assert steps.min() > 5e-5 and steps.max() < 3e-4, steps
