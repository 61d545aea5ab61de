"""CPU: the arithmetic behind the parity-mode network kernels of round 5, restated in numpy.

k_chain_s3 / k_conv_s3 multiply fp32 operands as three bf16 planes each (hi = rne(x), mid = rne(x - hi), lo = rne(x - hi - mid)) and
accumulate six of the nine plane products in fp32.  What the kernels rely on, checked here without a GPU:
  * the split is EXACT: hi + mid + lo == x for every finite fp32 in bf16's normal range (the two subtractions are exact);
  * every plane product is exact in fp32 (8 x 8 significant bits);
  * the dropped products (mid lo, lo mid, lo lo) are below 2^-24 of |x w|;
  * a 576-term dot product by six plane products, fp32 accumulation, is as close to binary64 as an ordinary fp32 dot product.
The device code under test is exercised by tests/test_nn_golden_gpu.py and tests/test_nn_gpu.py (1e-5 (1 + |x|) of the reference modules)."""
import numpy as np


def bf16_rne(x):
    """round-to-nearest-even to bfloat16, returned as float32 (finite inputs)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    hi = bf16_rne(x)
    r1 = x - hi
    mid = bf16_rne(r1)
    r2 = r1 - mid
    lo = bf16_rne(r2)
    return hi, mid, lo


def _samples(rng, n):
    mant = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    expo = rng.integers(127 - 40, 127 + 40, n, dtype=np.uint32)          # 2^-40 .. 2^40: activations and weights of a normalised network
    sign = rng.integers(0, 2, n, dtype=np.uint32)
    return ((sign << 31) | (expo << 23) | mant).view(np.float32)


def test_three_bf16_planes_are_the_fp32_number_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([_samples(rng, 2_000_000), np.float32([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, 1.2e-30, 255.0 / 256.0])])
    hi, mid, lo = split3(x)
    for p in (hi, mid, lo):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)                   # representable in bf16
    s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    # the planes shrink by 2^-8 each at least: |mid| <= 2^-8 |hi|, |lo| <= 2^-16 |hi| (what bounds the dropped products)
    nz = hi != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0 ** -8) and np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0 ** -16)


def test_plane_products_are_exact_in_fp32_and_the_dropped_ones_are_below_2_to_minus_24():
    rng = np.random.default_rng(1)
    x, w = _samples(rng, 500_000), _samples(rng, 500_000)
    xs, ws = split3(x), split3(w)
    for a in xs:
        for b in ws:
            assert np.array_equal((a * b).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))   # fp32 product == exact product
    exact = x.astype(np.float64) * w.astype(np.float64)
    dropped = sum(xs[i].astype(np.float64) * ws[j].astype(np.float64) for i, j in ((1, 2), (2, 1), (2, 2)))
    assert np.max(np.abs(dropped) / np.abs(exact)) < 2.0 ** -23.5
    kept = sum(xs[i].astype(np.float64) * ws[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)))
    assert np.allclose(kept + dropped, exact, rtol=0, atol=0)


def test_a_576_term_dot_product_by_six_plane_products_is_fp32_accurate():
    """the shape of one output of a 64-channel 3x3 convolution: K = 576 in k-steps of 32, products in the kernels' order, fp32 accumulation"""
    rng = np.random.default_rng(2)
    n, K = 4000, 576
    x = rng.standard_normal((n, K)).astype(np.float32)
    w = rng.standard_normal((n, K)).astype(np.float32) * np.float32(0.06)
    ref = np.sum(x.astype(np.float64) * w.astype(np.float64), axis=1)
    xs, ws = split3(x), split3(w)
    acc = np.zeros(n, np.float32)
    for k0 in range(0, K, 32):                                           # one MFMA k-step: 32 exact products summed into the fp32 accumulator
        for wi, xi in ((0, 2), (0, 1), (0, 0), (1, 1), (1, 0), (2, 0)):  # k_chain_s3's order of the six products
            blk = (ws[wi][:, k0:k0 + 32].astype(np.float64) * xs[xi][:, k0:k0 + 32].astype(np.float64)).sum(axis=1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)       # (the instruction rounds once per accumulation)
    plain = np.zeros(n, np.float32)
    for k in range(K):                                                   # an ordinary fp32 FMA chain
        plain = (plain.astype(np.float64) + x[:, k].astype(np.float64) * w[:, k].astype(np.float64)).astype(np.float32)
    scale = np.sqrt(np.sum((x.astype(np.float64) * w.astype(np.float64)) ** 2, axis=1))
    e_split, e_plain = np.abs(acc - ref) / scale, np.abs(plain - ref) / scale
    assert e_split.max() < 2e-6 and np.mean(e_split) <= 1.05 * np.mean(e_plain) + 1e-9
    assert np.max(np.abs(acc - ref) / (1.0 + np.abs(ref))) < 1e-6        # two orders inside north_star's 1e-5 (1 + |x|)
