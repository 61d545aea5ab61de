"""CPU: the Winograd F(2x2, 3x3) pieces the convolution kernels rely on (Lavin & Gray 2016), against an independent NumPy statement:
  * lz_wino_weights (the library's host-side weight transform, binary64 then one rounding) == G g G^T computed here;
  * with the input / output transforms the kernels hard-code (k_conv_wino / k_chain_w in lz_nn.hip: B^T d B and A^T M A, written out as
    additions there, as matrices here), sum_ci U * V reproduces the direct 3x3 / pad-1 correlation the reference's Conv2d computes
    (common.py:309-327 ResBlock convolutions) on whole 6x6 and 8x8 feature maps."""
import numpy as np

G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _lib_transform(w):
    from lightzero_amd import _lib as L
    cout, cin = w.shape[:2]
    u = np.zeros((16, cin, cout), np.float32)
    L.check(L.lib().lz_wino_weights(np.ascontiguousarray(w, np.float32), cout, cin, u))
    return u


def test_weight_transform_is_G_g_Gt_rounded_once():
    rng = np.random.default_rng(3)
    w = rng.standard_normal((5, 7, 3, 3)).astype(np.float32)
    u = _lib_transform(w)
    want = np.einsum("ik,ockl,jl->ijco", G, w.astype(np.float64), G).reshape(16, 7, 5).astype(np.float32)
    assert np.array_equal(u, want)


def _direct(x, w):  # x [cin][H][W], w [cout][cin][3][3]: correlation, pad 1 (torch Conv2d)
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W))
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, dy, dx], xp[:, dy:dy + H, dx:dx + W])
    return out


def _winograd(x, u):  # u [16][cin][cout] from the library
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((u.shape[2], H, W))
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]              # patch rows 2 ty - 1 .. 2 ty + 2 of the unpadded map
            V = np.einsum("ik,ckl,jl->ijc", BT, d, BT).reshape(16, cin)  # B^T d B per channel
            M = np.einsum("pc,pco->po", V, u.astype(np.float64)).reshape(4, 4, -1)
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ik,klo,jl->oij", AT, M, AT)
    return out


def test_winograd_pipeline_equals_the_direct_convolution():
    rng = np.random.default_rng(4)
    for hw in (6, 8):
        x = rng.standard_normal((8, hw, hw))
        w = rng.standard_normal((4, 8, 3, 3)).astype(np.float32)
        got = _winograd(x, _lib_transform(w))
        want = _direct(x, w.astype(np.float64))
        assert np.abs(got - want).max() < 2e-6 * (1 + np.abs(want).max())   # only the float32 rounding of U separates them
