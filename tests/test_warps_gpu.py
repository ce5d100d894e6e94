"""Offline warps (flip / rotate / translate): device kernel + host metadata math against
the numpy oracle (oracle/warp_ref.py; unpinned at the OpenCV boundary)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import warp_ref as WR


def _frame(seed=0, H=384, W=512):
    rs = np.random.RandomState(seed)
    img = (rs.rand(H, W, 1) * 255).astype(np.uint8)
    return np.repeat(img, 3, axis=2)


MD = [[100, 140, 120, 60, 30.0, 7], [400, 300, 50, 20, 170.0, 2], [256, 192, 80, 80, 0.0, 11]]


def test_flip_image():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import augmentation as A
    img = _frame(1)
    for code, suffix in ((0, "_v"), (1, "_h"), (-1, "_vh")):
        out, md, prefix = A.flip_image(img, MD, "f", code)
        np.testing.assert_array_equal(out, WR.flip(img, code))          # pure permutation: bit-exact
        assert md == WR.flip_meta(MD, code, 512, 384) and prefix == "f" + suffix
    out, md, prefix = A.flip_image(img, MD, "f", -2)
    np.testing.assert_array_equal(out, img)
    assert md == MD and prefix == "f"


def test_rotate_image():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import augmentation as A
    img = _frame(2)
    for ang in (17.5, -20.0, 3.25):
        out, md, prefix = A.rotate_image(img, MD, "r", ang)
        want = WR.warp_affine(img, WR.rotation_matrix((256, 192), ang))
        want_u8 = np.clip(np.floor(want + 0.5), 0, 255).astype(np.uint8)
        # same bilinear formula in f32 on both sides; allow a grey level where floor(x+0.5) sits on a tie
        assert np.abs(out.astype(int) - want_u8.astype(int)).max() <= 1
        assert (out != want_u8).mean() < 1e-3
        assert md == WR.rotate_meta(MD, ang, 512, 384)
        assert prefix == "r_r{:>.2f}".format(ang)
    out, md, prefix = A.rotate_image(img, MD, "r", 0)
    np.testing.assert_array_equal(out, img)


def test_translate_image():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import augmentation as A
    img = _frame(3)
    np.random.seed(5)
    xt = int(round(40 * (2 * np.random.random() - 1)))
    yt = int(round(40 * (2 * np.random.random() - 1)))
    np.random.seed(5)
    out, md, prefix = A.translate_image(img, MD, "t", 1)
    want = np.zeros_like(img)
    H, W = img.shape[:2]
    ys, xs = slice(max(yt, 0), H + min(yt, 0)), slice(max(xt, 0), W + min(xt, 0))
    ysrc, xsrc = slice(max(-yt, 0), H + min(-yt, 0)), slice(max(-xt, 0), W + min(-xt, 0))
    want[ys, xs] = img[ysrc, xsrc]
    np.testing.assert_array_equal(out, want)
    assert md == [[cx + xt, cy + yt, a, b, ang, r] for cx, cy, a, b, ang, r in MD]
    assert prefix == "t_t%d,%d" % (xt, yt)
