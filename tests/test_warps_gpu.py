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
        # uint8 frames: OpenCV's fixed-point algorithm, integer arithmetic on both sides -> bit-exact
        np.testing.assert_array_equal(out, WR.warp_affine_cv2(img, WR.rotation_matrix((256, 192), ang)))
        # ... which stays within the quantisation of the exact-float bilinear result (1/32-pixel coordinates)
        want = WR.warp_affine(img, WR.rotation_matrix((256, 192), ang))
        assert np.abs(out.astype(np.float64) - want).mean() < 1.5
        # float frames keep the exact-float bilinear kernel
        outf, _, _ = A.rotate_image(img.astype(np.float32), MD, "r", ang)
        assert np.abs(outf - want).max() < 0.1        # grey levels 0..255; fp32 source coordinates (~1e-5 px)
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
