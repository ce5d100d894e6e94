"""numpy restatement of the offline warps (TEST INFRASTRUCTURE; PARITY UNPINNED -- OpenCV is absent).

  cv2.flip / cv2.getRotationMatrix2D / cv2.warpAffine as the reference calls them in
  spnet/augmentation.py:82-112 (flip_image), :184-207 (rotate_image), :216-239 (translate_image),
  driven by augment_preproc.py:74-99.
warp_affine: bilinear interpolation with exact float weights and a zero border (float images).
warp_affine_cv2: OpenCV's fixed-point algorithm for 8-bit images (1/32-pixel coordinates, 15-bit weights), the path the
reference's uint8 frames take.
"""
import numpy as np


def rotation_matrix(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D: positive angle = counter-clockwise (image y axis pointing down)."""
    a = scale * np.cos(np.deg2rad(angle_deg))
    b = scale * np.sin(np.deg2rad(angle_deg))
    return np.array([[a, b, (1 - a) * center[0] - b * center[1]],
                     [-b, a, b * center[0] + (1 - a) * center[1]]], np.float64)


def warp_affine(img, M):
    """dst(x,y) = src(M^-1 (x,y)) for a forward 2x3 matrix M, bilinear, zero outside.  img [H,W,C]."""
    H, W = img.shape[:2]
    Minv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
    ys, xs = np.mgrid[0:H, 0:W]
    sx = (Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]).astype(np.float32)
    sy = (Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]).astype(np.float32)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    wx, wy = (sx - x0)[..., None], (sy - y0)[..., None]
    src = img.astype(np.float32)

    def at(y, x):
        ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
        out = np.zeros(ys.shape + src.shape[2:], np.float32)
        out[ok] = src[y[ok], x[ok]]
        return out
    top = at(y0, x0) + wx * (at(y0, x0 + 1) - at(y0, x0))
    bot = at(y0 + 1, x0) + wx * (at(y0 + 1, x0 + 1) - at(y0 + 1, x0))
    return top + wy * (bot - top)


def warp_affine_cv2(img, M):
    """cv2.warpAffine(uint8 img [H,W,C], M, (W,H)) restated from OpenCV 3.4's published algorithm (modules/imgproc/src/
    imgwarp.cpp: warpAffine, WarpAffineInvoker, remapBilinear with the fixed-point table; INTER_LINEAR, BORDER_CONSTANT
    0) in plain integer numpy -- PARITY UNPINNED like the rest of this file (OpenCV cannot run here), but the
    arithmetic is integer, so a faithful restatement is bit-exact by construction:
      M inverted in double; adelta/bdelta = cvRound(M00|M10 * x * 2^10); X0/Y0 = cvRound((M01|M11 * y + M02|M12) * 2^10)
      + 16; X = (X0 + adelta) >> 5 (1/32 px); weights (32-fy)(32-fx)*32 of 2^15 (int16 table: 32768 saturates to
      32767, harmless for 8-bit pixels); result = (sum w p + 2^14) >> 15."""
    H, W = img.shape[:2]
    m = np.asarray(M, np.float64).reshape(2, 3).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = A11, m[0, 1] * -D, m[1, 0] * -D, A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    adelta, bdelta = np.rint(m[0, 0] * xs * 1024).astype(np.int64), np.rint(m[1, 0] * xs * 1024).astype(np.int64)
    X0 = np.rint((m[0, 1] * ys + m[0, 2]) * 1024).astype(np.int64) + 16
    Y0 = np.rint((m[1, 1] * ys + m[1, 2]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    src = img.astype(np.int64)

    def at(y, x):
        ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
        out = np.zeros(y.shape + src.shape[2:], np.int64)
        out[ok] = src[y[ok], x[ok]]
        return out
    w00 = np.minimum((32 - fy) * (32 - fx) * 32, 32767)[..., None]
    w01, w10, w11 = ((32 - fy) * fx * 32)[..., None], (fy * (32 - fx) * 32)[..., None], (fy * fx * 32)[..., None]
    v = (at(sy, sx) * w00 + at(sy, sx + 1) * w01 + at(sy + 1, sx) * w10 + at(sy + 1, sx + 1) * w11 + (1 << 14)) >> 15
    return np.clip(v, 0, 255).astype(np.uint8)


def flip(img, code):
    """cv2.flip: 0 vertical (rows reversed), 1 horizontal, -1 both."""
    if code == 0:
        return img[::-1]
    if code == 1:
        return img[:, ::-1]
    return img[::-1, ::-1]


def cleanup_angle(angle):
    while angle < 0:
        angle += 180
    while angle >= 180:
        angle -= 180
    return angle


def flip_meta(md, code, width, height):
    """Metadata half of flip_image (augmentation.py:91-104): [cx,cy,a,b,angle,rings] rows."""
    out = []
    for cx, cy, a, b, angle, rings in md:
        if code in (0, -1):
            cy, angle = height - cy, -angle
        angle = cleanup_angle(angle)
        if code in (1, -1):
            cx, angle = width - cx, 180 - angle
        angle = cleanup_angle(angle)
        out.append([cx, cy, a, b, angle, rings])
    return out


def rotate_meta(md, rot_angle, width, height):
    """Metadata half of rotate_image (augmentation.py:196-204)."""
    M = rotation_matrix((width / 2, height / 2), rot_angle)
    out = []
    for cx, cy, a, b, angle, rings in md:
        p = M @ np.array([cx, cy, 1.0])
        out.append([int(round(p[0])), int(round(p[1])), a, b, cleanup_angle(angle + rot_angle), rings])
    return out
