"""numpy restatement of the offline warps (TEST INFRASTRUCTURE; PARITY UNPINNED -- OpenCV is absent).

  cv2.flip / cv2.getRotationMatrix2D / cv2.warpAffine as the reference calls them in
  spnet/augmentation.py:82-112 (flip_image), :184-207 (rotate_image), :216-239 (translate_image),
  driven by augment_preproc.py:74-99.
Bilinear interpolation with exact float weights and a zero border (OpenCV quantises the weights to
1/32 pixel and rounds in fixed point, so its uint8 result can differ by 1 grey level).
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
