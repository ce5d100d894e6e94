"""TEST INFRASTRUCTURE -- CPU restatement of the reference's fake-ESPI generator (gen_fake_espi.py), the input distribution
of the benchmark.  Only tests/ may import this module; the product generator is
spnet_amd/fake_espi.py (host) + spnet_amd/csrc/espi.hip (device).

What is restated, and how it is pinned:

* PARAMETER DRAWS in the reference's own RNG call order -- Python's `random` module and numpy's global generator,
  interleaved exactly as the reference interleaves them: draw_waves (gen_fake_espi.py:60-80), the antinode count
  (:250-251), draw_antinodes incl. its rejection loop (:145-206), draw_rings' rand_start and ring geometry (:101-114),
  draw_ellipse's argument conversion (spnet/utils.py:35-53).  PINNED: tests/golden/make_goldens.py runs the reference's
  own draw_waves / draw_antinodes (imported under inert stubs, their cv2.polylines / draw_ellipse calls recorded) from
  seeded generators and stores every argument they passed to OpenCV plus the generators' next outputs;
  tests/test_oracle_numpy.py requires this module to reproduce them exactly.
* RASTERISATION of those calls in numpy: cv2.polylines(thickness = t) and cv2.ellipse(thickness = t, LINE_AA, shift = 10)
  draw the set of pixels within t / 2 of the curve (OpenCV: a filled quadrilateral per segment plus round joins, in
  fixed-point coordinates).  OpenCV (opencv_python 3.4.0.12, requirements.txt) is not in the image and not vendored, so
  this part is a restatement of the published geometry -- distance to the polyline <= t / 2 -- WITHOUT OpenCV's sub-pixel
  rounding and without the anti-aliased rim of LINE_AA ellipses (the rim's intermediate greys are replaced by the nearer
  of the two levels): PARITY UNPINNED at pixel level, by a fraction of a pixel along outlines.
* SENSOR MODEL (:258-264): additive cv2.randn(mean 40, sigma 40) saturated to uint8, then a Bernoulli(0.5) pixel mask
  from np.random.choice.  cv2.randn's generator is OpenCV's own and cannot be reproduced: the noise here is numpy's
  normal generator with the same saturation (distribution parity only); the mask follows the reference's call.
  blur_inplace (:254, spnet/augmentation.py:66-70) discards its result -- a no-op on the pixels that still consumes one
  np.random.random() and, three times in ten, one random.choice: restated as such.
  bandpass_mixup (:267) needs the author's private images: out of scope.
"""
import numpy as np

IM_W, IM_H = 512, 384                    # gen_fake_espi.py:31-32
GREY, BLACK, RING = 128, 0, 138          # :39-41, :111 (grey + 10)
MIN_LINE_WIDTH = 4                       # :46


def draw_waves_params(rnd, nprnd):
    """gen_fake_espi.py:64-69 -> (amp, x_wavelength, thickness, slope, y_spacing, numlines)."""
    amp = rnd.randint(10, 200)
    x_wavelength = rnd.randint(100, int(IM_W / 2))
    thickness = rnd.randint(15, 40)
    slope = 3 * (nprnd.rand() - .5)
    y_spacing = rnd.randint(thickness + thickness * int(np.abs(1.5 * slope)), int(IM_H / 3))
    numlines = 60 + int(IM_H / y_spacing)
    return amp, x_wavelength, thickness, slope, y_spacing, numlines


def wave_polylines(params):
    """The int32 point lists draw_waves hands to cv2.polylines (:71-78): [numlines][512][2]."""
    amp, x_wavelength, thickness, slope, y_spacing, numlines = params
    xs = np.arange(0, IM_W)
    out = np.zeros((numlines, IM_W, 2), np.int32)
    for j in range(numlines):
        y_start = j * y_spacing - IM_W * abs(slope)              # img.shape[1] is the width of the (H, W, 1) canvas
        for i in range(IM_W):
            out[j, i] = (int(xs[i]), int(y_start + slope * xs[i] + amp * np.cos(xs[i] / x_wavelength)))
    return out


def ellipse_box(center, axes, angle):
    """get_ellipse_box (:82-98)."""
    rad = np.radians(angle)
    a, b = axes
    dx = np.sqrt(a ** 2 * np.cos(rad) ** 2 + b ** 2 * np.sin(rad) ** 2)
    dy = np.sqrt(a ** 2 * np.sin(rad) ** 2 + b ** 2 * np.cos(rad) ** 2)
    return [center[0] - dx, center[1] - dy, center[0] + dx, center[1] + dy]


def _overlap(a, b):                        # does_overlap (:116-129)
    return not (a[2] < b[0] or a[0] > b[2] or a[3] < b[1] or a[1] > b[3])


def ring_calls(center, axes, angle, num_rings, nprnd):
    """draw_rings (:101-114): the (center, axes, angle, color, thickness) of every draw_ellipse call, innermost first."""
    nwb = 2 * num_rings
    if nwb == 0:
        nwb = 1
    thickness = int(round(min(axes) / nwb))
    rand_start = nprnd.choice([0, 1])
    calls = []
    for j in range(nwb):
        color = BLACK if (rand_start + j) % 2 == 0 else RING
        ring_axes = [axes[i] * (j + 1) * 1.0 / (nwb + 1) for i in range(2)]
        calls.append((center, ring_axes, angle, color, thickness))
    return calls


def draw_antinodes_params(num_antinodes, rnd, nprnd):
    """draw_antinodes (:145-206) -> (rows [(cx, cy, a, b, angle, rings)], every draw_ellipse call in drawing order)."""
    boxes, rows, calls = [], [], []
    for _ in range(num_antinodes):
        axes = sorted((rnd.randint(15, int(IM_W / 3.5)), rnd.randint(15, int(IM_H / 3.5))), reverse=True)
        max_rings = min(axes[1] // 8, 11)
        num_rings = rnd.randint(1, max_rings)
        if axes[1] / num_rings < MIN_LINE_WIDTH:
            num_rings = axes[1] // MIN_LINE_WIDTH
        center = (rnd.randint(axes[0], IM_W - axes[0]), rnd.randint(axes[1], IM_H - axes[1]))
        angle = rnd.randint(1, 179)
        box = ellipse_box(center, axes, angle)
        trycount, maxtries = 0, 2000
        while (any(_overlap(box, b) for b in boxes) or box[0] < 0 or box[2] > IM_W or box[1] < 0 or box[3] > IM_H) \
                and trycount < maxtries:
            trycount += 1
            axes = sorted((rnd.randint(25, int(IM_W / 3)), rnd.randint(25, int(IM_H / 3))), reverse=True)
            if axes[1] / num_rings < MIN_LINE_WIDTH:
                num_rings = axes[1] // MIN_LINE_WIDTH
            center = (rnd.randint(axes[0], IM_W - axes[0]), rnd.randint(axes[1], IM_H - axes[1]))
            angle = rnd.randint(1, 180)
            box = ellipse_box(center, axes, angle)
        if trycount < maxtries:
            calls += ring_calls(center, axes, angle, num_rings, nprnd)
            rows.append((center[0], center[1], axes[0], axes[1], angle, num_rings))
            boxes.append(box)
    return rows, calls


def frame_params(rnd, nprnd, count_range=(1, 7)):
    """One frame's draws in gen_images' order (:243-252): waves, antinode count, antinodes."""
    waves = draw_waves_params(rnd, nprnd)
    n = rnd.randint(count_range[0], count_range[1])
    rows, calls = draw_antinodes_params(n, rnd, nprnd)
    return waves, rows, calls


def caption(rows):
    """the CSV text gen_images writes (:197, :274-276)"""
    return "\n".join("{0},{1},{2},{3},{4},{5}".format(*r) for r in rows)


def cv_ellipse_args(center, axes, angle, shift=10):
    """What draw_ellipse (spnet/utils.py:41-52) passes to cv2.ellipse: fixed-point centre and axes, the NEGATED angle."""
    c = (int(round(center[0] * 2 ** shift)), int(round(center[1] * 2 ** shift)))
    a = (int(round(axes[0] * 2 ** shift)), int(round(axes[1] * 2 ** shift)))
    return c, a, -angle


# ------------------------------------------------------------------------------------------------ rasterisation (numpy)
def _dist2_to_segments(px, py, x0, y0, x1, y1):
    """squared distance of pixels (px, py: [P]) to segments ([S]): [P] min over segments, in blocks of segments"""
    best = np.full(px.shape, np.inf)
    for lo in range(0, len(x0), 64):
        ax, ay, bx, by = (v[lo:lo + 64][None, :] for v in (x0, y0, x1, y1))
        dx, dy = bx - ax, by - ay
        ll = np.maximum(dx * dx + dy * dy, 1e-12)
        t = np.clip(((px[:, None] - ax) * dx + (py[:, None] - ay) * dy) / ll, 0.0, 1.0)
        d2 = (px[:, None] - (ax + t * dx)) ** 2 + (py[:, None] - (ay + t * dy)) ** 2
        best = np.minimum(best, d2.min(axis=1))
    return best


def _stroke(img, pts, thickness, color, closed):
    """pixels within thickness / 2 of the polyline `pts` ([n][2], float) take `color`"""
    pts = np.asarray(pts, np.float64)
    r = thickness / 2.0
    x0, y0 = pts[:-1, 0], pts[:-1, 1]
    x1, y1 = pts[1:, 0], pts[1:, 1]
    if closed:
        x0, y0 = np.append(x0, pts[-1, 0]), np.append(y0, pts[-1, 1])
        x1, y1 = np.append(x1, pts[0, 0]), np.append(y1, pts[0, 1])
    lo_x, hi_x = int(np.floor(pts[:, 0].min() - r)), int(np.ceil(pts[:, 0].max() + r))
    lo_y, hi_y = int(np.floor(pts[:, 1].min() - r)), int(np.ceil(pts[:, 1].max() + r))
    lo_x, lo_y, hi_x, hi_y = max(lo_x, 0), max(lo_y, 0), min(hi_x, IM_W - 1), min(hi_y, IM_H - 1)
    if lo_x > hi_x or lo_y > hi_y:
        return
    yy, xx = np.mgrid[lo_y:hi_y + 1, lo_x:hi_x + 1]
    d2 = _dist2_to_segments(xx.ravel().astype(np.float64), yy.ravel().astype(np.float64), x0, y0, x1, y1)
    sub = img[lo_y:hi_y + 1, lo_x:hi_x + 1]
    sub[(d2 <= r * r).reshape(sub.shape)] = color


def raster(waves, calls):
    """The noise-free canvas (:243-252): grey 128, the wave bands (black), then every ring in drawing order."""
    img = np.full((IM_H, IM_W), GREY, np.uint8)
    amp, x_wavelength, thickness, slope, y_spacing, numlines = waves
    lines = wave_polylines(waves)
    for j in range(numlines):
        ys = lines[j, :, 1]
        if ys.max() < -thickness or ys.min() > IM_H + thickness:
            continue
        _stroke(img, lines[j].astype(np.float64), thickness, BLACK, closed=False)
    for center, axes, angle, color, thickness in calls:
        t = np.deg2rad(np.arange(360))
        th = np.deg2rad(-angle)                          # draw_ellipse passes -angle (spnet/utils.py:50)
        u, v = axes[0] * np.cos(t), axes[1] * np.sin(t)
        pts = np.stack([center[0] + u * np.cos(th) - v * np.sin(th), center[1] + u * np.sin(th) + v * np.cos(th)], 1)
        _stroke(img, pts, thickness, color, closed=True)
    return img


def sensor(img, rnd, nprnd):
    """blur_inplace's RNG consumption (no-op on the pixels), saturated N(40, 40) noise, Bernoulli(0.5) mask (:254-264)."""
    if nprnd.random_sample() <= 0.3:                     # spnet/augmentation.py:67-69
        rnd.choice([3, 7])
    noise = np.clip(np.rint(nprnd.normal(40, 40, img.shape)), 0, 255)
    out = np.minimum(img.astype(np.float64) + noise, 255)
    mask = nprnd.choice([0, 1], size=img.shape).astype(np.float32)
    return (out * mask).astype(np.uint8)
