"""Synthetic 'fake-ESPI' frames + labels: the benchmark / test input distribution.

Follows the reference generator gen_fake_espi.py:60-279 up to, but excluding, the band-pass mix-up
with the author's private real images (augmentation.py:10-62): 512x384 grey canvas at 128, wavy dark
bands (draw_waves :60-80), 1..7 non-overlapping ringed ellipses (draw_antinodes :145-206, draw_rings
:101-114), additive clipped N(40,40) noise, 50 % Bernoulli pixel dropout.  Rasterisation uses PIL
(OpenCV is not available); the parameter draws follow the reference's RNG call order (one `random.Random` +
one numpy RandomState per frame: checked against oracle/espi_ref.py, which is pinned to the reference's own
draws), the pixels are statistically, not bitwise, OpenCV's.  Exactly one PNG per CSV is written (the reference also writes
a *_bp.png that would trip build_dataset's file-count assertion, utils.py:455-459).
"""
import os

import numpy as np
from PIL import Image, ImageDraw

IM_W, IM_H = 512, 384
MIN_LINE_WIDTH = 4


def _ellipse_pts(center, axes, angle_deg, n=90):
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    th = np.deg2rad(-angle_deg)            # the reference passes -angle to cv2.ellipse (utils.py:50)
    x, y = axes[0] * np.cos(t), axes[1] * np.sin(t)
    return [(float(center[0] + u * np.cos(th) - v * np.sin(th)), float(center[1] + u * np.sin(th) + v * np.cos(th)))
            for u, v in zip(x, y)]


def _ellipse_box(center, axes, angle):
    rad = np.radians(angle)
    dx = np.sqrt(axes[0] ** 2 * np.cos(rad) ** 2 + axes[1] ** 2 * np.sin(rad) ** 2)
    dy = np.sqrt(axes[0] ** 2 * np.sin(rad) ** 2 + axes[1] ** 2 * np.cos(rad) ** 2)
    return [center[0] - dx, center[1] - dy, center[0] + dx, center[1] + dy]


def _overlaps(a, b):
    return not (a[2] < b[0] or a[0] > b[2] or a[3] < b[1] or a[1] > b[3])


def draw_params(seed, count_range=(1, 7)):
    """All random PARAMETERS of one frame, drawn in the reference generator's own RNG call order -- it interleaves Python's
    `random` module and numpy's generator, and so does this: one random.Random(seed) and one RandomState(seed) per frame
    (the reference seeds both global generators once per run, gen_fake_espi.py:317-318; a generator pair per frame lets
    frames be generated in any order and in parallel).
    waves = (amp, wavelength, thickness, slope, spacing) (draw_waves, gen_fake_espi.py:60-80) and
    nodes = [(cx, cy, a, b, angle, rings, ring_start)] (draw_antinodes :145-206 incl. the non-overlap rejection
    loop, draw_rings' rand_start :107).  Returns (waves, nodes, (rnd, rs)) -- the generators continue with the sensor-model
    draws of the host rasteriser.  count_range = (lo, hi) inclusive: the number of antinodes drawn per frame -- (1, 7) is
    the reference's current generator (gen_fake_espi.py:250-251); its comments there date that to 'Nov 11 2020 increasing
    from 6 to 7 ... elminating 0', i.e. the published Dataset-A run was generated with (0, 6): 3.0 objects per frame."""
    import random
    rnd, rs = random.Random(seed), np.random.RandomState(seed)
    amp = rnd.randint(10, 200)
    wavelength = rnd.randint(100, IM_W // 2)
    thick = rnd.randint(15, 40)
    slope = 3 * (rs.rand() - .5)
    spacing = rnd.randint(thick + thick * int(abs(1.5 * slope)), IM_H // 3)
    waves = (amp, wavelength, thick, slope, spacing)
    boxes, nodes = [], []
    for _ in range(rnd.randint(count_range[0], count_range[1])):
        axes = sorted((rnd.randint(15, int(IM_W / 3.5)), rnd.randint(15, int(IM_H / 3.5))), reverse=True)
        rings = rnd.randint(1, min(axes[1] // 8, 11))
        if axes[1] / rings < MIN_LINE_WIDTH:
            rings = axes[1] // MIN_LINE_WIDTH
        center = (rnd.randint(axes[0], IM_W - axes[0]), rnd.randint(axes[1], IM_H - axes[1]))
        angle = rnd.randint(1, 179)
        box = _ellipse_box(center, axes, angle)
        tries = 0
        while (any(_overlaps(box, b) for b in boxes) or box[0] < 0 or box[2] > IM_W or box[1] < 0 or box[3] > IM_H) \
                and tries < 2000:
            tries += 1
            axes = sorted((rnd.randint(25, IM_W // 3), rnd.randint(25, IM_H // 3)), reverse=True)
            if axes[1] / rings < MIN_LINE_WIDTH:
                rings = axes[1] // MIN_LINE_WIDTH
            center = (rnd.randint(axes[0], IM_W - axes[0]), rnd.randint(axes[1], IM_H - axes[1]))
            angle = rnd.randint(1, 180)
            box = _ellipse_box(center, axes, angle)
        if tries < 2000:
            nodes.append((center[0], center[1], axes[0], axes[1], angle, rings, int(rs.choice([0, 1]))))
            boxes.append(box)
    return waves, nodes, (rnd, rs)


def _draw_waves(d, waves):
    amp, wavelength, thick, slope, spacing = waves
    xs = np.arange(IM_W)
    base = slope * xs + amp * np.cos(xs / wavelength)
    for j in range(60 + IM_H // spacing):
        y0 = j * spacing - IM_W * abs(slope)
        ys = (y0 + base).astype(np.int64)
        if ys.max() < -thick or ys.min() > IM_H + thick:
            continue
        d.line(list(zip(xs.tolist(), ys.tolist())), fill=0, width=thick, joint="curve")


def _draw_rings(d, center, axes, angle, rings, start):
    nwb = max(2 * rings, 1)
    thick = max(int(round(min(axes) / nwb)), 1)
    for j in range(nwb):
        col = 0 if (start + j) % 2 == 0 else 138
        ax = [a * (j + 1) / (nwb + 1) for a in axes]
        pts = _ellipse_pts(center, ax, angle)
        d.line(pts + [pts[0]], fill=col, width=thick, joint="curve")


def raster_host(waves, nodes):
    """The noise-free canvas of one frame, rasterised with PIL: uint8 [384,512]."""
    img = Image.new("L", (IM_W, IM_H), 128)
    d = ImageDraw.Draw(img)
    _draw_waves(d, waves)
    for cx, cy, a, b, angle, rings, start in nodes:
        _draw_rings(d, (cx, cy), (a, b), angle, rings, start)
    return np.asarray(img, dtype=np.uint8)


def gen_frame(seed):
    """One frame: (uint8 [384,512] image, [(cx,cy,a,b,angle,rings), ...])."""
    waves, nodes, (rnd, rs) = draw_params(seed)
    a = raster_host(waves, nodes).astype(np.float32)
    if rs.random_sample() <= 0.3:             # blur_inplace (augmentation.py:66-70): a no-op on the pixels, consumes RNG
        rnd.choice([3, 7])
    noise = np.clip(np.rint(rs.normal(40, 40, a.shape)), 0, 255)      # cv2.randn into a uint8 image saturates
    a = np.minimum(a + noise, 255)
    a *= rs.choice([0, 1], size=a.shape)                                # drop half of the pixels (:262-264)
    return a.astype(np.uint8), [n[:6] for n in nodes]


def frame_seeds(n, seed):
    return [seed * 1000003 + i for i in range(n)]


def generate_device(n, seed=0, device="cuda:0", noise=True, want_u8=False, chunk=1024, count_range=(1, 7)):
    """n frames rasterised directly in HBM (csrc/espi.hip): the SAME per-frame parameters as generate(n, seed)
    (so the labels are identical), pixels from the analytic device rasteriser, sensor noise / dropout from a
    counter-based RNG.  Returns (float32 device tensor [n,384,512,1] in [-1,1], label rows[, uint8 device tensor])."""
    import torch
    from . import _lib as L
    dev = torch.device(device)
    X = torch.empty((n, IM_H, IM_W, 1), dtype=torch.float32, device=dev)
    U = torch.empty((n, IM_H, IM_W), dtype=torch.uint8, device=dev) if want_u8 else None
    labels = []
    st = torch.cuda.current_stream(dev).cuda_stream
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        waves = np.zeros((hi - lo, 5), np.float32)
        nodes = np.zeros((hi - lo, 7, 8), np.float32)
        nn = np.zeros(hi - lo, np.int32)
        for k, s in enumerate(frame_seeds(n, seed)[lo:hi]):
            w, nd, _ = draw_params(s, count_range)
            waves[k] = w
            nn[k] = len(nd)
            for j, node in enumerate(nd):
                nodes[k, j, :7] = node
                nodes[k, j, 7] = 1.0
            labels.append([node[:6] for node in nd])
        wd, ndd, nnd = (torch.from_numpy(a).to(dev) for a in (waves, nodes, nn))
        L.spnet_fake_espi(wd.data_ptr(), ndd.data_ptr(), nnd.data_ptr(), hi - lo, IM_H, IM_W,
                          (seed * 2654435761 + lo * 97 + 12345) & 0xFFFFFFFF, int(bool(noise)), X[lo:hi].data_ptr(),
                          U[lo:hi].data_ptr() if U is not None else None, st)
        torch.cuda.current_stream(dev).synchronize()       # wd / ndd / nnd are freed on return
    return (X, labels, U) if want_u8 else (X, labels)


def rows_to_csv(rows):
    if not rows:
        return "0,0,0,0,0,0.0"
    return "\n".join("{0},{1},{2},{3},{4},{5}".format(*r) for r in rows)


def gpu_may_be_live():
    """True when this process may already hold a HIP context, i.e. when fork() is unsafe: torch has initialised the
    GPU, or a profiler's preloaded tool library has (rocprofv3 initialises the runtime before Python starts, and
    torch.cuda.is_initialized() stays False then)."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return True
    env = os.environ
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in env):
        return True
    return any(t in env.get(k, "") for k in ("LD_PRELOAD", "HSA_TOOLS_LIB") for t in ("rocprof", "roctracer", "rocprofiler"))


def generate(n, seed=0, workers=None):
    """n frames -> (uint8 [n,384,512], list of label rows).  Deterministic in (n, seed)."""
    seeds = frame_seeds(n, seed)
    workers = workers or min(os.cpu_count() or 1, 16)
    # Worker processes are FORKED: safe only while this process has not initialised the GPU (a forked copy of a
    # HIP-initialised, multi-threaded process crashes or hangs).  Afterwards: small sets serially, large ones from
    # freshly spawned interpreters.
    ctx = None
    if workers > 1 and n >= 16:
        import multiprocessing
        if not gpu_may_be_live():
            ctx = multiprocessing.get_context("fork")
        elif n >= 512:
            ctx = multiprocessing.get_context("spawn")
    if ctx is not None:
        p = ctx.Pool(workers)
        try:
            res = p.map(gen_frame, seeds, chunksize=max(1, n // (workers * 4)))
        finally:                     # let the workers exit normally (Pool.__exit__ would terminate() = SIGTERM them)
            p.close()
            p.join()
    else:
        res = [gen_frame(s) for s in seeds]
    X = np.stack([r[0] for r in res])
    return X, [r[1] for r in res]


def to_network_input(X_u8):
    """uint8 [n,H,W] -> float32 [n,H,W,1] in [-1,1] exactly as load_X_one_proc scales (utils.py:340-342)."""
    x = X_u8.astype(np.float32) / 255.0
    x -= 0.5
    x *= 2.0
    return x[..., None]


def write_dataset(path, n, seed=0, start=0):
    """steelpan_NNNNNNN.png + .csv pairs under `path` (the layout build_dataset reads)."""
    os.makedirs(path, exist_ok=True)
    X, labels = generate(n, seed)
    for i in range(n):
        stem = os.path.join(path, "steelpan_" + str(start + i).zfill(7))
        Image.fromarray(X[i]).save(stem + ".png")
        with open(stem + ".csv", "w") as f:
            f.write(rows_to_csv(labels[i]))
    return X, labels
