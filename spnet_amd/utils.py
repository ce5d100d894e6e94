"""Data / target codec and output helpers -- the host-side surface of the reference's spnet/utils.py.

  grid codec      setup_means_and_ranges, true_to_pred_grid, norm_Y, denorm_Y   (utils.py:144-244)
  metadata        parse_meta_file, build_Y                                      (utils.py:260-320)
  frames          build_X (PNG -> float32 [N,H,W,1] in [-1,1]), build_dataset   (utils.py:325-482)
  outputs         cleanup_antinode_vars, show_pred_ellipses (overlay PNGs + hawley_spnet.csv)
                                                                                (utils.py:56-137)
OpenCV is not available on the target image: overlays are drawn with PIL (anti-aliased polygon
outline instead of cv2.ellipse with shift=10), so the PNGs are visually equivalent, not bit-identical;
the CSV rows are computed exactly as the reference does.
"""
import errno
import glob
import math
import os
import random
from functools import partial
from multiprocessing import cpu_count
from operator import itemgetter

import numpy as np
import pandas as pd
from PIL import Image, ImageDraw

from . import config as cf

orig_img_dims = [512, 384]
means = []
ranges = []


def make_sure_path_exists(path):
    try:
        os.makedirs(path)
    except OSError as exc:
        if exc.errno != errno.EEXIST:
            raise


# ----------------------------------------------------------------------------- grid codec
def setup_means_and_ranges(pred_shape):
    """Per-cell means / ranges / 'blank' defaults of the predictor grid (utils.py:144-176).
    NOTE the first grid axis indexes x (image width), the second y."""
    global means, ranges
    cx_min, cy_min, cx_max, cy_max = 40, 40, 470, 350
    nx, ny = int(pred_shape[0]), int(pred_shape[1])
    xbinsize = int((cx_max - cx_min) / nx)
    ybinsize = int((cy_max - cy_min) / ny)
    shape = tuple(int(v) for v in pred_shape)
    gridmeans = np.zeros(shape, dtype=cf.dtype)
    gridranges = np.zeros(shape, dtype=cf.dtype)
    gridYi = np.zeros(shape, dtype=cf.dtype)
    xs = np.arange(nx) * xbinsize + cx_min + xbinsize / 2
    ys = np.arange(ny) * ybinsize + cy_min + ybinsize / 2
    gx, gy = np.meshgrid(xs, ys, indexing="ij")
    for slot in range(shape[2]):
        #                      cx   cy   a             b             cos2t sin2t noobj rings
        gridYi[:, :, slot, 0], gridYi[:, :, slot, 1] = gx, gy
        gridYi[:, :, slot, 2:] = [xbinsize / 2, ybinsize / 2, -1, 0, 1, 0]
        gridmeans[:, :, slot, 0], gridmeans[:, :, slot, 1] = gx, gy
        gridmeans[:, :, slot, 2:] = [xbinsize / 2, ybinsize / 2, 0, 0, 0, 5]
        gridranges[:, :, slot, :] = [xbinsize, ybinsize, xbinsize, ybinsize, 2, 2, 1, 10]
    means = gridmeans.flatten()
    ranges = gridranges.flatten()
    return cx_min, cy_min, cx_max, cy_max, xbinsize, ybinsize, gridYi


def norm_Y(Y, set_means_ranges=False):
    return (Y - means) / ranges


def denorm_Y(normY):
    return normY * ranges + means


def true_to_pred_grid(true_arr, pred_shape, num_classes=11, img_filename=None):
    """Scatter one image's ellipses [cx,cy,a,b,cos2t,sin2t,noobj,rings] over the grid of predictors
    (utils.py:191-244): cell = clip(int((c - 40)/bin)), next free slot, AssertionError on overflow."""
    cx_min, cy_min, _, _, xbinsize, ybinsize, gridYi = setup_means_and_ranges(pred_shape)
    true_arr = np.asarray(true_arr)
    taken = np.zeros(gridYi.shape[0:2], dtype=int)
    if true_arr.ndim < 2:
        return gridYi
    for row in true_arr:
        ix = min(max(int((row[0] - cx_min) / xbinsize), 0), int(pred_shape[0]) - 1)
        iy = min(max(int((row[1] - cy_min) / ybinsize), 0), int(pred_shape[1]) - 1)
        assert taken[ix, iy] < pred_shape[2], \
            "grid cell (%d,%d) already holds %d ellipses (%s)" % (ix, iy, taken[ix, iy], img_filename)
        gridYi[ix, iy, taken[ix, iy]] = row
        taken[ix, iy] += 1
    return gridYi


def add_to_stack(a, b):
    return [b] if a is None else a + [b]


def nearest_multiple(a, b):
    return int(a / b) * b


def parse_meta_file(meta_filename):
    """CSV rows cx,cy,a,b,angle,rings -> sorted list of [cx,cy,a,b,cos2t,sin2t,0,rings] (utils.py:260-286)."""
    try:
        df = pd.read_csv(meta_filename, header=None, names=['cx', 'cy', 'a', 'b', 'angle', 'rings'])
    except pd.errors.EmptyDataError:
        return []
    df.drop_duplicates(inplace=True)
    out = []
    for cx, cy, a, b, angle, rings in df.itertuples(index=False):
        angle = float(angle)
        if b > a:
            a, b, angle = b, a, angle + 90
        if rings > 0.0:
            t = 2 * np.deg2rad(angle)
            out.append([cx, cy, a, b, np.cos(t), np.sin(t), 0, rings])
    return sorted(out, key=itemgetter(0, 1))


def build_Y(total_load, meta_file_list, img_file_list, pred_grid=[6, 6, 2], set_means_ranges=False):
    pred_shape = np.array([pred_grid[0], pred_grid[1], pred_grid[2], cf.vars_per_pred], dtype=int)
    num_outputs = int(np.prod(pred_shape))
    Y = np.zeros([total_load, num_outputs], dtype=cf.dtype)
    for i in range(total_load):
        if 0 == i % 5000:
            print("      Reading metadata file i =", i, "/", total_load, ":", meta_file_list[i])
        rows = np.array(parse_meta_file(meta_file_list[i]))
        Y[i, :] = true_to_pred_grid(rows, pred_shape, img_filename=img_file_list[i]).flatten()
    if total_load == 0:
        setup_means_and_ranges(pred_shape)
    return norm_Y(Y, set_means_ranges=set_means_ranges), pred_shape


# ----------------------------------------------------------------------------- frames
def _load_one(force_dim, grayscale, filename, as_uint8=False):
    img = Image.open(filename).convert("RGB")
    if force_dim is not None:
        img = img.resize((force_dim, force_dim), Image.LANCZOS)     # PIL.Image.ANTIALIAS of the reference
    if as_uint8:           # grey levels as decoded (channel 0); Model.predict scales them on the device
        return np.asarray(img, dtype=np.uint8)[:, :, 0:1]
    arr = np.asarray(img, dtype=np.float32)
    arr = (arr / 255.0 - 0.5) * 2.0
    return arr[:, :, 0:1] if grayscale else arr


def build_X(total_load, img_file_list, force_dim=224, grayscale=False, as_uint8=False):
    """PNG -> X float32 [N,H,W,C] scaled to [-1,1]; channel 0 only when grayscale (utils.py:325-421).
    as_uint8 (additive, grayscale only): keep the decoded grey levels, uint8 [N,H,W,1]; Model.predict applies the
    same scaling on the device (bit-identical), and the frames cross PCIe as bytes."""
    print("      Reading images and assigning as input X...")
    if as_uint8 and not grayscale:
        raise ValueError("as_uint8 is for grayscale input (model_type 'big' / 'monolithic')")
    first = _load_one(force_dim, grayscale, img_file_list[0], as_uint8)
    img_dims = first.shape
    X = np.zeros((total_load,) + img_dims, dtype=np.uint8 if as_uint8 else cf.dtype)
    worker = partial(_load_one, force_dim, grayscale, as_uint8=as_uint8)
    nproc = min(cpu_count(), max(1, total_load // 64))
    if nproc > 1:
        # forked workers are only safe while this process has not initialised the GPU (train_spnet.py evaluates and
        # predicts after training in one process): afterwards load from freshly spawned interpreters
        import multiprocessing
        from .fake_espi import gpu_may_be_live
        gpu_live = gpu_may_be_live()
        pool = multiprocessing.get_context("spawn" if gpu_live else "fork").Pool(nproc)
        try:
            for i, arr in enumerate(pool.imap(worker, img_file_list[0:total_load], chunksize=32)):
                X[i] = arr
        finally:
            pool.close()
            pool.join()
    else:
        for i in range(total_load):
            X[i] = worker(img_file_list[i])
    return X, img_dims


def build_dataset(path="Train/", load_frac=1.0, set_means_ranges=False, pred_grid=[6, 6, 2], batch_size=None,
                  shuffle=True):
    """X, Y, img_file_list, pred_shape for one directory of PNG + same-stem CSV (utils.py:425-482)."""
    if cf.model_type == 'simple':
        grayscale, force_dim = False, 224
    elif cf.model_type == 'big':
        grayscale, force_dim = True, None        # native 384x512 frames (predict_spnet.py:47-55)
    else:
        grayscale, force_dim = True, 331
    print("Loading data from", path, ", fraction =", load_frac)
    img_file_list = sorted(glob.glob(path + '*.png'))
    meta_file_list = sorted(glob.glob(path + '*' + cf.meta_extension))
    assert len(img_file_list) == len(meta_file_list), \
        "Error: len(img_file_list) = %d but len(meta_file_list) = %d" % (len(img_file_list), len(meta_file_list))
    if shuffle and img_file_list:
        pairs = list(zip(img_file_list, meta_file_list))
        random.shuffle(pairs)
        img_file_list, meta_file_list = zip(*pairs)
    total_files = len(img_file_list)
    total_load = int(total_files * load_frac)
    if batch_size is not None:
        total_load = nearest_multiple(total_load, batch_size)
    print("      Total files = ", total_files, ", going to load total_load = ", total_load)
    Y, pred_shape = build_Y(total_load, meta_file_list, img_file_list, pred_grid=pred_grid,
                            set_means_ranges=set_means_ranges)
    X, _ = build_X(total_load, img_file_list, force_dim=force_dim, grayscale=grayscale)
    return X, Y, img_file_list, pred_shape


# ----------------------------------------------------------------------------- outputs
def cleanup_antinode_vars(Y_subarr):
    """(cx,cy,a,b,angle_deg,noobj,rings) with integer geometry and angle in (0,180] (utils.py:56-64)."""
    cx, cy, a, b, cos2t, sin2t, noobj, rings = Y_subarr
    cx, cy, a, b, noobj = [int(round(v)) for v in (cx, cy, a, b, noobj)]
    angle = np.rad2deg(np.arctan2(sin2t, cos2t) / 2.0)
    angle = angle if angle > 0 else angle + 180
    return cx, cy, a, b, angle, noobj, rings


def ellipse_polygon(center, axes, angle_deg, n=72):
    """Vertices of the ellipse the reference draws with cv2.ellipse(..., -angle, ...) (utils.py:35-53):
    image y points down, so a positive annotation angle rotates counter-clockwise on screen."""
    t = np.linspace(0, 2 * np.pi, n, endpoint=False)
    th = np.deg2rad(-angle_deg)
    x = axes[0] * np.cos(t)
    y = axes[1] * np.sin(t)
    return np.stack([center[0] + x * np.cos(th) - y * np.sin(th), center[1] + x * np.sin(th) + y * np.cos(th)], 1)


def draw_ellipse(img, center, axes, angle, startAngle=0, endAngle=360, color=(0,), thickness=2, **_):
    """Draw on a PIL image (in place).  thickness < 0 fills.  `color` is BGR like the reference's."""
    pts = [tuple(p) for p in ellipse_polygon(center, axes, angle)]
    rgb = tuple(color[::-1]) if len(color) == 3 else color[0]
    d = ImageDraw.Draw(img)
    if thickness < 0:
        d.polygon(pts, fill=rgb)
    else:
        d.line(pts + [pts[0]], fill=rgb, width=int(thickness), joint="curve")
    return img


def show_pred_ellipses(Yt, Yp, file_list, num_draw=40, log_dir='./logs/', ind_extra=None, out_csv=None,
                       show_true=True, verbosity=0):
    """Overlay PNGs steelpan_pred_%05d.png and the zooniverse-style CSV (cx,cy,file,rings,a,b,angle)
    from de-normalised true / predicted grids (utils.py:67-137)."""
    m = Yt.shape[0]
    num_draw = min(num_draw, m, len(file_list))
    if out_csv is not None:
        open(out_csv, "w").close()
    n_pred = int(Yt[0].size / cf.vars_per_pred)
    for j in range(num_draw):
        in_filename = file_list[j]
        img = Image.open(in_filename).convert("RGB")
        todraw = ([dict(name='True', Y=Yt, color=cf.truecolor)] if show_true else []) + \
                 [dict(name='Pred', Y=Yp, color=cf.predcolor)]
        csv_str = ''
        d = ImageDraw.Draw(img)
        for an in range(n_pred):
            for td in todraw:
                cx, cy, a, b, angle, noobj, rings = cleanup_antinode_vars(
                    td['Y'][j, an * cf.vars_per_pred:(an + 1) * cf.vars_per_pred])
                if noobj == 0 and rings > 0 and a >= 0 and b >= 0:
                    draw_ellipse(img, [cx, cy], [a, b], angle, color=td['color'], thickness=3)
                    d.text((cx - 10, cy - 8 + (0 if td['name'] == 'True' else 14)), "{: >3.1f}".format(rings),
                           fill=tuple(td['color'][::-1]))
                    if td['name'] == 'Pred' and out_csv is not None:
                        csv_str += "{},{},{},{},{},{},{}\n".format(cx, cy, os.path.basename(in_filename), rings, a, b, angle)
        if out_csv is not None and csv_str == '':
            csv_str = '0,0,' + os.path.basename(in_filename) + ',0,0,0,0\n'
        d.text((5, orig_img_dims[1] - 14), os.path.basename(in_filename), fill=(255, 255, 255))
        img.save(log_dir + '/steelpan_pred_' + str(j).zfill(5) + '.png')
        if out_csv is not None:
            with open(out_csv, "a") as f:
                f.write(csv_str)
