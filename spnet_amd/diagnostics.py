"""Evaluation metrics with the names of the reference's spnet/diagnostics.py.

  calc_errors    count-based metrics on de-normalised grids            (diagnostics.py:13-59)
  compute_iou    filled-ellipse raster IoU of one predictor pair      (diagnostics.py:64-120)
  precision,
  calc_map       precision at an IoU threshold, mAP@[.5:.95]          (diagnostics.py:125-161)

The reference rasterises with cv2.ellipse (LINE_AA, shift=10); OpenCV is absent here, so the masks are
the analytic point-in-ellipse test at pixel centres -- IoU values agree with OpenCV's to about one
boundary pixel ring (~1e-2 for small ellipses), not bitwise.  Each pair's IoU is computed once and
shared by the ten mAP thresholds (the reference recomputes it per threshold with the same result).

`calc_map(..., device=True)` / `precision(..., device=True)` run the same raster test for all 72 x N pairs in
one HIP launch (`spnet_ellipse_iou`, csrc/metrics.hip: the reference spends minutes here on the CPU); without
the flag the metric is computed on the host exactly as before.  The two agree to a boundary pixel or two per
pair (fp32 sin/cos of different math libraries).
"""
import numpy as np

from . import config as cf


def _calc_errors_device(Yp, Yt):
    """calc_errors in one HIP launch (csrc/metrics.hip: spnet_calc_errors); same nine return values."""
    import torch
    from . import _lib as L
    dev = torch.device("cuda", torch.cuda.current_device())
    yp = torch.as_tensor(np.ascontiguousarray(Yp, dtype=np.float32)).to(dev)
    yt = torch.as_tensor(np.ascontiguousarray(Yt, dtype=np.float32)).to(dev)
    counts = torch.zeros(7, dtype=torch.int32, device=dev)
    pix = torch.empty(yp.shape[0], dtype=torch.float32, device=dev)
    L.spnet_calc_errors(yp.data_ptr(), yt.data_ptr(), yp.shape[0], yp.shape[1], counts.data_ptr(), pix.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
    c = [int(v) for v in counts.cpu().numpy()]
    pix_err = pix.cpu().numpy()
    return (c[0], c[1], c[2], c[3], c[4], c[5], c[6], pix_err, int(np.argmax(pix_err)))


def calc_errors(Yp, Yt, device=False):
    """device=True: the same counts from the GPU (vars_per_pred == 8 layouts)."""
    if device and cf.vars_per_pred == 8:
        return _calc_errors_device(Yp, Yt)
    n_pred = int(Yt.shape[1] / cf.vars_per_pred)
    diff = Yp - Yt
    pix_err = np.sqrt(diff[:, 0] ** 2 + diff[:, 1] ** 2)
    ipem = int(np.argmax(pix_err))
    ring_miscounts = ring_truecounts = total_obj = 0
    false_obj_pos = false_obj_neg = true_obj_pos = true_obj_neg = 0
    for j in range(Yt.shape[0]):
        for an in range(n_pred):
            ind = cf.ind_rings + an * cf.vars_per_pred
            i_noobj = cf.ind_noobj + an * cf.vars_per_pred
            there = (0 == int(round(Yt[j, i_noobj])))
            predicted = (0 == int(round(Yp[j, i_noobj])))
            if there:
                total_obj += 1
                if predicted:
                    true_obj_pos += 1
                    if np.abs(Yt[j, ind] - Yp[j, ind]) > 0.5:
                        ring_miscounts += 1
                    else:
                        ring_truecounts += 1
                else:
                    false_obj_neg += 1
            elif predicted:
                false_obj_pos += 1
            else:
                true_obj_neg += 1
    return (ring_miscounts, ring_truecounts, total_obj, false_obj_pos, false_obj_neg, true_obj_pos, true_obj_neg,
            pix_err, ipem)


def create_ellipse_image(args, nx=512, ny=384):
    """uint8 [ny,nx] mask (255 inside) of one predictor's ellipse, zeros when noobj >= 0.5."""
    img = np.zeros((ny, nx), np.uint8)
    cx, cy, a, b, cos2t, sin2t, noobj, rings = args
    if noobj < 0.5 and a > 0 and b > 0:
        ang = np.arctan2(sin2t, cos2t) / 2.0
        th = -ang                       # the reference draws with -angle (image y points down)
        r = float(max(a, b)) + 1
        x0, x1 = int(max(0, np.floor(cx - r))), int(min(nx, np.ceil(cx + r) + 1))
        y0, y1 = int(max(0, np.floor(cy - r))), int(min(ny, np.ceil(cy + r) + 1))
        if x1 > x0 and y1 > y0:
            xs, ys = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1))
            dx, dy = xs - cx, ys - cy
            u = dx * np.cos(th) + dy * np.sin(th)
            v = -dx * np.sin(th) + dy * np.cos(th)
            img[y0:y1, x0:x1][(u / a) ** 2 + (v / b) ** 2 <= 1.0] = 255
    return img


def compute_iou(args_p, args_t, display=False):
    """IoU of predicted vs true ellipse; -1 when nothing is supposed to be there (true noobj > 0.99)
    or both rasters are empty."""
    if args_t[-2] > 0.99:
        return -1
    img_p, img_t = create_ellipse_image(args_p), create_ellipse_image(args_t)
    num_i = int(np.count_nonzero(img_p & img_t))
    num_u = int(np.count_nonzero(img_p | img_t))
    if num_i == 0 and num_u == 0:
        return -1
    return num_i / num_u


def _pair_ious_device(Yp, Yt):
    """All predictor pairs on the GPU; same list of (iou, noobj_p, noobj_t) as _pair_ious."""
    import torch
    from . import _lib as L
    v = cf.vars_per_pred
    yp = torch.from_numpy(np.ascontiguousarray(Yp, dtype=np.float32)).cuda().reshape(-1, v)
    yt = torch.from_numpy(np.ascontiguousarray(Yt, dtype=np.float32)).cuda().reshape(-1, v)
    iou = torch.empty(yp.shape[0], dtype=torch.float64, device="cuda")
    L.spnet_ellipse_iou(yp.data_ptr(), yt.data_ptr(), yp.shape[0], 512, 384, iou.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
    iou = iou.cpu().numpy()
    P, T = np.asarray(Yp, np.float32).reshape(-1, v), np.asarray(Yt, np.float32).reshape(-1, v)
    keep = np.nonzero(iou >= 0)[0]
    return [(float(iou[k]), P[k, -2], T[k, -2]) for k in keep]


def _pair_ious(Yp, Yt):
    out = []
    for i in range(Yp.shape[0]):
        for j in np.arange(0, Yp.shape[1], cf.vars_per_pred):
            args_p, args_t = Yp[i, j:j + cf.vars_per_pred], Yt[i, j:j + cf.vars_per_pred]
            iou = compute_iou(args_p, args_t)
            if iou >= 0:
                out.append((iou, args_p[-2], args_t[-2]))
    return out


def precision(Yp, Yt, thresh=0.5, _pairs=None, device=False):
    pairs = ((_pair_ious_device if device else _pair_ious)(Yp, Yt)) if _pairs is None else _pairs
    tp = fp = fn = 0
    for iou, noobj_p, noobj_t in pairs:
        if iou > thresh:
            tp += 1
        elif noobj_p < 0.5 and noobj_t >= 0.5:
            fp += 1
        elif noobj_p >= 0.5 and noobj_t < 0.5:
            fn += 1
    print("precision: thresh = ", thresh, ",tp_count, fp_count, fn_count = ", tp, fp, fn)
    denom = tp + fp + fn
    prec = tp / denom if denom else 0.0
    return prec, tp, fp, fn


def calc_map(Yp, Yt, device=False):
    print("\ncalc_map: Calculating mean average precision. Yp.shape[0] =", Yp.shape[0])
    pairs = (_pair_ious_device if device else _pair_ious)(Yp, Yt)
    threshes = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
    return sum(precision(Yp, Yt, thresh=t, _pairs=pairs)[0] for t in threshes) / len(threshes)
