"""Augmentation of ESPI frames on the MI355X -- the surface of the reference's spnet/augmentation.py.

Train-time (AugmentOnTheFly, spnet/callbacks.py:272-341): cutout + salt-and-pepper + the reference's
no-op blur.  The random PARAMETERS are drawn on the host with numpy's global RNG in exactly the
reference's call order (so a seeded run reproduces the reference's augmented frames bit for bit),
the PIXEL work runs in HIP kernels on frames that never leave HBM (csrc/augment.hip).

Offline warps (augment_preproc.py:56-100): flip / rotate / translate = one inverse-affine bilinear
gather kernel plus the reference's host-side metadata arithmetic.
"""
import random

import numpy as np
import torch

from . import _lib as L

MAX_RECTS = 6


_stream = L.current_stream


def cleanup_angle(angle):
    """Wrap into [0,180) (spnet/augmentation.py:74-79)."""
    while angle < 0:
        angle += 180
    while angle >= 180:
        angle -= 180
    return angle


# ----------------------------------------------------------------------------- parameter draws
def draw_cutout(shape, lo, hi, max_regions=6, minsize=11, maxsize=75):
    """RNG call order of cutout_inplace (augmentation.py:117-134).  Returns [(r0,r1,c0,c1,value)]."""
    H, W = shape[0], shape[1]
    n = np.random.randint(0, high=max_regions + 1)
    out = []
    for _ in range(n):
        r0, c0 = np.random.randint(0, H - minsize), np.random.randint(0, W - minsize)
        dr, dc = np.random.randint(minsize, maxsize), np.random.randint(minsize, maxsize)
        r1, c1 = min(r0 + dr, H - 1), min(c0 + dc, W - 1)
        out.append((r0, r1, c0, c1, np.float32(np.random.uniform(lo, hi))))
    return out


def draw_saltpepper(shape, salt_vs_pepper=0.2, amount=0.004):
    """RNG call order of salt_n_pepa_inplace (augmentation.py:157-180).  None when the coin says skip,
    else int arrays (salt_rows, salt_cols, pepper_rows, pepper_cols)."""
    if np.random.choice(['good', 'not good']) != 'good':
        return None
    size = int(np.prod(shape))
    n_salt = int(np.ceil(amount * size * salt_vs_pepper))
    n_pepper = int(np.ceil(amount * size * (1.0 - salt_vs_pepper)))
    sr, sc = [np.random.randint(0, d - 1, n_salt) for d in shape[0:2]]
    pr, pc = [np.random.randint(0, d - 1, n_pepper) for d in shape[0:2]]
    return sr, sc, pr, pc


def saltpepper_counts(shape, salt_vs_pepper=0.2, amount=0.004):
    size = int(np.prod(shape))
    return int(np.ceil(amount * size * salt_vs_pepper)), int(np.ceil(amount * size * (1.0 - salt_vs_pepper)))


def draw_blur_gate(blur_prob_outer=0.4, blur_prob=0.3):
    """AugmentOnTheFly.blur + blur_inplace (callbacks.py:306-309, augmentation.py:66-70).  The
    reference discards cv2.GaussianBlur's result, so only the RNG consumption is reproduced.
    Returns the kernel size it would have used, or 0."""
    if np.random.rand() < blur_prob_outer:
        if np.random.random() <= blur_prob:
            return random.choice([3, 7])
    return 0


class DeviceAugmenter:
    """Keeps the pristine frames [N,H,W,1] in HBM and writes augmented batches into a device buffer."""

    def __init__(self, X_orig, real_blur=False):
        if not X_orig.is_cuda:
            raise RuntimeError("DeviceAugmenter needs device-resident frames (no CPU fallback)")
        self.X = X_orig.contiguous()
        self.N, self.H, self.W = X_orig.shape[0], X_orig.shape[1], X_orig.shape[2]
        self.shape = (self.H, self.W, 1)
        # False (default): the reference's blur, whose cv2.GaussianBlur result is discarded (augmentation.py:66-70:
        # RNG consumed, pixels untouched).  True: apply the blur that call computes (csrc/augment.hip).
        self.real_blur = real_blur
        mm = torch.empty(self.N, 2, device=self.X.device)
        scratch = torch.empty(self.N * 32, device=self.X.device)
        L.spnet_minmax(self.X.data_ptr(), self.N, self.H * self.W, mm.data_ptr(), scratch.data_ptr(), _stream())
        self.mm_host = mm.cpu().numpy()     # min/max of the pristine frames: cutout's fill range
        self.n_salt, self.n_pepper = saltpepper_counts(self.shape)
        self._upload = None

    def draw(self, indices, seeds=None):
        """Host-side parameter draw for the given frame indices, reference RNG order per frame.  seeds (optional,
        one per frame): each frame's draws come from numpy / python streams seeded with its own seed (data parallel:
        a sample's augmentation then depends on its seed only, not on what was drawn before it); the process-wide
        RNG states are saved and restored around the draw, so later consumers of np.random / random are unaffected."""
        B = len(indices)
        npts = self.n_salt + self.n_pepper
        rects = np.zeros((B, MAX_RECTS, 4), np.int32)
        vals = np.zeros((B, MAX_RECTS), np.float32)
        nrect = np.zeros(B, np.int32)
        coords = np.zeros((B, 2, npts), np.int32)
        flag = np.zeros(B, np.int32)
        ksize = np.zeros(B, np.int32)
        saved = (np.random.get_state(), random.getstate()) if seeds is not None else None
        try:
            return self._draw(indices, seeds, rects, vals, nrect, coords, flag, ksize)
        finally:
            if saved is not None:       # the per-sample streams must not leak into the process-wide RNGs
                np.random.set_state(saved[0])
                random.setstate(saved[1])

    def _draw(self, indices, seeds, rects, vals, nrect, coords, flag, ksize):
        for j, i in enumerate(indices):
            if seeds is not None:
                np.random.seed(int(seeds[j]))
                random.seed(int(seeds[j]))
            lo, hi = self.mm_host[i]
            rs = draw_cutout(self.shape, lo, hi)
            nrect[j] = len(rs)
            for k, (r0, r1, c0, c1, v) in enumerate(rs):
                rects[j, k] = (r0, r1, c0, c1)
                vals[j, k] = v
            sp = draw_saltpepper(self.shape)
            if sp is not None:
                flag[j] = 1
                coords[j, 0, :self.n_salt], coords[j, 1, :self.n_salt] = sp[0], sp[1]
                coords[j, 0, self.n_salt:], coords[j, 1, self.n_salt:] = sp[2], sp[3]
            ksize[j] = draw_blur_gate()
        return dict(index=np.asarray(indices, np.int32), rects=rects, vals=vals, nrect=nrect, coords=coords, flag=flag,
                    ksize=ksize)

    def apply(self, params, out):
        """out[j] = augmented copy of frame params['index'][j]; out is a device tensor [B,H,W,1]."""
        dev = self.X.device
        if self._upload is None:
            self._upload = L.AsyncUploader(dev)
        # ONE host -> device copy for the whole parameter set: every async copy from pinned memory is preceded by ~56 us
        # of idle GPU (rocprofv3 kernel trace: the gap in front of each __amd_rocclr_copyBuffer), so seven small uploads
        # cost a 12.5 ms step 0.4 ms.  All fields are 4-byte types: packed as int32 words, viewed back on the device.
        names = ("index", "rects", "vals", "nrect", "coords", "flag", "ksize")
        flat = [np.ascontiguousarray(params[k]).reshape(-1).view(np.int32) for k in names]
        packed = self._upload("params", np.concatenate(flat))      # pinned ring + one async copy
        up, off = {}, 0
        for k, f in zip(names, flat):
            t = packed[off:off + f.size]
            up[k] = t.view(torch.float32) if params[k].dtype == np.float32 else t
            off += f.size
        B = len(params["index"])
        self._keep = up                    # keep the upload alive until the kernels have consumed it
        self.index_dev = up["index"]       # int32 frame indices of this batch on the device (label gathers reuse them)
        L.spnet_cutout(self.X.data_ptr(), up["index"].data_ptr(), out.data_ptr(), B, self.H, self.W,
                       up["rects"].data_ptr(), up["vals"].data_ptr(), up["nrect"].data_ptr(), _stream())
        mm = torch.empty(B * (2 + 32), device=dev)   # [B,2] result followed by B*32 floats of reduction scratch
        self._mm = mm
        L.spnet_minmax(out.data_ptr(), B, self.H * self.W, mm.data_ptr(), mm[2 * B:].data_ptr(), _stream())
        L.spnet_saltpepper(out.data_ptr(), B, self.H, self.W, up["coords"].data_ptr(), self.n_salt, self.n_pepper,
                           up["flag"].data_ptr(), mm.data_ptr(), _stream())
        if self.real_blur and params["ksize"].any():
            tmp = torch.empty_like(out)
            L.spnet_gaussian_blur(out.data_ptr(), tmp.data_ptr(), B, self.H, self.W, up["ksize"].data_ptr(), _stream())
            out.copy_(tmp)
        return out

    def augment(self, indices, out):
        return self.apply(self.draw(indices), out)


# ----------------------------------------------------------------------------- in-place API on device frames
def cutout_inplace(img, max_regions=6, minsize=11, maxsize=75):
    """Reference signature (augmentation.py:117); `img` is ONE device frame [H,W,1]."""
    _require_cuda(img)
    H, W = img.shape[0], img.shape[1]
    mm = torch.empty(2 + 32, device=img.device)
    n = np.random.randint(0, high=max_regions + 1)
    if n == 0:
        return
    L.spnet_minmax(img.data_ptr(), 1, H * W, mm.data_ptr(), mm[2:].data_ptr(), _stream())
    lo, hi = mm[:2].cpu().numpy()
    rects = np.zeros((1, MAX_RECTS, 4), np.int32)
    vals = np.zeros((1, MAX_RECTS), np.float32)
    for k in range(n):
        r0, c0 = np.random.randint(0, H - minsize), np.random.randint(0, W - minsize)
        dr, dc = np.random.randint(minsize, maxsize), np.random.randint(minsize, maxsize)
        rects[0, k] = (r0, min(r0 + dr, H - 1), c0, min(c0 + dc, W - 1))
        vals[0, k] = np.random.uniform(lo, hi)
    r, v = torch.from_numpy(rects).to(img.device), torch.from_numpy(vals).to(img.device)
    nr = torch.tensor([n], dtype=torch.int32, device=img.device)
    L.spnet_cutout(img.data_ptr(), None, img.data_ptr(), 1, H, W, r.data_ptr(), v.data_ptr(), nr.data_ptr(), _stream())
    torch.cuda.current_stream().synchronize()


def salt_n_pepa_inplace(img, salt_vs_pepper=0.2, amount=0.004):
    """Reference signature (augmentation.py:157); `img` is ONE device frame [H,W,1]."""
    _require_cuda(img)
    sp = draw_saltpepper(tuple(img.shape), salt_vs_pepper, amount)
    if sp is None:
        return
    H, W = img.shape[0], img.shape[1]
    ns, npep = len(sp[0]), len(sp[2])
    coords = np.zeros((1, 2, ns + npep), np.int32)
    coords[0, 0, :ns], coords[0, 1, :ns], coords[0, 0, ns:], coords[0, 1, ns:] = sp
    c = torch.from_numpy(coords).to(img.device)
    flag = torch.ones(1, dtype=torch.int32, device=img.device)
    mm = torch.empty(2 + 32, device=img.device)
    L.spnet_minmax(img.data_ptr(), 1, H * W, mm.data_ptr(), mm[2:].data_ptr(), _stream())
    L.spnet_saltpepper(img.data_ptr(), 1, H, W, c.data_ptr(), ns, npep, flag.data_ptr(), mm.data_ptr(), _stream())
    torch.cuda.current_stream().synchronize()


def blur_inplace(img, blur_prob=0.3, kernel_size=None):
    """Bug-compatible with the reference (augmentation.py:66-70): consumes the RNG, leaves pixels alone."""
    if np.random.random() <= blur_prob:
        _ = kernel_size if kernel_size else random.choice([3, 7])


def _require_cuda(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("spnet_amd.augmentation operates on device tensors (no CPU fallback)")


# ----------------------------------------------------------------------------- offline warps
def _warp(img, minv):
    """img: uint8/float numpy [H,W,C]; minv: 2x3 destination->source map.  Bilinear, zero border."""
    if not torch.cuda.is_available():
        raise RuntimeError("spnet_amd.augmentation warps run on the GPU (no CPU fallback)")
    H, W, C = img.shape
    src = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda()
    dst = torch.empty_like(src)
    m = torch.tensor(np.asarray(minv, np.float32).reshape(1, 6)).cuda()
    L.spnet_warp_affine(src.data_ptr(), dst.data_ptr(), 1, H, W, C, m.data_ptr(), _stream())
    out = dst.cpu().numpy()
    if img.dtype == np.uint8:
        out = np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    return out


def cv2_fixed_point_terms(M, H, W):
    """Row / column terms of cv2.warpAffine's fixed-point coordinate computation for a FORWARD 2x3 matrix M (OpenCV 3.4
    imgwarp.cpp, warpAffine + WarpAffineInvoker): M is inverted in double precision exactly as OpenCV does it, then
    adelta[x] = cvRound(M00*x*1024), bdelta[x] = cvRound(M10*x*1024), X0[y] = cvRound((M01*y + M02)*1024) + 16,
    Y0[y] = cvRound((M11*y + M12)*1024) + 16 (cvRound = round-half-even = np.rint).  Returns int32 arrays
    xrow [H,2] = (X0, Y0), xcol [W,2] = (adelta, bdelta)."""
    m = np.asarray(M, np.float64).reshape(2, 3).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = A11, m[0, 1] * -D, m[1, 0] * -D, A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    xcol = np.stack([np.rint(m[0, 0] * xs * 1024.0), np.rint(m[1, 0] * xs * 1024.0)], 1).astype(np.int32)
    xrow = np.stack([np.rint((m[0, 1] * ys + m[0, 2]) * 1024.0) + 16, np.rint((m[1, 1] * ys + m[1, 2]) * 1024.0) + 16],
                    1).astype(np.int32)
    return xrow, xcol


def _warp_cv2(img, M):
    """cv2.warpAffine(img, M, (W,H)) for a uint8 image [H,W,C] and a FORWARD matrix: OpenCV's fixed-point algorithm on
    the device (csrc/augment.hip: warp_affine_fixed_kernel)."""
    if not torch.cuda.is_available():
        raise RuntimeError("spnet_amd.augmentation warps run on the GPU (no CPU fallback)")
    H, W, C = img.shape
    xrow, xcol = cv2_fixed_point_terms(M, H, W)
    src = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda()
    dst = torch.empty_like(src)
    xr, xc = torch.from_numpy(xrow).cuda(), torch.from_numpy(xcol).cuda()
    L.spnet_warp_affine_fixed(src.data_ptr(), dst.data_ptr(), 1, H, W, C, xr.data_ptr(), xc.data_ptr(), _stream())
    return dst.cpu().numpy().astype(np.uint8)


def _invert_affine(M):
    A = np.vstack([np.asarray(M, np.float64), [0, 0, 1]])
    return np.linalg.inv(A)[:2]


def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """Same matrix as cv2.getRotationMatrix2D (positive angle = counter-clockwise on screen)."""
    a = scale * np.cos(np.deg2rad(angle_deg))
    b = scale * np.sin(np.deg2rad(angle_deg))
    cx, cy = center
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], np.float64)


def flip_image(img, metadata, file_prefix, flip_param):
    """flip_param: -2 none, 0 vertical, 1 horizontal, -1 both (augmentation.py:82-112)."""
    if flip_param == -2:
        return img.copy(), list(metadata), file_prefix[:]
    height, width, _ = img.shape
    sx = -1.0 if flip_param in (1, -1) else 1.0
    sy = -1.0 if flip_param in (0, -1) else 1.0
    minv = [[sx, 0, (width - 1) if sx < 0 else 0], [0, sy, (height - 1) if sy < 0 else 0]]
    out = _warp(img, minv)
    new_md = []
    for cx, cy, a, b, angle, rings in metadata:
        if flip_param in (0, -1):
            cy, angle = height - cy, -angle
        angle = cleanup_angle(angle)
        if flip_param in (1, -1):
            cx, angle = width - cx, 180 - angle
        angle = cleanup_angle(angle)
        new_md.append([cx, cy, a, b, angle, rings])
    suffix = {0: "_v", 1: "_h"}.get(flip_param, "_vh")
    return out, new_md, file_prefix + suffix


def rotate_image(img, metadata, file_prefix, rot_angle, rot_origin=None):
    """Rotate about the centre, bilinear, zero fill; centres mapped through the same 2x3 matrix and
    rounded to int (augmentation.py:184-207)."""
    if rot_angle == 0:
        return img.copy(), list(metadata), file_prefix
    height, width, _ = img.shape
    if rot_origin is None:
        rot_origin = (width / 2, height / 2)
    M = rotation_matrix_2d(rot_origin, rot_angle, 1.0)
    # 8-bit images (the offline set, augment_preproc.py) take OpenCV's fixed-point path like the reference's call does
    out = _warp_cv2(img, M) if img.dtype == np.uint8 else _warp(img, _invert_affine(M))
    new_md = []
    for cx, cy, a, b, angle, rings in metadata:
        angle = cleanup_angle(angle + rot_angle)
        p = M @ np.array([cx, cy, 1.0])
        new_md.append([int(round(p[0])), int(round(p[1])), a, b, angle, rings])
    return out, new_md, file_prefix[:] + "_r{:>.2f}".format(rot_angle)


def translate_image(img, metadata, file_prefix, trans_index):
    """Integer shift of up to +-40 px per axis (augmentation.py:216-239)."""
    if trans_index == 0:
        return img.copy(), list(metadata), file_prefix
    trans_max = 40
    xt = int(round(trans_max * (2 * np.random.random() - 1)))
    yt = int(round(trans_max * (2 * np.random.random() - 1)))
    out = _warp_cv2(img, [[1, 0, xt], [0, 1, yt]]) if img.dtype == np.uint8 else _warp(img, [[1, 0, -xt], [0, 1, -yt]])
    new_md = [[cx + xt, cy + yt, a, b, angle, rings] for cx, cy, a, b, angle, rings in metadata]
    return out, new_md, file_prefix[:] + "_t" + str(xt) + ',' + str(yt)


def invert_image(img, metadata, file_prefix):
    return 255 - img, list(metadata), file_prefix + "_i"


def bandpass_mixup(img_in, path_real=None):
    """Out of scope: the reference mixes in private real frames from a
    hard-coded path (augmentation.py:10-62)."""
    raise NotImplementedError("bandpass_mixup needs the reference author's private real images")
