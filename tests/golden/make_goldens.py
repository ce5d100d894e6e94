#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own pure-numpy functions.

Runs ONLY in the build container (needs /root/reference).  The reference's third-party
imports that are absent here (keras, tensorflow, cv2, numba) are satisfied with inert
stub modules so that the pure-numpy functions can be called unmodified; nothing from the
reference is copied into this repo -- only the numeric inputs/outputs are saved, as
tests/golden/reference_numpy.npz.

Usage:  python tests/golden/make_goldens.py
"""
import importlib.abc
import importlib.machinery
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_numpy.npz")

_STUB_ROOTS = ("keras", "tensorflow", "cv2", "numba")


class _Anything:
    """Attribute sink: any attribute / call returns another sink; usable as a base class."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:   # used as a bare decorator
            return a[0]
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def update(self, *a, **k):
        return None


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name in ("Layer", "Callback", "Activation"):
            return type(name, (object,), {"__init__": lambda self, *a, **k: None})
        if name == "jit":
            def jit(*a, **k):
                if len(a) == 1 and callable(a[0]) and not k:
                    return a[0]
                return lambda f: f
            return jit
        return _Anything()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _import_reference():
    sys.meta_path.insert(0, _StubFinder())
    if not hasattr(np, "int"):
        np.int = int            # numpy 2 removed the alias the reference uses
    sys.path.insert(0, REF)
    import spnet.config as cf
    from spnet import models, utils, augmentation, diagnostics, callbacks
    return cf, models, utils, augmentation, diagnostics, callbacks


def main():
    cf, models, utils, aug, diag, cb = _import_reference()
    G = {}

    # ---- loss: my_loss ('same' and 'hybrid') -----------------------------------------
    rs = np.random.RandomState(0)
    yt = rs.rand(4, 576).astype(np.float32)
    yt[:, 6::8] = (yt[:, 6::8] > 0.5)
    yp = rs.rand(4, 576).astype(np.float32)
    G["loss_yt"], G["loss_yp"] = yt, yp
    cf.loss_type = "same"
    tot, parts = models.my_loss(yt, yp)
    G["loss_same_total"], G["loss_same_parts"] = np.float64(tot), np.asarray(parts, np.float64)
    cf.loss_type = "hybrid"
    tot, parts = models.my_loss(yt, yp)
    G["loss_hybrid_total"], G["loss_hybrid_parts"] = np.float64(tot), np.asarray(parts, np.float64)
    cf.loss_type = "same"
    # a second, larger, signed case (predictions are unbounded in practice)
    rs = np.random.RandomState(7)
    yt2 = (rs.randn(16, 576) * 0.5).astype(np.float32)
    yt2[:, 6::8] = (rs.rand(16, 72) > 0.7)
    yp2 = (rs.randn(16, 576) * 0.7).astype(np.float32)
    G["loss2_yt"], G["loss2_yp"] = yt2, yp2
    tot, parts = models.my_loss(yt2, yp2)
    G["loss2_same_total"], G["loss2_same_parts"] = np.float64(tot), np.asarray(parts, np.float64)
    cf.loss_type = "hybrid"
    tot, parts = models.my_loss(yt2, yp2)
    G["loss2_hybrid_total"], G["loss2_hybrid_parts"] = np.float64(tot), np.asarray(parts, np.float64)
    cf.loss_type = "same"

    # ---- codec: means/ranges, grid encode, denorm, cleanup -----------------------------
    ret = utils.setup_means_and_ranges([6, 6, 2, 8])
    G["mr_scalars"] = np.asarray(ret[:6], np.float64)
    G["mr_gridYi"] = np.asarray(ret[6], np.float32)
    G["means"], G["ranges"] = np.asarray(utils.means), np.asarray(utils.ranges)

    rs = np.random.RandomState(3)
    # up to 5 ellipses, spread so no cell takes more than 2
    ell = []
    for cx, cy in [(100, 140), (101, 141), (300, 60), (469, 349), (20, 20), (250, 200)]:
        a, b = rs.randint(15, 140), rs.randint(15, 100)
        a, b = max(a, b), min(a, b)
        ang = rs.randint(1, 179)
        ell.append([cx, cy, a, b, np.cos(2 * np.deg2rad(ang)), np.sin(2 * np.deg2rad(ang)), 0, rs.randint(1, 11)])
    ell = np.array(ell, dtype=np.float64)
    G["grid_in"] = ell
    grid = utils.true_to_pred_grid(ell, np.array([6, 6, 2, 8]))
    G["grid_out"] = np.asarray(grid, np.float32)
    Yn = utils.norm_Y(grid.flatten()[None, :])
    G["grid_norm"] = np.asarray(Yn)
    G["grid_denorm"] = np.asarray(utils.denorm_Y(Yn))

    sub = utils.denorm_Y(np.tile(np.array([.25, -.1, .5, .2, .6, -.8, .2, .13], np.float32), 72)[None, :])[0, :8]
    G["denorm8"] = np.asarray(sub, np.float64)
    G["cleanup8"] = np.asarray(utils.cleanup_antinode_vars(sub), np.float64)
    # a few more cleanup cases, including the angle<=0 branch
    cases = np.array([[93.4, 60.5, 71.2, 35.7, 1.2, -1.6, 0.2, 6.3],
                      [10.5, 11.5, 3.49, 2.5, -1.0, 0.0, 0.6, 0.0],
                      [400.2, 300.7, 50.0, 20.0, 0.5, 0.5, 0.49, 3.2],
                      [1.0, 2.0, 3.0, 4.0, 1.0, 0.0, 0.0, 1.0],
                      [1.0, 2.0, 3.0, 4.0, -0.3, -0.9, 1.0, 11.0]], np.float64)
    G["cleanup_in"] = cases
    G["cleanup_out"] = np.array([utils.cleanup_antinode_vars(c) for c in cases], np.float64)

    # parse_meta_file on a small CSV (duplicates, b>a swap, rings<=0 dropped, sort)
    import tempfile
    csv = "300,200,40,80,30,5\n100,50,60,20,170,3\n100,50,60,20,170,3\n100,40,30,30,10,0\n100,45,25,26,45,2.0\n"
    G["meta_csv"] = np.array(csv)
    with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
        f.write(csv)
    G["meta_out"] = np.asarray(utils.parse_meta_file(f.name), np.float64)
    os.unlink(f.name)

    G["nearest_multiple_720_31"] = np.int64(utils.nearest_multiple(720, 31))

    # ---- 1-cycle schedule ---------------------------------------------------------------
    lrs = cb.get_1cycle_schedule(lr_max=4e-5, n_data_points=40000, epochs=100, batch_size=16)
    G["lrs_len"] = np.int64(len(lrs))
    idx = np.array([0, 1, 2499, 12499, 74998, 74999, 75000, 75001, 150000, 249998, 249999])
    G["lrs_idx"], G["lrs_val"] = idx, np.asarray(lrs[idx], np.float64)
    lrs2 = cb.get_1cycle_schedule(lr_max=1e-3, n_data_points=1000, epochs=3, batch_size=8)
    G["lrs2_full"] = np.asarray(lrs2, np.float64)

    # ---- diagnostics.calc_errors ----------------------------------------------------------
    rs = np.random.RandomState(11)
    Yt = rs.rand(6, 576) * 10
    Yt[:, 6::8] = (rs.rand(6, 72) > 0.6)
    Yp = Yt + rs.randn(6, 576) * 0.4
    Yp[:, 6::8] = np.clip(Yt[:, 6::8] + rs.randn(6, 72) * 0.35, -0.2, 1.2)
    out = diag.calc_errors(Yp, Yt)
    G["ce_Yt"], G["ce_Yp"] = Yt, Yp
    G["ce_counts"] = np.asarray(out[:7], np.int64)
    G["ce_pix_err"], G["ce_ipem"] = np.asarray(out[7]), np.int64(out[8])

    # ---- cleanup_angle -------------------------------------------------------------------
    angs = np.array([-190.0, -1.0, 0.0, 45.5, 179.9, 180.0, 361.0, 540.0])
    G["angle_in"], G["angle_out"] = angs, np.array([aug.cleanup_angle(a) for a in angs])

    # ---- augmentation: cutout + salt&pepper with the reference's RNG call order -------------
    # (the AugmentOnTheFly loop body: cutout -> salt_n_pepa -> blur gate, callbacks.py:326-332)
    # Inputs are re-derivable from the seed (RandomState(5).rand(6,H,W,1)*2-1, float32), so only the
    # changed pixels of each output are stored: flat index + new value.
    for tag, (H, W) in {"a": (331, 331), "b": (96, 128)}.items():
        rs = np.random.RandomState(5)
        X = (rs.rand(6, H, W, 1).astype(np.float32) * 2 - 1)
        G[f"aug_{tag}_shape"] = np.array(X.shape, np.int64)
        np.random.seed(1234)
        random.seed(1234)
        outs = []
        for i in range(X.shape[0]):
            img = X[i].copy()
            aug.cutout_inplace(img)
            aug.salt_n_pepa_inplace(img)
            if np.random.rand() < 0.4:       # AugmentOnTheFly.blur gate (callbacks.py:306-309)
                aug.blur_inplace(img)        # no-op on the pixels (result discarded), consumes RNG
            outs.append(img)
        out = np.stack(outs)
        changed = np.flatnonzero(out.ravel() != X.ravel())
        G[f"aug_{tag}_changed_idx"] = changed.astype(np.int64)
        G[f"aug_{tag}_changed_val"] = out.ravel()[changed]
        G[f"aug_{tag}_rng_after"] = np.float64(np.random.rand())   # pins total RNG consumption

    # ---- zooniverse-style CSV of predictions (utils.show_pred_ellipses, utils.py:67-137): the image / drawing calls go
    #      to the inert keras / cv2 stubs, the CSV text is the reference's own string formatting -----------------
    import tempfile
    rs = np.random.RandomState(21)
    n_img = 5
    Yt_csv = np.zeros((n_img, 576), np.float32)
    Yp_csv = np.zeros((n_img, 576), np.float32)
    for arr in (Yt_csv, Yp_csv):
        arr[:, 0::8] = rs.uniform(20, 490, (n_img, 72))
        arr[:, 1::8] = rs.uniform(20, 360, (n_img, 72))
        arr[:, 2::8] = rs.uniform(-5, 140, (n_img, 72))
        arr[:, 3::8] = rs.uniform(-5, 100, (n_img, 72))
        arr[:, 4::8] = rs.uniform(-1, 1, (n_img, 72))
        arr[:, 5::8] = rs.uniform(-1, 1, (n_img, 72))
        arr[:, 6::8] = rs.uniform(-0.2, 1.2, (n_img, 72))
        arr[:, 7::8] = rs.uniform(-1, 11, (n_img, 72))
    Yp_csv[3, 6::8] = 0.9                      # an image without any predicted object -> the all-zeros row
    names = ["steelpan_%07d.png" % (100 + i) for i in range(n_img)]
    with tempfile.TemporaryDirectory() as td:
        out_csv = os.path.join(td, "hawley_spnet.csv")
        utils.show_pred_ellipses(Yt_csv, Yp_csv, [os.path.join(td, n) for n in names], num_draw=n_img, log_dir=td,
                                 out_csv=out_csv, show_true=False)
        G["csv_text"] = np.array(open(out_csv).read())
    G["csv_Yt"], G["csv_Yp"], G["csv_names"] = Yt_csv, Yp_csv, np.array(names)

    # ---- fake-ESPI generator (gen_fake_espi.py:60-206): the reference's own draw_waves / draw_antinodes run from seeded
    #      generators; their OpenCV calls go to recorders, every argument they pass is stored (the parameter draws and the
    #      geometry handed to cv2.polylines / cv2.ellipse are pure Python; the pixels are OpenCV's and are not stored) ----
    import gen_fake_espi as gfe
    for s in (0, 1, 2, 3, 11):
        rec_lines, rec_ell = [], []
        gfe.cv2.polylines = lambda img, pts, closed, color, thickness=1: rec_lines.append((np.array(pts[0]), closed, color, thickness))
        gfe.draw_ellipse = lambda img, center, axes, angle, color=0, thickness=2, **k: rec_ell.append(
            (center[0], center[1], axes[0], axes[1], angle, color, thickness))
        random.seed(s)
        np.random.seed(s)
        img = 128 * np.ones((gfe.imHeight, gfe.imWidth, 1), np.uint8)
        gfe.draw_waves(img)
        n_an = random.randint(1, 7)                       # gen_images (:250-251)
        _, cap = gfe.draw_antinodes(img, num_antinodes=n_an)
        G["espi%d_n_lines" % s] = np.int64(len(rec_lines))
        G["espi%d_line_thickness" % s] = np.int64(rec_lines[0][3])
        G["espi%d_line_first" % s] = rec_lines[0][0].astype(np.int32)
        G["espi%d_line_last" % s] = rec_lines[-1][0].astype(np.int32)
        G["espi%d_line_y0" % s] = np.array([l[0][0, 1] for l in rec_lines], np.int32)
        G["espi%d_ellipses" % s] = np.array(rec_ell, np.float64).reshape(-1, 7)
        G["espi%d_caption" % s] = np.array(cap)
        G["espi%d_rng_after" % s] = np.array([random.random(), np.random.rand()])     # pins total RNG consumption

    # ---- flip metadata transform (pure python part of flip_image is not separable from cv2.flip,
    #      so only cleanup_angle is pinned above) -------------------------------------------

    np.savez_compressed(OUT, **G)
    print("wrote", OUT, "with", len(G), "arrays")
    for k in ("loss_same_total", "loss_hybrid_total", "lrs_val", "ce_counts"):
        print(k, G[k])


if __name__ == "__main__":
    main()
