#!/usr/bin/env python3
"""SURVEY section 8(c) statistical acceptance: train the product the way the reference's published run was trained
and set the outcome beside that run's log (paper/run_logs/log_DatasetA_FakeLarge_MSEloss_100ep_...txt).

Layout of the reference run (its log :62-66,:176,:208): model_type 'monolithic' = every 512x384 frame resized to
331x331 (PIL Lanczos, spnet/utils.py:330-342), Xception, 40,000 train / 4,992 val / 4,992 test fake-ESPI frames,
batch 16, lr_max 4e-5 1-cycle, augmentation on the fly, freeze_fac 0 (`Freezing 0 / 144 layers`), loss_type 'same'.
Here: the same flow through the product's own API (models.setup_model -> Model.fit with MyProgressCallback,
OneCycleScheduler, AugmentOnTheFly; evaluate_spnet.py's metrics at the end).  Frames come from the device
fake-ESPI generator (csrc/espi.hip; labels identical to the host generator's), are resized on the host with the
codec's own PIL call and never touch the disk.  gpurun allows 20 minutes per call, so the number of epochs is an
argument (the 1-cycle schedule is built for that many epochs, as train_spnet.py -e would).

Round 4 (like-for-like): --density 0-6 draws the number of antinodes per frame the way the generator did when the
published dataset was made (gen_fake_espi.py:250-251 dates the change to 1-7 to Nov 2020; 14,965 objects in 4,992
frames = 3.0 per frame), --schedule-epochs builds the 1-cycle table for more epochs than are run (the reference's own
250,000-entry table, stopped after --epochs), --adam-eps a,b runs one variant per value on the SAME frames with the
same initial weights, and --test 0 skips the held-out metrics (early-dynamics runs).

Writes <out>/acceptance[_eps<e>].json (per-epoch rows + final metrics + the reference's figures + band verdicts) and
<out>/acceptance[_eps<e>]_table.txt."""
import argparse
import json
import os
import sys
import time
from multiprocessing.pool import ThreadPool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the reference's published run (file:line in /root/reference/paper/run_logs/log_DatasetA_FakeLarge_MSEloss_100ep_...txt)
REF = {
    "epoch_rows": {   # epoch index: train_total val_total center size angle noobj rings   (:195, :217, ... one block per epoch)
        0: (2.582e-01, 2.323e-01, 2.971e-03, 2.122e-03, 1.085e-03, 3.410e-03, 2.331e-03),
        1: (2.160e-01, 2.016e-01, 2.223e-03, 1.741e-03, 8.848e-04, 2.268e-03, 1.640e-03),      # :217
        2: (1.877e-01, 1.726e-01, 1.617e-03, 1.451e-03, 7.196e-04, 1.487e-03, 9.644e-04),      # :239
        3: (1.546e-01, 1.354e-01, 1.256e-03, 1.297e-03, 6.500e-04, 1.164e-03, 6.663e-04),      # :261
        4: (1.152e-01, 9.569e-02, 1.036e-03, 1.170e-03, 5.647e-04, 9.983e-04, 5.510e-04),      # :283
        5: (7.819e-02, 6.253e-02, 8.259e-04, 9.932e-04, 4.818e-04, 8.974e-04, 4.584e-04),
        10: (6.465e-03, 4.995e-03, 2.235e-04, 2.207e-04, 1.649e-04, 3.991e-04, 1.863e-04),
        20: (9.174e-04, 8.666e-04, 8.632e-05, 7.927e-05, 8.209e-05, 1.337e-04, 6.309e-05),
        30: (5.992e-04, 6.863e-04, 8.115e-05, 7.517e-05, 6.689e-05, 1.140e-04, 8.983e-05),
        40: (4.548e-04, 5.224e-04, 6.049e-05, 5.126e-05, 6.708e-05, 1.011e-04, 4.182e-05),
        50: (3.604e-04, 4.389e-04, 5.883e-05, 4.077e-05, 5.490e-05, 9.374e-05, 3.178e-05),
        60: (2.759e-04, 3.666e-04, 4.788e-05, 3.767e-05, 3.840e-05, 8.948e-05, 2.680e-05),
        70: (2.056e-04, 2.993e-04, 3.687e-05, 3.417e-05, 3.187e-05, 7.861e-05, 1.884e-05),
        80: (1.527e-04, 2.442e-04, 2.679e-05, 2.717e-05, 2.757e-05, 7.081e-05, 1.253e-05),
        90: (1.239e-04, 2.133e-04, 2.188e-05, 2.365e-05, 2.412e-05, 6.339e-05, 1.084e-05),
        99: (1.154e-04, 2.054e-04, 2.071e-05, 2.262e-05, 2.317e-05, 6.197e-05, 9.557e-06),      # :2447-2448
    },
    "final": {"loss": 1.1543e-04, "val_loss": 2.0538e-04,                    # :2463
              "mAP": 0.9687651654815838, "mean_pixel_error": 5.0598817,      # :2487-2509, test set of 4,992 frames
              "ring_correct": 14432, "total_obj": 14965, "tp_rate": 97.35382559305044,
              "false_pos": 413, "false_neg": 396, "ring_miscounts": 137},
    "epochs": 100,
}

# Acceptance bands, fixed BEFORE the run.  Not equality: the frames come from another rasteriser
# (analytic device kernel vs OpenCV drawing calls, no bandpass_mixup), dropout uses another RNG, and the run may be shorter.
BANDS = {
    "l2_after_epoch1": (0.75, 1.25),        # (val_total - sum of the five val terms) / the reference's 0.2204 at epoch index 0
    "final_loss_max_ratio": 3.0,            # train total  <= 3 x 1.154e-4  (for runs of >= 50 epochs)
    "final_val_loss_max_ratio": 3.0,        # val total    <= 3 x 2.054e-4
    "per_term_max_ratio": 3.0,              # each of the five val terms <= 3 x the reference's
    "mAP_min": 0.93, "mean_pixel_error_max": 7.5, "ring_accuracy_min": 93.0, "tp_rate_min": 95.0,
}


def make_set(n, seed, dev, size, threads, count_range=(1, 7)):
    """n fake-ESPI frames: device rasteriser -> uint8 on the host -> the input codec's resize (PIL Lanczos to
    size x size, utils._load_one) -> float32 [n,size,size,1] in [-1,1]; labels -> normalised grid targets."""
    import torch
    from PIL import Image
    import bench
    from spnet_amd import fake_espi as F
    X = np.empty((n, size, size, 1), np.float32)
    labels = []
    pool = ThreadPool(threads)

    def resize(a):
        return np.asarray(Image.fromarray(a).resize((size, size), Image.LANCZOS), dtype=np.float32)

    block = 4096
    for k, lo in enumerate(range(0, n, block)):
        m = min(block, n - lo)
        _, lab, U = F.generate_device(m, seed=seed + k, device=str(dev), want_u8=True, count_range=count_range)
        u = U.cpu().numpy()
        del U
        torch.cuda.empty_cache()
        for i, arr in enumerate(pool.imap(resize, list(u), chunksize=16)):
            X[lo + i, :, :, 0] = arr
        labels += lab
    pool.close()
    X /= 255.0
    X -= 0.5
    X *= 2.0
    return X, bench.labels_to_Y(labels)


def run_variant(args, tag, adam_eps, data, t_start, t_data, count_range):
    """One training run on the prepared frames: fresh model (same seed -> same initial weights for every variant),
    `epochs` epochs of a 1-cycle table built for `schedule_epochs`, tables + json under args.out with suffix `tag`."""
    import torch
    from spnet import callbacks, diagnostics, models, utils
    X_train, Y_train, X_val, Y_val, X_test, Y_test = data
    np.random.seed(args.seed)
    models.ADAM_EPS = float(adam_eps)
    model, _ = models.setup_model(X_train, Y_train[0].size, no_cp_fatal=False, weights_file="no_such_weights.hdf5",
                                  parallel=False, freeze_fac=0.0)
    log_dir = os.path.join(args.out, "log" + tag)
    if os.path.exists(log_dir + "/losses.dat"):
        os.remove(log_dir + "/losses.dat")
    sched_epochs = args.schedule_epochs or args.epochs
    cbs = [callbacks.MyProgressCallback(X_val=X_val, Y_val=Y_val, val_file_list=None, log_dir=log_dir,
                                        pred_shape=[6, 6, 2, 8], num_draw=0, make_plots=False),
           callbacks.OneCycleScheduler(lr_max=args.lrmax, n_data_points=X_train.shape[0], epochs=sched_epochs,
                                       batch_size=args.batch, verbose=1),
           callbacks.AugmentOnTheFly(X_train, Y_train, aug_every=1, seed=args.seed)]

    class Clock(callbacks.Callback):
        def __init__(self):
            super().__init__()
            self.t, self.rows, self.l2 = [], [], []

        def on_epoch_begin(self, epoch, logs=None):
            self.t0 = time.time()

        def on_epoch_end(self, epoch, logs=None):
            self.t.append(time.time() - self.t0)
            # 1e-4 * sum(w^2) of each regularised kernel (which of them carry the decay of the l2 penalty)
            r = self.model._root
            self.l2.append({n: 1e-4 * float(r.theta[off:off + cnt].double().square().sum().item())
                            for n, (off, cnt, _) in r.p_off.items() if off < r.l2_n})
            if args.time_budget > 0 and (time.time() - t_start) + 1.3 * max(self.t) + 45 > args.time_budget:
                print("time budget: stopping after epoch", epoch, flush=True)
                self.model.stop_training = True
            with open(os.path.join(args.out, "progress%s.txt" % tag), "a") as f:     # survives a killed call
                f.write("epoch %d  %.1f s  loss %.4e  val_loss %.4e  lr %.6e\n" % (
                    epoch, self.t[-1], logs.get("loss"), logs.get("val_loss"), logs.get("lr", float("nan"))))

    clock = Clock()
    t_fit = time.time()
    # (Clock goes last: it reads logs['lr'], which OneCycleScheduler.on_epoch_end sets)
    hist = model.fit(X_train, Y_train, batch_size=args.batch, epochs=args.epochs, shuffle=True, verbose=1,
                     validation_data=(X_val, Y_val), callbacks=cbs + [clock])
    t_fit = time.time() - t_fit

    rows = []
    for line in open(log_dir + "/losses.dat"):
        if not line.startswith("#"):
            v = line.split()
            rows.append([int(v[0])] + [float(x) for x in v[1:]])
    # rows: epoch train_total my_val_loss center size angle noobj rings ; Keras' val_loss (with l2) from history
    for r, vl in zip(rows, hist["val_loss"]):
        r.insert(2, float(vl))
    lr_end = [float(cbs[1].lrs[min((e + 1) * (X_train.shape[0] // args.batch), len(cbs[1].lrs)) - 1]) for e in range(len(rows))]

    final = {"loss": rows[-1][1], "val_loss": rows[-1][2]}
    if X_test is not None and X_test.shape[0]:
        # ---- evaluate_spnet.py's metrics on the held-out test set
        t0 = time.time()
        Y_pred = model.predict(X_test, batch_size=args.batch)
        fps = X_test.shape[0] / (time.time() - t0)
        Yt, Yp = utils.denorm_Y(Y_test), utils.denorm_Y(Y_pred)
        mAP = float(diagnostics.calc_map(Yp, Yt, device=True))
        (ring_miscounts, ring_truecounts, total_obj, false_pos, false_neg, true_pos, true_neg, pix_err,
         ipem) = diagnostics.calc_errors(Yp, Yt)
        tot = max(int(total_obj), 1)
        final.update({"mAP": mAP, "mean_pixel_error": float(np.mean(pix_err)),
                      "max_pixel_error": float(pix_err[ipem]), "ring_correct": int(ring_truecounts),
                      "total_obj": int(total_obj), "ring_accuracy": 100.0 * int(ring_truecounts) / tot,
                      "ring_miscounts": int(ring_miscounts), "false_pos": int(false_pos), "false_neg": int(false_neg),
                      "tp_rate": 100.0 * int(true_pos) / tot, "test_predict_fps": fps})

    # ---- bands
    E = sched_epochs
    full = len(rows) >= E                       # the schedule was run to its end: final-quality bands apply
    ref0 = REF["epoch_rows"][0]
    ref_l2_0 = ref0[1] - sum(ref0[2:])
    l2_0 = rows[0][2] - sum(rows[0][4:9])
    reff = REF["epoch_rows"][99]
    lm = args.loss_band
    checks = {"l2_after_epoch1": {"ours": l2_0, "reference": ref_l2_0, "ratio": l2_0 / ref_l2_0,
                                  "ok": BANDS["l2_after_epoch1"][0] <= l2_0 / ref_l2_0 <= BANDS["l2_after_epoch1"][1]}}
    if full:
        checks["final_loss"] = {"ours": final["loss"], "reference": reff[0], "ratio": final["loss"] / reff[0],
                                "ok": final["loss"] <= lm * reff[0]}
        checks["final_val_loss"] = {"ours": final["val_loss"], "reference": reff[1], "ratio": final["val_loss"] / reff[1],
                                    "ok": final["val_loss"] <= lm * reff[1]}
        for j, name in enumerate(("center", "size", "angle", "noobj", "rings")):
            ours, ref = rows[-1][4 + j], reff[2 + j]
            checks["val_" + name] = {"ours": ours, "reference": ref, "ratio": ours / ref, "ok": ours <= lm * ref}
    if "mAP" in final:
        checks.update({
            "mAP": {"ours": final["mAP"], "reference": REF["final"]["mAP"], "ok": final["mAP"] >= BANDS["mAP_min"]},
            "mean_pixel_error": {"ours": final["mean_pixel_error"], "reference": REF["final"]["mean_pixel_error"],
                                 "ok": final["mean_pixel_error"] <= BANDS["mean_pixel_error_max"]},
            "ring_accuracy": {"ours": final["ring_accuracy"],
                              "reference": 100.0 * REF["final"]["ring_correct"] / REF["final"]["total_obj"],
                              "ok": final["ring_accuracy"] >= BANDS["ring_accuracy_min"]},
            "tp_rate": {"ours": final["tp_rate"], "reference": REF["final"]["tp_rate"],
                        "ok": final["tp_rate"] >= BANDS["tp_rate_min"]}})
    result = {"args": vars(args), "adam_eps": float(adam_eps), "antinodes_per_frame": list(count_range),
              "schedule_epochs": E, "epochs_run": len(rows), "stopped_early": len(rows) < args.epochs,
              "reference_epochs": REF["epochs"],
              "seconds": {"data": t_data, "fit": t_fit, "per_epoch_median": float(np.median(clock.t)), "total": time.time() - t_start},
              "train_images_per_sec_incl_validation_and_augmentation": args.train * len(rows) / t_fit,
              "columns": ["epoch", "train_total", "val_total", "my_val_loss", "center", "size", "angle", "noobj", "rings"],
              "rows": rows, "lr_at_epoch_end": lr_end, "l2_by_kernel_at_epoch_end": clock.l2, "final": final, "bands": dict(BANDS, loss_max_ratio=lm),
              "checks": checks, "all_ok": all(c["ok"] for c in checks.values()),
              "reference": {"final": REF["final"], "epoch_rows": {str(k): v for k, v in REF["epoch_rows"].items()}}}
    with open(os.path.join(args.out, "acceptance%s.json" % tag), "w") as f:
        json.dump(result, f, indent=1)

    # ---- table: our epoch e of E beside the reference's epoch at the same fraction of its 1-cycle schedule
    lines = ["acceptance run%s: %d epochs (1-cycle table for %d) x %d frames at %dx%d, batch %d, lr_max %g, Adam eps %g, "
             "%d-%d antinodes per frame (reference: 100 epochs)" %
             (tag, len(rows), E, args.train, args.size, args.size, args.batch, args.lrmax, adam_eps, count_range[0],
              count_range[1]),
             "fit %.0f s (%.1f s/epoch median; %.0f images/s including validation, progress callback and augmentation)" %
             (t_fit, float(np.median(clock.t)), result["train_images_per_sec_incl_validation_and_augmentation"]), "",
             "%5s %5s | %-10s %-10s | %-10s %-10s | %-9s %-9s | five val terms (ours / reference): center size angle noobj rings" %
             ("ep", "refep", "train", "ref", "val", "ref", "l2", "ref")]
    for ref_ep in sorted(REF["epoch_rows"]):
        e = int(round(ref_ep * (E - 1) / 99.0))
        if e >= len(rows):
            continue
        r, q = rows[e], REF["epoch_rows"][ref_ep]
        lines.append("%5d %5d | %.3e  %.3e | %.3e  %.3e | %.3e %.3e | %s" % (
            e, ref_ep, r[1], q[0], r[2], q[1], r[2] - sum(r[4:9]), q[1] - sum(q[2:]),
            "  ".join("%.2e/%.2e" % (r[4 + j], q[2 + j]) for j in range(5))))
    names = list(clock.l2[0]) if clock.l2 else []
    lines += ["", "l2 penalty by kernel at epoch end (1e-4 * sum w^2): " + "  ".join(n.split("/")[0] for n in names)]
    for e in sorted(set([0, 1, 2, 3, 4] + list(range(9, len(clock.l2), 10)) + [len(clock.l2) - 1])):
        if 0 <= e < len(clock.l2):
            lines.append("  %3d  %s   total %.4f" % (e, "  ".join("%.4f" % clock.l2[e][n] for n in names), sum(clock.l2[e].values())))
    lines += ["", "checks (ours / reference / band):"]
    for k, c in checks.items():
        lines.append("  %-18s %.5g / %.5g   %s" % (k, c["ours"], c["reference"], "ok" if c["ok"] else "OUT OF BAND"))
    lines.append("all_ok = %s" % result["all_ok"])
    open(os.path.join(args.out, "acceptance%s_table.txt" % tag), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines), flush=True)
    # free this variant's device memory (frames of AugmentOnTheFly, plans, optimizer state) before the next one
    del model, cbs, hist
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=40000)
    ap.add_argument("--val", type=int, default=4992)
    ap.add_argument("--test", type=int, default=4992, help="0: skip the held-out metrics (early-dynamics runs)")
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--schedule-epochs", type=int, default=0,
                    help="build the 1-cycle table for this many epochs and stop after --epochs (default: = --epochs); 100 "
                         "is the reference run's own 250,000-entry table")
    ap.add_argument("--density", default="1-7", help="antinodes per frame, lo-hi inclusive: 1-7 = the reference's current "
                                                     "generator, 0-6 = the one the published dataset was made with")
    ap.add_argument("--adam-eps", default="1e-7", help="comma-separated: one run per value on the same frames")
    ap.add_argument("--loss-band", type=float, default=3.0, help="final losses must be <= this x the reference's")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lrmax", type=float, default=4e-5)
    ap.add_argument("--size", type=int, default=331)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "acceptance"))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--time-budget", type=float, default=0.0,
                    help="seconds from process start; training stops early (schedule cut short, recorded) when the "
                         "next epoch would not fit -- a guard against the 20-minute limit of a gpurun call")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    t_start = time.time()

    import torch
    import spnet.config as cf
    from bench import host_cpu_share
    cf.model_type, cf.loss_type, cf.basemodel = "monolithic", "same", "Xception"
    threads, _ = host_cpu_share()
    torch.set_num_threads(threads)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lo, hi = (int(v) for v in args.density.split("-"))
    count_range = (lo, hi)

    # (generator seeds: frame seed = seed * 1000003 + i must stay below 2**32; one seed per block of 4,096 frames)
    X_train, Y_train = make_set(args.train, 100 + 1000 * args.seed, dev, args.size, threads, count_range)
    X_val, Y_val = make_set(args.val, 300 + 1000 * args.seed, dev, args.size, threads, count_range)
    X_test = Y_test = None
    if args.test:
        X_test, Y_test = make_set(args.test, 400 + 1000 * args.seed, dev, args.size, threads, count_range)
    t_data = time.time() - t_start
    print("data: %d/%d/%d frames at %dx%d in %.1f s" % (args.train, args.val, args.test, args.size, args.size, t_data),
          flush=True)
    data = (X_train, Y_train, X_val, Y_val, X_test, Y_test)
    eps_list = [float(v) for v in args.adam_eps.split(",")]
    for eps in eps_list:
        tag = "" if len(eps_list) == 1 else "_eps%g" % eps
        run_variant(args, tag, eps, data, t_start, t_data, count_range)


if __name__ == "__main__":
    main()
