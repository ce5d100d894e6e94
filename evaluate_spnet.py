#! /usr/bin/env python3
"""Score a trained model on a labelled directory (mAP, count metrics, overlays) -- entry point and flags
of the reference's evaluate_spnet.py (:38-120), running on the MI355X engine."""
import argparse
import time

import numpy as np

from spnet import diagnostics, models
from spnet.utils import build_dataset, denorm_Y, make_sure_path_exists, show_pred_ellipses
import spnet.config as cf


def evaluate_network(model=None, weights_file="", datapath="Test/", fraction=1.0, log_dir="", batch_size=32,
                     pred_grid=[6, 6, 2], set_means_ranges=True):
    np.random.seed(1)
    print("Getting data..., fraction = ", fraction)
    X_test, Y_test, test_file_list, pred_shape = build_dataset(
        path=datapath, load_frac=fraction, set_means_ranges=set_means_ranges, batch_size=batch_size, shuffle=False,
        pred_grid=pred_grid)
    if model is None:
        print("Loading full model from full_model.h5")
        model = models.load_model("full_model.h5")

    m = X_test.shape[0]
    print("    Predicting... (m = ", m, " frames in this (Test?) dataset)", sep="")
    start_time = time.time()
    Y_pred = model.predict(X_test, batch_size=batch_size)
    elapsed = time.time() - start_time
    print("    ...elapsed time to predict = ", elapsed, "s.   FPS = ", m * 1.0 / elapsed)

    if cf.loss_type != 'same':
        Y_pred[:, cf.ind_noobj::cf.vars_per_pred] = 1.0 / (1.0 + np.exp(-Y_pred[:, cf.ind_noobj::cf.vars_per_pred]))
    Yt, Yp = denorm_Y(Y_test), denorm_Y(Y_pred)
    print("mAP = ", diagnostics.calc_map(Yp, Yt, device=True))      # 72 x N raster IoUs in one HIP launch

    (ring_miscounts, ring_truecounts, total_obj, false_obj_pos, false_obj_neg, true_obj_pos, true_obj_neg, pix_err,
     ipem) = diagnostics.calc_errors(Yp, Yt)
    mistakes = ring_miscounts + false_obj_pos + false_obj_neg
    tot = max(total_obj, 1)
    class_acc = (total_obj - mistakes) * 100.0 / tot
    print('Mean pixel error =', np.mean(pix_err))
    print("    Ring correct counts = ", ring_truecounts, ' / ', total_obj, '.   = ', 100 * ring_truecounts / tot,
          ' % ring-class accuracy', sep="")
    print("         Ring miscounts = ", ring_miscounts, ' / ', total_obj, '.   = ', 100 * ring_miscounts / tot, ' %', sep="")
    print("        False positives = ", false_obj_pos, ' / ', total_obj, '.   = ', 100 * false_obj_pos / tot, ' %', sep="")
    print("        False negatives = ", false_obj_neg, ' / ', total_obj, '.   = ', 100 * false_obj_neg / tot, ' %', sep="")
    print("         True positives = ", true_obj_pos, ' / ', total_obj, '.   = ', 100 * true_obj_pos / tot, ' %', sep="")
    print("         True negatives = ", true_obj_neg, sep="")
    print("    Total Mistakes = ", mistakes, ' / ', total_obj, '.   => ', class_acc,
          ' % class. accuracy rate (lack of mistakes)', sep="")

    make_sure_path_exists(log_dir)
    print("    Drawing sample ellipse images...")
    show_pred_ellipses(Yt, Yp, test_file_list, num_draw=m, log_dir=log_dir, out_csv=log_dir + 'hawley_spnet.csv')
    return model


if __name__ == '__main__':
    p = argparse.ArgumentParser(description="tests network on test dataset",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-w', '--weights', default="weights.hdf5", help='weights file')
    p.add_argument('-d', '--datapath', default="Test/", help='Test dataset directory')
    p.add_argument('-f', '--fraction', type=float, default=1.0, help='Fraction of dataset to use')
    p.add_argument('-l', '--logdir', default='logs/Testing/', help='Directory to write log files into')
    p.add_argument('-b', '--batch_size', type=int, default=16, help='Batch size to use')
    p.add_argument('--model_type', default=None, help="override spnet.config.model_type ('monolithic' | 'big')")
    p.add_argument('--loss_type', default=None, help="override spnet.config.loss_type")
    args = p.parse_args()
    for attr, val in (("model_type", args.model_type), ("loss_type", args.loss_type)):
        if val is not None:
            setattr(cf, attr, val)
    model = evaluate_network(weights_file=args.weights, datapath=args.datapath + '/', fraction=args.fraction,
                             log_dir=args.logdir, batch_size=args.batch_size)
    weights2name = "eval_end_weights.hdf5"
    print("Saving model to", weights2name)
    model.save_weights(weights2name)
