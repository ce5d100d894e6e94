#! /usr/bin/env python3
"""Predict ellipses + ring counts for a directory of frames (no scoring) -- entry point and flags of the
reference's predict_spnet.py (:40-115), running on the MI355X engine."""
import argparse
import glob
import time

import numpy as np

from spnet.models import load_model, setup_model
from spnet.utils import (build_X, denorm_Y, make_sure_path_exists, nearest_multiple, setup_means_and_ranges,
                         show_pred_ellipses)
import spnet.config as cf

default_image_dir = '/home/shawley/datasets/zooniverse_steelpan/'


def predict_network(weights_file="spnet.model", datapath=default_image_dir, fraction=1.0, log_dir='logs/Predicting/',
                    batch_size=16, model=None, X_pred='', u8_frames=False):
    img_file_list = []
    if isinstance(X_pred, str) and X_pred == '':
        print(f"Getting data from {datapath}, fraction = {fraction}.")
        if cf.model_type == 'simple':
            grayscale, force_dim = False, 224
        elif cf.model_type == 'big':
            grayscale, force_dim = True, None      # native 384x512 frames
        else:
            grayscale, force_dim = True, 331
        img_file_list = sorted(glob.glob(datapath + '/*.png')) or sorted(glob.glob(datapath + '/*.bmp'))
        total_files = len(img_file_list)
        total_load = int(total_files * fraction)
        if batch_size is not None:
            total_load = nearest_multiple(total_load, batch_size)
        print("      Total files = ", total_files, ", going to load total_load = ", total_load)
        X_pred, _ = build_X(total_load, img_file_list, force_dim=force_dim, grayscale=grayscale,
                            as_uint8=bool(u8_frames) and grayscale)
        print("")

    if model is None:
        print("Loading model from", weights_file)
        if '.hdf5' in weights_file:
            print("   Defining model, then loading weights")
            model, _ = setup_model(X_pred, try_checkpoint=True, no_cp_fatal=True, weights_file=weights_file,
                                   parallel=False, freeze_fac=0.0, quick_setup=True)
        else:
            print("   Loading whole model")
            model = load_model(weights_file)

    m = X_pred.shape[0]
    print("    Predicting... (m = ", m, " frames in dataset)", sep="")
    start_time = time.time()
    Y_pred = model.predict(X_pred, batch_size=batch_size)
    elapsed = time.time() - start_time
    print("    ...elapsed time to predict = ", elapsed, "s.   FPS = ", m * 1.0 / elapsed)

    print("    Drawing ellipse images...")
    make_sure_path_exists(log_dir)
    setup_means_and_ranges([6, 6, 2, cf.vars_per_pred])
    if cf.loss_type != 'same':
        Y_pred[:, cf.ind_noobj::cf.vars_per_pred] = 1.0 / (1.0 + np.exp(-Y_pred[:, cf.ind_noobj::cf.vars_per_pred]))
    Yp = denorm_Y(Y_pred)
    if img_file_list:
        show_pred_ellipses(Yp, Yp, img_file_list, num_draw=m, log_dir=log_dir, out_csv=log_dir + 'hawley_spnet.csv',
                           show_true=False)
    return model


if __name__ == '__main__':
    np.random.seed(1)
    p = argparse.ArgumentParser(description="predicts ellipses for a directory of images",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-w', '--weights', default="spnet.model", help='weights file (.hdf5) or whole-model file')
    p.add_argument('-d', '--datapath', default=default_image_dir, help='Dataset directory with list of images')
    p.add_argument('-f', '--fraction', type=float, default=1.0, help='Fraction of dataset to use')
    p.add_argument('-l', '--logdir', default='logs/Predicting/', help='Directory of log/output files')
    p.add_argument('-b', '--batch_size', type=int, default=16, help='Batch size to use')
    p.add_argument('--model_type', default=None, help="override spnet.config.model_type ('monolithic' | 'big')")
    p.add_argument('--loss_type', default=None, help="override spnet.config.loss_type")
    p.add_argument('--u8_frames', action='store_true',
                   help="(additive) keep the decoded frames as uint8 and scale them to [-1,1] on the GPU: same "
                        "predictions, a quarter of the host-to-device bytes")
    args = p.parse_args()
    for attr, val in (("model_type", args.model_type), ("loss_type", args.loss_type)):
        if val is not None:
            setattr(cf, attr, val)
    predict_network(weights_file=args.weights, datapath=args.datapath, fraction=args.fraction, log_dir=args.logdir,
                    batch_size=args.batch_size, u8_frames=args.u8_frames)
