#! /usr/bin/env python3
"""Train SPNet on a directory with Train/ and Val/ (PNG + same-stem CSV) -- entry point and flags of the
reference's train_spnet.py, running on the MI355X engine.

Additive flags (not in the reference): --model_type / --loss_type / --backbone override the
spnet.config globals; multi-GPU = launch with `python -m torch.distributed.run --nproc-per-node N`.
"""
import argparse
import os
import random
import shutil
import sys
import time

import numpy as np

from spnet import callbacks, models, multi_gpu, utils  # noqa: F401
import spnet.config as cf
from evaluate_spnet import evaluate_network
from predict_spnet import default_image_dir, predict_network


def train_network(weights_file="weights.hdf5", datapath=".", fraction=1.0, batch_size=32, epochs=30, pred_grid=[6, 6, 2],
                  noaugment=False, log_dir=".", lr_max=4e-5, freeze_fac=0.7, frozen_epochs=4, random_seed=1,
                  augment_blur=False):
    np.random.seed(random_seed)
    # Data parallel (launched by torch.distributed.run): choose this rank's GPU and join the process group before
    # anything touches the device; rank 0 alone logs, validates and writes checkpoints.
    rank, _, world = multi_gpu.parallel.init_distributed()
    print("pred_grid = ", pred_grid)
    X_train, Y_train, train_file_list, pred_shape = utils.build_dataset(
        path=datapath + "/Train/", load_frac=fraction, set_means_ranges=True, batch_size=batch_size, pred_grid=pred_grid)
    X_val, Y_val, val_file_list, pred_shape = utils.build_dataset(
        path=datapath + "/Val/", load_frac=1.0, set_means_ranges=False, batch_size=batch_size, pred_grid=pred_grid)

    print("Seting up NN model.  model_type = ", cf.model_type)
    parallel = world > 1
    model, serial_model = models.setup_model(X_train, Y_train[0].size, no_cp_fatal=False, weights_file=weights_file,
                                             parallel=parallel, freeze_fac=freeze_fac)

    callback_list = []
    if rank == 0:           # logging / validation / checkpoints: one writer
        callback_list += [
            callbacks.MyProgressCallback(X_val=X_val, Y_val=Y_val, val_file_list=val_file_list, log_dir=log_dir,
                                         pred_shape=pred_shape),
            callbacks.ParallelCheckpointCallback(model, filepath=weights_file, save_every=5, dir=log_dir)]
    # one optimizer iteration consumes batch_size frames on EVERY rank
    callback_list.append(callbacks.OneCycleScheduler(lr_max=lr_max, n_data_points=X_train.shape[0], epochs=epochs,
                                                     batch_size=batch_size * world, verbose=int(rank == 0)))
    if not noaugment:
        print("Adding callback for augment on the fly")
        callback_list.append(callbacks.AugmentOnTheFly(X_train, Y_train, aug_every=1, seed=random_seed,
                                                       real_blur=augment_blur))

    fit_args = dict(batch_size=batch_size, shuffle=True, verbose=1, validation_data=(X_val, Y_val), callbacks=callback_list)
    if frozen_epochs > 0 and freeze_fac > 0.0:        # warm-up phase with the first layers frozen
        model.fit(X_train, Y_train, epochs=frozen_epochs, **fit_args)
    if freeze_fac > 0.0:
        model = models.unfreeze_model(model, X_train, Y_train, parallel=parallel)
    model.fit(X_train, Y_train, epochs=epochs - frozen_epochs, **fit_args)
    if os.environ.get("SPNET_DUMP_WEIGHT_SUM"):       # test hook: one checksum of the trained weights per rank
        import hashlib
        h = hashlib.sha256()
        for k, v in model.state_dict().items():
            if "moving_" not in k:           # BatchNorm statistics are per replica (tower semantics), weights are not
                h.update(v.numpy().tobytes())
        with open("%s.%d" % (os.environ["SPNET_DUMP_WEIGHT_SUM"], rank), "w") as f:
            f.write(h.hexdigest())
    return model


if __name__ == '__main__':
    seed = 1
    np.random.seed(seed)
    random.seed(seed)
    p = argparse.ArgumentParser(description="trains network on training dataset",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-b', '--batch_size', type=int, default=16, help='Batch size to use')
    p.add_argument('-d', '--datapath', default="./", help='Directory with images in Train/ and Val/ subdirs')
    p.add_argument('-e', '--epochs', type=int, default=100, help='Number of epochs to run')
    p.add_argument('-f', '--fraction', type=float, default=1.0, help='Fraction of dataset to use')
    p.add_argument('--freeze_fac', type=float, default=0.0, help='Fraction of base model to freeze')
    p.add_argument('--frozen_epochs', type=int, default=0, help='Starting epochs to run while base model is frozen')
    p.add_argument('-g', '--grid', default="6x6x2", help='Shape of predictor grid')
    p.add_argument('-w', '--weights', default="weights.hdf5", help='Weights file')
    p.add_argument('-l', '--lrmax', type=float, default=4e-5, help='Maximum learning rate value')
    p.add_argument('-n', '--noaugment', action='store_true', help="don't augment on the fly")
    p.add_argument('--name', default='', help='Descriptive name of the run, prepended to the log directory name')
    p.add_argument('-r', '--random_seed', type=int, default=1, help="Random seed value")
    p.add_argument('--model_type', default=None, help="override spnet.config.model_type ('monolithic' | 'big')")
    p.add_argument('--loss_type', default=None, help="override spnet.config.loss_type ('same' | 'hybrid')")
    p.add_argument('--backbone', default=None, help="override spnet.config.basemodel")
    p.add_argument('--augment_blur', action='store_true',
                   help="apply the Gaussian blur of the on-the-fly augmentation (the reference computes and discards it)")
    args = p.parse_args()
    print("Command line ~= \n", ' '.join(sys.argv))
    print("args = ", args)
    for attr, val in (("model_type", args.model_type), ("loss_type", args.loss_type), ("basemodel", args.backbone)):
        if val is not None:
            setattr(cf, attr, val)

    pred_grid = [int(i) for i in args.grid.split('x')]
    now = time.strftime("%c").replace('  ', '_').replace(' ', '_')
    log_dir = './logs/' + (args.name + '_' + now if args.name else now)
    print("Logging will go to ", log_dir)

    print("\n----------------------------\nStarting training...")
    model = train_network(weights_file=args.weights, datapath=args.datapath, fraction=args.fraction,
                          batch_size=args.batch_size, epochs=args.epochs, pred_grid=pred_grid, noaugment=args.noaugment,
                          log_dir=log_dir, lr_max=args.lrmax, freeze_fac=args.freeze_fac,
                          frozen_epochs=args.frozen_epochs, random_seed=args.random_seed,
                          augment_blur=args.augment_blur)

    if int(os.environ.get("RANK", "0")) == 0:
        print("\n----------------------------\nStarting model evaluation...")
        testpath = args.datapath + '/Test/'
        if not os.path.isdir(testpath):
            testpath = args.datapath + '/Val/'
        evaluate_network(model=model, weights_file="", datapath=testpath, fraction=1.0, log_dir="logs/Evaluation/",
                         batch_size=args.batch_size, pred_grid=pred_grid, set_means_ranges=False)
        if os.path.isdir(default_image_dir):      # the reference predicts on the author's unlabeled set here
            print("\n----------------------------\nStarting Zooniverse predictions...")
            predict_network(weights_file="", fraction=args.fraction, log_dir='logs/Predicting/',
                            batch_size=args.batch_size, model=model, X_pred='')
        weights2name = "final_" + args.weights
        print("Just to be sure: Saving model to", weights2name)
        model.save_weights(weights2name)
        print("And saving full model too")
        model.save("full_model.h5")
        for f in (weights2name, "full_model.h5", "nohup.out"):
            if os.path.exists(f):
                shutil.copy(f, log_dir)
        print("SPNet execution completed.")
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
