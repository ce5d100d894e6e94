"""Training-loop callbacks with the protocol and names of the reference's spnet/callbacks.py.

  Callback                    minimal Keras-style base (set_model / on_train_begin / on_epoch_begin /
                              on_batch_begin / on_epoch_end)
  AugmentOnTheFly             callbacks.py:272-341 -- per-epoch cutout + salt&pepper of the WHOLE training
                              set from pristine frames, frame order 0..N-1 (reference RNG order), but the
                              frames live in HBM and the pixel work runs in HIP kernels
  get_1cycle_schedule,
  OneCycleScheduler           callbacks.py:346-406 -- per-batch learning-rate table
  MyProgressCallback          callbacks.py:58-265 -- validation predict, per-term losses.dat, count
                              metrics, progress.png, overlay drawings
  ParallelCheckpointCallback  callbacks.py:20-41 -- weights + full-model checkpoints every N epochs
"""
import os
import time

import numpy as np

from . import config as cf
from . import diagnostics, utils


class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


# ----------------------------------------------------------------------------- LR schedule
def get_1cycle_schedule(lr_max=1e-3, n_data_points=8000, epochs=200, batch_size=40, verbose=0):
    """Look-up table of per-iteration learning rates: linear warm-up over the first 30 % of the
    iterations from lr_max/25 to lr_max, then cosine annealing down to lr_max/25/1e4
    (callbacks.py:346-377)."""
    pct_start, div_factor = 0.3, 25.
    lr_start = lr_max / div_factor
    lr_end = lr_start / 1e4
    n_iter = n_data_points * epochs // batch_size
    a1 = int(n_iter * pct_start)
    a2 = n_iter - a1
    warm = np.linspace(lr_start, lr_max, a1)
    anneal = (lr_max - lr_end) * (1 + np.cos(np.linspace(0, np.pi, a2))) / 2 + lr_end
    return np.concatenate((warm, anneal))


class OneCycleScheduler(Callback):
    """Sets model.optimizer.lr from the table before every batch; the iteration counter is never reset
    between fit() calls (callbacks.py:380-406)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.verbose = kwargs.get('verbose', 0)
        self.lrs = get_1cycle_schedule(**kwargs)
        self.iteration = 0

    def on_batch_begin(self, batch, logs=None):
        self.model.optimizer.lr = float(self.lrs[min(self.iteration, len(self.lrs) - 1)])
        self.iteration += 1

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None:
            logs['lr'] = self.model.optimizer.lr
        if self.verbose > 0:
            print('\nLearning rate =', self.model.optimizer.lr)


# ----------------------------------------------------------------------------- augmentation
class AugmentOnTheFly(Callback):
    """Each epoch: X[i] = augment(X_orig[i]) for i = 0..N-1, Y unchanged (callbacks.py:319-338).

    X may be a numpy array (uploaded once; the reference's host-RAM copy X_orig becomes a device
    tensor) or already a device tensor.  The augmented set is a second device tensor that the model's
    fit() reads batches from (`model.set_train_frames`)."""

    def __init__(self, X, Y, orig_img_shape=(384, 512), aug_every=1, chunk=256, seed=1, real_blur=False):
        """real_blur=False reproduces the reference, whose Gaussian blur is a no-op (the result of cv2.GaussianBlur is
        discarded, augmentation.py:66-70); True applies it (train_spnet.py --augment_blur)."""
        super().__init__()
        import torch
        from . import parallel
        from .augmentation import DeviceAugmenter
        self.X, self.Y = X, Y
        self.aug_every = aug_every
        self.orig_img_shape = orig_img_shape
        self.chunk = chunk
        self.seed = seed
        parallel.init_distributed()          # this rank's GPU (no-op if the model already did it)
        dev = parallel.local_device()
        self.X_orig = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X)).to(dev)
        self.X_aug = self.X_orig.clone()      # rows no epoch has augmented yet hold the pristine frame, not garbage
        self.augmenter = DeviceAugmenter(self.X_orig, real_blur=real_blur)

    def set_model(self, model):
        super().set_model(model)
        if hasattr(model, "set_train_frames"):
            model.set_train_frames(self.X, self.X_aug)

    def on_epoch_begin(self, epoch, logs=None):
        if 0 != epoch % self.aug_every:
            return
        shard = getattr(self.model, "epoch_indices", None) if getattr(self.model, "world", 1) > 1 else None
        if shard is not None:
            return self._augment_shard(np.asarray(shard), getattr(self.model, "_epochs_seen", epoch))
        n = self.X_orig.shape[0]
        for lo in range(0, n, self.chunk):
            hi = min(n, lo + self.chunk)
            if (lo // self.chunk) % 16 == 0 or hi == n:
                print("   Augmenting on the fly: ", hi, "/", n, "\r", sep="", end="")
            self.augmenter.augment(list(range(lo, hi)), self.X_aug[lo:hi])
        print("")


    def _augment_shard(self, shard, epoch):
        """Data parallel: only the samples this rank trains on in this epoch are augmented, each from its own RNG
        stream seeded by (seed, epoch, sample index) -- the augmented frame of a sample does not depend on the world
        size or on the rank that draws it."""
        import torch
        from . import parallel
        tmp = None
        for lo in range(0, len(shard), self.chunk):
            idx = shard[lo:lo + self.chunk]
            if tmp is None or tmp.shape[0] != len(idx):
                tmp = torch.empty((len(idx),) + tuple(self.X_aug.shape[1:]), device=self.X_aug.device)
            seeds = [parallel.sample_seed(self.seed, epoch, i) for i in idx]
            self.augmenter.apply(self.augmenter.draw(list(idx), seeds=seeds), tmp)
            self.X_aug.index_copy_(0, torch.as_tensor(idx, dtype=torch.int64, device=self.X_aug.device), tmp)


# ----------------------------------------------------------------------------- checkpoints
class ParallelCheckpointCallback(Callback):
    def __init__(self, serial_model, filepath="weights.hdf5", save_every=1, dir='.'):
        super().__init__()
        self.model_to_save = serial_model
        self.save_every = save_every
        self.weights_path = dir + '/' + filepath
        self.model_path = dir + '/' + "spnet.model"

    def on_epoch_end(self, epoch, logs=None):
        if (1 == self.save_every) or ((0 == ((epoch + 1) % self.save_every)) and (epoch > 0)):
            os.makedirs(os.path.dirname(self.weights_path) or ".", exist_ok=True)
            print("Saving weights checkpoint to", self.weights_path)
            self.model_to_save.save_weights(self.weights_path)
            print("Saving entire model checkpoint to", self.model_path)
            self.model_to_save.save(self.model_path)


# ----------------------------------------------------------------------------- progress
hist, train_loss_hist, val_loss_hist, my_val_loss_hist, acc_hist = [], [], [], [], []
center_loss_hist, size_loss_hist, angle_loss_hist, noobj_loss_hist, rings_loss_hist = [], [], [], [], []
global_count = 0


class MyProgressCallback(Callback):
    def __init__(self, X_val=None, Y_val=None, val_file_list=None, log_dir="./logs", use_tb=False,
                 pred_shape=[3, 3, 4, 6], num_draw=40, make_plots=True):
        super().__init__()
        self.X_val, self.Y_val, self.val_file_list = X_val, Y_val, val_file_list
        self.log_dir, self.pred_shape = log_dir, pred_shape
        self.num_draw, self.make_plots = num_draw, make_plots
        os.makedirs(log_dir, exist_ok=True)
        self.loss_file = open(log_dir + '/losses.dat', 'a')
        self.loss_file.write('# epoch Train_total Val_total center size angle noobj class\n')

    def on_epoch_end(self, epoch, logs=None):
        from . import models
        global global_count
        logs = logs or {}
        global_count += 1
        hist.append(global_count)
        train_loss_total, val_loss_total = logs.get('loss'), logs.get('val_loss')
        train_loss_hist.append(train_loss_total)
        val_loss_hist.append(val_loss_total)
        m = self.Y_val.shape[0]
        print("\n MyProgressCallback: predicting, testing & saving plot")
        print("    Predicting... (m = ", m, " frames in val set)", sep="")
        t0 = time.time()
        Y_pred = self.model.predict(self.X_val)
        elapsed = time.time() - t0
        print("    ...elapsed time to predict = ", elapsed, "s.   FPS = ", m * 1.0 / elapsed)
        my_val_loss, parts = models.my_loss(self.Y_val, Y_pred)
        center_loss, size_loss, angle_loss, noobj_loss, rings_loss = parts
        for lst, v in zip((my_val_loss_hist, center_loss_hist, size_loss_hist, angle_loss_hist, noobj_loss_hist,
                           rings_loss_hist), (my_val_loss, center_loss, size_loss, angle_loss, noobj_loss, rings_loss)):
            lst.append(v)
        self.loss_file.write(f'{epoch} {train_loss_total} {my_val_loss} {center_loss} {size_loss} {angle_loss} '
                             f'{noobj_loss} {rings_loss}\n')
        self.loss_file.flush()
        if cf.loss_type != 'same':
            Y_pred[:, cf.ind_noobj::cf.vars_per_pred] = 1.0 / (1.0 + np.exp(-Y_pred[:, cf.ind_noobj::cf.vars_per_pred]))
        Yv, Yp = utils.denorm_Y(self.Y_val), utils.denorm_Y(Y_pred)
        (ring_miscounts, ring_truecounts, total_obj, false_obj_pos, false_obj_neg, true_obj_pos, true_obj_neg,
         pix_err, ipem) = diagnostics.calc_errors(Yp, Yv)
        mistakes = ring_miscounts + false_obj_pos + false_obj_neg
        class_acc = (total_obj - mistakes) * 100.0 / max(total_obj, 1)
        acc_hist.append(class_acc)
        print('    Losses: Epoch Train:total Val:total   center     size       angle      noobj      rings')
        print(f'          {epoch:3d}     {train_loss_total}   {val_loss_total}   {center_loss:.3e} '
              f' {size_loss:.3e}  {angle_loss:.3e}  {noobj_loss:.3e}  {rings_loss:.3e}')
        if self.make_plots:
            self._plot(class_acc)
        if self.val_file_list is not None and self.num_draw > 0:
            try:
                utils.show_pred_ellipses(Yv, Yp, self.val_file_list, num_draw=self.num_draw, log_dir=self.log_dir,
                                         ind_extra=ipem)
            except (FileNotFoundError, OSError) as exc:      # synthetic in-memory sets have no image files
                print("    (skipping overlay drawings:", exc, ")")
        print("    In whole val dataset:")
        print('        Mean pixel error =', np.mean(pix_err))
        print("        Max pixel error =", pix_err[ipem], " (index =", ipem, ").")
        tot = max(total_obj, 1)
        print("    Ring correct counts = ", ring_truecounts, ' / ', total_obj, '.   = ', 100 * ring_truecounts / tot,
              ' % ring-class accuracy', sep="")
        print("         Ring miscounts = ", ring_miscounts, ' / ', total_obj, sep="")
        print("        False positives = ", false_obj_pos, ' / ', total_obj, sep="")
        print("        False negatives = ", false_obj_neg, ' / ', total_obj, sep="")
        print("         True positives = ", true_obj_pos, ' / ', total_obj, sep="")
        print("         True negatives = ", true_obj_neg, sep="")
        print("    Total Mistakes = ", mistakes, ' / ', total_obj, '.   => ', class_acc,
              ' % class. accuracy rate (lack of mistakes)', sep="")

    def _plot(self, class_acc):
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        fig = plt.figure(figsize=(10, 3.75))
        ax = plt.subplot(121)
        for series, label in ((train_loss_hist, "Train"), (my_val_loss_hist, "Val: Total"), (center_loss_hist, "Val: Center"),
                              (size_loss_hist, "Val: Size"), (angle_loss_hist, "Val: Angle"),
                              (noobj_loss_hist, "Val: NoObj"), (rings_loss_hist, "Val: Rings")):
            ys = [np.nan if v is None else v for v in series]
            ax.loglog(hist, ys, '-', label=label)
        ax.set_xlabel('(Global) Epoch')
        ax.set_ylabel('Loss')
        ax.legend(loc='lower left', fancybox=True, framealpha=0.8)
        ax = plt.subplot(122, ylim=[0, 100])
        ax.plot(hist, acc_hist, '-', color='orange', label='Acc = {:5.2f} %'.format(class_acc))
        ax.set_xlabel('(Global) Epoch')
        ax.set_ylabel('Accuracy (%)')
        ax.legend(loc='lower right', fancybox=True, framealpha=0.8)
        fig.tight_layout()
        plt.savefig(self.log_dir + '/progress.png')
        plt.close(fig)
