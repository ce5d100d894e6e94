"""Model construction, loss and the Keras-like model object -- the surface of the reference's
spnet/models.py, backed by the HIP engine (spnet_amd/engine.py).

  setup_model(X, Y0size, try_checkpoint, no_cp_fatal, weights_file, freeze_fac, parallel, quick_setup)
      -> (model, serial_model)                                          models.py:461-507
  create_model_functional(X, Y0size, freeze_fac, quick_setup)          models.py:302-424
  build_model(...)          north-star alias of create_model_functional (not in the reference)
  unfreeze_model(model, X, Y, parallel)                                 models.py:510-552
  custom_loss(y_true, y_pred), my_loss(y_true, y_pred)                  models.py:557-633
  SelectiveSigmoid, InterleaveColumns                                    models.py:223-298
  Model: fit / predict / evaluate / save_weights / load_weights / save / get_weights / set_weights /
         optimizer.lr / layers / trainable                              (Keras Model protocol used by
                                                                         train/predict/evaluate_spnet.py)
Checkpoints: h5py is not available on the target image, so weight files are safetensors containers
keyed by the Keras weight names (conv2d_1/kernel, block5_sepconv2/pointwise_kernel, ...), written to
the same file names the reference uses (weights.hdf5, spnet.model, full_model.h5, ...).
"""
import json
import os
import time
from os.path import isfile

import numpy as np

from . import config as cf
from . import multi_gpu, utils  # noqa: F401  (same import surface as the reference module)

lambda_center = 2.0
lambda_size = 1.0
lambda_angle = 3.0
lambda_noobj = 0.3
lambda_class = 5.0
logeps = 1e-10


def _torch():
    import torch
    return torch


def _require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("spnet_amd.models needs a HIP device: the hot path has no CPU fallback")
    return torch


# ----------------------------------------------------------------------------- loss functions
def _loss_on_device(y_true, y_pred):
    torch = _require_gpu()
    from . import _lib as L
    dev = torch.device("cuda", torch.cuda.current_device())
    yt = torch.as_tensor(np.ascontiguousarray(y_true) if isinstance(y_true, np.ndarray) else y_true, dtype=torch.float32).to(dev).contiguous()
    yp = torch.as_tensor(np.ascontiguousarray(y_pred) if isinstance(y_pred, np.ndarray) else y_pred, dtype=torch.float32).to(dev).contiguous()
    B, n = yp.shape
    parts = torch.empty(B, 5, device=dev)
    out = torch.empty(6, device=dev)
    L.spnet_ellipse_loss(yt.data_ptr(), yp.data_ptr(), None, parts.data_ptr(), out.data_ptr(), B, n,
                         0 if cf.loss_type == 'same' else 1, torch.cuda.current_stream().cuda_stream)
    return out.cpu().numpy().astype(np.float64)


def custom_loss(y_true, y_pred):
    """lambda-weighted MSE over the predictor grid, angle term weighted by (a-b)^2, optional
    BCE-with-logits on noobj when cf.loss_type != 'same' (models.py:564-589).  Scalar."""
    return float(_loss_on_device(y_true, y_pred)[5])


def my_loss(y_true, y_pred, verbosity=0):
    """(total, [center, size, angle, noobj, class]) -- models.py:594-633."""
    o = _loss_on_device(y_true, y_pred)
    return float(o[5]), o[:5].copy()


# ----------------------------------------------------------------------------- output layers (legacy heads)
class SelectiveSigmoid:
    """Sigmoid on columns start::skip, identity elsewhere (models.py:277-298)."""

    def __init__(self, **kwargs):
        self.start = kwargs.get('start', cf.ind_noobj)
        self.end = kwargs.get('end', None)
        self.skip = kwargs.get('skip', cf.vars_per_pred)
        self.sigmoid_stretch = 1

    def __call__(self, x):
        torch = _torch()
        if isinstance(x, torch.Tensor) and x.is_cuda and self.end is None and x.dtype == torch.float32:
            from . import _lib as L             # device tensors: the HIP kernel, out of place like the Keras layer
            y = x.contiguous().clone()
            L.spnet_selective_sigmoid(y.data_ptr(), None, y.shape[0], y.shape[1], self.start, self.skip, 0,
                                      torch.cuda.current_stream().cuda_stream)
            return y
        y = np.array(x, dtype=np.float32, copy=True)
        sl = slice(self.start, self.end, self.skip)
        y[:, sl] = self.sigmoid_stretch / (1.0 + np.exp(-y[:, sl]))
        return y

    call = __call__


class InterleaveColumns:
    """[n_preds sigmoid columns | remaining columns] -> per-predictor order with the first group at
    `start_index` inside each group of vars_per_pred (models.py:223-274; the reference's bare
    `vars_per_pred` NameError at :242 is not reproduced)."""

    def __init__(self, start_index=6, **kwargs):
        self.start_index = start_index

    def column_map(self, n_vars):
        v = cf.vars_per_pred
        if n_vars % v != 0:
            raise ValueError("n_vars (=%d) must be a multiple of vars_per_pred (=%d)" % (n_vars, v))
        n_preds = n_vars // v
        cml = [self.start_index + x * v for x in range(n_preds)]
        for i in range(n_preds):
            cml += [x + i * v for x in range(self.start_index)]
            cml += [1 + x + i * v + self.start_index for x in range(v - self.start_index - 1)]
        return cml

    def __call__(self, x):
        x = np.asarray(x)
        out = np.empty_like(x)
        out[:, self.column_map(x.shape[-1])] = x
        return out

    call = __call__


# ----------------------------------------------------------------------------- model object
# epsilon of keras.optimizers.Adam as the reference constructs it (Adam(lr=0.00001), spnet/models.py:494): Keras 2.1.3's
# default is epsilon=None -> K.epsilon() = 1e-7 (earlier releases: 1e-8).  No Keras run can pin it here; the early
# training dynamics of the reference's published run select 1e-7 (l2 penalty after epoch 1: 0.2203 against the log's
# 0.2204, 0.2187 with 1e-8; DESIGN.md section 1c, profiles/r04_early2_d06_*).
ADAM_EPS = 1e-7


def _l2_coef():
    from .engine import L2_COEF
    return L2_COEF


class _Optimizer:
    def __init__(self, lr=1e-5):
        self.lr = lr


def keras_layer_table(basemodel=None):
    """Ordered (layer_name, [weight-name prefixes]) list approximating base_model.layers of the
    reference (144 entries: 13 stem layers incl. the input + 131 Xception layers; run log
    'Freezing 0 / 144 layers').  Used only to translate freeze_fac into a set of frozen tensors."""
    from .engine import MOBILENET_BLOCKS, irv2_program, xception_plan
    t = [("input_1", []), ("conv2d_1", ["conv2d_1"]), ("average_pooling2d_1", []),
         ("batch_normalization_1", ["batch_normalization_1"]), ("leaky_re_lu_1", []), ("conv2d_2", ["conv2d_2"]),
         ("batch_normalization_2", ["batch_normalization_2"]), ("leaky_re_lu_2", []), ("conv2d_3", ["conv2d_3"]),
         ("batch_normalization_3", ["batch_normalization_3"]), ("average_pooling2d_2", []), ("add_1", []),
         ("dropout_1", [])]
    if (basemodel or cf.basemodel) == 'InceptionResNetV2':     # conv / bn / activation triples, pools, concats, lambdas
        for op in irv2_program():
            if op[0] == "conv":
                t.append((op[1], [op[1]]))
                if op[2]:
                    t.append((op[2], [op[2]]))
                if op[10]:
                    t.append((op[1] + "_ac", []))
            elif op[0] == "resadd":
                t += [("block_lambda", [])] + ([("block_ac", [])] if op[5] else [])
            else:
                t.append((op[0], []))
        return t
    if (basemodel or cf.basemodel) == 'MobileNet':       # keras.applications.mobilenet layer order
        t += [("conv1", ["conv1"]), ("conv1_bn", ["conv1_bn"]), ("conv1_relu", [])]
        for i in range(1, len(MOBILENET_BLOCKS) + 1):
            for kind in ("dw", "pw"):
                n = "conv_%s_%d" % (kind, i)
                t += [(n, [n]), (n + "_bn", [n + "_bn"]), (n + "_relu", [])]
        return t
    for n in ("block1_conv1", "block1_conv2"):
        t += [(n, [n]), (n + "_bn", [n + "_bn"]), (n + "_act", [])]

    def sep(name, act_before):
        out = [(name + "_act", [])] if act_before else []
        return out + [(name, [name]), (name + "_bn", [name + "_bn"])]

    for blk in xception_plan():
        if blk[0] == "strided":
            _, b, cin, c1, c2, first_relu, cn, bnn = blk
            t += sep("block%d_sepconv1" % b, first_relu) + sep("block%d_sepconv2" % b, True)
            t += [(cn, [cn]), ("block%d_pool" % b, []), (bnn, [bnn]), ("add_b%d" % b, [])]
        elif blk[0] == "middle":
            for k in (1, 2, 3):
                t += sep("block%d_sepconv%d" % (blk[1], k), True)
            t += [("add_b%d" % blk[1], [])]
        elif blk[0] == "exit":
            t += [("block14_sepconv1", ["block14_sepconv1"]), ("block14_sepconv1_bn", ["block14_sepconv1_bn"]),
                  ("block14_sepconv1_act", []), ("block14_sepconv2", ["block14_sepconv2"]),
                  ("block14_sepconv2_bn", ["block14_sepconv2_bn"]), ("block14_sepconv2_act", [])]
    return t


class Model:
    """Keras-Model-like handle over the HIP engine.  Frames are [N,H,W,1] float32 in [-1,1]."""

    def __init__(self, input_shape, Y0size=576, freeze_fac=0.0, seed=None, device=None):
        torch = _require_gpu()
        from . import parallel
        from .engine import Engine
        # Select this rank's GPU (and join the torchrun process group) BEFORE anything is allocated: every plan,
        # callback buffer and kernel launch of this process then lives on cuda:LOCAL_RANK.
        self.rank, _, self.world = parallel.init_distributed()
        if cf.basemodel not in ('Xception', 'MobileNet', 'InceptionResNetV2'):
            raise NotImplementedError("this build implements the Xception, MobileNet and InceptionResNetV2 backbones "
                                      "(cf.basemodel=%r)" % cf.basemodel)
        self.basemodel = cf.basemodel
        self.input_shape = tuple(int(v) for v in input_shape)
        H, W = self.input_shape[0], self.input_shape[1]
        self.H, self.W, self.Y0size = H, W, int(Y0size)
        self.device = device or str(parallel.local_device())
        self.seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else seed
        if self.world > 1:                 # replicas must start from identical weights: rank 0's seed rules
            import torch.distributed as dist
            box = [self.seed]
            dist.broadcast_object_list(box, src=0)
            self.seed = int(box[0])
        self.optimizer = _Optimizer(1e-5)
        self.trainable = True
        self.stop_training = False
        self.layers = [name for name, _ in keras_layer_table(self.basemodel)] + ["flatten_1", "FinalOutput"]
        self.freeze_fac = freeze_fac
        # 'compound' head (models.py:379-386): Dense(n_preds, sigmoid) 'SigmoidOutput' + Dense(rest) 'DenseOutput',
        # concatenated and re-ordered by InterleaveColumns(start_index=cf.ind_noobj) = one [K, Y0size] dense layer in
        # FINAL column order whose columns ind_noobj::vars_per_pred pass through a sigmoid (engine: sigmoid_cols).
        self.compound = (cf.model_type == 'compound')
        if self.compound and self.Y0size % cf.vars_per_pred != 0:
            raise ValueError("Y0size (=" + str(self.Y0size) + ") must be a multiple of cf.vars_per_pred (=" +
                             str(cf.vars_per_pred) + ")")
        self._Engine = Engine
        self._engines = {}
        self._root = None
        self._train_frames = None          # (id of host array, device tensor) registered by AugmentOnTheFly
        self._uploaded = {}
        self._rings = {}                   # predict(): pinned staging rings per (batch size, device)
        self._epochs_seen = 0              # epochs trained so far (over all fit() calls): the shard permutation's index
        self.epoch_indices = None          # data parallel: this rank's sample indices of the current epoch
        self._base = self._engine(1, train=False)      # owns the weights
        self._apply_freeze(freeze_fac)

    # -- engines ---------------------------------------------------------------------------------
    def _engine(self, batch, train):
        """One launch plan per (batch size, train?) -- all plans share the first one's weight buffers."""
        key = (int(batch), bool(train))
        if key not in self._engines:
            root = self._root
            if root is not None and train and not hasattr(root, "grad"):
                torch = _torch()
                for a in ("grad", "m", "v"):       # optimizer state lives beside the weights it updates
                    setattr(root, a, torch.zeros(root.n_theta, device=root.dev, dtype=torch.float32))
            eng = self._Engine(self.H, self.W, batch, n_out=self.Y0size, device=self.device, loss_type=cf.loss_type,
                               seed=self.seed, train=train, adam_eps=ADAM_EPS, share_from=root, rank=self.rank,
                               sigmoid_cols=(cf.ind_noobj, cf.vars_per_pred) if self.compound else None,
                               backbone=self.basemodel, pointwise=getattr(cf, "pointwise_gemm", "bf16x3"))
            if root is None:
                self._root = eng
            self._engines[key] = eng
        eng = self._engines[key]
        eng.loss_type = cf.loss_type
        return eng

    def _apply_freeze(self, freeze_fac):
        torch = _torch()
        self._frozen_prefixes = []
        table = keras_layer_table(self.basemodel)
        n_freeze = int(len(table) * freeze_fac)
        if freeze_fac == 1.0:
            n_freeze = len(table)
        for _, prefixes in table[:n_freeze]:
            self._frozen_prefixes += prefixes
        self._mask = None
        if self._frozen_prefixes:
            r = self._root
            mask = torch.ones(r.n_theta, device=r.dev, dtype=torch.float32)
            for name, (off, n, _) in r.p_off.items():
                if name.split("/")[0] in self._frozen_prefixes:
                    mask[off:off + n] = 0
            self._mask = mask

    # -- weights ----------------------------------------------------------------------------------
    def get_weights(self):
        return [v.numpy() for v in self.state_dict().values()]

    def set_weights(self, weights):
        self.load_state_dict(dict(zip(list(self.state_dict().keys()), weights)))

    def _head_columns(self):
        """(sigmoid columns, dense columns) of the engine's FinalOutput kernel for the 'compound' head."""
        sig = list(range(cf.ind_noobj, self.Y0size, cf.vars_per_pred))
        return sig, [c for c in range(self.Y0size) if c not in set(sig)]

    def state_dict(self):
        """name -> CPU tensor.  'compound' models expose the reference's two head layers (SigmoidOutput, DenseOutput:
        the column blocks InterleaveColumns interleaves) instead of the engine's single FinalOutput."""
        sd = self._root.state_dict()
        if not self.compound:
            return sd
        sig, rest = self._head_columns()
        out = type(sd)()
        for k, v in sd.items():
            if k == "FinalOutput/kernel":
                out["SigmoidOutput/kernel"], out["DenseOutput/kernel"] = v[:, sig].contiguous(), v[:, rest].contiguous()
            elif k == "FinalOutput/bias":
                out["SigmoidOutput/bias"], out["DenseOutput/bias"] = v[sig].contiguous(), v[rest].contiguous()
            else:
                out[k] = v
        return out

    def load_state_dict(self, sd):
        if self.compound and "SigmoidOutput/kernel" in sd:
            torch = _torch()
            sig, rest = self._head_columns()
            sd = dict(sd)
            K = sd["SigmoidOutput/kernel"].shape[0]
            w, b = torch.empty(K, self.Y0size), torch.empty(self.Y0size)
            w[:, sig], w[:, rest] = torch.as_tensor(sd.pop("SigmoidOutput/kernel")), torch.as_tensor(sd.pop("DenseOutput/kernel"))
            b[sig], b[rest] = torch.as_tensor(sd.pop("SigmoidOutput/bias")), torch.as_tensor(sd.pop("DenseOutput/bias"))
            sd["FinalOutput/kernel"], sd["FinalOutput/bias"] = w, b
        self._root.load_state_dict(sd)

    def count_params(self):
        sd = self._root.state_dict()
        tr = sum(v.numel() for k, v in sd.items() if not ("moving_" in k))
        return tr + sum(v.numel() for k, v in sd.items() if "moving_" in k), tr

    def save_weights(self, path):
        from safetensors.torch import save_file
        meta = {"format": "spnet_amd-weights-v1", "input_shape": json.dumps(self.input_shape), "Y0size": str(self.Y0size),
                "basemodel": self.basemodel}
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, path, metadata=meta)

    def load_weights(self, path, by_name=False):
        from safetensors.torch import load_file
        self.load_state_dict(load_file(path))

    def save(self, path):
        """'Whole model' file = weights + the configuration needed to rebuild the plan."""
        from safetensors.torch import save_file
        meta = {"format": "spnet_amd-model-v1", "input_shape": json.dumps(self.input_shape), "Y0size": str(self.Y0size),
                "basemodel": self.basemodel, "model_type": "compound" if self.compound else cf.model_type,
                "loss_type": cf.loss_type,
                "optimizer_iterations": str(self._root.t)}
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, path, metadata=meta)

    # -- data plumbing ----------------------------------------------------------------------------
    def set_train_frames(self, host_array, device_tensor):
        """AugmentOnTheFly registers the device tensor that shadows the host training array."""
        self._train_frames = (id(host_array), device_tensor)

    def _device_frames(self, X):
        torch = _torch()
        # (the registered training array first: AugmentOnTheFly given a device tensor keeps it pristine and augments a
        # second tensor, which is the one fit() must read -- the reference's in-place aliasing, callbacks.py:289,336)
        if self._train_frames is not None and self._train_frames[0] == id(X):
            return self._train_frames[1]
        if isinstance(X, torch.Tensor):
            return X if X.is_cuda else X.to(self.device)
        # One-entry upload cache so that fit() does not re-send an unchanged training array every epoch.  The
        # entry keeps a reference to the host array: id() of a dead temporary (X[:n]) can be handed to the next
        # one, and would otherwise return the wrong frames.
        key = (id(X), X.shape, X.__array_interface__["data"][0])
        hit = self._uploaded.get(key)
        if hit is None or hit[0] is not X:
            self._uploaded = {key: (X, torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(self.device))}
        return self._uploaded[key][1]

    # -- inference --------------------------------------------------------------------------------
    def predict(self, X, batch_size=32, verbose=0):
        """model.predict(X, batch_size) (predict_spnet.py:85, evaluate_spnet.py:66): [N,576] float32.

        One static launch plan per batch size, replayed as a hipGraph (Engine.predict_step).  Host frames are STREAMED:
        batch k+1 travels host -> pinned ring -> HBM on a copy stream while batch k is being computed, so neither
        the whole set nor a second copy of it has to fit anywhere and the PCIe transfer hides behind the forward
        passes; device-resident frames are read in place.  uint8 frames -- host arrays or device tensors alike -- are grey
        levels 0..255 and are scaled to [-1,1] on the device (an addition of this build: Keras would cast them unscaled);
        float frames are taken as they are."""
        torch = _torch()
        N = int(X.shape[0])
        bs = max(1, min(int(batch_size), max(N, 1)))
        eng = self._engine(bs, train=False)
        dev = eng.dev
        out = torch.empty(N, self.Y0size, device=dev)
        if N == 0:
            return out.cpu().numpy()
        resident = X if isinstance(X, torch.Tensor) else None
        if resident is None and self._train_frames is not None and self._train_frames[0] == id(X):
            resident = self._train_frames[1]
        if resident is not None and not resident.is_cuda:
            resident = resident.to(dev)

        from . import _lib as L

        def run(lo, hi, frames):
            """frames: device tensor holding hi-lo frames; float32 in [-1,1], or uint8 grey levels, which are scaled on
            the device exactly as the input codec scales them on the host (spnet_u8_to_input: utils.py:340-342) --
            the same frames give the same predictions whatever container they arrive in"""
            n = hi - lo
            if frames.dtype == torch.uint8:
                if frames.data_ptr() & 15:          # a slice of a resident uint8 tensor whose frame size is not a multiple of
                    frames = frames.clone()         # 16 bytes (331 x 331): the kernel wants 16-byte aligned bytes
                L.spnet_u8_to_input(frames.data_ptr(), eng.x_in.data_ptr(), n * self.H * self.W, L.current_stream())
                if n < bs:                           # ragged tail: pad with the last frame, drop the extras
                    eng.x_in[n:].copy_(eng.x_in[n - 1:n].expand(bs - n, -1, -1, -1))
            elif n == bs:
                eng.x_in.copy_(frames.reshape(eng.x_in.shape))
            else:                                   # ragged tail: pad with the last frame, drop the extras
                eng.x_in[:n].copy_(frames.reshape(n, self.H, self.W, 1))
                eng.x_in[n:].copy_(frames.reshape(n, self.H, self.W, 1)[n - 1:n].expand(bs - n, -1, -1, -1))
            y = eng.predict_step()
            out[lo:hi].copy_(y[:n])

        if resident is not None:
            resident = resident.contiguous()
            for lo in range(0, N, bs):
                hi = min(N, lo + bs)
                run(lo, hi, resident[lo:hi])
            return out.cpu().numpy()

        # host frames: pinned ring + copy stream.  uint8 frames (grey levels as the PNGs hold them) travel as one byte
        # per pixel and are scaled to [-1,1] on the device (spnet_u8_to_input: the arithmetic of utils.py:340-342,
        # bit-identical to the host conversion) straight into the plan's input buffer.
        u8 = isinstance(X, np.ndarray) and X.dtype == np.uint8
        dt_h = torch.uint8 if u8 else torch.float32
        Xh = np.ascontiguousarray(X, dtype=np.uint8 if u8 else np.float32).reshape(N, self.H, self.W, 1)
        depth = 3
        key = (bs, str(dev), u8)
        ring = self._rings.get(key)
        if ring is None:
            ring = [(torch.empty((bs, self.H, self.W, 1), dtype=dt_h).pin_memory(),
                     torch.empty((bs, self.H, self.W, 1), dtype=dt_h, device=dev),
                     torch.cuda.Event(), torch.cuda.Event()) for _ in range(depth)]
            self._rings[key] = ring
            self._copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream()
        for k, lo in enumerate(range(0, N, bs)):
            hi = min(N, lo + bs)
            n = hi - lo
            host, devbuf, landed, consumed = ring[k % depth]
            consumed.synchronize()                   # the forward that read this slot `depth` batches ago is done
            # (numpy's memcpy: 4 ms per 100 MB batch; torch's CPU copy_ spreads the same bytes over every host thread and
            # takes 23 ms on the 128-thread GPU box, which made the HOST the limiter of the streamed path: 2,600 frames/s)
            np.copyto(host.numpy()[:n], Xh[lo:hi])
            with torch.cuda.stream(self._copy_stream):
                devbuf[:n].copy_(host[:n], non_blocking=True)
                landed.record(self._copy_stream)
            main.wait_event(landed)
            run(lo, hi, devbuf[:n])
            consumed.record(main)
        return out.cpu().numpy()

    def predict_u8(self, X_u8, batch_size=32, verbose=0):
        """predict() over uint8 grey-level frames [N,H,W(,1)] (what load_img yields before utils.py:340-342 scales it):
        same result as predict(to_network_input(X_u8)), a quarter of the bytes over PCIe."""
        X_u8 = np.asarray(X_u8)
        if X_u8.dtype != np.uint8:
            raise TypeError("predict_u8 expects uint8 frames, got %s" % X_u8.dtype)
        return self.predict(X_u8, batch_size=batch_size, verbose=verbose)

    def evaluate(self, X, Y, batch_size=32, verbose=0):
        return custom_loss(Y, self.predict(X, batch_size=batch_size))

    # -- training ---------------------------------------------------------------------------------
    def fit(self, X, Y, batch_size=32, epochs=1, shuffle=True, verbose=1, validation_data=None, callbacks=None,
            initial_epoch=0):
        """Keras Model.fit.  Under torchrun (world > 1) `batch_size` is the PER-GPU batch: every rank trains on its
        rank-strided slice of a shared-seed epoch permutation (parallel.shard_indices), gradients are all-reduced
        bucket by bucket during backward, rank 0 alone prints, validates and feeds the logging callbacks."""
        torch = _torch()
        from . import _lib as L
        from . import parallel
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.set_model(self)
        eng = self._engine(batch_size, train=True)
        eng.update_mask = self._mask
        upload = L.AsyncUploader(eng.dev)
        world, rank = self.world, self.rank
        reducer = eng.make_reducer() if world > 1 else None
        Yd = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float32)).to(eng.dev) if isinstance(Y, np.ndarray) else Y
        Yd = Yd.to(eng.dev).float().contiguous()
        N = Yd.shape[0]
        chatty = verbose and rank == 0
        for cb in callbacks:
            cb.on_train_begin({})
        history = {"loss": [], "val_loss": []}
        for epoch in range(initial_epoch, epochs):
            t_epoch = time.time()
            if world > 1:
                # sharded BEFORE the callbacks run, so that AugmentOnTheFly augments this rank's samples only
                self.epoch_indices = parallel.shard_indices(N, self._epochs_seen, rank, world, seed=self.seed,
                                                            batch_size=batch_size)
                if not shuffle:
                    self.epoch_indices = np.sort(self.epoch_indices)
            for cb in callbacks:
                cb.on_epoch_begin(epoch, {})
            Xd = self._device_frames(X)
            if Xd.dtype == torch.uint8:
                # grey levels: the same scaling predict() applies to uint8 frames (utils.py:340-342) -- a container must
                # not change what the network sees
                raise TypeError("fit(): uint8 frames are grey levels; pass float32 frames in [-1, 1] "
                                "(utils.build_X) -- predict() scales uint8 frames on the device, fit() trains in place on "
                                "the caller's float array as the reference does")
            if Xd.dtype != torch.float32 or not Xd.is_contiguous():
                Xd = Xd.float().contiguous()
            if world > 1:
                index = self.epoch_indices
            else:
                index = np.arange(N)
                if shuffle:
                    np.random.shuffle(index)           # Keras shuffles with numpy's global RNG
            self._epochs_seen += 1
            nb = len(index) // batch_size
            loss_sum = torch.zeros(2, device=eng.dev, dtype=torch.float64)
            for b in range(nb):
                for cb in callbacks:
                    cb.on_batch_begin(b, {})
                idx = upload("idx", index[b * batch_size:(b + 1) * batch_size])
                L.gather_rows(Xd, idx, eng.x_in)
                L.gather_rows(Yd, idx, eng.y_true)
                out = eng.train_step(None, None, float(self.optimizer.lr), reducer=reducer)
                loss_sum += out[5:7].double()
                if chatty and (b % max(1, nb // 20) == 0 or b == nb - 1):
                    print("\rEpoch %d/%d  batch %d/%d" % (epoch + 1, epochs, b + 1, nb), end="", flush=True)
            ls = (loss_sum / max(nb, 1)).cpu().numpy()
            data_loss = parallel.all_reduce_scalar_mean(float(ls[0]))    # mean over the replicas' shards
            logs = {"loss": data_loss + float(ls[1])}         # Keras reports data loss + regularisation
            if validation_data is not None and rank == 0:
                Xv, Yv = validation_data[0], validation_data[1]
                # Keras evaluates the validation loss at the END of the epoch with the weights as they then are: the
                # regularisation term of val_loss is the l2 penalty of the current kernels (round 4: it used to be the
                # epoch MEAN of the training steps' penalties, which made the logged penalty lag half an epoch behind the
                # reference's -- 0.2415 logged against 0.2171 actual after epoch 1, reference log 0.2204)
                l2_now = _l2_coef() * float(eng.theta[:eng.l2_n].double().square().sum().item())
                logs["val_loss"] = custom_loss(Yv, self.predict(Xv, batch_size=batch_size)) + l2_now
            history["loss"].append(logs["loss"])
            history["val_loss"].append(logs.get("val_loss"))
            if chatty:
                n_img = nb * batch_size * world
                print("\r%d/%d - %ds - loss: %.4e%s" % (n_img, n_img, time.time() - t_epoch, logs["loss"],
                                                          (" - val_loss: %.4e" % logs["val_loss"]) if "val_loss" in logs else ""))
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end({})
        return history


# ----------------------------------------------------------------------------- constructors
def create_model_functional(X, Y0size=576, freeze_fac=0.75, quick_setup=False):
    """Stem (3x3 convs + average pooling + BN/LeakyReLU residual block, Dropout 0.1) -> cf.basemodel
    backbone -> Flatten -> Dense(Y0size, 'FinalOutput'); l2(1e-4) on the ten regularised kernels
    (models.py:302-424)."""
    print("Using functional API model, cf.basemodel =", cf.basemodel)
    print("X[0].shape = ", X[0].shape)
    model = Model(X[0].shape, Y0size=Y0size, freeze_fac=freeze_fac)
    n_layers = len(keras_layer_table(model.basemodel))
    print("Freezing ", int(n_layers * freeze_fac), "/", n_layers, " layers of base_model")
    total, trainable = model.count_params()
    frozen = 0
    if model._mask is not None:
        frozen = int((model._mask == 0).sum().item())
    print('create_model_functional: Total params: {:,}'.format(total))
    print('create_model_functional: Trainable params: {:,}'.format(trainable - frozen))
    print('create_model_functional: Non-trainable params: {:,}'.format(total - trainable + frozen))
    return model


def build_model(X, Y0size=576, freeze_fac=0.0, quick_setup=False):
    """Alias named by the build's north star; the reference's symbol is create_model_functional."""
    return create_model_functional(X, Y0size=Y0size, freeze_fac=freeze_fac, quick_setup=quick_setup)


def create_model_simple(X, Y0size=576, freeze_fac=0.75):
    raise NotImplementedError("model_type 'simple' (NASNetMobile with imagenet weights, models.py:428-458) needs a "
                              "network download and is 'not recommended' by the reference; out of scope")


def setup_model(X, Y0size=576, try_checkpoint=True, no_cp_fatal=False, weights_file='weights.hdf5', freeze_fac=0.75,
                parallel=False, quick_setup=False):
    """Build the model, optionally initialise it from `weights_file`, attach Adam(lr=1e-5) + custom_loss.
    Returns (model, serial_model) -- the same object twice: data parallelism here is one process per
    GPU (spnet_amd/parallel.py), not an in-graph wrapper."""
    print("Initializing blank model: Y0size =", Y0size)
    if parallel:            # pick this rank's GPU and join the process group before anything is allocated
        multi_gpu.make_parallel(None)
    if cf.model_type == 'simple':
        model = create_model_simple(X, Y0size=Y0size, freeze_fac=freeze_fac)
    else:
        model = create_model_functional(X, Y0size=Y0size, freeze_fac=freeze_fac, quick_setup=quick_setup)
    if try_checkpoint:
        if isfile(weights_file):
            print('Weights file detected. Loading from', weights_file)
            model.load_weights(weights_file)
        elif no_cp_fatal:
            raise Exception("*** No weights file detected; can't do anything.  Aborting.")
        else:
            print('    No weights file detected, so starting from scratch.')
    model.optimizer = _Optimizer(lr=0.00001)
    print("Compiling the model")
    return model, model


def load_model(path, custom_objects=None):
    """Rebuild a model from a 'whole model' file written by Model.save()."""
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
    shape = tuple(json.loads(meta.get("input_shape", "[331, 331, 1]")))
    saved_type, old_type, old_base = meta.get("model_type"), cf.model_type, cf.basemodel
    try:                                   # head variant and backbone are part of the saved model, not of the caller's config
        if saved_type in ("compound", "monolithic", "big"):
            cf.model_type = saved_type
        if meta.get("basemodel") in ("Xception", "MobileNet", "InceptionResNetV2"):
            cf.basemodel = meta["basemodel"]
        model = Model(shape, Y0size=int(meta.get("Y0size", 576)), freeze_fac=0.0)
    finally:
        cf.model_type, cf.basemodel = old_type, old_base
    model.load_weights(path)
    return model


def unfreeze_model(model, X, Y, parallel=False):
    """Fresh, fully trainable model with the same weights and a new optimizer (models.py:510-552)."""
    print("Unfreezing Model: make a new identical model, then copy the layer weights.")
    new_model = create_model_functional(X, Y[0].size, freeze_fac=0)
    new_model.set_weights(multi_gpu.get_serial_part(model, parallel=parallel).get_weights())
    new_model.optimizer = _Optimizer(lr=0.00001)
    print("  ...finished un-freezing model")
    return new_model
