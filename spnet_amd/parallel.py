"""Data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.

Replaces the reference's (disabled) TF1 tower replication, spnet/multi_gpu.py:35-88: there the batch
is sliced per GPU inside one graph and outputs are concatenated on the CPU; here every rank owns a
full replica and a shard of the minibatch, BatchNorm statistics stay per replica (the tower
semantics), and the only exchange step is the all-reduce(sum) of the flat fp32 gradient, bucketed:

  * the Dense-head kernel gradient (73 % of the bytes) is produced FIRST in backward; its ~32 MB
    pieces are launched right then and run on RCCL's stream underneath the whole backbone backward
  * the rest follows in reverse layer order, ~32 MB per bucket, each launched as soon as the node that
    completes it has been back-propagated (Engine.grad_buckets)
  * the small remainder (stem / entry-flow parameters, l2 kernels, depthwise kernels) when backward ends
  * the 1/world_size averaging is folded into the fused Adam kernel (grad_scale), no extra pass

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of S bytes moves 2(N-1)/N * S per
GPU and is bound by one link's rate per ring, so buckets are kept LARGE (tens of MB: latency-free, RCCL can
run several rings) and few, instead of the many small per-layer messages an NVSwitch-tuned framework sends.

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def local_device():
    """The HIP device of this process: cuda:LOCAL_RANK (modulo the visible devices, so several gloo ranks can
    rehearse on one GPU)."""
    _, local_rank, _ = env_world()
    return torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))


def init_distributed(backend=None, force=False):
    """Select this rank's GPU and initialise the default process group from the torchrun environment.  Must run
    BEFORE any model, engine or callback allocates device memory.  No-op for world size 1 unless `force` (a
    one-rank RCCL group, used to exercise the collective path on a single GPU)."""
    rank, local_rank, world = env_world()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device())
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # SPNET_DIST_BACKEND=gloo lets several ranks rehearse on ONE GPU (RCCL wants one device per rank)
            backend = os.environ.get("SPNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            # a failed RCCL init / collective then says why in this rank's stderr (bench.py relays the failing rank's tail).
            # The channel count is left at RCCL's default: what a collective costs the step underneath it depends on how
            # LONG it holds CUs, hardly on how many (tools/channel_hog.py) -- fewer channels = a longer
            # collective = more of the backward disturbed.
            os.environ.setdefault("NCCL_DEBUG", "WARN")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradReducer:
    """Sums a flat gradient buffer across ranks, bucket by bucket, while backward is still running.

    buckets: [(lo, hi, trigger)] float ranges in launch order; a bucket is launched by on_node_done(trigger).
    tail:    [(lo, hi)] ranges launched by finish().
    Legacy form GradReducer(flat, (lo, hi)): one 'head' bucket launched by launch_head(), the two pieces
    around it by finish().

    A collective is enqueued from the weight-gradient stream (after it has joined the data-gradient stream), so
    that RCCL's stream waits for BOTH producers of the bucket without stalling either of them; finish() makes the
    caller's stream wait for every piece."""

    def __init__(self, flat_grad, buckets, tail=None, group=None, force=False):
        self.g = flat_grad
        if len(buckets) == 2 and all(isinstance(v, (int, np.integer)) for v in buckets):
            lo, hi = int(buckets[0]), int(buckets[1])
            buckets, tail = [(lo, hi, "head")], [(0, lo), (hi, flat_grad.numel())]
        self.buckets = [(int(lo), int(hi), trig) for lo, hi, trig in buckets]
        self.tail = [(int(lo), int(hi)) for lo, hi in (tail or [])]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.side_stream = None
        self.pending = []
        self.launched = 0           # collectives launched in the current step (tests / diagnostics)
        covered = sorted([(lo, hi) for lo, hi, _ in self.buckets] + self.tail)
        pos = 0
        for lo, hi in covered:
            if lo != pos:
                raise ValueError("gradient buckets must tile the buffer: gap or overlap at %d (next %d)" % (pos, lo))
            pos = hi
        if pos != flat_grad.numel():
            raise ValueError("gradient buckets cover %d of %d elements" % (pos, flat_grad.numel()))

    def _launch(self, lo, hi):
        if hi <= lo or not self.active:
            return
        self.launched += 1
        piece = self.g[lo:hi]
        side = self.side_stream
        if side is not None and piece.is_cuda:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append(w)

    def on_node_done(self, node):
        """Engine.backward() calls this after each node's launches are enqueued."""
        for lo, hi, trig in self.buckets:
            if trig is node:
                self._launch(lo, hi)

    def launch_head(self):
        self.on_node_done("head")

    def finish(self):
        """Reduce the tail, wait for all pieces (stream-level wait on GPUs); returns the scale (1/world) the
        optimizer must apply to the summed gradient."""
        for lo, hi in self.tail:
            self._launch(lo, hi)
        for w in self.pending:
            w.wait()
        self.pending = []
        self.launched = 0
        return 1.0 / self.world


def shard_indices(n_total, epoch, rank, world, seed=1, batch_size=None):
    """This rank's sample indices for one epoch: every rank draws the SAME permutation (seed, epoch)
    and takes every world-th element, so the union over ranks is the whole epoch and the result for a
    sample does not depend on which rank processes it.  Truncated to a multiple of batch_size."""
    perm = np.random.RandomState((seed * 1000003 + epoch) & 0x7FFFFFFF).permutation(n_total)
    mine = perm[rank::world]
    per_rank = n_total // world
    mine = mine[:per_rank]
    if batch_size:
        mine = mine[:len(mine) // batch_size * batch_size]
    return mine


def sample_seed(seed, epoch, index):
    """Seed of the per-sample augmentation stream under data parallelism: a function of (seed, epoch, sample)
    only, so the augmented frame does not depend on the world size or on which rank draws it."""
    return (int(seed) * 1000003 + int(epoch) * 7919 + int(index) * 104729 + 12345) & 0x7FFFFFFF


def all_reduce_scalar_mean(value, device=None):
    """Mean of a python float over ranks (logging only)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = local_device() if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()
