"""Data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI.

Replaces the reference's (disabled) TF1 tower replication, spnet/multi_gpu.py:35-88: there the batch
is sliced per GPU inside one graph and outputs are concatenated on the CPU; here every rank owns a
full replica and a shard of the minibatch, BatchNorm statistics stay per replica (the tower
semantics), and the only exchange step is one all-reduce(sum) of the flat fp32 gradient per step:

  * the Dense-head kernel gradient (73 % of the bytes) is produced FIRST in backward, so its
    all-reduce is launched right then and runs on RCCL's stream underneath the whole backbone backward
  * the remaining ~20 M gradient values follow in two contiguous pieces when backward ends
  * the 1/world_size averaging is folded into the fused Adam kernel (grad_scale), no extra pass

Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # SPNET_DIST_BACKEND=gloo lets several ranks rehearse on ONE GPU (RCCL wants one device per rank)
            backend = os.environ.get("SPNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradReducer:
    """Sum a flat gradient buffer across ranks in three contiguous pieces, head piece first."""

    def __init__(self, flat_grad, head_range, group=None):
        self.g = flat_grad
        self.lo, self.hi = head_range
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pending = []

    def _launch(self, lo, hi):
        if hi > lo and self.world > 1:
            self.pending.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch_head(self):
        """Call as soon as the head gradient's kernels are enqueued."""
        self._launch(self.lo, self.hi)

    def finish(self):
        """Reduce everything else, wait for all pieces (stream-level wait on GPUs); returns the scale
        (1/world) the optimizer must apply to the summed gradient."""
        self._launch(0, self.lo)
        self._launch(self.hi, self.g.numel())
        for w in self.pending:
            w.wait()
        self.pending = []
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


def all_reduce_scalar_mean(value, device=None):
    """Mean of a python float over ranks (logging only)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()
