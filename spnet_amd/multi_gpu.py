"""Multi-GPU surface of the reference's spnet/multi_gpu.py (:15-88), re-based on one process per GPU.

The reference replicates the Keras graph once per GPU inside ONE process (TF1 towers, disabled by
default at train_spnet.py:55).  Here data parallelism is `torchrun` + RCCL: every process builds the
same model, `make_parallel` joins the process group, and Model.fit() shards each epoch's permutation
by rank and all-reduces the flat gradient (spnet_amd/parallel.py).  The functions keep the reference's
names so train_spnet.py reads the same.
"""
from . import parallel


def get_available_gpus():
    import torch
    return ["/device:GPU:%d" % i for i in range(torch.cuda.device_count())]


def make_parallel(model):
    """Select this rank's GPU and join the torchrun process group (no-op when WORLD_SIZE is 1); the model
    object is returned unchanged.  Call it BEFORE building the model (setup_model does)."""
    rank, local_rank, world = parallel.init_distributed()
    if world > 1:
        print("make_parallel: rank %d of %d, RCCL gradient all-reduce over xGMI" % (rank, world))
    return model


def get_serial_part(model, parallel=True):
    """The reference unwraps its tower model here; replicas are already 'serial' models."""
    return model
