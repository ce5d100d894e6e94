"""Helper of tests/test_dp_gpu.py: train a few steps on the SAME batch on every rank and save the weights.
Launched by torch.distributed.run (gloo backend on one GPU, as bench.py's rehearsal mode) or stand-alone.
argv: out_path steps [mode]   mode 'rccl1' = ONE rank with a forced one-rank RCCL ("nccl") group: every bucket of the
gradient all-reduce really goes through RCCL and its stream hand-off."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spnet_amd import parallel  # noqa: E402
from spnet_amd.engine import Engine  # noqa: E402

out_path, steps = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else ""
rank, local_rank, world = parallel.init_distributed(backend="nccl" if mode == "rccl1" else None, force=(mode == "rccl1"))
H, W, B = 96, 128, 4
eng = Engine(H, W, B, device="cuda:0", seed=3, rank=rank)
eng.drop_seed = 12345       # the same dropout masks on every rank: equal batches must give equal gradients
rs = np.random.RandomState(0)
X = torch.tensor(rs.rand(B, H, W, 1) * 2 - 1, dtype=torch.float32).cuda()
Y = torch.tensor(rs.rand(B, 576), dtype=torch.float32)
Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
Y = Y.cuda()
# small buckets so that this small network is cut into many pieces (head pieces, reverse-layer buckets, tail)
reducer = eng.make_reducer(force=(mode == "rccl1"), bucket_bytes=2 << 20) if (world > 1 or mode == "rccl1") else None
nlaunch = 0
for s in range(steps):
    if reducer is not None:
        count = []
        fin = reducer.finish
        reducer.finish = lambda: (count.append(reducer.launched + len(reducer.tail)), fin())[1]
    out = eng.train_step(X, Y, 1e-3, reducer=reducer)
    if reducer is not None:
        reducer.finish = fin
        nlaunch = count[0]
torch.cuda.synchronize()
if reducer is not None and rank == 0:
    nb, nt = len(reducer.buckets), len(reducer.tail)
    print("collectives per step: %d (buckets %d + tail %d), backend %s" % (nlaunch, nb, nt, torch.distributed.get_backend()))
    assert nlaunch == nb + nt and nb >= 4, (nlaunch, nb, nt)
if rank == 0:
    sd = eng.state_dict()
    np.savez(out_path, loss=out.cpu().numpy(), **{k.replace("/", "__"): v.numpy() for k, v in sd.items()})
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
