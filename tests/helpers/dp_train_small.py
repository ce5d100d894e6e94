"""Helper of tests/test_dp_gpu.py: train a few steps on the SAME batch on every rank and save the weights.
Launched by torch.distributed.run (gloo backend on one GPU, as bench.py's rehearsal mode) or stand-alone."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spnet_amd import parallel  # noqa: E402
from spnet_amd.engine import Engine  # noqa: E402

out_path, steps = sys.argv[1], int(sys.argv[2])
rank, local_rank, world = parallel.init_distributed()
H, W, B = 96, 128, 4
eng = Engine(H, W, B, device="cuda:0", seed=3)
rs = np.random.RandomState(0)
X = torch.tensor(rs.rand(B, H, W, 1) * 2 - 1, dtype=torch.float32).cuda()
Y = torch.tensor(rs.rand(B, 576), dtype=torch.float32)
Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
Y = Y.cuda()
reducer = parallel.GradReducer(eng.grad, eng.head_grad_range()) if world > 1 else None
for s in range(steps):
    out = eng.train_step(X, Y, 1e-3, reducer=reducer)
torch.cuda.synchronize()
if rank == 0:
    sd = eng.state_dict()
    np.savez(out_path, loss=out.cpu().numpy(), **{k.replace("/", "__"): v.numpy() for k, v in sd.items()})
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
