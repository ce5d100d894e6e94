#!/usr/bin/env python3
"""Dev tool: worst per-tensor gradient mismatch (engine vs torch-CPU oracle) for one geometry and several seeds
-- separates a systematic error (same tensors every seed) from ReLU / max-pool ties flipped by rounding."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import torch_ref as T
from tests.test_engine_gpu import dropout_mask
from spnet_amd.engine import Engine
H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for seed in range(4):
    eng = Engine(H, W, B, device="cuda:0", seed=seed)
    P = T.init_params(H, W, seed=seed + 10)
    if len(sys.argv) > 4:      # non-trivial BatchNorm parameters / moving statistics
        g = torch.Generator().manual_seed(seed)
        for k in P:
            if k.endswith("/gamma"):
                P[k] = 0.5 + torch.rand(P[k].shape, generator=g)
            elif k.endswith("/beta") or k.endswith("/moving_mean") or k.endswith("/bias"):
                P[k] = 0.2 * torch.randn(P[k].shape, generator=g)
            elif k.endswith("/moving_variance"):
                P[k] = 0.5 + torch.rand(P[k].shape, generator=g)
    rs = np.random.RandomState(seed)
    X = torch.tensor(rs.rand(B, H, W, 1) * 2 - 1, dtype=torch.float32)
    Y = torch.tensor(rs.rand(B, 576), dtype=torch.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    eng.load_state_dict(P)
    eng.set_drop_seed(5)
    mask = torch.tensor(dropout_mask(B * (H // 2) * (W // 2) * 3, 5).reshape(B, H // 2, W // 2, 3))
    tr = T.Trainer({k: v.clone() for k, v in P.items()})
    data, total, grads, yp = tr.grads(X, Y, drop_mask=mask, include_l2=False)
    eng.forward(X.cuda(), training=True); eng.loss(Y.cuda()); eng.backward(); torch.cuda.synchronize()
    gd = eng.grad_dict()
    errs = sorted(((float(np.abs(gd[k].numpy() - g.numpy()).max()) / max(float(np.abs(g.numpy()).max()), 1e-12), k) for k, g in grads.items()), reverse=True)
    print("seed %d worst:" % seed, ["%s %.2e" % (k, e) for e, k in errs[:4]], flush=True)
