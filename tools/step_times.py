#!/usr/bin/env python3
"""Dev tool (GPU box): what one benchmark step costs beyond Engine.train_step on a fixed batch.
modes: fixed (one resident batch), pool (a different batch of device-generated frames every step, no augmentation),
       augment (bench.py's step: DeviceAugmenter + label gather)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spnet_amd.engine import Engine
from spnet_amd.augmentation import DeviceAugmenter
from spnet_amd import fake_espi as F
from spnet_amd import _lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
eng = Engine(384, 512, 32, device="cuda:0", seed=0, train=True)
POOL = 1024
X_pool, _ = F.generate_device(POOL, seed=1, device="cuda:0")
Y_pool = torch.rand(POOL, 576, device="cuda")
Xf = torch.rand(32, 384, 512, 1, device="cuda") * 2 - 1
aug = DeviceAugmenter(X_pool)
upload = L.AsyncUploader(torch.device("cuda:0"))
order = np.random.RandomState(7).permutation(POOL)


def run(mode, noise_labels=False):
    it = [0]

    def step():
        i = it[0]
        it[0] += 1
        idx = order[(np.arange(32) + i * 32) % POOL]
        if mode == "fixed":
            eng.x_in.copy_(Xf)
        elif mode == "fixed_frames":
            eng.x_in.copy_(X_pool[:32])
        elif mode == "pool":
            torch.index_select(X_pool, 0, upload("idx64", idx.astype(np.int64)), out=eng.x_in)
        else:
            aug.augment(idx, eng.x_in)
        torch.index_select(Y_pool, 0, upload("idx", idx), out=eng.y_true)
        eng.train_step(None, None, 1e-5)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


for mode in ("fixed", "fixed_frames", "pool", "augment", "fixed", "augment"):
    print("%-13s %.3f ms per step" % (mode, run(mode)), flush=True)

# per-step GPU time of the very first steps of the bench.py step (events), no warm-up
if len(sys.argv) > 2 and sys.argv[2] == "first":
    eng2 = Engine(384, 512, 32, device="cuda:0", seed=1, train=True)
    aug2 = DeviceAugmenter(X_pool)
    up2 = L.AsyncUploader(torch.device("cuda:0"))
    n = 24
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    th = []
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        t0 = time.perf_counter()
        idx = order[(np.arange(32) + i * 32) % POOL]
        aug2.augment(idx, eng2.x_in)
        torch.index_select(Y_pool, 0, up2("idx", idx), out=eng2.y_true)
        eng2.train_step(None, None, 1e-5)
        ev[i + 1].record()
        th.append(1e3 * (time.perf_counter() - t0))
        if i == 2:
            torch.cuda.synchronize()          # bench.py's fence between warm-up and the timed region
    torch.cuda.synchronize()
    print("fresh engine, bench step, sync after step 3: GPU ms per step:", " ".join("%.2f" % ev[i].elapsed_time(ev[i + 1]) for i in range(n)))
    print("host ms per step:", " ".join("%.1f" % v for v in th))
