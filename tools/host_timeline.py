#!/usr/bin/env python3
"""Dev tool: where the HOST spends a train step while the GPU is busy (draw / upload+augment launches /
train_step enqueue), and how far it runs ahead of the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spnet_amd.engine import Engine
from spnet_amd.augmentation import DeviceAugmenter
B, H, W = 32, 384, 512
dev = torch.device("cuda:0")
X_pool = torch.rand(256, H, W, 1, device=dev) * 2 - 1
eng = Engine(H, W, B, device="cuda:0", seed=0)
aug = DeviceAugmenter(X_pool)
eng.y_true.copy_(torch.rand(B, 576, device=dev))
np.random.seed(0)
for i in range(3):
    aug.augment(np.arange(B), eng.x_in); eng.train_step(None, None, 1e-5)
torch.cuda.synchronize()
rec = []
t_start = time.perf_counter()
for i in range(12):
    t0 = time.perf_counter()
    p = aug.draw(np.arange(B) + i)
    t1 = time.perf_counter()
    aug.apply(p, eng.x_in)
    t2 = time.perf_counter()
    eng.train_step(None, None, 1e-5)
    t3 = time.perf_counter()
    rec.append((t0 - t_start, t1 - t0, t2 - t1, t3 - t2))
torch.cuda.synchronize()
t_end = time.perf_counter()
for r in rec:
    print("step starts at %7.2f ms: draw %5.2f ms, upload+augment %5.2f ms, train_step enqueue %5.2f ms" % tuple(1e3 * v for v in r))
print("host finished enqueueing at %.2f ms, GPU finished at %.2f ms (%.2f ms per step)" % (1e3 * (rec[-1][0] + sum(rec[-1][1:])), 1e3 * (t_end - t_start), 1e3 * (t_end - t_start) / 12))
