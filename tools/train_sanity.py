#!/usr/bin/env python3
"""Dev tool: a few hundred full-size train steps (batch 32, 384x512, 256 fake-ESPI frames, device augmentation,
1-cycle LR of a short run) -- the loss must fall steadily; prints data loss / l2 penalty every 50 steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from spnet_amd import fake_espi as F
from spnet_amd.augmentation import DeviceAugmenter
from spnet_amd.callbacks import get_1cycle_schedule
from spnet_amd.engine import Engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
X_u8, labels = F.generate(256, seed=1, workers=8)
Y = torch.from_numpy(bench.labels_to_Y(labels)).to(dev)
X = torch.from_numpy(F.to_network_input(X_u8)).to(dev)
eng = Engine(384, 512, 32, device="cuda:0", seed=0)
aug = DeviceAugmenter(X)
lrs = get_1cycle_schedule(lr_max=2e-4, n_data_points=256, epochs=max(1, steps * 32 // 256), batch_size=32)
np.random.seed(1)
t0 = time.time()
hist = []
for i in range(steps):
    idx = np.random.permutation(256)[:32]
    aug.augment(idx, eng.x_in)
    torch.index_select(Y, 0, torch.from_numpy(idx).to(dev), out=eng.y_true)
    out = eng.train_step(None, None, float(lrs[min(i, len(lrs) - 1)]))
    if i % 50 == 0 or i == steps - 1:
        o = out.cpu().numpy()
        hist.append(float(o[5]))
        print("step %4d  data loss %.5f  l2 %.5f  lr %.2e" % (i, o[5], o[6], lrs[min(i, len(lrs) - 1)]), flush=True)
print("%.1f s; loss %.5f -> %.5f" % (time.time() - t0, hist[0], hist[-1]))
assert hist[-1] < 0.5 * hist[0], "training does not converge"
