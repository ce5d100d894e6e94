#!/usr/bin/env python3
"""Dev tool (GPU box): how much of the Inception-ResNet-v2 train step (batch 16, 512x384) is the dependent chain of small
elementwise launches?  TIMING ONLY: one kernel family at a time is knocked out (its C-ABI entry replaced by a no-op, so the
results are garbage) and the step is timed again -- the upper bound of what fusing that family into its neighbours could
buy, before anything is built."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spnet_amd import _lib as L
from spnet_amd import engine as E

H, W, B = 384, 512, 16
eng = E.Engine(H, W, B, device="cuda:0", seed=0, backbone="InceptionResNetV2")
X = torch.rand(B, H, W, 1, device="cuda") * 2 - 1
Y = torch.rand(B, 576, device="cuda")


def run(steps=30):
    for _ in range(5):
        eng.train_step(X, Y, 1e-6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(X, Y, 1e-6)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


base = run()
print("baseline                          %.3f ms per step  (%.0f images/s)" % (base, 1e3 * B / base), flush=True)
noop = lambda *a, **k: None
groups = {
    "bn backward apply (148/step)": ["spnet_bn_bwd_from_partials"],
    "bn finalize+apply fwd (145/step)": ["spnet_bn_finalize_apply_ld"],
    "masked gradient + bn sums (136/step)": ["spnet_patches_bwd_bnsums_ld", "spnet_copy_cols_bnsums_ld"],
    "resadd fwd + bwd (80/step)": ["spnet_resadd", "spnet_resadd_bwd"],
    "all four": ["spnet_bn_bwd_from_partials", "spnet_bn_finalize_apply_ld", "spnet_patches_bwd_bnsums_ld", "spnet_copy_cols_bnsums_ld",
                 "spnet_resadd", "spnet_resadd_bwd"],
}
for name, fns in groups.items():
    saved = {f: getattr(L, f) for f in fns}
    for f in fns:
        setattr(L, f, noop)
    ms = run()
    for f, v in saved.items():
        setattr(L, f, v)
    print("without %-38s %.3f ms per step  (%+.3f ms, %.0f images/s)" % (name, ms, ms - base, 1e3 * B / ms), flush=True)
