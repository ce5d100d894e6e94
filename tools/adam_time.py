#!/usr/bin/env python3
"""Dev tool (GPU box): the optimizer kernel alone at the benchmark's parameter count (50.35 M; 28 bytes per parameter)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_350_000
n -= n % 4
p, g, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(4))
v.abs_()
sq, out = torch.empty(2048, device="cuda"), torch.empty(1, device="cuda")
spoil = torch.empty(160 << 20, device="cuda")          # 640 MB written between launches, as backward does in the step


def one():
    L.spnet_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, n - 1024, 1e-5, 0.9, 0.999, 1e-7, 1e-4, 1.0,
                      None, sq.data_ptr(), out.data_ptr(), None, st())


for label, between in (("back to back", lambda: None), ("caches spoiled between launches", lambda: spoil.zero_())):
    for _ in range(3):
        between(); one()
    ts = []
    for _ in range(10):
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); one(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print("adam %d params, %s: %.1f us, %.2f TB/s" % (n, label, t, 28.0 * n / t / 1e6), flush=True)
