#!/usr/bin/env python3
"""Host-side cost of enqueueing ONE train step on an idle GPU (eager launches vs hipGraph replay)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spnet_amd.engine import Engine
eng = Engine(384, 512, 32, device="cuda:0", seed=0)
X = torch.rand(32, 384, 512, 1, device="cuda") * 2 - 1
Y = torch.rand(32, 576, device="cuda")
for mode in ("eager", "graph", "eager-noverlap"):
    eng.use_graph = (mode == "graph")
    if mode == "eager-noverlap":
        eng.wgrad_stream = None
    for _ in range(3):
        eng.train_step(X, Y, 1e-5)
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_step(None, None, 1e-5)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    print(mode, "enqueue ms / total ms:", ["%.2f/%.2f" % r for r in res], flush=True)
