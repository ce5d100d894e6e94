#!/usr/bin/env python3
"""Dev tool (GPU box): the benchmark train step (Xception, batch 32, 384x512, fixed frames) on two engines that differ in one
constructor flag, alternating in ONE process (cdna_hip_programming.md rule 24).
usage: ab_engine_flags.py flag=valueA,valueB [rounds] [steps]     e.g.  fuse_dw_bwd=True,False
       ab_engine_flags.py attr:name=valueA,valueB ...             sets an attribute after construction (attr:use_graph=True,False)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd.engine import Engine

flag, vals = sys.argv[1].split("=")
vals = [eval(v) for v in vals.split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
if flag.startswith("attr:"):
    engs = [Engine(384, 512, 32, device="cuda:0", seed=0) for v in vals]
    for e, v in zip(engs, vals):
        setattr(e, flag[5:], v)
else:
    engs = [Engine(384, 512, 32, device="cuda:0", seed=0, **{flag: v}) for v in vals]
X = torch.rand(32, 384, 512, 1, device="cuda") * 2 - 1
Y = torch.rand(32, 576, device="cuda")
res = [[] for _ in vals]
for e in engs:
    for _ in range(5):
        e.train_step(X, Y, 1e-6)
torch.cuda.synchronize()
for r in range(rounds):
    for i, e in enumerate(engs):
        for _ in range(3):
            e.train_step(X, Y, 1e-6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            e.train_step(X, Y, 1e-6)
        torch.cuda.synchronize()
        res[i].append(1e3 * (time.perf_counter() - t0) / steps)
for v, r in zip(vals, res):
    print("%s=%r: ms per step %s  (median %.3f)" % (flag, v, " ".join("%.3f" % t for t in r), sorted(r)[len(r) // 2]))
