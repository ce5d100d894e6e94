#!/usr/bin/env python3
"""Dev tool (GPU box): the benchmark train step with the data-gradient chain on a HIGH-priority stream (the weight-gradient
stream stays at normal priority; this platform offers priorities -1 and 0 only) against the default, alternating in one
process -- or, with a third argument `hi` / `default`, ONE engine in ONE mode (alternate the processes on one box: two
engines + a high-priority stream in one process end up sharing hardware queues and the default-stream engine measures 15.7 ms).
usage: ab_stream_priority.py [rounds] [steps] [hi|default]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd.engine import Engine

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
X = torch.rand(32, 384, 512, 1, device="cuda") * 2 - 1
Y = torch.rand(32, 576, device="cuda")
hi = torch.cuda.Stream(priority=-1) if (len(sys.argv) <= 3 or sys.argv[3] == "hi") else torch.cuda.current_stream()
print("priorities: high stream %d, default %d" % (hi.priority, torch.cuda.current_stream().priority))
only = sys.argv[3] if len(sys.argv) > 3 else None
engs = {}
if only in (None, "hi"):
    engs["main chain on a high-priority stream"] = (Engine(384, 512, 32, device="cuda:0", seed=0), hi)
if only in (None, "default"):
    engs["default stream"] = (Engine(384, 512, 32, device="cuda:0", seed=0), None)
torch.cuda.synchronize()


def run(e, s, n):
    if s is None:
        for _ in range(n):
            e.train_step(X, Y, 1e-6)
    else:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n):
                e.train_step(X, Y, 1e-6)


res = {k: [] for k in engs}
for k, (e, s) in engs.items():
    run(e, s, 5)
torch.cuda.synchronize()
for r in range(rounds):
    for k, (e, s) in engs.items():
        run(e, s, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(e, s, steps)
        torch.cuda.synchronize()
        res[k].append(1e3 * (time.perf_counter() - t0) / steps)
for k, r in res.items():
    print("%s: ms per step %s  (median %.3f)" % (k, " ".join("%.3f" % t for t in r), sorted(r)[len(r) // 2]))
