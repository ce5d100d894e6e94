#!/usr/bin/env python3
"""Dev tool (GPU box): per-shape in-step time of the pointwise forward (+ statistics) and data-gradient GEMMs, exact fp32
kernels against the bf16x3 kernel (Engine.pointwise toggled on one engine), HIP events inside the benchmark train step (and the predict plan).
Decides which shapes go to which kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spnet_amd.engine import Engine

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
B = 32 if mode == "train" else 128
# (one engine per arithmetic: a bf16x3 plan keeps its GEMM operands as planes, the mode is fixed at construction;
# x3_min_tiles=0: every eligible shape on the bf16x3 kernels, also those the product plan keeps exact)
engs = {"f32": Engine(384, 512, B, device="cuda:0", seed=0, train=(mode == "train"), pointwise="f32"),
        "bf16x3": Engine(384, 512, B, device="cuda:0", seed=0, train=(mode == "train"), x3_min_tiles=0)}
X = torch.rand(B, 384, 512, 1, device="cuda") * 2 - 1
Y = torch.rand(B, 576, device="cuda")
res = {}
for name, alt in (("exact", "f32"), ("x3", "bf16x3")):
    eng = engs[alt]
    eng.wgrad_stream = None

    def step():
        if mode == "train":
            eng.train_step(X, Y, 1e-6)
        else:
            eng.forward(X, training=False)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    tg = bench.time_families(eng, step, 8, torch.cuda.synchronize).tagged()
    for tag, (n, ms, w) in tg.items():
        if not (isinstance(tag, tuple) and len(tag) == 4):
            continue
        form = str(tag[0]).replace("x3p ", "").replace("x3w ", "").replace("x3 ", "")
        if form not in ("aB", "aB+stats", "ab") and not form.startswith("AB"):
            continue
        res.setdefault((form,) + tuple(tag[1:]), {})[name] = (n / 8, 1e3 * ms / n)
print("%-10s %8s %6s %6s | %5s | %9s %9s | %s" % ("form", "M", "N", "K", "n", "exact us", "x3 us", "x3/exact"))
tot = {"exact": 0.0, "x3": 0.0, "best": 0.0}
for key in sorted(res, key=lambda k: -res[k].get("exact", (0, 0))[1] * res[k].get("exact", (0, 0))[0]):
    r = res[key]
    if "exact" in r and "x3" in r:
        n, e = r["exact"]
        x = r["x3"][1]
        tot["exact"] += n * e; tot["x3"] += n * x; tot["best"] += n * min(e, x)
        print("%-10s %8d %6d %6d | %5.1f | %9.1f %9.1f | %.2f" % (key + (n, e, x, x / e)))
print("per step: exact %.3f ms, bf16x3 %.3f ms, best of both per shape %.3f ms" % (tot["exact"] / 1e3, tot["x3"] / 1e3, tot["best"] / 1e3))
