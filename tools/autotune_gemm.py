#!/usr/bin/env python3
"""Dev tool (GPU box): per-shape tile choice for the GEMMs of the benchmark train step, measured IN the step.
For every distinct plain GEMM launch (operand form, BatchNorm statistics or not, M, N, K) each tile id is forced in turn
(engine.TILE_PROBE) for a few whole train steps and that shape's launches are timed with HIP events; a tile is recorded
only where it beats the library's own choice by more than 3 % in BOTH of two alternating passes.
writes spnet_amd/gemm_tiles.json      usage: autotune_gemm.py [steps per probe]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPNET_OVERLAP_WGRAD"] = "0"
os.environ["SPNET_GEMM_TILES"] = "0"
import torch
from spnet_amd import engine as E

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 32
eng = E.Engine(384, 512, B, device="cuda:0", seed=0, train=True)
X = torch.rand(B, 384, 512, 1, device="cuda") * 2 - 1
Y = torch.rand(B, 576, device="cuda")


def measure():
    """{key: us per launch} over `steps` train steps."""
    t = E.KernelTimer()
    eng.prof = t
    for _ in range(steps):
        eng.train_step(X, Y, 1e-5)
    torch.cuda.synchronize()
    eng.prof = None
    acc = {}
    for i, key in t.keys.items():
        _, s, e, _ = t.records[i]
        n, ms = acc.get(key, (0, 0.0))
        acc[key] = (n + 1, ms + s.elapsed_time(e))
    return {k: 1e3 * ms / n for k, (n, ms) in acc.items()}, {k: n // steps for k, (n, ms) in acc.items()}


for _ in range(3):
    eng.train_step(X, Y, 1e-5)
base1, count = measure()
base2, _ = measure()
keys = sorted(base1, key=lambda k: -base1[k] * count[k])
chosen = {}
for key in keys:
    a_major, b_major, stats, M, N, K = key
    form = 2 if a_major == 1 else (1 if b_major == 0 else 0)
    base = min(base1[key], base2[key])
    best = (base, 0)
    res = []
    for tile in (1, 2, 3, 5, 6, 7, 8):
        if M <= 32:
            continue
        E.TILE_PROBE.clear()
        E.TILE_PROBE[key] = tile
        try:
            a, _ = measure()
            b, _ = measure()
        except Exception as ex:          # a tile the shape cannot use
            torch.cuda.synchronize()
            continue
        us = max(a[key], b[key])         # must win in both passes
        res.append("t%d %.1f" % (tile, us))
        if us < best[0]:
            best = (us, tile)
    E.TILE_PROBE.clear()
    gain = (base - best[0]) * count[key]
    mark = ""
    if best[1] and best[0] < 0.97 * base:
        chosen[",".join(str(v) for v in key)] = best[1]
        mark = "  -> tile %d (%.1f us x %d per step saved)" % (best[1], base - best[0], count[key])
    print("form %d stats %d M=%-6d N=%-5d K=%-6d x%-2d | auto %6.1f us | %s%s" % (form, stats, M, N, K, count[key], base, "  ".join(res), mark), flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spnet_amd", "gemm_tiles.json")
if len(sys.argv) > 2:
    out = sys.argv[2]
with open(out, "w") as f:
    json.dump({"note": "tools/autotune_gemm.py on MI355X: tile ids measured inside the 384x512 batch-32 train step; key = a_major,b_major,stats,M,N,K",
               "tiles": chosen}, f, indent=1, sort_keys=True)
print("wrote %s: %d shapes" % (out, len(chosen)))
