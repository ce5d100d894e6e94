#!/usr/bin/env python3
"""Dev tool (GPU box): per-shape tile choice for the GEMMs of the benchmark train step, measured IN the step.
For every distinct plain GEMM launch (operand form, BatchNorm statistics or not, M, N, K) each tile id is forced in turn
(engine.TILE_PROBE) for a few whole train steps and that shape's launches are timed with HIP events; a tile is recorded
only where it beats the library's own choice by more than 3 % in BOTH of two alternating passes.
MERGES into spnet_amd/gemm_tiles.json (keys carry the shape, so geometries do not collide).
usage: autotune_gemm.py [steps per probe] [out.json] [H W B] [train|predict] [backbone]
       (default 384 512 32 train = the benchmark step; 331 331 16 train = the reference's own layout;
        384 512 128 predict = BASELINE configs[4])"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPNET_OVERLAP_WGRAD"] = "0"
os.environ["SPNET_GEMM_TILES"] = "0"
import torch
from spnet_amd import engine as E

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W, B = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (384, 512, 32)
MODE = sys.argv[6] if len(sys.argv) > 6 else "train"
BACKBONE = sys.argv[7] if len(sys.argv) > 7 else "Xception"
eng = E.Engine(H, W, B, device="cuda:0", seed=0, train=(MODE == "train"), backbone=BACKBONE)
X = torch.rand(B, H, W, 1, device="cuda") * 2 - 1
Y = torch.rand(B, 576, device="cuda")


def one_step():
    if MODE == "train":
        eng.train_step(X, Y, 1e-5)
    else:
        eng.forward(X, training=False)


def measure():
    """{key: us per launch} over `steps` steps."""
    t = E.KernelTimer()
    eng.prof = t
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    eng.prof = None
    acc = {}
    for i, key in t.keys.items():
        _, s, e, _ = t.records[i]
        n, ms = acc.get(key, (0, 0.0))
        acc[key] = (n + 1, ms + s.elapsed_time(e))
    return {k: 1e3 * ms / n for k, (n, ms) in acc.items()}, {k: n // steps for k, (n, ms) in acc.items()}


for _ in range(3):
    one_step()
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
    for tile in (1, 2, 3, 5, 6, 7, 8) + ((9, 10) if M <= 16384 and form != 2 else ()):
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
if len(sys.argv) > 2 and sys.argv[2] not in ("", "-"):
    out = sys.argv[2]
tiles = {}
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spnet_amd", "gemm_tiles.json")
if os.path.exists(src):
    tiles = dict(json.load(open(src)).get("tiles", {}))
tiles.update(chosen)
with open(out, "w") as f:
    json.dump({"note": "tools/autotune_gemm.py on MI355X: tile ids measured inside the step of each geometry (384x512 batch 32 "
                       "train, 331x331 batch 16 train, 384x512 batch 128 predict); key = a_major,b_major,stats,M,N,K",
               "tiles": tiles}, f, indent=1, sort_keys=True)
print("wrote %s: %d new shapes for %s %dx%d batch %d %s, %d in all" % (out, len(chosen), BACKBONE, H, W, B, MODE, len(tiles)))
