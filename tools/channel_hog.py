#!/usr/bin/env python3
"""Rehearsal on ONE GPU of what the first 8-GPU run will meet: RCCL's channel kernels hold CUs
underneath the backbone backward while the Dense-head gradient bucket (226 MB of the 310 MB) is in flight.  The GEMM tile
plan is fitted to 256 free CUs (768 workgroups = exactly 3 per CU on the middle-flow shapes), so the question is what the
train step loses when N CUs are taken away for the span of that collective.

A stand-in (tools/probes/hog.hip): N workgroups of 256 threads spinning for `--us` microseconds on a third stream,
launched from Engine.backward's on_node_done hook exactly where parallel.GradReducer launches the head bucket (after
the Dense node).  Prints ms per step for every N.   usage: channel_hog.py [--us 2000] [--n 0,8,16,32,64]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--us", type=float, default=2000.0)
ap.add_argument("--n", default="0,8,16,32,64")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()

from spnet_amd.engine import Engine

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "lib", "libhog.so"))
lib.probe_hog.restype = ctypes.c_int
lib.probe_hog.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

H, W, B = 384, 512, args.batch
eng = Engine(H, W, B, device="cuda:0", seed=0)
X = torch.rand(B, H, W, 1, device="cuda") * 2 - 1
Y = torch.rand(B, 576, device="cuda")
buf = torch.rand(1 << 20, device="cuda")
sink = torch.zeros(4, device="cuda")
hog_stream = torch.cuda.Stream()
head = eng.nodes[-1]


class Hog:
    """Quacks like parallel.GradReducer for Engine.train_step: launches the hog where the head bucket would start."""

    def __init__(self, n):
        self.n, self.side_stream = n, None

    def on_node_done(self, node):
        if node is head and self.n:
            hog_stream.wait_stream(torch.cuda.current_stream())
            rc = lib.probe_hog(self.n, args.us, buf.data_ptr(), buf.numel(), sink.data_ptr(), hog_stream.cuda_stream)
            assert rc == 0, rc

    def finish(self):
        torch.cuda.current_stream().wait_stream(hog_stream)      # the optimizer waits for the "collective"
        return 1.0


def run(n):
    r = Hog(n)
    for _ in range(5):
        eng.train_step(X, Y, 1e-5, reducer=r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step(X, Y, 1e-5, reducer=r)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / args.steps


print("train step, Xception %dx%d batch %d; hog = N workgroups x 256 threads for %.0f us from the Dense node's backward on" % (H, W, B, args.us))
base = None
for n in [int(v) for v in args.n.split(",")]:
    ms = [run(n) for _ in range(3)]
    m = float(np.median(ms))
    base = m if base is None else base
    print("N = %3d   %.3f ms per step (median of 3 x %d steps: %s)   %+.2f %% vs N = 0" % (
        n, m, args.steps, " ".join("%.3f" % v for v in ms), 100.0 * (m / base - 1.0)), flush=True)
