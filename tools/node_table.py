#!/usr/bin/env python3
"""Dev tool (GPU box): where a train step's time goes, node by node.  One stream (weight gradients on the main stream), HIP
events around every node's forward and backward; the deferred weight-gradient launches and the optimizer are listed apart.
usage: node_table.py [batch] [H] [W] [predict]      (predict: the inference forward of an inference-only plan, eager)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
W = int(sys.argv[3]) if len(sys.argv) > 3 else 512
PREDICT = len(sys.argv) > 4 and sys.argv[4] == "predict"
eng = Engine(H, W, B, device="cuda:0", seed=0, train=not PREDICT)
eng.wgrad_stream = None
X = torch.rand(B, H, W, 1, device="cuda") * 2 - 1
Y = torch.rand(B, eng.n_out, device="cuda")
if PREDICT:
    step = lambda: eng.forward(X, training=False)
else:
    step = lambda: eng.train_step(X, Y, 1e-4)
for _ in range(3):
    step()
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = {}


def wrap(obj, meth, key):
    f = getattr(obj, meth)

    def g(*a, **k):
        e0, e1 = ev(), ev()
        e0.record()
        r = f(*a, **k)
        e1.record()
        rows.setdefault(key, []).append((e0, e1))
        return r
    setattr(obj, meth, g)


names = []
for i, n in enumerate(eng.nodes):
    nm = "%02d %s" % (i, getattr(n, "name", None) or type(n).__name__)
    names.append(nm)
    wrap(n, "fwd", (nm, "fwd"))
    if not PREDICT:
        wrap(n, "bwd", (nm, "bwd"))
for m in (() if PREDICT else ("flush_deferred_wgrads", "reduce_depthwise_wgrads", "adam_step", "loss", "refresh_planes")):
    wrap(eng, m, (m, ""))
R = 5
e0, e1 = ev(), ev()
e0.record()
for _ in range(R):
    step()
e1.record()
torch.cuda.synchronize()
tot = e0.elapsed_time(e1) / R
ms = lambda k: sum(a.elapsed_time(b) for a, b in rows.get(k, [])) / R
print("one-stream step %.3f ms" % tot)
print("%-34s %8s %8s" % ("node", "fwd ms", "bwd ms"))
sf = sb = 0.0
for nm in names:
    f, b = ms((nm, "fwd")), ms((nm, "bwd"))
    sf += f
    sb += b
    print("%-34s %8.3f %8.3f" % (nm, f, b))
print("%-34s %8.3f %8.3f" % ("sum over nodes", sf, sb))
for m in (() if PREDICT else ("flush_deferred_wgrads", "reduce_depthwise_wgrads", "adam_step", "loss", "refresh_planes")):
    print("%-34s %8.3f" % (m, ms((m, ""))))
