#!/usr/bin/env python3
"""Dev tool (GPU box): every GEMM launch of one benchmark train step (Xception, 384x512, batch 32) grouped by
(operand form, M, N, K), with its HIP-event time against the fp32-MFMA peak -- sorted by the time LOST to the peak,
i.e. the order in which the GEMM family is worth attacking.  Same one-stream replay as bench.py's roofline leg.
usage: gemm_table.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPNET_OVERLAP_WGRAD"] = "0"
import torch
from spnet_amd.engine import Engine, KernelTimer

PEAK = 157.3e12
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(os.environ.get("GT_BATCH", "32"))
eng = Engine(384, 512, B, device="cuda:0", seed=0, train=True)
X = torch.rand(B, 384, 512, 1, device="cuda") * 2 - 1
Y = torch.rand(B, 576, device="cuda")
for _ in range(3):
    eng.train_step(X, Y, 1e-5)
torch.cuda.synchronize()
t = KernelTimer()
eng.prof = t
for _ in range(steps):
    eng.train_step(X, Y, 1e-5)
torch.cuda.synchronize()
eng.prof = None
rows = []
dwrows = []
for tag, (n, ms, work) in t.tagged().items():
    if tag[0].startswith("dw"):
        dwrows.append((tag, n, ms, work))
        continue
    ideal = work / PEAK * 1e3
    rows.append((ms - ideal, tag, n, ms, work))
rows.sort(key=lambda r: -r[0])
tot_ms = sum(r[3] for r in rows) / steps
tot_w = sum(r[4] for r in rows) / steps
print("GEMM launches carrying a tag: %.3f ms/step, %.1f GFLOP/step, %.1f TFLOP/s (%.1f %% of peak)"
      % (tot_ms, tot_w / 1e9, tot_w / tot_ms / 1e9, 100 * tot_w / tot_ms / 1e9 / 157.3))
print("%-18s %8s %6s %8s | %5s %9s %9s %6s %9s" % ("form", "M", "N", "K", "n/stp", "us/launch", "ms/step", "TF/s", "lost ms"))
for lost, tag, n, ms, work in rows:
    form, M, N, K = tag
    print("%-18s %8d %6d %8d | %5.1f %9.1f %9.3f %6.1f %9.3f"
          % (form, M, N, K, n / steps, 1e3 * ms / n, ms / steps, work / ms / 1e9, lost / steps), flush=True)
print("depthwise launches (algorithmic bytes: fwd 8 B, bwd 12 B per element)")
for tag, n, ms, work in sorted(dwrows, key=lambda r: -r[2]):
    print("%-8s %4dx%-4d C=%-5d | %5.1f /step %8.1f us/launch %8.3f ms/step %7.0f GB/s (%.2f of 8 TB/s)"
          % (tag[0], tag[1], tag[2], tag[3], n / steps, 1e3 * ms / n, ms / steps, work / ms / 1e6, work / ms / 1e6 / 8000))
tot = t.totals()
for fam, (n, ms, w) in tot.items():
    print("family %-5s %6.1f launches/step %8.3f ms/step" % (fam, n / steps, ms / steps))
