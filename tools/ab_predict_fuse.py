import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from spnet_amd.engine import Engine
res = {}
X = torch.rand(128, 384, 512, 1, device="cuda") * 2 - 1
engs = {v: Engine(384, 512, 128, device="cuda:0", seed=0, train=False, fuse_dw_bwd=v) for v in (True, False)}
outs = {}
for v, e in engs.items():
    e.x_in.copy_(X)
    for _ in range(3):
        e.predict_step()
    torch.cuda.synchronize()
    outs[v] = e.out.clone()
print("fused == unfused outputs:", bool(torch.equal(outs[True], outs[False])), float((outs[True]-outs[False]).abs().max()))
for r in range(4):
    for v, e in engs.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            e.predict_step()
        torch.cuda.synchronize()
        res.setdefault(v, []).append(1e3 * (time.perf_counter() - t0) / 20)
for v, r in res.items():
    print("fuse=%r ms per 128 frames: %s -> %.0f frames/s" % (v, " ".join("%.3f" % t for t in r), 128e3 / sorted(r)[len(r)//2]))
