#!/usr/bin/env python3
"""Tiled vs streaming depthwise kernels on the network's planes (dev tool, GPU box).

Every timed launch works on ANOTHER set of tensors out of a rotation whose total size is beyond the 256 MiB Infinity
Cache, so the rates are HBM rates (inside the step the operands were written milliseconds earlier).
usage: dw_stream_sweep.py [batch] [fwd|bwd|both]   (batch 32: the train plan, 128: the predict plan)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
what = sys.argv[2] if len(sys.argv) > 2 else "both"
shapes = [("b2u1", 93, 125, 64), ("b2u2", 93, 125, 128), ("b3u1", 47, 63, 128), ("b3u2", 47, 63, 256), ("b4u1", 24, 32, 256),
          ("b4u2", 24, 32, 728), ("mid", 12, 16, 728), ("b14u1", 6, 8, 1024), ("b14u2", 6, 8, 1536)]
if os.environ.get("DW_ONLY"):
    shapes = [s_ for s_ in shapes if s_[0] in os.environ["DW_ONLY"].split(",")]
# rows per wave[:prefetch depth override] (0 = the library's choice)
def _code(tok):
    r, _, d = tok.partition(":")
    return (int(r) & 0xffff) | (int(d or 0) << 16)


RPS = [_code(v) for v in os.environ.get("DW_RPS", "0,1000,48,24,12").split(",")]
RPS_B = [_code(v) for v in os.environ.get("DW_RPS_BWD", os.environ.get("DW_RPS", "0,1000,48,24,12")).split(",")]


def _lab(c):
    return "s%d%s" % (c & 0xffff, (":%d" % (c >> 16)) if c >> 16 else "")


def timeit(fn, nrot, iters=30):
    for i in range(3):
        fn(i % nrot)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nrot)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, H, W, C in shapes:
    nbytes = B * H * W * C * 4
    T = nbytes / 1e6
    ntens = 2 if what == "fwd" else 4
    nrot = max(2, int(700e6 / (ntens * nbytes)) + 1)
    xs = [torch.randn(B, H, W, C, device="cuda") for _ in range(nrot)]
    ys = [torch.empty_like(xs[0]) for _ in range(nrot)]
    w = torch.randn(3, 3, C, device="cuda")
    sc, sh, mu, isd = (torch.rand(C, device="cuda") for _ in range(4))
    line = "%-6s %3dx%3dx%4d T=%6.1fMB rot %2d |" % (name, H, W, C, T, nrot)
    if what in ("fwd", "both"):
        t0 = timeit(lambda i: L.spnet_dwconv3x3_tiled_fwd(xs[i].data_ptr(), w.data_ptr(), ys[i].data_ptr(), B, H, W, C, 1,
                                                          sc.data_ptr(), sh.data_ptr(), st()), nrot)
        line += " fwd tiled %6.1fus %.2fTB/s |" % (t0, 2 * T / t0)
        for rps in RPS:
            if (rps & 0xffff) > H and (rps & 0xffff) != 1000:
                continue
            t1 = timeit(lambda i: L.spnet_dwconv3x3_stream_fwd(xs[i].data_ptr(), w.data_ptr(), ys[i].data_ptr(), B, H, W, C, 1,
                                                               sc.data_ptr(), sh.data_ptr(), rps, st()), nrot)
            line += " %-7s %6.1fus %.2f |" % (_lab(rps), t1, 2 * T / t1)
    print(line, flush=True)
    if what in ("bwd", "both"):
        dys = [torch.randn(B, H, W, C, device="cuda") for _ in range(nrot)]
        dxs = [torch.empty_like(xs[0]) for _ in range(nrot)]
        ws = torch.empty(max(L.spnet_dwconv3x3_tiled_bwd_ws(B, H, W, C),
                             max(L.spnet_dwconv3x3_stream_bwd_ws(B, H, W, C, r) for r in RPS_B)), device="cuda")
        rows = max(L.spnet_dwconv3x3_tiled_rows(B, H, W, C), max(L.spnet_dwconv3x3_stream_rows(B, H, W, C, r) for r in RPS_B))
        bnp = torch.empty(rows * 2 * C, device="cuda")
        line = "%-6s %28s |" % (name, "")
        t0 = timeit(lambda i: L.spnet_dwconv3x3_tiled_bwd(dys[i].data_ptr(), xs[i].data_ptr(), w.data_ptr(), dxs[i].data_ptr(), None, B,
                                                          H, W, C, 1, None, ws.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                          mu.data_ptr(), isd.data_ptr(), bnp.data_ptr(), None, st()), nrot)
        line += " bwd tiled %6.1fus %.2fTB/s |" % (t0, 3 * T / t0)
        for rps in RPS_B:
            if (rps & 0xffff) > H and (rps & 0xffff) != 1000:
                continue
            t1 = timeit(lambda i: L.spnet_dwconv3x3_stream_bwd(dys[i].data_ptr(), xs[i].data_ptr(), w.data_ptr(), dxs[i].data_ptr(), None,
                                                               B, H, W, C, 1, None, ws.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                               mu.data_ptr(), isd.data_ptr(), bnp.data_ptr(), None, rps, st()), nrot)
            line += " %-7s %6.1fus %.2f |" % (_lab(rps), t1, 3 * T / t1)
        print(line, flush=True)
        del dys, dxs
    del xs, ys
    torch.cuda.empty_cache()
print("--- calibration: a plain copy of the same tensors (read T + write T)")
for name, H, W, C in shapes:
    nbytes = B * H * W * C * 4
    nrot = max(2, int(700e6 / (2 * nbytes)) + 1)
    xs = [torch.randn(B, H, W, C, device="cuda") for _ in range(nrot)]
    ys = [torch.empty_like(xs[0]) for _ in range(nrot)]
    c = timeit(lambda i: ys[i].copy_(xs[i]), nrot)
    print("%-6s T=%6.1fMB | torch copy %6.1fus (%.2f TB/s)" % (name, nbytes / 1e6, c, 2 * nbytes / 1e6 / c), flush=True)
    del xs, ys
    torch.cuda.empty_cache()
