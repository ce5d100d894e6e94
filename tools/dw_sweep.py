#!/usr/bin/env python3
"""Time the tiled depthwise kernels on the network's layer shapes (dev tool, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L
st = lambda: torch.cuda.current_stream().cuda_stream
B = 32
import os as _os
shapes = [("b2s2", 93, 125, 128), ("b3s2", 47, 63, 256), ("b4s2", 24, 32, 728), ("mid", 12, 16, 728), ("b14", 6, 8, 1536)]
if _os.environ.get("DW_ONLY"):
    shapes = [s_ for s_ in shapes if s_[0] == _os.environ["DW_ONLY"]]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, H, W, C in shapes:
    x = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(3, 3, C, device="cuda")
    y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x); dw = torch.empty(3, 3, C, device="cuda")
    add = torch.randn_like(x)
    sc, sh, mu, isd = (torch.rand(C, device="cuda") for _ in range(4))
    ws = torch.empty(L.spnet_dwconv3x3_tiled_bwd_ws(B, H, W, C), device="cuda")
    rows = L.spnet_dwconv3x3_tiled_rows(B, H, W, C)
    bnp = torch.empty(rows * 2 * C, device="cuda")
    T = x.numel() * 4 / 1e6
    f0 = timeit(lambda: L.spnet_dwconv3x3_tiled_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, 1, None, None, st()))
    f1 = timeit(lambda: L.spnet_dwconv3x3_tiled_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, 1, sc.data_ptr(), sh.data_ptr(), st()))
    b0 = timeit(lambda: L.spnet_dwconv3x3_tiled_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), B, H, W, C, 1, None, ws.data_ptr(), None, None, None, None, None, None, st()))
    b1 = timeit(lambda: L.spnet_dwconv3x3_tiled_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), B, H, W, C, 1, add.data_ptr(), ws.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), bnp.data_ptr(), None, st()))
    o0 = timeit(lambda: L.spnet_dwconv3x3_strided(0, x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, 1, None, st()))
    print("%-5s T=%6.1fMB | fwd %6.1fus (%.2f TB/s) fwd+aff %6.1fus | bwd %6.1fus (%.2f TB/s alg 3T) bwd fused %6.1fus | gather fwd (stride-1 form of the strided kernel) %6.1fus"
          % (name, T, f0, 2 * T / f0, f1, b0, 3 * T / b0, b1, o0), flush=True)
print("--- calibration: elementwise kernels on the same tensors")
for name, H, W, C in shapes:
    x = torch.randn(B, H, W, C, device="cuda"); y = torch.empty_like(x)
    ss = torch.rand(2 * C, device="cuda")
    T = x.numel() * 4 / 1e6
    c = timeit(lambda: y.copy_(x))
    a = timeit(lambda: L.spnet_bn_apply(x.data_ptr(), B * H * W, C, ss.data_ptr(), 1, None, 0, y.data_ptr(), st()))
    print("%-5s T=%6.1fMB | torch copy %6.1fus (%.2f TB/s) | bn_apply %6.1fus (%.2f TB/s)" % (name, T, c, 2 * T / c, a, 2 * T / a), flush=True)
