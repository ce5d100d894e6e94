#!/usr/bin/env python3
"""Dev tool: time the small-channel direct convolutions at the benchmark's sizes (batch 32, 384x512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
WS = 4 * 1024 * 1024
ws = torch.empty(WS, device="cuda")


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


B = 32
for (cin, cout, stride, same, H, W) in [(1, 3, 1, 1, 384, 512), (3, 3, 1, 1, 384, 512), (3, 3, 1, 1, 192, 256), (3, 32, 2, 0, 192, 256)]:
    OH = H if same else (H - 3) // stride + 1
    OW = W if same else (W - 3) // stride + 1
    x = torch.randn(B, H, W, cin, device="cuda")
    w = torch.randn(3, 3, cin, cout, device="cuda")
    y = torch.empty(B, OH, OW, cout, device="cuda")
    dy = torch.randn(B, OH, OW, cout, device="cuda")
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    nb = 4 * (x.numel() + y.numel())
    t0 = timeit(lambda: L.spnet_conv3x3_small(0, cin, cout, stride, same, x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, ws.data_ptr(), WS, st()))
    t1 = timeit(lambda: L.spnet_conv3x3_small(1, cin, cout, stride, same, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, W, ws.data_ptr(), WS, st()))
    t2 = timeit(lambda: L.spnet_conv3x3_small(2, cin, cout, stride, same, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ws.data_ptr(), WS, st()))
    print("conv %d->%d s%d %dx%d: tensors %.1f MB | fwd %.1f us (%.2f TB/s) | dX %.1f us (%.2f TB/s) | dW %.1f us (%.2f TB/s)"
          % (cin, cout, stride, H, W, nb / 1e6, t0, nb / t0 / 1e6, t1, nb / t1 / 1e6, t2, nb / t2 / 1e6), flush=True)

# block1_conv2 input gradient: implicit GEMM vs (dgrad GEMM into dcol + col2im)
H, W = 95, 127
dy = torch.randn(B, H - 2, W - 2, 64, device="cuda")
w = torch.randn(3, 3, 32, 64, device="cuda")
dx = torch.empty(B, H, W, 32, device="cuda")
t = timeit(lambda: L.spnet_conv3x3_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, W, 32, 64, st()))
fl = 2.0 * B * H * W * 576 * 32
print("block1_conv2 dX implicit GEMM: %.1f us (%.1f TFLOP/s)" % (t, fl / t / 1e6), flush=True)
x = torch.randn(B, H, W, 32, device="cuda")
y = torch.empty(B, H - 2, W - 2, 64, device="cuda")
t = timeit(lambda: L.spnet_conv3x3_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, 32, 64, st()))
fl = 2.0 * B * (H - 2) * (W - 2) * 288 * 64
print("block1_conv2 Y  implicit GEMM: %.1f us (%.1f TFLOP/s)" % (t, fl / t / 1e6), flush=True)
n = L.spnet_conv3x3_wgrad_ws(B, H, W, 32, 64)
wsb = torch.empty(n, device="cuda")
dw = torch.empty(3, 3, 32, 64, device="cuda")
t = timeit(lambda: L.spnet_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, 32, 64, wsb.data_ptr(), n, st()))
print("block1_conv2 dW implicit GEMM: %.1f us (%.1f TFLOP/s)" % (t, fl / t / 1e6), flush=True)
