#!/usr/bin/env python3
"""Dev tool: the 24 middle-flow weight gradients (728x728, K = 6144 pixels) as one batched launch vs 24
split-K launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L
st = lambda: torch.cuda.current_stream().cuda_stream
nb, M, N, K = 24, 728, 728, 6144
Xs = [torch.randn(K, M, device="cuda") for _ in range(nb)]
Ds = [torch.randn(K, N, device="cuda") for _ in range(nb)]
Cs = [torch.empty(M, N, device="cuda") for _ in range(nb)]
a0, b0, c0 = Xs[0].data_ptr(), Ds[0].data_ptr(), Cs[0].data_ptr()
table = torch.tensor([v for b in range(nb) for v in ((Xs[b].data_ptr() - a0) // 4, (Ds[b].data_ptr() - b0) // 4, (Cs[b].data_ptr() - c0) // 4)], dtype=torch.int64, device="cuda")
WS = 16 * 1024 * 1024
ws = torch.empty(WS, device="cuda")
def timeit(f, iters=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
fl = 2.0 * nb * M * N * K
for tile in (0, 1, 2, 3, 5, 6, 7, 8):
    t = timeit(lambda: L.spnet_gemm_f32_batched(a0, b0, c0, table.data_ptr(), nb, 1, M, 1, N, N, M, N, K, tile, st()))
    print("batched tile %d: %.1f us total, %.1f us per layer, %.1f TFLOP/s" % (tile, t, t / nb, fl / t / 1e6), flush=True)
def loop():
    for b in range(nb):
        L.spnet_gemm_f32(Xs[b].data_ptr(), 1, M, Ds[b].data_ptr(), 1, N, Cs[b].data_ptr(), N, M, N, K, 0, ws.data_ptr(), WS, None, 0, st())
t = timeit(loop)
print("24 split-K launches (+reduce): %.1f us total, %.1f us per layer, %.1f TFLOP/s" % (t, t / nb, fl / t / 1e6))
