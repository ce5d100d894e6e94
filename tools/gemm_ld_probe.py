#!/usr/bin/env python3
"""Dev tool (GPU box): does the 2,912-byte row stride of the 728-channel tensors cost the middle-flow GEMMs anything?
A pixel row of 728 floats starts at 96 / 64 / 32 bytes past a 128-byte line three times out of four, so every 32-float K
tile of the A operand (and of W) straddles two lines.  Same GEMMs (M x N x K = 6144 x 728 x 728, the three operand
forms) with the operands' leading dimensions at 728 (as the network has them) and at 768 (rows line-aligned)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
M, N, K = 6144, 728, 728


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for tile in (0, 6, 5):
    for ld in (728, 736, 768):
        A = torch.randn(M, ld, device="cuda")          # activations [M][K] / gradients [M][N], row stride ld
        W = torch.randn(K, ld, device="cuda")          # pointwise kernel [K][N], row stride ld
        C = torch.empty(M, ld, device="cuda")
        G = torch.empty(K, ld, device="cuda")
        ws = torch.empty(16 << 20, device="cuda")
        fwd = timeit(lambda: L.spnet_gemm_f32(A.data_ptr(), 0, ld, W.data_ptr(), 1, ld, C.data_ptr(), ld, M, N, K, 1, None, 0, None, tile, st()))
        dgr = timeit(lambda: L.spnet_gemm_f32(A.data_ptr(), 0, ld, W.data_ptr(), 0, ld, C.data_ptr(), ld, M, K, N, 1, None, 0, None, tile, st()))
        wgr = timeit(lambda: L.spnet_gemm_f32(A.data_ptr(), 1, ld, C.data_ptr(), 1, ld, G.data_ptr(), ld, K, N, M, 0, ws.data_ptr(), ws.numel(), None, tile, st()))
        print("tile %d  ld %d | fwd %6.1f us  dgrad %6.1f us  wgrad %6.1f us" % (tile, ld, fwd, dgr, wgr), flush=True)
