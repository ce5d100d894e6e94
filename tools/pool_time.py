#!/usr/bin/env python3
"""Dev tool: max-pool forward/backward at the benchmark's four pooling layers (batch 32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
B = 32
for (H, W, C) in [(93, 125, 128), (47, 63, 256), (24, 32, 728), (12, 16, 1024)]:
    OH, OW = (H + 1) // 2, (W + 1) // 2
    x = torch.randn(B, H, W, C, device="cuda"); r = torch.randn(B, OH, OW, C, device="cuda")
    y = torch.empty_like(r); idx = torch.empty(B * OH * OW * C // 4, dtype=torch.int32, device="cuda")
    dy = torch.randn_like(r); dx = torch.empty_like(x)
    ss = torch.rand(2 * C, device="cuda")
    tf = timeit(lambda: L.spnet_maxpool3x3s2_add_fwd(x.data_ptr(), r.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, C, ss.data_ptr(), ss.data_ptr(), st()))
    tb = timeit(lambda: L.spnet_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), B, H, W, C, st()))
    print("pool %dx%dx%d: fwd %.1f us  bwd %.1f us" % (H, W, C, tf, tb), flush=True)
