#!/usr/bin/env python3
"""Dev tool: what the vendor fp32 GEMM (torch.mm -> rocBLAS/hipBLASLt) reaches on the network's GEMM
shapes, beside spnet_gemm_f32's auto tile.  Only a yardstick: the product never calls the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_sweep import run

torch.backends.cuda.matmul.allow_tf32 = False


def lib(form, M, N, K, iters=20):
    if form == "fwd":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        f = lambda: torch.mm(A, B)
    elif form == "dgrad":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        f = lambda: torch.mm(A, B.t())
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        f = lambda: torch.mm(A.t(), B)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6


shapes = [("mid", 6144, 728, 728), ("b2s1", 372000, 128, 64), ("b2s2", 372000, 128, 128), ("b1c2", 372000, 64, 288),
          ("b3s2", 94752, 256, 256), ("b4s2", 24576, 728, 728), ("b14b", 1536, 2048, 1536), ("b13r", 1536, 1024, 728),
          ("big", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    for form in ("fwd", "dgrad", "wgrad"):
        m, n, k = (M, N, K) if form == "fwd" else ((M, K, N) if form == "dgrad" else (K, N, M))
        lu, lt = lib(form, m, n, k)
        us, tf = run(form, m, n, k, 0)
        print("%-5s %-5s M=%-6d N=%-5d K=%-6d | library %7.1fus %5.1fTF | spnet %7.1fus %5.1fTF" % (name, form, m, n, k, lu, lt, us, tf), flush=True)
