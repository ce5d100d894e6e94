#!/usr/bin/env python3
"""Dev tool (GPU box): the bf16x3 probe beside the exact fp32 MFMA GEMM on the network's dominant forward shape
(6144 x 728 x 728, the Xception middle flow at batch 32) and two larger ones: time per launch, TFLOP/s of algorithmic
work, and the error of both against float64 (max and rms, relative to |a_row| * |w_col|)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream


def t(fn, iters=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for M, N, K in ((6144, 728, 728), (24576, 728, 728), (6144, 1024, 728), (1536, 2048, 1536)):
    rs = np.random.RandomState(M)
    A, W = rs.randn(M, K).astype(np.float32), (rs.randn(K, N) * 0.05).astype(np.float32)
    a, w = torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()
    Kp = int(L.spnet_bf16x3_kp(K))
    planes = torch.zeros(3 * N * Kp, dtype=torch.int16, device="cuda")
    L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st())
    c3, c1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    f3 = lambda: L.spnet_gemm_bf16x3_fwd(a.data_ptr(), K, planes.data_ptr(), c3.data_ptr(), N, M, N, K, st())
    f1 = lambda: L.spnet_gemm_f32(a.data_ptr(), 0, K, w.data_ptr(), 1, N, c1.data_ptr(), N, M, N, K, 1, None, 0, None, 0, st())
    fs = lambda: L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st())
    t3, t1, ts = t(f3), t(f1), t(fs)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.sqrt((A.astype(np.float64) ** 2).sum(1))[:, None] * np.sqrt((W.astype(np.float64) ** 2).sum(0))[None, :]
    e3 = np.abs(c3.cpu().double().numpy() - ref) / scale
    e1 = np.abs(c1.cpu().double().numpy() - ref) / scale
    fl = 2.0 * M * N * K
    print("M=%-6d N=%-5d K=%-5d | exact fp32 MFMA %6.1f us %6.1f TF | bf16x3 %6.1f us %6.1f TF (x%.2f) | weight split %5.1f us | "
          "error vs fp64 (max / rms, relative to |a||w|): exact %.2e / %.2e, bf16x3 %.2e / %.2e"
          % (M, N, K, t1, fl / t1 / 1e6, t3, fl / t3 / 1e6, t1 / t3, ts, e1.max(), np.sqrt((e1 ** 2).mean()), e3.max(),
             np.sqrt((e3 ** 2).mean())), flush=True)
