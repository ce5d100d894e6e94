#!/usr/bin/env python3
"""Dev tool (GPU box): 20 launches of the bf16x3 kernel and of the exact fp32 MFMA kernel on the middle flow's GEMM
(6144 x 728 x 728), for counter passes (tools/x3_pmc.py runs this under rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
M, N, K = 6144, 728, 728
a = [torch.randn(M, K, device="cuda") for _ in range(4)]
w = torch.randn(K, N, device="cuda") * 0.05
Kp = int(L.spnet_bf16x3_kp(K))
planes = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(N, K)), dtype=torch.int16, device="cuda")
L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st())
c = torch.empty(M, N, device="cuda")
ap = []
for x in a:                                    # A as planes in 1-KiB pieces (what the depthwise kernels write in the product)
    pl = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(M, K)), dtype=torch.int16, device="cuda")
    L.spnet_split_rows_bf16x3(x.data_ptr(), K, pl.data_ptr(), M, K, st())
    ap.append(pl)
for i in range(20):
    L.spnet_gemm_bf16x3_pp(ap[i % 4].data_ptr(), planes.data_ptr(), c.data_ptr(), N, M, N, K, None, 0, st())
for i in range(20):
    L.spnet_gemm_bf16x3_fwd(a[i % 4].data_ptr(), K, planes.data_ptr(), c.data_ptr(), N, M, N, K, st())
for i in range(20):
    L.spnet_gemm_f32(a[i % 4].data_ptr(), 0, K, w.data_ptr(), 1, N, c.data_ptr(), N, M, N, K, 1, None, 0, None, 0, st())
torch.cuda.synchronize()
