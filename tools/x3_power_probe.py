#!/usr/bin/env python3
"""Dev tool (GPU box): is the bf16x3 GEMM bound by the power limit?  The same launch (6144 x 728 x 728, 300 back-to-back
launches) on random operands and on all-zero operands: the instruction stream and the memory traffic are identical, only the
switching activity of the datapaths differs, so a time that falls with zeros is clock lost to power, not to the schedule.
(MI355X_MICROARCH.md: zero-filled inputs ran +19 % at the same wave cycles.)  The exact fp32 kernel beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
M, N, K = 6144, 728, 728
Kp = int(L.spnet_bf16x3_kp(K))
c = torch.empty(M, N, device="cuda")


def timeit(fn, iters=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for label, a, w in (("random normal", torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda") * 0.05),
                    ("all zeros", torch.zeros(M, K, device="cuda"), torch.zeros(K, N, device="cuda")),
                    ("random normal again", torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda") * 0.05)):
    planes = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(N, K)), dtype=torch.int16, device="cuda")
    L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st())
    t3 = timeit(lambda: L.spnet_gemm_bf16x3_fwd(a.data_ptr(), K, planes.data_ptr(), c.data_ptr(), N, M, N, K, st()))
    t1 = timeit(lambda: L.spnet_gemm_f32(a.data_ptr(), 0, K, w.data_ptr(), 1, N, c.data_ptr(), N, M, N, K, 1, None, 0, None, 0, st()))
    print("%-20s bf16x3 %6.1f us   exact fp32 %6.1f us" % (label, t3, t1), flush=True)
