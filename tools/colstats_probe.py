#!/usr/bin/env python3
"""Dev tool (GPU box): the forward GEMM with BatchNorm column sums in its epilogue beside the plain forward GEMM, on the
middle-flow shape and two entry-flow shapes (run once per library build: SPNET_HIP_LIB selects it)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L
st = lambda: torch.cuda.current_stream().cuda_stream

def timeit(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

print("library:", L.LIB_PATH)
for (M, N, K, tile) in ((6144, 728, 728, 6), (6144, 728, 728, 5), (94752, 256, 256, 0), (372000, 128, 128, 0), (24576, 728, 728, 7)):
    A, W = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    cs = torch.empty((M + 31) // 32 * 2 * N, device="cuda")
    rows = ctypes.c_int(0)
    t0 = timeit(lambda: L.spnet_gemm_f32(A.data_ptr(), 0, K, W.data_ptr(), 1, N, C.data_ptr(), N, M, N, K, 1, None, 0, None, tile, st()))
    t1 = timeit(lambda: L.spnet_gemm_f32_colstats(A.data_ptr(), 0, K, W.data_ptr(), 1, N, C.data_ptr(), N, M, N, K, tile, cs.data_ptr(), ctypes.addressof(rows), st()))
    print("M=%-6d N=%-4d K=%-4d tile %d | plain %7.1f us | with column sums %7.1f us (%+.1f)" % (M, N, K, tile, t0, t1, t1 - t0), flush=True)
