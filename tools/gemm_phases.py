#!/usr/bin/env python3
"""Dev tool (GPU box): where the time of ONE GEMM launch goes.  Needs the diagnostic build of the library
(gemm.hip compiled with -DSP_STAMPS -> tools/var/libstamps.so, selected through SPNET_HIP_LIB): every workgroup stamps
the 100 MHz real-time counter at entry / first tile in LDS / main loop done / epilogue issued / stores drained.
usage: SPNET_HIP_LIB=$PWD/tools/var/libstamps.so gemm_phases.py form M N K tile [stats]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spnet_amd import _lib as L

form, M, N, K, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
stats = "stats" in sys.argv[6:]
zeros = "zeros" in sys.argv[6:]
st = lambda: torch.cuda.current_stream().cuda_stream
if form == "fwd":
    A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    args = (A.data_ptr(), 0, K, B.data_ptr(), 1, N)
elif form == "dgrad":
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    args = (A.data_ptr(), 0, K, B.data_ptr(), 0, K)
else:
    A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    args = (A.data_ptr(), 1, M, B.data_ptr(), 1, N)
if zeros:
    A.zero_(); B.zero_()
C = torch.empty(M, N, device="cuda")
cs = torch.empty((M + 31) // 32 * 2 * N, device="cuda")
rows = ctypes.c_int(0)
if stats:
    call = lambda: L.spnet_gemm_f32_colstats(*args, C.data_ptr(), N, M, N, K, tile, cs.data_ptr(), ctypes.addressof(rows), st())
else:
    call = lambda: L.spnet_gemm_f32(*args, C.data_ptr(), N, M, N, K, 1, None, 0, None, tile, st())
for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
lib = ctypes.CDLL(L.LIB_PATH)
n = 16384 * 8
buf = (ctypes.c_ulonglong * n)()
rc = lib.spnet_debug_read_stamps(buf, n)
assert rc == 0, rc
s = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
s = s[s[:, 0] > 0]
# only the workgroups of the LAST launch: entry stamps within 1 ms of the newest
s = s[s[:, 0] > s[:, 0].max() - 100000]
t0 = s[:, 0].min()
tick = 0.01     # us per tick
q = lambda v: "min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % tuple(tick * np.percentile(v, p) for p in (0, 50, 90, 100))
print("%s M=%d N=%d K=%d tile=%d stats=%d: %.1f us per launch (back to back), %d workgroups stamped; ideal at the fp32 MFMA peak %.1f us"
      % (form, M, N, K, tile, stats, us, len(s), 2.0 * M * N * K / 157.3e6))
print("entry after first entry      ", q(s[:, 0] - t0))
print("prologue (entry -> tile 0)   ", q(s[:, 1] - s[:, 0]))
print("main loop                    ", q(s[:, 2] - s[:, 1]))
print("epilogue issue (stats+stores)", q(s[:, 3] - s[:, 2]))
print("store drain                  ", q(s[:, 4] - s[:, 3]))
print("workgroup lifetime           ", q(s[:, 4] - s[:, 0]))
print("last exit after first entry   %.2f us" % (tick * (s[:, 4].max() - t0)))
clk = (s[:, 6] - s[:, 5]) / np.maximum(s[:, 2] - s[:, 1], 1) * 100.0
print("shader clock in the main loop  MHz: min %.0f  p50 %.0f  max %.0f   (zeros=%d)" % (clk.min(), np.median(clk), clk.max(), zeros))
print("main loop start skew          %.2f us, end skew %.2f us" % (tick * (s[:, 1].max() - s[:, 1].min()), tick * (s[:, 2].max() - s[:, 2].min())))
