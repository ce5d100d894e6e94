#!/usr/bin/env python3
"""Dev tool (GPU box): time-before-build of "depthwise 3x3 inside the pointwise GEMM's A staging" for one Xception
middle-flow unit.  Two processes:

  python tools/dwfuse_mock.py                                   production library: the launch pair it would replace
                                                                (depthwise forward + pointwise GEMM with statistics),
                                                                as a dependent chain on one stream
  SPNET_HIP_LIB=$PWD/tools/var/libdwmock.so python tools/dwfuse_mock.py mock
                                                                diagnostic build (gemm.hip -DSP_DWMOCK): the GEMM with the
                                                                fused A path's traffic and instruction mix (timing only,
                                                                wrong results), and the same binary with the mock off
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

B, H, W, C = 32, 12, 16, 728
M, N, K = B * H * W, 728, 728
st = lambda: torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, C, device="cuda")
z = torch.empty_like(x)
wd = torch.randn(3, 3, C, device="cuda") * 0.3
wp = torch.randn(K, N, device="cuda") * 0.05
y = torch.empty(M, N, device="cuda")
ss = torch.rand(2 * C, device="cuda")
cs = torch.empty((M + 31) // 32 * 2 * N, device="cuda")
rows = ctypes.c_int(0)


def t(fn, iters=200):
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


def dw():
    L.spnet_dwconv3x3_tiled_fwd(x.data_ptr(), wd.data_ptr(), z.data_ptr(), B, H, W, C, 1, ss.data_ptr(), ss[C:].data_ptr(), st())


def gemm(tile, a=z):
    L.spnet_gemm_f32_colstats(a.data_ptr(), 0, K, wp.data_ptr(), 1, N, y.data_ptr(), N, M, N, K, tile, cs.data_ptr(),
                              ctypes.addressof(rows), st())


if len(sys.argv) > 1 and sys.argv[1] == "mock":
    setm = L._lib.spnet_debug_set_dwmock
    setm.restype, setm.argtypes = ctypes.c_int, [ctypes.c_void_p]
    x2 = torch.randn(M + 64, K, device="cuda")
    for tile in (6, 5):
        setm(None)
        off = t(lambda: gemm(tile, x))
        setm(x2.data_ptr())
        on = t(lambda: gemm(tile, x))
        print("mock build, tile %d: GEMM+stats %.1f us with the mock off, %.1f us with the fused-A-path mock" % (tile, off, on),
              flush=True)
else:
    d = t(dw)
    for tile in (6, 5):
        g = t(lambda: gemm(tile))

        def pair():
            dw()
            gemm(tile)
        p = t(pair)
        print("production, tile %d: depthwise fwd alone %.1f us, GEMM+stats alone %.1f us, the dependent pair %.1f us"
              % (tile, d, g, p), flush=True)
