#!/usr/bin/env python3
"""Dev tool (GPU box): the pointwise data-gradient GEMM + depthwise backward of one Xception middle-flow unit
(6144 pixels of 12 x 16 planes, 728 channels), as the two launches (spnet_gemm_bf16x3_pp into dz, spnet_dwconv3x3_tiled_bwd)
and as the fused launch (spnet_gemm_bf16x3_pp_dwbwd), interleaved in one process, operands rotating over four buffer sets.
SPNET_HIP_LIB selects a variant library (tools/build_variant_lib.sh, e.g. -DFB_SKIP_EPILOGUE: the fused kernel's main loop
alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
B, H, W, cin, cout = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 12, 16, 728, 728)))
M = B * H * W
nb = 4
pe = lambda r, k: 3 * int(L.spnet_bf16x3_plane_elems(r, k))
dy = [torch.randn(M, cout, device="cuda") * 0.5 for _ in range(nb)]
dyp = [torch.zeros(pe(M, cout), dtype=torch.int16, device="cuda") for _ in range(nb)]
for a, p in zip(dy, dyp):
    L.spnet_split_rows_bf16x3(a.data_ptr(), cout, p.data_ptr(), M, cout, st())
w = torch.randn(cin, cout, device="cuda") * 0.05
wp = torch.zeros(pe(cin, cout), dtype=torch.int16, device="cuda")
jobs = torch.tensor([w.data_ptr(), wp.data_ptr(), cout, cin, cout, 1], dtype=torch.int64, device="cuda")
L.spnet_split_bf16x3_batched(jobs.data_ptr(), 1, pe(cin, cout) // 3, st())
x = [torch.randn(B, H, W, cin, device="cuda") for _ in range(nb)]
add = [torch.randn(B, H, W, cin, device="cuda") for _ in range(nb)]
wd = torch.randn(3, 3, cin, device="cuda")
sc, sh, mu, isd = (torch.rand(cin, device="cuda") + 0.5 for _ in range(4))
dz = torch.empty(M, cin, device="cuda")
dx = torch.empty(M, cin, device="cuda")
rows0 = int(L.spnet_dwconv3x3_tiled_rows(B, H, W, cin))
ws0 = torch.zeros(int(L.spnet_dwconv3x3_tiled_bwd_ws(B, H, W, cin)), device="cuda")
bnp = torch.zeros(max(rows0, 64) * 2 * cin, device="cuda")
part = torch.zeros(int(L.spnet_gemm_bf16x3_dwbwd_rows(M)) * 9 * cin, device="cuda")


def gemm(i):
    L.spnet_gemm_bf16x3_pp(dyp[i % nb].data_ptr(), wp.data_ptr(), dz.data_ptr(), cin, M, cin, cout, None, None, st())


def dwb(i, with_add=False):
    L.spnet_dwconv3x3_tiled_bwd(dz.data_ptr(), x[i % nb].data_ptr(), wd.data_ptr(), dx.data_ptr(), None, B, H, W, cin, 1,
                                add[i % nb].data_ptr() if with_add else None, ws0.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                mu.data_ptr(), isd.data_ptr(), bnp.data_ptr(), None, st())


def fused(i, with_add=False):
    L.spnet_gemm_bf16x3_pp_dwbwd(dyp[i % nb].data_ptr(), wp.data_ptr(), B, H, W, cin, cout, x[i % nb].data_ptr(), wd.data_ptr(),
                                 dx.data_ptr(), 1, add[i % nb].data_ptr() if with_add else None, part.data_ptr(), sc.data_ptr(),
                                 sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), bnp.data_ptr(), None, st())


def timed(fn, n=100):
    for i in range(10):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


cases = {"gemm (96x96 tiles) -> dz": gemm, "depthwise backward over dz": dwb, "the pair": lambda i: (gemm(i), dwb(i)),
         "fused": fused, "the pair, with residual gradient": lambda i: (gemm(i), dwb(i, True)),
         "fused, with residual gradient": lambda i: fused(i, True)}
res = {k: [] for k in cases}
for r in range(5):
    for k, f in cases.items():
        res[k].append(timed(f))
print("%d x %d x %d planes, %d -> %d channels (M = %d), lib %s" % (B, H, W, cin, cout, M, os.path.basename(L.LIB_PATH)))
for k, v in res.items():
    v = sorted(v)
    print("  %-40s median %6.1f us  min %6.1f" % (k, v[len(v) // 2], v[0]))
