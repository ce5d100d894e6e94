#!/usr/bin/env python3
"""Dev tool (GPU box): how fast the VENDOR's bf16 GEMM (torch.mm on bf16 tensors -> hipBLASLt / rocBLAS) executes the same
number of bf16 MFMAs as one bf16x3 product -- C[M][N] = A[M][6K] B[6K][N] in bf16 with fp32 accumulation is six
K-deep bf16 GEMMs, the matrix work of a 2 M N K fp32-equivalent bf16x3 GEMM (its operand bytes are 12 (M + N) K against
the planes' 6 (M + N) K, so the vendor kernel moves MORE) -- interleaved with spnet_gemm_bf16x3_pp in one process, random
and all-zero operands.  Only a yardstick for `roofline.frac_vs_real_operand_bf16_rate`: the product never calls the library.
usage: x3_vendor_yardstick.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timed(f, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def run(M, N, K, zeros):
    nb = 4
    mk = torch.zeros if zeros else torch.randn
    a = [mk(M, K, device="cuda") for _ in range(nb)]
    w = mk(K, N, device="cuda") * 0.05
    wp = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(N, K)), dtype=torch.int16, device="cuda")
    L.spnet_split_bf16x3(w.data_ptr(), wp.data_ptr(), K, N, st())
    ap = []
    for x in a:
        pl = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(M, K)), dtype=torch.int16, device="cuda")
        L.spnet_split_rows_bf16x3(x.data_ptr(), K, pl.data_ptr(), M, K, st())
        ap.append(pl)
    c = torch.empty(M, N, device="cuda")
    a6 = [mk(M, 6 * K, device="cuda").to(torch.bfloat16) for _ in range(nb)]
    w6 = (mk(6 * K, N, device="cuda") * 0.05).to(torch.bfloat16)
    w6t = w6.t().contiguous()                      # [N][6K]: the "NT" form the planes kernel has (both operands K-major)
    fns = {"spnet_gemm_bf16x3_pp": lambda i: L.spnet_gemm_bf16x3_pp(ap[i % nb].data_ptr(), wp.data_ptr(), c.data_ptr(), N, M, N, K,
                                                                     None, None, st()),
           "vendor bf16 [M][6K] x [6K][N]": lambda i: torch.mm(a6[i % nb], w6),
           "vendor bf16 [M][6K] x [N][6K]^T": lambda i: torch.mm(a6[i % nb], w6t.t())}
    res = {k: [] for k in fns}
    for f in fns.values():
        timed(f, 20)
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(timed(f))
    flops = 2.0 * M * N * K
    print("%d x %d x %d %s" % (M, N, K, "(all-zero operands)" if zeros else "(random operands)"))
    for k, v in res.items():
        v = sorted(v)
        med = v[len(v) // 2]
        print("  %-34s median %7.1f us  min %7.1f  (%6.1f TFLOP/s fp32-equivalent = %6.0f TFLOP/s of bf16 MFMA work)"
              % (k, med, v[0], flops / med / 1e6, 6 * flops / med / 1e6))
    sys.stdout.flush()


if __name__ == "__main__":
    torch.manual_seed(0)
    for shp in ((6144, 728, 728), (24576, 728, 728), (6144, 1024, 728)):
        run(*shp, zeros=False)
        run(*shp, zeros=True)
