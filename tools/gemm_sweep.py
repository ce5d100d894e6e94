#!/usr/bin/env python3
"""Time spnet_gemm_f32 on the network's real GEMM shapes for every tile / form (dev tool, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

WS = 64 * 1024 * 1024
ws = torch.empty(WS, device="cuda")
st = lambda: torch.cuda.current_stream().cuda_stream


def run(form, M, N, K, tile, split=0, iters=20):
    # form: fwd A[M,K] K-major, B[K,N]; dgrad A[M,K] K-major, B[N,K] K-major; wgrad A[K,M] out-major, B[K,N]
    if form == "fwd":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        args = (A.data_ptr(), 0, K, B.data_ptr(), 1, N)
    elif form == "dgrad":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        args = (A.data_ptr(), 0, K, B.data_ptr(), 0, K)
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        args = (A.data_ptr(), 1, M, B.data_ptr(), 1, N)
    C = torch.empty(M, N, device="cuda")
    call = lambda: L.spnet_gemm_f32(*args, C.data_ptr(), N, M, N, K, split, ws.data_ptr(), WS, None, tile, st())
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        form, m, n, k, tile = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        us, tf = run(form, m, n, k, tile, iters=50)
        print("one %s M=%d N=%d K=%d tile=%d: %.1f us %.1f TF" % (form, m, n, k, tile, us, tf))
        sys.exit(0)

    if len(sys.argv) > 1 and sys.argv[1] == "blend":
        # BatchNorm backward folded into the data-gradient GEMM vs the separate elementwise pass it replaces
        def t(fn, iters=30):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters
        for name, M, N, K in (("mid", 6144, 728, 728), ("b4s2", 24576, 728, 728), ("b3s2", 94752, 256, 256), ("b2s2", 372000, 128, 128),
                              ("b14a", 1536, 1024, 1536)):
            g, yp = torch.randn(M, K, device="cuda"), torch.randn(M, K, device="cuda")
            wt = torch.randn(K, N, device="cuda")
            dx, dy = torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda")
            cld = (K + 63) // 64 * 64
            coef = torch.rand(3 * cld, device="cuda")
            co = torch.rand(3 * K, device="cuda")
            sv = torch.rand(2 * K, device="cuda")
            part = torch.rand(64 * 2 * K, device="cuda")
            dga, dbe = torch.empty(K, device="cuda"), torch.empty(K, device="cuda")
            plain = t(lambda: L.spnet_gemm_f32(g.data_ptr(), 0, K, wt.data_ptr(), 1, N, dx.data_ptr(), N, M, N, K, 1, ws.data_ptr(), WS, None, 0, st()))
            blend = t(lambda: L.spnet_gemm_f32_bnblend(g.data_ptr(), yp.data_ptr(), coef.data_ptr(), cld, K, wt.data_ptr(), N, dx.data_ptr(), N, M, N, K, 0, dy.data_ptr(), st()))
            blend0 = t(lambda: L.spnet_gemm_f32_bnblend(g.data_ptr(), yp.data_ptr(), coef.data_ptr(), cld, K, wt.data_ptr(), N, dx.data_ptr(), N, M, N, K, 0, None, st()))
            apply_ = t(lambda: L.spnet_bn_bwd_from_partials(yp.data_ptr(), g.data_ptr(), M, K, sv.data_ptr(), sv.data_ptr(), sv.data_ptr(), sv[K:].data_ptr(), 32, part.data_ptr(), dy.data_ptr(), dga.data_ptr(), dbe.data_ptr(), co.data_ptr(), st()))
            fin = t(lambda: L.spnet_bn_bwd_coeffs_from_partials(32, part.data_ptr(), M, K, sv.data_ptr(), sv.data_ptr(), sv[K:].data_ptr(), dga.data_ptr(), dbe.data_ptr(), coef.data_ptr(), cld, st()))
            print("blend %-5s M=%-6d N=%-4d K=%-4d | plain gemm %6.1f us | blend+dy_out %6.1f | blend only %6.1f | finalize+apply %6.1f | finalize only %5.1f"
                  % (name, M, N, K, plain, blend, blend0, apply_, fin), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nsweep":
        # Does a narrower last column tile pay?  The middle-flow GEMM (6144 x 728 x 728) on the 96x64 tile is 64 x 12 = 768
        # workgroups = exactly 3 per CU, all resident at once; the kernel ends when the busiest CU ends.  N = 704 is
        # 11 column tiles (704 workgroups: 2.75 per CU -> most CUs still hold 3), N = 640 ten (2.5 per CU), N = 512
        # eight (exactly 2 per CU).  Time falls with N only where a whole workgroup per CU disappears.
        for form in ("fwd", "dgrad"):
            res = []
            for N in (768, 736, 728, 704, 640, 576, 512):
                m, n, k = (6144, N, 728) if form == "fwd" else (6144, N, 728)
                us, tf = run(form, m, n, k, 6, split=1, iters=40)
                res.append("N=%d %5.1fus (%d wgs)" % (N, us, 64 * ((N + 63) // 64)))
            print("nsweep %-5s tile 96x64 | %s" % (form, " | ".join(res)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ksweep":
        # fixed cost vs per-K-tile cost of the middle-flow GEMM: time(K) = t0 + slope*K
        for form in ("fwd", "dgrad", "wgrad"):
            for tile in (6, 5):
                res = []
                for K in (104, 728, 1456, 2912, 5824):
                    m, n, k = (6144, 728, K) if form != "wgrad" else (728, 728, K * 8)
                    us, tf = run(form, m, n, k, tile, split=1, iters=30)
                    res.append("K=%d %6.1fus %5.1fTF" % (k, us, tf))
                print("ksweep %-5s tile %d | %s" % (form, tile, " | ".join(res)), flush=True)
        sys.exit(0)
    shapes = [("mid", 6144, 728, 728), ("b2s1", 372000, 128, 64), ("b2s2", 372000, 128, 128), ("b1c2", 372000, 64, 288),
              ("b3s2", 94752, 256, 256), ("b4s2", 24576, 728, 728), ("b14b", 1536, 2048, 1536), ("b13r", 1536, 1024, 728)]
    for name, M, N, K in shapes:
        for form in ("fwd", "dgrad", "wgrad"):
            if form == "fwd":
                m, n, k = M, N, K
            elif form == "dgrad":
                m, n, k = M, K, N
            else:
                m, n, k = K, N, M
            res = []
            for tile in (0, 1, 5, 2, 3, 6, 7, 8):
                try:
                    us, tf = run(form, m, n, k, tile)
                    res.append("t%d %7.1fus %5.1fTF" % (tile, us, tf))
                except Exception as ex:
                    res.append("t%d ERR" % tile)
            print("%-5s %-5s M=%-6d N=%-5d K=%-6d | %s" % (name, form, m, n, k, " | ".join(res)), flush=True)
    # head
    for form, m, n, k in (("fwd", 32, 576, 98304), ("dgrad", 32, 98304, 576), ("wgrad", 98304, 576, 32)):
        us, tf = run(form, m, n, k, 0)
        print("head  %-5s M=%-6d N=%-6d K=%-6d | auto %7.1fus %5.1fTF" % (form, m, n, k, us, tf), flush=True)


if __name__ == "__main__":
    main()
