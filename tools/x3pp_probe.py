#!/usr/bin/env python3
"""Dev tool (GPU box): the planes x planes bf16x3 GEMM probe (tools/probes/x3pp.hip) against the product kernel
(gemm_bf16x3_fwd_kernel, fp32 A split in the kernel) and the exact fp32 MFMA kernel, interleaved in ONE process on the
shapes of the Xception middle flow (cdna_hip_programming.md rule 24), random and all-zero operands (rule 25).
A's planes are produced offline by the product library's batched split; results are compared bit for bit.

usage: x3pp_probe.py [variants, e.g. 0,1,2] [rounds]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "probes", "lib", "libx3pp.so"))
P, I, LG = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.probe_x3pp.restype = I
lib.probe_x3pp.argtypes = [I, P, LG, I, P, LG, I, P, I, I, I, I, P]
lib.probe_x3pp_retile.restype = I
lib.probe_x3pp_retile.argtypes = [P, P, I, I, P]
st = lambda: torch.cuda.current_stream().cuda_stream
TILED = lambda v: 10 <= v < 100
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def split_rows(a):
    """fp32 [R][K] -> int16 ROW-MAJOR planes [3][R][Kp] (round to nearest even, zero padded): the layout this probe started
    from (the product library writes the pieces layout since the probe's result was adopted)."""
    R, K = a.shape
    Kp = int(L.spnet_bf16x3_kp(K))
    planes = torch.zeros(3, R, Kp, dtype=torch.bfloat16, device="cuda")
    h = a.to(torch.bfloat16)
    r1 = a - h.float()
    m = r1.to(torch.bfloat16)
    l = (r1 - m.float()).to(torch.bfloat16)
    planes[0, :, :K], planes[1, :, :K], planes[2, :, :K] = h, m, l
    return planes.view(torch.int16)


def split_w(w):
    """Keras pointwise kernel [K][N] -> (row-major planes [3][N][Kp] for the probe kernels, the product library's planes)"""
    K, N = w.shape
    prod = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(N, K)), dtype=torch.int16, device="cuda")
    L.spnet_split_bf16x3(w.data_ptr(), prod.data_ptr(), K, N, st())
    return split_rows(w.t().contiguous()).reshape(-1), prod


def retile(planes, R, Kp):
    """planes [3][R][Kp] -> the tiled layout of x3pp2_kernel<TILED> ([3][ceil(R/16)][Kp/32][512])"""
    rgs = (R + 15) // 16
    out = torch.empty(3 * rgs * (Kp // 32) * 512, dtype=torch.int16, device="cuda")
    rc = lib.probe_x3pp_retile(planes.data_ptr(), out.data_ptr(), R, Kp, st())
    assert rc == 0, rc
    return out


def launch(v, apl, atl, bpl, btl, c, M, N, Kp):
    """one launch of variant v on (planes | tiled planes)"""
    if v == 20:         # the product kernel's operand form (fp32 A) on tiled B planes; `apl` is the fp32 A here
        return lib.probe_x3pp(v, apl.data_ptr(), 0, apl.shape[1], btl.data_ptr(), btl.numel() // 3, Kp, c.data_ptr(), N, M, N, Kp, st())
    if TILED(v):
        return lib.probe_x3pp(v, atl.data_ptr(), atl.numel() // 3, Kp, btl.data_ptr(), btl.numel() // 3, Kp, c.data_ptr(), N, M, N, Kp, st())
    return lib.probe_x3pp(v, apl.data_ptr(), M * Kp, Kp, bpl.data_ptr(), N * Kp, Kp, c.data_ptr(), N, M, N, Kp, st())


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def run(M, N, K, zeros=False):
    nb = 4                                             # rotate operand buffers: 4 x (17.9 + 26.9) MB at the middle-flow shape
    if zeros:
        a = [torch.zeros(M, K, device="cuda") for _ in range(nb)]
        w = torch.zeros(K, N, device="cuda")
    else:
        a = [torch.randn(M, K, device="cuda") for _ in range(nb)]
        w = torch.randn(K, N, device="cuda") * 0.05
    Kp = int(L.spnet_bf16x3_kp(K))
    bpl, bprod = split_w(w)
    apl = [split_rows(x) for x in a]
    atl = [retile(x, M, Kp) for x in apl]
    btl = retile(bpl, N, Kp)
    c0 = torch.empty(M, N, device="cuda")
    c1 = torch.empty(M, N, device="cuda")
    L.spnet_gemm_bf16x3_fwd(a[0].data_ptr(), K, bprod.data_ptr(), c0.data_ptr(), N, M, N, K, st())
    names = {}
    for v in variants:
        c1.fill_(float("nan"))
        rc = launch(v, a[0] if v == 20 else apl[0], atl[0], bpl, btl, c1, M, N, Kp)
        assert rc == 0, rc
        torch.cuda.synchronize()
        same = bool(torch.equal(c0, c1))
        err = float((c0 - c1).abs().max())
        names[v] = "bit-identical" if same else "MAX DIFF %.3e" % err
    fns = {"product x3": lambda i: L.spnet_gemm_bf16x3_fwd(a[i % nb].data_ptr(), K, bprod.data_ptr(), c0.data_ptr(), N, M, N, K, st()),
           "exact f32": lambda i: L.spnet_gemm_f32(a[i % nb].data_ptr(), 0, K, w.data_ptr(), 1, N, c0.data_ptr(), N, M, N, K, 1,
                                                   None, 0, None, 0, st())}
    for v in variants:
        fns["pp v%d" % v] = (lambda v_: lambda i: launch(v_, a[i % nb] if v_ == 20 else apl[i % nb], atl[i % nb], bpl, btl, c1, M, N, Kp))(v)
    res = {k: [] for k in fns}
    for k, f in fns.items():
        timed(f, 20)
    for r in range(rounds):
        for k, f in fns.items():
            res[k].append(timed(f, 50))
    flops = 2.0 * M * N * K
    print("%d x %d x %d %s" % (M, N, K, "(all-zero operands)" if zeros else "(random operands)"))
    for k, v in res.items():
        v = sorted(v)
        med = v[len(v) // 2]
        extra = ""
        if k.startswith("pp v"):
            extra = "  " + names[int(k[4:])]
        print("  %-12s median %7.1f us  min %7.1f  (%6.1f TFLOP/s fp32-equivalent)%s" % (k, med, v[0], flops / med / 1e6, extra))
    sys.stdout.flush()


def ksweep(M, N, zeros):
    """time = fixed + per K step: launch + prologue + epilogue against the main loop"""
    print("K sweep %d x %d %s: us per launch" % (M, N, "(all-zero operands)" if zeros else "(random operands)"))
    print("  %6s %6s | %10s | %s" % ("K", "steps", "product", " ".join("%8s" % ("pp v%d" % v) for v in variants)))
    rows = []
    for K in (32, 64, 128, 256, 384, 512, 728, 1024, 1456, 2912):
        nb = 4
        mk = torch.zeros if zeros else torch.randn
        a = [mk(M, K, device="cuda") for _ in range(nb)]
        w = mk(K, N, device="cuda") * 0.05
        Kp = int(L.spnet_bf16x3_kp(K))
        bpl, bprod = split_w(w)
        apl = [split_rows(x) for x in a]
        atl = [retile(x, M, Kp) for x in apl]
        btl = retile(bpl, N, Kp)
        c0 = torch.empty(M, N, device="cuda")
        fns = [lambda i: L.spnet_gemm_bf16x3_fwd(a[i % nb].data_ptr(), K, bprod.data_ptr(), c0.data_ptr(), N, M, N, K, st())]
        for v in variants:
            fns.append((lambda v_: lambda i: launch(v_, a[i % nb] if v_ == 20 else apl[i % nb], atl[i % nb], bpl, btl, c0, M, N, Kp))(v))
        for f in fns:
            timed(f, 20)
        t = [min(timed(f, 50) for _ in range(3)) for f in fns]
        rows.append((Kp // 32, t))
        print("  %6d %6d | %10.1f | %s" % (K, Kp // 32, t[0], " ".join("%8.1f" % x for x in t[1:])))
    # least-squares line through the points with >= 8 steps
    import numpy as np
    xs = np.array([r[0] for r in rows if r[0] >= 8], dtype=float)
    for j, name in enumerate(["product"] + ["pp v%d" % v for v in variants]):
        ys = np.array([r[1][j] for r in rows if r[0] >= 8])
        b, a0 = np.polyfit(xs, ys, 1)
        print("  %-8s fixed %5.1f us + %6.3f us per K step" % (name, a0, b))
    sys.stdout.flush()


def edge(M, N, K):
    """ragged shapes: every variant against the product kernel, bit for bit"""
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda")
    Kp = int(L.spnet_bf16x3_kp(K))
    bpl, bprod = split_w(w)
    apl = split_rows(a)
    atl, btl = retile(apl, M, Kp), retile(bpl, N, Kp)
    c0 = torch.empty(M, N, device="cuda")
    L.spnet_gemm_bf16x3_fwd(a.data_ptr(), K, bprod.data_ptr(), c0.data_ptr(), N, M, N, K, st())
    out = []
    for v in variants:
        c1 = torch.full((M + 1, N), 7.0, device="cuda")            # a guard row behind C
        rc = launch(v, a if v == 20 else apl, atl, bpl, btl, c1, M, N, Kp)
        assert rc == 0, rc
        torch.cuda.synchronize()
        ok = bool(torch.equal(c0, c1[:M])) and bool((c1[M] == 7.0).all())
        out.append("v%d %s" % (v, "ok" if ok else "MISMATCH %.3e" % float((c0 - c1[:M]).abs().max())))
    print("edge %d x %d x %d: %s" % (M, N, K, ", ".join(out)))


if __name__ == "__main__":
    torch.manual_seed(0)
    for shp in ((1000, 300, 100), (97, 97, 36), (96, 96, 32), (5, 3, 4), (6144, 728, 728)):
        edge(*shp)
    if len(sys.argv) > 3 and sys.argv[3] == "ksweep":
        ksweep(6144, 728, False)
        ksweep(6144, 728, True)
        sys.exit(0)
    run(6144, 728, 728)
    run(6144, 728, 728, zeros=True)
    run(24576, 728, 728)
    run(1536, 1024, 728)
    run(1536, 1536, 1024)
    run(94752, 256, 256)
