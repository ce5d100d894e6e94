#!/usr/bin/env python3
"""Dev tool (GPU box): the folded BatchNorm kernels against the launch pairs they replace, on the middle-flow shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spnet_amd import _lib as L

st = lambda: torch.cuda.current_stream().cuda_stream


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for M, C, B, H, W in ((6144, 728, 32, 12, 16), (1536, 1536, 32, 6, 8), (24576, 728, 32, 24, 32)):
    x, g, r = (torch.randn(M, C, device="cuda") for _ in range(3))
    out = torch.empty(M, C, device="cuda")
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    mm, mv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    save, ss, co = torch.rand(2 * C, device="cuda"), torch.rand(2 * C, device="cuda"), torch.empty(3 * C, device="cuda")
    dga, dbe = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    part = torch.rand(256, 2, C, device="cuda") + 1
    w = torch.randn(3, 3, C, device="cuda")
    bwd = lambda P: L.spnet_bn_bwd_from_partials(x.data_ptr(), g.data_ptr(), M, C, gam.data_ptr(), bet.data_ptr(), save.data_ptr(),
                                                 save[C:].data_ptr(), P, part.data_ptr(), out.data_ptr(), dga.data_ptr(),
                                                 dbe.data_ptr(), co.data_ptr(), st())
    fused = lambda: L.spnet_bn_finalize_apply(part.data_ptr(), 64, x.data_ptr(), M, C, gam.data_ptr(), bet.data_ptr(), mm.data_ptr(),
                                              mv.data_ptr(), save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 0, r.data_ptr(),
                                              out.data_ptr(), 1e-3, 0.99, st())

    def pair():
        L.spnet_bn_finalize_fwd(part.data_ptr(), 64, M, C, gam.data_ptr(), bet.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
        L.spnet_bn_apply(x.data_ptr(), M, C, ss.data_ptr(), 0, r.data_ptr(), 0, out.data_ptr(), st())

    x4, y4 = x.view(B, H, W, C), out.view(B, H, W, C)
    dwp = lambda: L.spnet_dwconv3x3_tiled_fwd(x4.data_ptr(), w.data_ptr(), y4.data_ptr(), B, H, W, C, 1, ss.data_ptr(), ss[C:].data_ptr(), st())
    dwf = lambda: L.spnet_dwconv3x3_tiled_fwd_bnfin(x4.data_ptr(), w.data_ptr(), y4.data_ptr(), B, H, W, C, 1, part.data_ptr(), 64, M,
                                                    gam.data_ptr(), bet.data_ptr(), mm.data_ptr(), mv.data_ptr(), save.data_ptr(),
                                                    save[C:].data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
    fin = lambda: L.spnet_bn_finalize_fwd(part.data_ptr(), 64, M, C, gam.data_ptr(), bet.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                          save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
    print("M=%-6d C=%-5d | bwd: one launch (P=32) %5.1f us, finalize+apply (P=160) %5.1f | fwd: finalize_apply %5.1f, finalize + apply %5.1f"
          " | dw fwd plain %5.1f, finalize alone %4.1f, dw with the finalize folded in %5.1f"
          % (M, C, t(lambda: bwd(32)), t(lambda: bwd(160)), t(fused), t(pair), t(dwp), t(fin), t(dwf)), flush=True)
