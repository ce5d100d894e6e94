"""Measurement hook for spnet_amd.engine.Engine.pw_alt: the forward and data-gradient GEMMs of every pointwise (1x1)
convolution on the bf16 matrix cores by operand splitting (tools/probes/gemm_bf16x3.hip: fp32 operands as three bf16
pieces, six bf16 MFMAs per product block, fp32 accumulate; fp32-accurate, NOT the k-ordered fmaf chain of the product).

Not product code: bench.py uses it for the `roofline_alt` leg (the whole train step and the GEMM family with these two
operand forms replaced, beside the exact kernels' figures -- VERDICT r3 item 8), tests/test_engine_gpu.py checks that a
train step through it still matches the fp64 oracle at the suite's gradient tolerance.  Weight gradients, the blended
data-gradient GEMMs of blocks 2-3 (BatchNorm backward in the A operand), block1's 3x3 convolutions and the Dense head
stay on the exact kernels.

The weights are split ONCE per optimizer step into their bf16 planes, in both operand forms, by one batched launch:
  forward        y[M][cout] = x[M][cin]  W          B element (n = cout, k = cin)  = W[k][n]
  data gradient  dx[M][cin] = dy[M][cout] W^T       B element (n = cin,  k = cout) = W[n][k]"""
import ctypes

import torch

from spnet_amd import engine as E
from tools.probes import probe_lib


class Bf16x3Pointwise:
    def __init__(self, eng):
        self.e = eng
        self.lib = probe_lib.load()
        self.lib.spnet_gemm_bf16x3_fwd_colstats.restype = ctypes.c_int
        self.lib.spnet_gemm_bf16x3_fwd_colstats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.spnet_split_bf16x3_batched.restype = ctypes.c_int
        self.lib.spnet_split_bf16x3_batched.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_void_p]
        self.planes = {}          # wname -> (forward planes, data-gradient planes)
        self.ver = -1
        self._rows = ctypes.c_int(0)
        jobs, mx = [], 1
        for pw in self._pointwise_layers():
            if pw.wname in self.planes:
                continue
            kf, kd = self._kp(pw.cin), self._kp(pw.cout)
            pf = torch.zeros(3 * pw.cout * kf, dtype=torch.int16, device=eng.dev)
            pd = torch.zeros(3 * pw.cin * kd, dtype=torch.int16, device=eng.dev)
            self.planes[pw.wname] = (pf, pd)
            jobs += [pw.w.data_ptr(), pf.data_ptr(), pw.cin, pw.cout, 1, pw.cout]          # forward form
            jobs += [pw.w.data_ptr(), pd.data_ptr(), pw.cout, pw.cin, pw.cout, 1]          # data-gradient form
            mx = max(mx, pw.cout * kf, pw.cin * kd)
        self.jobs = torch.tensor(jobs, dtype=torch.int64, device=eng.dev)
        self.njobs, self.max_elems = len(jobs) // 6, mx

    @staticmethod
    def _kp(k):
        return (k + 31) // 32 * 32

    def _pointwise_layers(self):
        for node in self.e.nodes:
            for name in ("u1", "u2", "u3"):
                u = getattr(node, name, None)
                if u is not None and hasattr(u, "pw"):
                    yield u.pw
            if hasattr(node, "pwr"):
                yield node.pwr

    def ensure_fresh(self):
        """Planes of the current weights (theta changes with every optimizer step, load or re-initialisation).  The engine
        calls this at the start of forward() / backward(), on the main stream: the split must be ordered in front of
        every consumer on every stream (a strided block's residual convolution runs on the side stream)."""
        if self.ver != self.e._tver[0]:
            rc = self.lib.spnet_split_bf16x3_batched(self.jobs.data_ptr(), self.njobs, self.max_elems, E._stream())
            assert rc == 0, rc
            self.ver = self.e._tver[0]

    def _run(self, tag, A, lda, planes, C, ldc, M, N, K, colstats=None):
        prof = self.e.prof
        t0 = prof.start() if prof is not None else None
        if colstats is None:
            rc = self.lib.spnet_gemm_bf16x3_fwd(A.data_ptr(), lda, planes.data_ptr(), C.data_ptr(), ldc, M, N, K, E._stream())
        else:
            rc = self.lib.spnet_gemm_bf16x3_fwd_colstats(A.data_ptr(), lda, planes.data_ptr(), C.data_ptr(), ldc, M, N, K, colstats,
                                                         ctypes.addressof(self._rows), E._stream())
        assert rc == 0, rc
        if prof is not None:
            prof.stop("gemm", t0, 2.0 * M * N * K, (tag, M, N, K))

    def fwd(self, pw, x, y):
        self.ensure_fresh()
        self._run("bf16x3 aB", x, pw.cin, self.planes[pw.wname][0], y, pw.cout, pw.M, pw.cout, pw.cin)

    def fwd_colstats(self, pw, x, y, region):
        self.ensure_fresh()
        if (pw.M + 95) // 96 * 2 * pw.cout > region[1]:
            raise RuntimeError("BatchNorm partial region too small")
        self._run("bf16x3 aB+stats", x, pw.cin, self.planes[pw.wname][0], y, pw.cout, pw.M, pw.cout, pw.cin,
                  colstats=self.e.ws_ptr(region))
        return self._rows.value

    def dgrad(self, pw, dy, dx):
        self.ensure_fresh()
        self._run("bf16x3 aA", dy, pw.cout, self.planes[pw.wname][1], dx, pw.cin, pw.M, pw.cin, pw.cout)
