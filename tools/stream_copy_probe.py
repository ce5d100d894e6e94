#!/usr/bin/env python3
"""Dev tool (GPU box): the streaming ceiling of this chip for a read-T + write-T pass, by organisation of the bytes in
flight (tools/probes/stream_copy.hip): registers (U loads per lane), nontemporal, wave-private LDS ring filled by
LDS-DMA.  Buffers rotate beyond the Infinity Cache.  usage: stream_copy_probe.py [MB per tensor]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "lib", "libstream_copy.so"))
lib.probe_copy.restype = ctypes.c_int
lib.probe_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
MB = float(sys.argv[1]) if len(sys.argv) > 1 else 190.5
n4 = int(MB * 1e6 / 16) // 64 * 64
nrot = max(2, int(700e6 / (2 * n4 * 16)) + 1)
xs = [torch.randn(n4 * 4, device="cuda") for _ in range(nrot)]
ys = [torch.empty_like(xs[0]) for _ in range(nrot)]
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=30):
    for i in range(3):
        fn(i % nrot)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nrot)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def check(mode, depth, blocks):
    ys[0].zero_()
    rc = lib.probe_copy(xs[0].data_ptr(), ys[0].data_ptr(), n4, mode, depth, blocks, st)
    torch.cuda.synchronize()
    return rc == 0 and torch.equal(xs[0], ys[0])


T = n4 * 16 / 1e6
print("tensor %.1f MB, rotation %d" % (T, nrot))
t = timeit(lambda i: ys[i].copy_(xs[i]))
print("torch copy_            %7.1f us  %.2f TB/s" % (t, 2 * T / t))
for mode, name in ((0, "regs"), (1, "regs nontemporal"), (2, "LDS ring (LDS-DMA)")):
    for depth in ((1, 2, 4, 8, 16) if mode < 2 else (2, 4, 8, 16)):
        line = "%-20s depth %2d |" % (name, depth)
        for blocks in (256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32):
            if mode == 2 and 4 * depth * 1024 * (blocks // 256) > 160 * 1024 and False:
                continue
            ok = check(mode, depth, blocks)
            t = timeit(lambda i: lib.probe_copy(xs[i].data_ptr(), ys[i].data_ptr(), n4, mode, depth, blocks, st))
            line += " %5d wg %6.1fus %.2f%s |" % (blocks, t, 2 * T / t, "" if ok else " WRONG")
        print(line, flush=True)
