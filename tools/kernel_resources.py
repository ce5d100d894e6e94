#!/usr/bin/env python3
"""Dev tool: VGPR / scratch / occupancy / LDS table of every kernel in one HIP source
(hipcc -Rpass-analysis=kernel-resource-usage).   usage: kernel_resources.py csrc/gemm.hip [name filter]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I", os.path.dirname(src), "-I", inc,
                    "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print("%-90s VGPR %3d AGPR %3d scratch %4d occ %d LDS %6d SGPR %3d" % (k[-90:], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1),
                                                                        v.get("Occupancy", -1), v.get("LDS", -1), v.get("TotalSGPRs", -1)))
