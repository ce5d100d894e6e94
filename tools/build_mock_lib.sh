#!/bin/bash
# Dev tool: diagnostic builds of the GEMM.  The product gemm.hip carries no diagnostic code (round 4); the two
# instrumented variants live in tools/diag/gemm_diag.patch (SP_DWMOCK: the fused depthwise-in-GEMM A path as a timing
# mock, tools/dwfuse_mock.py; SP_STAMPS: per-workgroup phase stamps, tools/gemm_phases.py).  This script applies the
# patch to a temporary copy and builds tools/var/libdwmock.so / tools/var/libstamps.so (git-ignored; they travel to the
# GPU box with gpurun).  The patch was cut at the commit that removed the code; re-cut it if gemm.hip has moved on.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/spnet_amd/csrc" >/dev/null
mkdir -p "$ROOT/tools/var" /tmp/mockobj /tmp/mocksrc
cp "$ROOT"/spnet_amd/csrc/*.h "$ROOT"/spnet_amd/csrc/gemm.hip /tmp/mocksrc/
(cd /tmp/mocksrc && patch -p3 gemm.hip < "$ROOT/tools/diag/gemm_diag.patch")
OBJS=$(ls "$ROOT"/spnet_amd/lib/obj/*.o | grep -v '/gemm.o')
for V in DWMOCK:libdwmock STAMPS:libstamps; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I/tmp/mocksrc -I"$ROOT/include" -DSP_${V%%:*} -c /tmp/mocksrc/gemm.hip -o /tmp/mockobj/gemm.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/var/${V##*:}.so" /tmp/mockobj/gemm.o $OBJS
  echo built tools/var/${V##*:}.so
done
