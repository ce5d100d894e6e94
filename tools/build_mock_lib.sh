#!/bin/bash
# Dev tool: diagnostic build of the library with gemm.hip compiled -DSP_DWMOCK (the fused depthwise-in-GEMM A path as a
# timing mock, tools/dwfuse_mock.py) -> tools/var/libdwmock.so (git-ignored; travels to the GPU box with gpurun).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/spnet_amd/csrc" >/dev/null
mkdir -p "$ROOT/tools/var" /tmp/mockobj
cd "$ROOT/spnet_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I. -I../../include -DSP_DWMOCK -c gemm.hip -o /tmp/mockobj/gemm.o 2>/dev/null
OBJS=$(ls ../lib/obj/*.o | grep -v '/gemm.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/var/libdwmock.so" /tmp/mockobj/gemm.o $OBJS
echo built tools/var/libdwmock.so
