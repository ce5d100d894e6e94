#!/bin/bash
# Dev tool: build the WORKING TREE's library with extra compiler flags, and optionally a patch from tools/diag/ applied, into
# tools/lib<name>.so (A/B on one box with SPNET_HIP_LIB=$PWD/tools/lib<name>.so, tools/ab_libs.sh).
# usage: build_variant_lib.sh NAME "-DFLAG ..." [tools/diag/x3_roles.patch]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/spnet_amd" && cp -r "$ROOT/spnet_amd/csrc" "$TMP/spnet_amd/csrc" && cp -r "$ROOT/include" "$TMP/include"
mkdir -p "$TMP/spnet_amd/lib"
if [ -n "$3" ]; then (cd "$TMP" && patch -p1 < "$ROOT/$3"); fi
make -C "$TMP/spnet_amd/csrc" OUT="$ROOT/tools/lib$1.so" FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I. -I../../include $2" >/dev/null
rm -rf "$TMP"
echo built tools/lib$1.so
