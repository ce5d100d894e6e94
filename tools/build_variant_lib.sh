#!/bin/bash
# Dev tool: build the WORKING TREE's library with extra compiler flags into tools/lib<name>.so (A/B on one box with
# SPNET_HIP_LIB=$PWD/tools/lib<name>.so).  usage: build_variant_lib.sh NAME "-DFLAG ..."
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/spnet_amd" && cp -r "$ROOT/spnet_amd/csrc" "$TMP/spnet_amd/csrc" && cp -r "$ROOT/include" "$TMP/include"
mkdir -p "$TMP/spnet_amd/lib"
make -C "$TMP/spnet_amd/csrc" OUT="$ROOT/tools/lib$1.so" FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I. -I../../include $2" >/dev/null
rm -rf "$TMP"
echo built tools/lib$1.so
