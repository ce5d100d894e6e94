#!/bin/bash
# Dev tool: build the library of the last COMMIT into tools/libhead.so (A/B against the working tree with
# SPNET_HIP_LIB=$PWD/tools/libhead.so; only valid while the C ABI is unchanged).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive HEAD spnet_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/spnet_amd/csrc" OUT="$ROOT/tools/libhead.so" >/dev/null
rm -rf "$TMP"
echo built tools/libhead.so
