#!/bin/bash
# quick A/B of GEMM build variants on the dominant shapes (dev tool)
for lib in "" tools/libvar_storeend.so; do
  echo "== lib=${lib:-default}"
  for args in "fwd 6144 728 728 5" "dgrad 6144 728 728 5" "wgrad 728 728 6144 5" "fwd 24576 728 728 5" "fwd 372000 128 128 3" "fwd 372000 128 128 1" "fwd 94752 256 256 3"; do
    SPNET_HIP_LIB=${lib:+$PWD/$lib} python tools/gemm_sweep.py one $args 2>&1 | grep "^one"
  done
done
