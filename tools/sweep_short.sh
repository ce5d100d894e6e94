#!/bin/bash
# quick A/B of GEMM build variants on the dominant shapes (dev tool)
for lib in "" tools/libvar_st18.so tools/libvar_st36.so tools/libvar_st72.so; do
  echo "== lib=${lib:-default}"
  for args in "fwd 6144 728 728 5" "dgrad 6144 728 728 5" "fwd 24576 728 728 5"; do
    SPNET_HIP_LIB=${lib:+$PWD/$lib} python tools/gemm_sweep.py one $args 2>&1 | grep "^one"
  done
done
