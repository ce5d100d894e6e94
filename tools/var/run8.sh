for d in 4 6 12 3 5 2 1; do
  echo "== SP_DBG=$d (1 no frag reads, 2 no LDS stores, 4 no global loads, 8 no barrier)"
  SPNET_HIP_LIB=$PWD/tools/var/libdbg$d.so timeout -k 5 100 python tools/gemm_sweep.py one fwd 6144 728 2912 6 2>&1 | grep "^one"
done
echo "== full"; python tools/gemm_sweep.py one fwd 6144 728 2912 6 2>&1 | grep "^one"
