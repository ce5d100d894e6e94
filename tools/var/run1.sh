python tools/gemm_table.py 5 > gpurun_out/gt1.log 2>&1
export SPNET_HIP_LIB=$PWD/tools/var/libstamps.so
for a in "fwd 6144 728 728 6 stats" "fwd 6144 728 728 6" "fwd 6144 728 728 5" "dgrad 6144 728 728 6" "fwd 6144 728 2912 6" "fwd 24576 728 728 6 stats" "wgrad 728 728 6144 5"; do
  timeout -k 5 120 python tools/gemm_phases.py $a >> gpurun_out/ph1.log 2>&1
done
