export SPNET_HIP_LIB=$PWD/tools/var/libstamps.so
for a in "fwd 6144 728 728 6" "fwd 6144 728 728 6 zeros" "fwd 6144 728 5824 6" "fwd 6144 728 5824 6 zeros"; do
  timeout -k 5 120 python tools/gemm_phases.py $a >> gpurun_out/ph3.log 2>&1
done
grep -v amdgpu.ids gpurun_out/ph3.log
