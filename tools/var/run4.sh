set -o pipefail
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1 || { tail -30 gpurun_out/t_gemm.log; exit 1; }
tail -3 gpurun_out/t_gemm.log
python tools/gemm_table.py 5 > gpurun_out/gt4.log 2>&1
SPNET_HIP_LIB=$PWD/tools/libhead.so python tools/gemm_table.py 5 > gpurun_out/gt4_head.log 2>&1
head -14 gpurun_out/gt4.log; head -5 gpurun_out/gt4_head.log
export SPNET_HIP_LIB=$PWD/tools/var/libstamps.so
rm -f gpurun_out/ph4.log
for a in "fwd 6144 728 728 6 stats" "dgrad 6144 728 728 6" "fwd 6144 728 728 5" "fwd 6144 728 5824 6"; do
  timeout -k 5 120 python tools/gemm_phases.py $a >> gpurun_out/ph4.log 2>&1
done
grep -v amdgpu.ids gpurun_out/ph4.log
