set -o pipefail
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1 || { tail -30 gpurun_out/t_gemm.log; exit 1; }
tail -3 gpurun_out/t_gemm.log
python tools/gemm_table.py 5 > gpurun_out/gt9.log 2>&1
SPNET_HIP_LIB=$PWD/tools/var/libnogang.so python tools/gemm_table.py 5 > gpurun_out/gt9_nogang.log 2>&1
SPNET_HIP_LIB=$PWD/tools/libhead.so python tools/gemm_table.py 5 > gpurun_out/gt9_head.log 2>&1
head -30 gpurun_out/gt9.log; head -8 gpurun_out/gt9_nogang.log; head -5 gpurun_out/gt9_head.log
rm -f gpurun_out/ph9.log
export SPNET_HIP_LIB=$PWD/tools/var/libstamps.so
for a in "fwd 6144 728 728 6 stats" "dgrad 6144 728 728 6" "fwd 6144 728 2912 6"; do
  timeout -k 5 120 python tools/gemm_phases.py $a >> gpurun_out/ph9.log 2>&1
done
grep -v amdgpu.ids gpurun_out/ph9.log
