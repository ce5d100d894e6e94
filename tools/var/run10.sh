set -o pipefail
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/t10.log 2>&1 || { tail -40 gpurun_out/t10.log; exit 1; }
tail -3 gpurun_out/t10.log
for i in 1 2; do
for v in 1 0; do
  a=$(SPNET_BN_FOLD=$v timeout -k 5 200 python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 --sustained-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(SPNET_BN_FOLD=$v timeout -k 5 200 python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 --sustained-seconds 0 --no-overlap 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "fold=$v overlap $a ms  no-overlap $b ms" | tee -a gpurun_out/ab10.log
done; done
