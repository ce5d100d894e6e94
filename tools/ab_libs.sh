# usage: ab_libs.sh [lib.so ...] -- working-tree library vs the given ones on one box (ms per step, three rounds)
mkdir -p gpurun_out
for i in 1 2 3; do for v in new "$@"; do
  if [ $v = new ]; then unset SPNET_HIP_LIB; else export SPNET_HIP_LIB=$PWD/$v; fi
  a=$(timeout -k 5 120 python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$v: $a ms" | tee -a gpurun_out/ab.log
done; done
