# usage: ab_lib.sh  -- working-tree library vs tools/libhead.so on one box (ms per step)
mkdir -p gpurun_out
for i in 1 2 3; do for v in new head; do
  if [ $v = head ]; then export SPNET_HIP_LIB=$PWD/tools/libhead.so; else unset SPNET_HIP_LIB; fi
  a=$(timeout -k 5 120 python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$v: $a ms" | tee -a gpurun_out/ab.log
done; done
