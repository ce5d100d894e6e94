# usage: ab_bench.sh ENVVAR [reps] -- A/B of an engine env switch on one box (ms per step, overlap / no-overlap)
# Every bench run is bounded by `timeout`, and results are appended to gpurun_out/ab.log as they come.
V=$1
mkdir -p gpurun_out
for i in $(seq ${2:-2}); do for d in 1 0; do
  a=$(env $V=$d timeout -k 5 120 python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $V=$d timeout -k 5 120 python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 --no-overlap 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$V=$d overlap $a ms  no-overlap $b ms" | tee -a gpurun_out/ab.log
done; done
