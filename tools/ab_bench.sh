# usage: ab_bench.sh ENVVAR [reps] -- A/B of an engine env switch on one box (ms per step, overlap / no-overlap)
V=$1
for i in $(seq ${2:-2}); do for d in 1 0; do
  a=$(env $V=$d python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $V=$d python bench.py --no-cpu-baseline --no-kernel-timers --steps 20 --no-overlap 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$V=$d overlap $a ms  no-overlap $b ms"
done; done
