#!/bin/bash
# round-3 GPU call: tests -> bench -> diagnostics -> profiles (each step only if the previous one passed)
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t2.log 2>&1; rc=$?; tail -4 gpurun_out/r3_t2.log; [ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r3_b_bench.json 2> gpurun_out/r3_b_bench.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || { tail -20 gpurun_out/r3_b_bench.err; exit $rc; }
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r3_b_bench.json") if l.startswith("{")][-1])
print("value", r["value"], "ms", r["ms_per_step"], "sustained", r["sustained"]["images_per_sec"])
print("gemm", r["roofline"]["frac"], r["roofline"]["ms_per_step"], "dw", r["roofline_secondary"]["frac"], r["roofline_secondary"]["ms_per_step"])
print("dw sub", json.dumps(r["roofline_secondary"].get("sub_families")))
print("predict", json.dumps({k: v for k, v in r["predict"].items() if "roofline" not in k and "note" not in k}))
print("331", json.dumps(r["layout_331"]))
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["sample"][-60:])
PY
SPNET_POOL_STATS=0 python bench.py --no-cpu-baseline --no-secondary --sustained-seconds 3 > gpurun_out/r3_b_bench_nopoolstats.json 2>/dev/null; echo "no-pool-stats rc=$?"
python tools/gemm_sweep.py nsweep > gpurun_out/r3_b_nsweep.txt 2>&1; cat gpurun_out/r3_b_nsweep.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_irv2 -- python3 bench.py --backbone InceptionResNetV2 --batch 16 --steps 5 --warmup 2 > gpurun_out/r3_b_irv2_under_rocprof.json 2> gpurun_out/r3_b_irv2.err; echo "irv2 prof rc=$?"
bash tools/profile_round.sh r03_a > gpurun_out/profile_r03_a.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/profile_r03_a.log
