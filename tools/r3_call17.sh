#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t12.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t12.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t12.log; exit $rc; }
bash tools/profile_round.sh r03_d > gpurun_out/profile_r03_d.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/profile_r03_d.log
cp gpurun_out/profiles_r03_d/hbm_traffic_current.json profiles/hbm_traffic_current.json
python bench.py --sustained-seconds 30 > gpurun_out/r3_e_bench.json 2> gpurun_out/r3_e_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r3_e_bench.json") if l.startswith("{")][-1])
print("value", r["value"], "ms", r["ms_per_step"], "sustained", r["sustained"], "gemm", r["roofline"]["frac"], r["roofline"]["traffic"], "dw", r["roofline_secondary"]["frac"], r["roofline_secondary"]["traffic"])
print("predict", r["predict"]["frames_per_sec"], r["predict"]["host_streamed_frames_per_sec"], r["predict"]["host_streamed_u8_frames_per_sec"], "331", r["layout_331"]["train"]["images_per_sec"], r["layout_331"]["predict"]["images_per_sec"], "cpu", r["cpu_baseline"]["value"])
PY
python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 > gpurun_out/r3_e_irv2_bench.json 2>/dev/null; cat gpurun_out/r3_e_irv2_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_e_irv2 -- python3 bench.py --backbone InceptionResNetV2 --batch 16 --steps 5 --warmup 2 --no-kernel-timers > gpurun_out/r3_e_irv2_under_rocprof.json 2> gpurun_out/r3_e_irv2.err; echo "irv2 prof rc=$?"
