#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t16.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t16.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t16.log; exit $rc; }
bash tools/profile_round.sh r03_e > gpurun_out/profile_r03_e.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/profile_r03_e.log
cp gpurun_out/profiles_r03_e/hbm_traffic_current.json profiles/hbm_traffic_current.json
python bench.py --sustained-seconds 30 > gpurun_out/r3_f_bench.json 2> gpurun_out/r3_f_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r3_f_bench.json") if l.startswith("{")][-1])
print("value", r["value"], "ms", r["ms_per_step"], "sustained", r["sustained"], "gemm", r["roofline"]["frac"], r["roofline"]["traffic"], "dw", r["roofline_secondary"]["frac"], r["roofline_secondary"]["traffic"])
print("predict", r["predict"]["frames_per_sec"], r["predict"]["host_streamed_frames_per_sec"], r["predict"]["host_streamed_u8_frames_per_sec"], "331", r["layout_331"]["train"]["images_per_sec"], r["layout_331"]["predict"]["images_per_sec"], "cpu", r["cpu_baseline"]["value"])
PY
python bench.py --backbone MobileNet --batch 8 --steps 50 --warmup 5 > gpurun_out/r3_f_mobilenet_bench.json 2>/dev/null; cat gpurun_out/r3_f_mobilenet_bench.json | cut -c1-260
