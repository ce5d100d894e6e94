#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t6.log 2>&1; rc=$?; tail -4 gpurun_out/r3_t6.log; [ $rc -eq 0 ] || exit $rc
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
/usr/bin/time -v python bench.py > gpurun_out/r3_c_bench.json 2> gpurun_out/r3_c_bench.err; echo "bench rc=$?"; grep -E "Elapsed|Maximum resident" gpurun_out/r3_c_bench.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r3_c_bench.json") if l.startswith("{")][-1])
print("value", r["value"], "ms", r["ms_per_step"], "sustained", r["sustained"]["images_per_sec"], "gemm", r["roofline"]["frac"], "dw", r["roofline_secondary"]["frac"])
print("predict", r["predict"]["frames_per_sec"], r["predict"]["host_streamed_frames_per_sec"], r["predict"]["host_streamed_u8_frames_per_sec"], "331", r["layout_331"]["train"]["images_per_sec"], r["layout_331"]["predict"]["images_per_sec"], "cpu", r["cpu_baseline"]["value"])
PY
python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 > gpurun_out/r3_c_irv2_bench.json 2>/dev/null; cat gpurun_out/r3_c_irv2_bench.json
python bench.py --backbone MobileNet --batch 8 --steps 50 --warmup 5 > gpurun_out/r3_c_mobilenet_bench.json 2>/dev/null; cat gpurun_out/r3_c_mobilenet_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c_irv2 -- python3 bench.py --backbone InceptionResNetV2 --batch 16 --steps 5 --warmup 2 --no-kernel-timers > gpurun_out/r3_c_irv2_under_rocprof.json 2> gpurun_out/r3_c_irv2.err; echo "irv2 prof rc=$?"
bash tools/profile_round.sh r03_c > gpurun_out/profile_r03_c.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/profile_r03_c.log
