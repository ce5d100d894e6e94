#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r3_t7.log 2>&1; rc=$?; tail -4 gpurun_out/r3_t7.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r3_t7.log; exit $rc; }
SECONDS=0
python bench.py > gpurun_out/r3_c_bench.json 2> gpurun_out/r3_c_bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r3_c_bench.json") if l.startswith("{")][-1])
print("value", r["value"], "ms", r["ms_per_step"], "sustained", r["sustained"]["images_per_sec"], "gemm", r["roofline"]["frac"], r["roofline"]["traffic"], "dw", r["roofline_secondary"]["frac"], r["roofline_secondary"]["traffic"])
print("predict", r["predict"]["frames_per_sec"], r["predict"]["host_streamed_frames_per_sec"], r["predict"]["host_streamed_u8_frames_per_sec"], "331", r["layout_331"]["train"]["images_per_sec"], r["layout_331"]["predict"]["images_per_sec"], "cpu", r["cpu_baseline"]["value"])
PY
