#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_bench_geometry_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/r3_t3.log 2>&1; rc=$?; tail -4 gpurun_out/r3_t3.log; [ $rc -eq 0 ] || exit $rc
for i in 1 2; do
python bench.py --no-cpu-baseline --no-secondary --sustained-seconds 3 > gpurun_out/r3_c_bench_$i.json 2>/dev/null; echo "bench rc=$?"
SPNET_POOL_STATS=0 python bench.py --no-cpu-baseline --no-secondary --sustained-seconds 3 > gpurun_out/r3_c_bench_nopool_$i.json 2>/dev/null; echo "nopool rc=$?"
done
python - <<'PY'
import json
for f in ("r3_c_bench_1","r3_c_bench_nopool_1","r3_c_bench_2","r3_c_bench_nopool_2"):
    r = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
    print(f, r["value"], r["ms_per_step"], r["sustained"]["images_per_sec"])
PY
bash tools/profile_round.sh r03_b > gpurun_out/profile_r03_b.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/profile_r03_b.log
