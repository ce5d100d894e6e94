#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for v in 1 0 1; do
SPNET_GEMM_TILES=$v python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 --no-kernel-timers > gpurun_out/r3_n_irv2_tt$v.json 2> gpurun_out/r3_n_irv2_tt$v.err; echo "tile table $v rc=$?"; python -c "
import json; r=json.loads(open('gpurun_out/r3_n_irv2_tt$v.json').read()); print(r['value'], r['ms_per_step'])"
done
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t15.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t15.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t15.log; exit $rc; }
python bench.py --no-cpu-baseline --no-secondary --sustained-seconds 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('xception', r['value'], r['ms_per_step'])"
