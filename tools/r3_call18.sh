#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_irv2_gpu.py -m gpu -x -q > gpurun_out/r3_t13.log 2>&1; rc=$?; tail -5 gpurun_out/r3_t13.log; [ $rc -eq 0 ] || { tail -80 gpurun_out/r3_t13.log; exit $rc; }
for v in 1 0 1; do
SPNET_IR_MERGE_SIBLINGS=$v python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 --no-kernel-timers > gpurun_out/r3_l_irv2_ms$v.json 2> gpurun_out/r3_l_irv2_ms$v.err; echo "merge siblings $v rc=$?"; python -c "
import json; r=json.loads(open('gpurun_out/r3_l_irv2_ms$v.json').read()); print(r['value'], r['ms_per_step'])" || tail -5 gpurun_out/r3_l_irv2_ms$v.err
done
for v in 1 0; do
SPNET_IR_MERGE_SIBLINGS=$v python bench.py --backbone InceptionResNetV2 --mode predict --batch 16 --steps 50 --warmup 5 --no-kernel-timers 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('predict merge $v', r['value'], r['ms_per_step'])"
done
