#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_irv2_gpu.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; rc=$?; tail -6 gpurun_out/r3_t8.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t8.log; exit $rc; }
for v in 1 0 1; do
SPNET_IR_FUSE_BNSUMS=$v python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 --no-kernel-timers > gpurun_out/r3_h_irv2_fs$v.json 2> gpurun_out/r3_h_irv2_fs$v.err; echo "fused bn sums $v rc=$?"; python -c "
import json; r=json.loads(open('gpurun_out/r3_h_irv2_fs$v.json').read()); print(r['value'], r['ms_per_step'])"
done
