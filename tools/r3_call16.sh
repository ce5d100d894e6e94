#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_irv2_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "irv2 or inception or batchnorm or convolution or column" > gpurun_out/r3_t11.log 2>&1; rc=$?; tail -4 gpurun_out/r3_t11.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t11.log; exit $rc; }
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do
set -- $cfg
SPNET_IR_DIRECT_CONCAT=$1 SPNET_IR_IMPLICIT_FWD=$2 python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 --no-kernel-timers > gpurun_out/r3_k_irv2_$1_$2.json 2> gpurun_out/r3_k_irv2_$1_$2.err; echo "direct concat $1 implicit fwd $2 rc=$?"; python -c "
import json; r=json.loads(open('gpurun_out/r3_k_irv2_$1_$2.json').read()); print(r['value'], r['ms_per_step'])"
done
for v in 1 0; do
SPNET_IR_IMPLICIT_FWD=$v python bench.py --backbone InceptionResNetV2 --mode predict --batch 16 --steps 50 --warmup 5 --no-kernel-timers 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('predict implicit $v', r['value'], r['ms_per_step'])"
done
