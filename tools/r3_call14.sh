#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_irv2_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "irv2 or inception or gemm or convolution" > gpurun_out/r3_t10.log 2>&1; rc=$?; tail -4 gpurun_out/r3_t10.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t10.log; exit $rc; }
for v in 1 0 1; do
SPNET_IR_ACC_EPILOGUE=$v python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 --no-kernel-timers > gpurun_out/r3_i_irv2_acc$v.json 2> gpurun_out/r3_i_irv2_acc$v.err; echo "acc epilogue $v rc=$?"; python -c "
import json; r=json.loads(open('gpurun_out/r3_i_irv2_acc$v.json').read()); print(r['value'], r['ms_per_step'])"
done
for i in 1 2; do python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_flake_$i.log 2>&1; echo "full suite run $i rc=$?"; tail -2 gpurun_out/r3_flake_$i.log; done
