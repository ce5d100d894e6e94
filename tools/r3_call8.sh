#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py tests/test_irv2_gpu.py -m gpu -x -q -k "gemm or irv2 or inception" > gpurun_out/r3_t5.log 2>&1; rc=$?; tail -6 gpurun_out/r3_t5.log; [ $rc -eq 0 ] || exit $rc
for v in 1 0 1; do
SPNET_IR_BATCH_WGRAD=$v python bench.py --backbone InceptionResNetV2 --batch 16 --steps 30 --warmup 5 > gpurun_out/r3_g_irv2_bw$v.json 2> gpurun_out/r3_g_irv2_bw$v.err; echo "batched wgrad $v rc=$?"; cat gpurun_out/r3_g_irv2_bw$v.json
done
