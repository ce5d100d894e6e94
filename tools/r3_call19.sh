#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > gpurun_out/r3_t14.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t14.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r3_t14.log; exit $rc; }
python tools/autotune_gemm.py 4 gpurun_out/tiles_irv2_train.json 384 512 16 train InceptionResNetV2 > gpurun_out/r3_m_autotune_irv2_train.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/r3_m_autotune_irv2_train.txt
grep -c "\-> tile" gpurun_out/r3_m_autotune_irv2_train.txt; grep "\-> tile" gpurun_out/r3_m_autotune_irv2_train.txt | cut -c1-60,140-260 | head -40
