#!/bin/bash
export TMPDIR=/tmp
python tools/autotune_gemm.py 4 gpurun_out/tiles_irv2_pred.json 384 512 16 predict InceptionResNetV2 > gpurun_out/r3_o_autotune_irv2_pred.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/r3_o_autotune_irv2_pred.txt
python tools/autotune_gemm.py 4 gpurun_out/tiles_xc_train.json 384 512 32 train Xception > gpurun_out/r3_o_autotune_xc_train.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/r3_o_autotune_xc_train.txt
grep "\-> tile" gpurun_out/r3_o_autotune_xc_train.txt | cut -c1-70,150-260
