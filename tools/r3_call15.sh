#!/bin/bash
export TMPDIR=/tmp
python tools/autotune_gemm.py 4 gpurun_out/tiles_331_16.json 331 331 16 train > gpurun_out/r3_j_autotune_331_16.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3_j_autotune_331_16.txt
python tools/autotune_gemm.py 4 gpurun_out/tiles_pred128.json 384 512 128 predict > gpurun_out/r3_j_autotune_pred128.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3_j_autotune_pred128.txt
python tools/autotune_gemm.py 4 gpurun_out/tiles_pred331.json 331 331 32 predict > gpurun_out/r3_j_autotune_pred331.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3_j_autotune_pred331.txt
grep -h "\-> tile" gpurun_out/r3_j_autotune_*.txt | cut -c1-200
