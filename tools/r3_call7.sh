#!/bin/bash
export TMPDIR=/tmp
python tools/dwfuse_mock.py > gpurun_out/r3_f_dwfuse.txt 2>&1; echo rc=$?
SPNET_HIP_LIB=$PWD/tools/var/libdwmock.so python tools/dwfuse_mock.py mock >> gpurun_out/r3_f_dwfuse.txt 2>&1; echo rc=$?
python tools/dwfuse_mock.py >> gpurun_out/r3_f_dwfuse.txt 2>&1
SPNET_HIP_LIB=$PWD/tools/var/libdwmock.so python tools/dwfuse_mock.py mock >> gpurun_out/r3_f_dwfuse.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3_f_dwfuse.txt
