#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
{
TAG=base python tools/cfg_run.py C5 - 20
TAG=bf RMI_HIP_LIB=build_ab/librmi_hip_bf.so python tools/cfg_run.py C5 - 20
TAG=bf RMI_HIP_LIB=build_ab/librmi_hip_bf.so python tools/cfg_run.py C5 dups 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
