#!/bin/bash
O=gpurun_out/r5v; mkdir -p $O
{
TAG=now timeout 120 python tools/cfg_run.py C5 - 20
TAG=single timeout 120 env RMI_HIP_LIB=build_ab/librmi_hip_n1w2.so python tools/cfg_run.py C5 - 20
TAG=now timeout 120 python tools/cfg_run.py C5 dups 20
TAG=single timeout 120 env RMI_HIP_LIB=build_ab/librmi_hip_n1w2.so python tools/cfg_run.py C5 dups 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
