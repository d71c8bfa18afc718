#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
{
TAG=stop5_noagg RMI_HIP_LIB=build_ab/librmi_hip_stop5.so python tools/cfg_run.py C5 - 20
TAG=stop6_nostore RMI_HIP_LIB=build_ab/librmi_hip_stop6.so python tools/cfg_run.py C5 - 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
