#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
{
TAG=lean2 RMI_HIP_LIB=build_ab/librmi_hip_lean2.so python tools/cfg_run.py C5 - 20
TAG=lean3 RMI_HIP_LIB=build_ab/librmi_hip_lean3.so python tools/cfg_run.py C5 - 20
TAG=lean3 RMI_HIP_LIB=build_ab/librmi_hip_lean3.so python tools/cfg_run.py C5 dups 20
TAG=lean3_2048 RMI_HIP_SCAN_WAVES=2048 RMI_HIP_LIB=build_ab/librmi_hip_lean3.so python tools/cfg_run.py C5 - 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
