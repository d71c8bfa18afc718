#!/bin/bash
O=gpurun_out/r5v; mkdir -p $O
{
TAG=split python tools/cfg_run.py C5 dups 20
TAG=lean3 RMI_HIP_LIB=build_ab/librmi_hip_lean3.so python tools/cfg_run.py C5 dups 20
TAG=single RMI_HIP_LIB=build_ab/librmi_hip_n1w2.so python tools/cfg_run.py C5 dups 20
TAG=single RMI_HIP_LIB=build_ab/librmi_hip_n1w2.so python tools/cfg_run.py C5 - 20
TAG=split python tools/cfg_run.py C5 - 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
