#!/bin/bash
O=gpurun_out/r5r; mkdir -p $O
{
for v in n2w2 n1w2 n1w3; do TAG=$v RMI_HIP_LIB=build_ab/librmi_hip_$v.so python tools/cfg_run.py C5 - 20; done
for v in n2w2 n1w2 n1w3; do TAG=$v RMI_HIP_LIB=build_ab/librmi_hip_$v.so python tools/cfg_run.py C5 dups 20; done
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
