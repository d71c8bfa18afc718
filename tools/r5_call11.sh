#!/bin/bash
O=gpurun_out/r5k; mkdir -p $O
{
python tools/bw_both.py
for v in stop5 stop6 nont; do TAG=$v RMI_HIP_LIB=build_ab/librmi_hip_$v.so python tools/cfg_run.py C5 - 20; done
TAG=base python tools/cfg_run.py C5 - 20
TAG=w1536 RMI_HIP_SCAN_WAVES=1536 python tools/cfg_run.py C5 - 20
TAG=w1280 RMI_HIP_SCAN_WAVES=1280 python tools/cfg_run.py C5 - 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
