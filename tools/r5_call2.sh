#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
{
TAG=p5 python tools/cfg_run.py C5
TAG=p5 python tools/cfg_run.py C5 dups
TAG=p3 RMI_HIP_SCAN=0 python tools/cfg_run.py C5
TAG=p3 RMI_HIP_SCAN=0 python tools/cfg_run.py C5 dups
TAG=wpe2 RMI_HIP_LIB=build_ab/librmi_hip_wpe2.so RMI_HIP_SCAN_WAVES=2048 python tools/cfg_run.py C5
TAG=wpe2 RMI_HIP_LIB=build_ab/librmi_hip_wpe2.so RMI_HIP_SCAN_WAVES=2048 python tools/cfg_run.py C5 dups
TAG=wpe2_1024 RMI_HIP_LIB=build_ab/librmi_hip_wpe2.so RMI_HIP_SCAN_WAVES=1024 python tools/cfg_run.py C5
TAG=base python tools/cfg_run.py M
TAG=base python tools/cfg_run.py C3
TAG=base python tools/cfg_run.py Ms
TAG=base python tools/cfg_run.py C4s
TAG=base python tools/cfg_run.py D
python tools/scan_dbg2.py
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
