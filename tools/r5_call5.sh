#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_scan.py -x -q > $O/scan_tests.log 2>&1; tail -5 $O/scan_tests.log
{
TAG=p5 python tools/cfg_run.py C5
TAG=p5 python tools/cfg_run.py C5 dups
TAG=nofast RMI_HIP_LIB=build_ab/librmi_hip_nofast.so python tools/cfg_run.py C5
TAG=nofast RMI_HIP_LIB=build_ab/librmi_hip_nofast.so python tools/cfg_run.py C5 dups
TAG=wpe3 RMI_HIP_LIB=build_ab/librmi_hip_wpe3.so RMI_HIP_SCAN_WAVES=3072 python tools/cfg_run.py C5
TAG=wpe3 RMI_HIP_LIB=build_ab/librmi_hip_wpe3.so RMI_HIP_SCAN_WAVES=3072 python tools/cfg_run.py C5 dups
TAG=stop4 RMI_HIP_LIB=build_ab/librmi_hip_stop4.so python tools/cfg_run.py C5
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
