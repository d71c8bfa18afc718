#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_scan.py -x -q > $O/scan_tests.log 2>&1; tail -5 $O/scan_tests.log
{
TAG=p5 python tools/cfg_run.py C5
TAG=p5 python tools/cfg_run.py C5 dups
TAG=p5_w1024 RMI_HIP_SCAN_WAVES=1024 python tools/cfg_run.py C5
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
