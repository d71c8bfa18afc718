#!/bin/bash
mkdir -p gpurun_out/j7
INC='k_spline_scan|k_scan_gaps|k_lane_reduce|k_init'
python -m pytest tests/test_gpu_scan.py tests/test_gpu_regs.py tests/test_gpu_streamed.py -x -q 2>&1 | tail -4 | tee gpurun_out/j7/tests.txt
tools/kt_ab.sh C5 - 10 "$INC" intree 2>&1 | tee gpurun_out/j7/c5.txt
TAG=c5 python tools/cfg_run.py C5 - 50 2>&1 | grep -v Warn | tee gpurun_out/j7/c5_wall.txt
