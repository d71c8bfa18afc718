#!/bin/bash
mkdir -p gpurun_out/j6
INC='k_spline_scan|k_scan_gaps|k_lane_reduce|k_init'
python -m pytest tests/test_gpu_scan.py -x -q 2>&1 | tail -4 | tee gpurun_out/j6/tests_scan.txt
tools/kt_ab.sh C5 - 10 "$INC" intree build_ab/librmi_hip_fb0.so intree build_ab/librmi_hip_fb0.so 2>&1 | tee gpurun_out/j6/ab_c5.txt
tools/kt_ab.sh C5 dups 10 "$INC" intree build_ab/librmi_hip_fb0.so 2>&1 | tee gpurun_out/j6/ab_c5_dups.txt
