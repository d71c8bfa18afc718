#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
{
for v in stop1 stop3 stop4; do TAG=$v RMI_HIP_LIB=build_ab/librmi_hip_$v.so python tools/cfg_run.py C5 - 20; done
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
tools/profile_r05.sh C5 > $O/prof_c5.log 2>&1; tail -30 $O/prof_c5.log
grep -A10 "k_spline_scan (disp" gpurun_out/prof_r05/c5/r05_c5_summary.txt
