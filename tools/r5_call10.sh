#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
tools/profile_r05.sh C5 > $O/prof_c5.log 2>&1
grep -A9 "k_spline_scan (disp" gpurun_out/prof_r05/c5/r05_c5_summary.txt | head -44; tail -3 $O/prof_c5.log
