#!/bin/bash
# round 6, job 1: the search kernel's variants on M / Ms / U32, then the GPU tests
mkdir -p gpurun_out/j1
INC='k_leaf_search|k_leaf_samples|k_leaf_regs|k_regs_finalize|k_lane_reduce'
tools/kt_ab.sh M - 20 "$INC" intree build_ab/var/ls_ilp1.so build_ab/var/ls_ilp4.so build_ab/var/ls_i2b256.so build_ab/var/ls_i4b256.so build_ab/var/ls_i8b128.so 2>&1 | tee gpurun_out/j1/ab_M.txt
tools/kt_ab.sh Ms - 20 "$INC" intree build_ab/var/ls_ilp1.so build_ab/var/ls_i4b256.so 2>&1 | tee gpurun_out/j1/ab_Ms.txt
tools/kt_ab.sh U32 - 10 "$INC" intree build_ab/var/ls_ilp1.so build_ab/var/ls_i4b256.so 2>&1 | tee gpurun_out/j1/ab_U32.txt
tools/kt_ab.sh C3 - 10 "$INC" intree build_ab/var/ls_ilp1.so 2>&1 | tee gpurun_out/j1/ab_C3.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/j1/tests.txt
