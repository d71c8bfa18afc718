#!/bin/bash
# round 6, job 2: 4-byte keys on the register kernel at two waves per SIMD
mkdir -p gpurun_out/j2
INC='k_leaf_search|k_leaf_regs|k_regs_finalize|k_leaf_lanes'
python -m pytest tests/test_gpu_regs.py -x -q -k "u32" 2>&1 | tail -5 | tee gpurun_out/j2/tests_regs_u32.txt
RMI_HIP_REGS_U32=1 tools/kt_ab.sh U32 - 10 "$INC" intree 2>&1 | tee gpurun_out/j2/ab_U32_w1.txt
tools/kt_ab.sh U32 - 10 "$INC" intree build_ab/var/w2_160.so build_ab/var/w2_176.so 2>&1 | tee gpurun_out/j2/ab_U32_w2.txt
tools/kt_ab.sh U32r - 10 "$INC" intree build_ab/var/w2_160.so 2>&1 | tee gpurun_out/j2/ab_U32r_w2.txt
for v in 1 2; do RMI_HIP_REGS_U32=$v TAG=u32_$v python tools/cfg_run.py U32 - 30; done 2>&1 | grep -v Warn | tee gpurun_out/j2/wall.txt
RMI_HIP_LIB=$PWD/build_ab/var/w2_160.so TAG=w2_160 python tools/cfg_run.py U32 - 30 2>&1 | grep -v Warn | tee -a gpurun_out/j2/wall.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanes.py tests/test_gpu_regs.py -x -q 2>&1 | tail -5 | tee gpurun_out/j2/tests.txt
