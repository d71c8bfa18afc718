#!/bin/bash
O=gpurun_out/r5u; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_regs.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
{
TAG=m python tools/cfg_run.py M
TAG=d python tools/cfg_run.py D
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
