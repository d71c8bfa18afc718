#!/bin/bash
O=gpurun_out/r5y; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_gpu_regs.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
{
TAG=now timeout 120 python tools/cfg_run.py C3 - 20
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
