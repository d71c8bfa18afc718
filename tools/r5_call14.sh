#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_regs.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
{
TAG=new python tools/cfg_run.py D
TAG=regs0 RMI_HIP_REGS=0 python tools/cfg_run.py D
TAG=new python tools/cfg_run.py M
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
