#!/bin/bash
O=gpurun_out/r5p; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_regs.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q > $O/tests.log 2>&1; tail -6 $O/tests.log
{
TAG=new python tools/cfg_run.py C3
TAG=margin0 RMI_HIP_CUBIC_MARGIN=0 python tools/cfg_run.py C3
TAG=new python tools/cfg_run.py M
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
