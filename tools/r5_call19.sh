#!/bin/bash
O=gpurun_out/r5s; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_scan.py tests/test_gpu_sharded.py tests/test_gpu_cli.py tests/test_gpu_streamed.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
{
TAG=n1 python tools/cfg_run.py C5
TAG=n1 python tools/cfg_run.py C5 dups
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
