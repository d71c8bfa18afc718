#!/bin/bash
# round 5, GPU call 1: pipeline 5 parity tests, C5 timings (pipeline 5 against 3, wave counts), profiles of C5, the other configurations' times
O=gpurun_out/r5a; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_scan.py -x -q > $O/scan_tests.log 2>&1; tail -3 $O/scan_tests.log
{
TAG=p5 python tools/cfg_run.py C5
TAG=p5 python tools/cfg_run.py C5 dups
TAG=p3 RMI_HIP_SCAN=0 python tools/cfg_run.py C5
TAG=p3 RMI_HIP_SCAN=0 python tools/cfg_run.py C5 dups
TAG=w1024 RMI_HIP_SCAN_WAVES=1024 python tools/cfg_run.py C5
TAG=w2048 RMI_HIP_SCAN_WAVES=2048 python tools/cfg_run.py C5
TAG=w4096 RMI_HIP_SCAN_WAVES=4096 python tools/cfg_run.py C5
TAG=base python tools/cfg_run.py M
TAG=base python tools/cfg_run.py C3
TAG=base python tools/cfg_run.py Ms
TAG=base python tools/cfg_run.py C4s
TAG=base python tools/cfg_run.py D
} > $O/times.log 2>&1
cat $O/times.log
tools/profile_r05.sh C5 > $O/prof_c5.log 2>&1; tail -30 $O/prof_c5.log
tools/profile_r05.sh C5 dups --no-pmc > $O/prof_c5d.log 2>&1; tail -8 $O/prof_c5d.log
