#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_regs.py tests/test_gpu_lanes.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q > $O/tests.log 2>&1; tail -6 $O/tests.log
