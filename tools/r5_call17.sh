#!/bin/bash
O=gpurun_out/r5q; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_scan.py tests/test_gpu_cli.py tests/test_golden.py tests/test_gpu_streamed.py -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
