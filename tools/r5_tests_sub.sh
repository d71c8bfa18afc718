#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sigma.py tests/test_gpu_streamed.py -m gpu -x -q --durations=5 > $O/gpu_tests_sub.log 2>&1; tail -12 $O/gpu_tests_sub.log
