#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
timeout -k 5 420 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
