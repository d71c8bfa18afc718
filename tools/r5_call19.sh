#!/bin/bash
O=gpurun_out/r5s; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_scan.py tests/test_gpu_sharded.py tests/test_gpu_cli.py tests/test_gpu_streamed.py -m gpu -q -x -k "not eight_ranks" > $O/tests.log 2>&1; tail -3 $O/tests.log
