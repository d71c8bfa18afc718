#!/bin/bash
# BASELINE.json configs on one GPU (timing only; parity is covered by the tests at smaller sizes)
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"]); print("   keys/s %.4g  ms/step %.3f  device_us %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["pipeline_device_us"]), d["roofline"]["kernel_us"])'
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec linear,linear --leaves 262144 2>&1 | tail -1 | python -c "$P"
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec cubic,linear --leaves 1048576 2>&1 | tail -1 | python -c "$P"
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec linear,linear --leaves 2097152 --keys 800000000 2>&1 | tail -1 | python -c "$P"
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec radix,linear_spline --leaves 4194304 --keys 400000000 --dtype uint32 2>&1 | tail -1 | python -c "$P"
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec linear,linear --leaves 1048576 --dataset dups 2>&1 | tail -1 | python -c "$P"
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec linear,linear --leaves 262144 --dataset books 2>&1 | tail -1 | python -c "$P"
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --spec linear,linear --leaves 1048576 --dataset books 2>&1 | tail -1 | python -c "$P"
