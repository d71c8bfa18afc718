#!/bin/bash
A="--spec radix,linear_spline --leaves 4194304 --keys 400000000 --dtype uint32"
P='import sys,json; d=json.loads(sys.stdin.read()); print("   ms/step %.4f" % d["ms_per_step"], {k: round(v) for k, v in d["roofline"]["kernel_us"].items()})'
for l in c1 c2 c4; do echo $l; RMI_HIP_LIB=$PWD/build_ab/$l.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline $A 2>&1 | tail -1 | python -c "$P"; done
