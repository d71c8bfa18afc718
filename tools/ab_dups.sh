#!/bin/bash
A="--dataset dups"
P='import sys,json; d=json.loads(sys.stdin.read()); print("   ms/step %.4f" % d["ms_per_step"], {k: round(v) for k, v in d["roofline"]["kernel_us"].items()})'
for rep in 1 2; do
  echo "old"; RMI_HIP_LIB=$PWD/build_ab/old.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline $A 2>&1 | tail -1 | python -c "$P"
  echo "new"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline $A 2>&1 | tail -1 | python -c "$P"
done
