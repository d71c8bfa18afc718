#!/bin/bash
# A/B timing of the metric configuration: build_ab/old.so (a previous build) against the in-tree library,
# interleaved on the same box.
P='import sys,json; d=json.loads(sys.stdin.read()); print("   ms/step %.4f" % d["ms_per_step"], {k: round(v) for k, v in d["roofline"]["kernel_us"].items()})'
for rep in 1 2 3; do
  echo "old"; RMI_HIP_LIB=$PWD/build_ab/old.so python bench.py --steps 100 --warmup 30 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "$P"
  echo "new"; python bench.py --steps 100 --warmup 30 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "$P"
done
