#!/bin/bash
# The in-tree library under environment settings given as arguments ("VAR=val" or "VAR=val,VAR2=val2"), interleaved with the default.
P='import sys,json; d=json.loads(sys.stdin.read()); print("   ms/step %.4f" % d["ms_per_step"], {k: round(v) for k, v in d["roofline"]["kernel_us"].items()})'
for rep in 1 2; do
  echo "default"; python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  for e in "$@"; do echo "$e"; env ${e//,/ } python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; done
done
