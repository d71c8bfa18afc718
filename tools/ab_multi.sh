#!/bin/bash
# Timing of several builds of the library on the same box, interleaved: the in-tree one and every
# build_ab/var/*.so (built with an experiment macro).  usage: tools/ab_multi.sh [reps] [bench args...]
REPS=${1:-2}; shift || true
P='import sys,json; d=json.loads(sys.stdin.read()); print("   ms/step %.4f" % d["ms_per_step"], {k: round(v) for k, v in d["roofline"]["kernel_us"].items()})'
for rep in $(seq $REPS); do
  echo "base"; python bench.py --steps 100 --warmup 30 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "$P"
  for so in build_ab/var/*.so; do
    echo "$(basename $so .so)"; RMI_HIP_LIB=$PWD/$so python bench.py --steps 100 --warmup 30 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "$P"
  done
done
