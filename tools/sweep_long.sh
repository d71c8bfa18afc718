#!/bin/bash
# books-shaped data: hand-over threshold of pass A (leaves longer than this go to k_fit_long)
P='import sys,json; d=json.loads(sys.stdin.read()); print("   ms/step %.3f" % d["ms_per_step"], {k: round(v) for k, v in d["roofline"]["kernel_us"].items()})'
for lm in 1024 4096; do
  for L in 262144 1048576; do
    echo "long_min=$lm L=$L"
    RMI_HIP_LONG_MIN=$lm python bench.py --steps 3 --warmup 1 --no-cpu-baseline --spec linear,linear --leaves $L --dataset books 2>&1 | tail -1 | python -c "$P"
  done
done
echo "L=1024 uniform"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --spec linear,linear --leaves 1024 2>&1 | tail -1 | python -c "$P"
echo "L=65536 uniform"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --spec linear,linear --leaves 65536 2>&1 | tail -1 | python -c "$P"
echo "L=2^20 uniform"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
