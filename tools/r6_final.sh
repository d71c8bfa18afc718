#!/bin/bash
# round 6: the final evidence run (GPU box): tests, smoke, the profiles of every configuration, the bench lines (behind the profiles: they quote the
# traffic files of the same kernel sources), the sweep over the shapes between the configurations
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/final/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
for c in "M -" "C3 -" "C4s -" "Ms -" "C5 -" "C5 dups" "U32 -" "S64 -" "C2 -"; do
  set -- $c
  RND=r06 timeout 900 tools/profile_cfg.sh $1 $2 > gpurun_out/final/prof_$1_$2.txt 2>&1
done
cp gpurun_out/prof_r06/*/traffic_r06_*.json profiles/ 2>/dev/null
python bench.py > gpurun_out/final/r06_bench.json 2> gpurun_out/final/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/final/r06_bench_driver_flags.json 2> gpurun_out/final/bench20.err
timeout 900 python tools/sweep_shapes.py u64 u32 f64 dups64 iid64 8 2>&1 | grep -v amdgpu.ids > gpurun_out/final/r06_sweep.txt
tail -3 gpurun_out/final/tests.txt; cat gpurun_out/final/smoke.txt | tail -1
