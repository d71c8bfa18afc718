#!/bin/bash
# usage (GPU box): tools/regs_prof.sh <tag> [variants...]  -> gpurun_out/<tag>/kernel_stats_*.csv: kernel trace of a few trainings through
# pipeline 4 (and pipeline 3 beside it: the boxes of the pool differ by 7 %).  Every step under `timeout -k`, nothing reads stdin.
TAG=${1:-rgprof}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 5 100 rocprofv3 --kernel-trace --stats -T -d $OUT/kt_$name -o kt -f csv -- python $GRAFT_REPO_ROOT/tools/regs_ab.py 200000000 1048576 20 < /dev/null > $OUT/kt_$name.log 2>&1
  local F=$(find $OUT/kt_$name -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$F" ]; then cp "$F" $OUT/kernel_stats_$name.csv; echo "== $name"; grep "k_leaf_regs\|k_leaf_lanes\|k_leaf_search\|k_regs_finalize\|k_lane_reduce" $OUT/kernel_stats_$name.csv | cut -d, -f1,2,4 ; else echo "== $name: no kernel stats"; tail -3 $OUT/kt_$name.log; fi
}
run lanes RMI_HIP_REGS=0 RMI_AB_ONLY=in-tree
run regs RMI_HIP_REGS=1 RMI_AB_ONLY=in-tree
for v in "$@"; do run "$v" RMI_HIP_REGS=1 RMI_AB_ONLY=$v; done
