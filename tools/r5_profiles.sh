#!/bin/bash
# round 5: the profiles of every BASELINE configuration on one GPU (kernel trace, counters, traffic), then the bench line
O=gpurun_out/r5prof; mkdir -p $O
for c in "M -" "C5 -" "C5 dups" "C3 -" "C2 -"; do set -- $c; tools/profile_r05.sh $1 $2 > $O/prof_$1_$2.log 2>&1; tail -3 $O/prof_$1_$2.log; done
