#!/bin/bash
# usage (GPU box): tools/regs_pmc.sh <tag> [lib.so]  -> the SQ wait / active counters of k_leaf_regs for the in-tree library or an experiment build
TAG=${1:-rgpmc}; LIB=${2:-}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
if [ -n "$LIB" ]; then export RMI_AB_ONLY=$(basename $LIB .so); else export RMI_AB_ONLY=in-tree; fi
RMI_HIP_REGS=1 timeout -k 5 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
   --kernel-include-regex "k_leaf_regs" -d $OUT/pmc -o p -f csv -- python $GRAFT_REPO_ROOT/tools/regs_ab.py 200000000 1048576 2 < /dev/null > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/summarize_prof.py $OUT 2>&1 < /dev/null | grep -A9 "k_leaf_regs"
