#!/bin/bash
# kernel-trace statistics of one configuration under several builds (GPU box): tools/kt_ab.sh <cfg> <dataset|-> <steps> <kernel-regex> [lib.so|intree]...
CFG=$1; DS=$2; STEPS=$3; INC=$4; shift 4
R=$PWD; export TMPDIR=/tmp
for LIB in "$@"; do
  T=$(basename "$LIB" .so); OUT=$R/gpurun_out/kt_${CFG}_$T; rm -rf $OUT; mkdir -p $OUT
  if [ "$LIB" = intree ]; then unset RMI_HIP_LIB; else export RMI_HIP_LIB=$R/$LIB; fi
  (cd /tmp && RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT -o kt -f csv -- python $R/tools/cfg_run.py $CFG $DS $STEPS < /dev/null > $OUT/log 2>&1)
  echo "== $CFG $T"
  python $R/tools/summarize_prof.py $OUT | grep -E "$INC"
done
