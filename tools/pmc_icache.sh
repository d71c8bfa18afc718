#!/bin/bash
# instruction-cache counters of one configuration's kernels (GPU box): tools/pmc_icache.sh <cfg> [lib.so]
CFG=$1; LIB=${2:-}
R=$PWD; OUT=$R/gpurun_out/pmc_ic_$CFG; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$LIB" ] && export RMI_HIP_LIB=$R/$LIB
cd /tmp
W="python $R/tools/cfg_run.py $CFG -"
INC='k_leaf_regs|k_leaf_lanes|k_spline_scan'
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-include-regex "$INC" -d $OUT/p1 -o p -f csv -- $W 2 < /dev/null > $OUT/p1.log 2>&1
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 150 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-include-regex "$INC" -d $OUT/p2 -o p -f csv -- $W 2 < /dev/null > $OUT/p2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
tail -3 $OUT/p1.log
