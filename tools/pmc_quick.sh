#!/bin/bash
# a handful of SQ counters of one configuration's dominant kernels (GPU box): tools/pmc_quick.sh <cfg> [dataset] [lib.so]
CFG=$1; DS=${2:--}; LIB=${3:-}
R=$PWD; OUT=$R/gpurun_out/pmcq_${CFG}_$(basename "${LIB:-intree}" .so); mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$LIB" ] && export RMI_HIP_LIB=$R/$LIB
cd /tmp
W="python $R/tools/cfg_run.py $CFG $DS"
INC='k_leaf_regs|k_leaf_lanes|k_spline_scan|k_leaf_search|k_regs_finalize'
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-include-regex "$INC" -d $OUT/p1 -o p -f csv -- $W 2 < /dev/null > $OUT/p1.log 2>&1
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-include-regex "$INC" -d $OUT/p2 -o p -f csv -- $W 2 < /dev/null > $OUT/p2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    print("   ", {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
