#!/bin/bash
# usage (GPU box): tools/lanes_prof.sh <tag> <variant-substring> [n L]  -> gpurun_out/ln_<tag>/summary.txt   (every pass under `timeout`)
TAG=$1; VAR=$2; N=${3:-200000000}; L=${4:-1048576}
OUT=gpurun_out/ln_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
INC='k_leaf_lanes|k_leaf_search|k_err_range|k_finalize'
B="python tools/lanes_check.py time $N $L 4 uniform"
timeout 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "$INC" -d $OUT/p1 -o p -f csv -- $B "$VAR" > $OUT/p1.log 2>&1
timeout 90 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-include-regex "$INC" -d $OUT/p2 -o p -f csv -- $B "$VAR" > $OUT/p2.log 2>&1
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$INC" -d $OUT/p3 -o p -f csv -- $B "$VAR" > $OUT/p3.log 2>&1
timeout 90 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "$INC" -d $OUT/p4 -o p -f csv -- $B "$VAR" > $OUT/p4.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A9 "k_leaf_lanes\|k_leaf_search\|k_err_range" $OUT/summary.txt | head -120
