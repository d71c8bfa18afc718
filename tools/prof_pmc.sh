#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <kernel-regex> [env assignments...] ; prints per-kernel PMC summary
TAG=$1; INC=$2; shift 2
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
B="python bench.py --no-cpu-baseline --steps 2 --warmup 0"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "$INC" -d $OUT/p1 -o p -f csv -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-include-regex "$INC" -d $OUT/p2 -o p -f csv -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$INC" -d $OUT/p3 -o p -f csv -- $B > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "$INC" -d $OUT/p4 -o p -f csv -- $B > $OUT/p4.log 2>&1
python tools/summarize_prof.py $OUT
