#!/bin/bash
# usage (GPU box): tools/sigma/prof.sh <tag> <kernel-regex> <script args...>   -> gpurun_out/sg_<tag>/summary.txt
TAG=$1; INC=$2; shift 2
OUT=gpurun_out/sg_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="python tools/sigma/time_modes.py $*"
rocprofv3 --kernel-trace --stats -T -d $OUT/kt -o kt -f csv -- $B > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "$INC" -d $OUT/p1 -o p -f csv -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-include-regex "$INC" -d $OUT/p2 -o p -f csv -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$INC" -d $OUT/p3 -o p -f csv -- $B > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "$INC" -d $OUT/p4 -o p -f csv -- $B > $OUT/p4.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -80
