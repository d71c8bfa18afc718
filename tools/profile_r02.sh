#!/bin/bash
# rocprofv3 of the default bench command: kernel-trace stats, then PMC passes (separately; MI355X_MICROARCH.md)
set -u
OUT=gpurun_out/prof_r02; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -T -d $OUT/kt -o kt -f csv -- $B --steps 200 --warmup 50 > $OUT/kt.log 2>&1
INC='k_sigma2|k_list|k_finalize|k_fill|k_stats|k_init'
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
   --kernel-include-regex "$INC" -d $OUT/pmc1 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
   --kernel-include-regex "$INC" -d $OUT/pmc2 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$INC" -d $OUT/pmc3 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "$INC" -d $OUT/pmc4 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt
head -60 $OUT/summary.txt
