#!/bin/bash
# rocprofv3 of the default bench command (round 4): kernel-trace stats, then the PMC passes (each on its own and under `timeout -k`,
# nothing reads stdin), then the traffic file tied to the kernel sources' sha256.
# usage (GPU box): tools/profile_r04.sh   -> gpurun_out/prof_r04/{r04_kernel_stats.csv, r04_rocprofv3_summary.txt, traffic_r04_exact.json, r04_bench.json}
set -u
R=$PWD; OUT=$R/gpurun_out/prof_r04; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout -k 5 150 rocprofv3 --kernel-trace --stats -T -d $OUT/kt -o kt -f csv -- $B --steps 200 --warmup 50 < /dev/null > $OUT/kt.log 2>&1
INC='k_leaf_regs|k_regs_finalize|k_leaf_lanes|k_leaf_search|k_leaf_samples|k_lane_reduce|k_list|k_finalize|k_init'
timeout -k 5 100 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
   --kernel-include-regex "$INC" -d $OUT/pmc1 -o p -f csv -- $B --steps 2 --warmup 0 < /dev/null > $OUT/pmc1.log 2>&1
timeout -k 5 100 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
   --kernel-include-regex "$INC" -d $OUT/pmc2 -o p -f csv -- $B --steps 2 --warmup 0 < /dev/null > $OUT/pmc2.log 2>&1
timeout -k 5 100 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "$INC" -d $OUT/pmc4 -o p -f csv -- $B --steps 2 --warmup 0 < /dev/null > $OUT/pmc4.log 2>&1
cd $R
python tools/summarize_prof.py $OUT > $OUT/r04_rocprofv3_summary.txt 2>&1 < /dev/null
F=$(find $OUT/kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/r04_kernel_stats.csv
# ---- traffic: FETCH_SIZE / WRITE_SIZE per kernel, reads calibrated on k_read_bw (reads every key byte exactly once)
cd /tmp
K='k_read_bw|k_leaf_regs|k_regs_finalize|k_leaf_lanes|k_leaf_search|k_leaf_samples|k_lane_reduce'
timeout -k 5 100 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$K" -d $OUT/tf_exact -o p -f csv -- python $R/tools/traffic_r04.py < /dev/null > $OUT/tf_exact.log 2>&1
timeout -k 5 100 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$K" -d $OUT/tw_exact -o p -f csv -- python $R/tools/traffic_r04.py < /dev/null > $OUT/tw_exact.log 2>&1
cd $R
python tools/traffic_r04_json.py $OUT exact > $OUT/traffic_r04_exact.json 2> $OUT/traffic_json.err < /dev/null
timeout -k 5 200 python bench.py --steps 200 --warmup 50 --no-cpu-baseline < /dev/null > $OUT/r04_bench.json 2> $OUT/bench.err
head -40 $OUT/r04_rocprofv3_summary.txt
