#!/bin/bash
# rocprofv3 of the default bench command (round 3): kernel-trace stats, then the PMC passes (each on its own, each under
# `timeout`: a pass that combined FETCH_SIZE with TCC_*_sum hung a box for 15 minutes), then the traffic files (one per mode).
# usage (GPU box): tools/profile_r03.sh   -> gpurun_out/prof_r03/{r03_kernel_stats.csv, r03_rocprofv3_summary.txt, traffic_r03_*.json}
set -u
OUT=gpurun_out/prof_r03; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extras"
timeout 200 rocprofv3 --kernel-trace --stats -T -d $OUT/kt -o kt -f csv -- $B --steps 200 --warmup 50 > $OUT/kt.log 2>&1
INC='k_leaf_lanes|k_leaf_search|k_leaf_samples|k_lane_reduce|k_list|k_finalize|k_init'
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
   --kernel-include-regex "$INC" -d $OUT/pmc1 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
   --kernel-include-regex "$INC" -d $OUT/pmc2 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc2.log 2>&1
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "$INC" -d $OUT/pmc4 -o p -f csv -- $B --steps 2 --warmup 0 > $OUT/pmc4.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/r03_rocprofv3_summary.txt
cp $OUT/kt/*kernel_stats.csv $OUT/r03_kernel_stats.csv 2>/dev/null
# ---- traffic: FETCH_SIZE / WRITE_SIZE per kernel and mode, reads calibrated on k_read_bw (reads every key byte exactly once)
for MODE in exact onepass_guarded; do
  M=0; [ $MODE = onepass_guarded ] && M=1
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_read_bw|k_leaf_lanes|k_leaf_search|k_leaf_samples|k_lane_reduce|k_sigma2|k_finalize|k_list' -d $OUT/tf_$MODE -o p -f csv -- python tools/traffic_r03.py $M > $OUT/tf_$MODE.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'k_read_bw|k_leaf_lanes|k_leaf_search|k_leaf_samples|k_lane_reduce|k_sigma2|k_finalize|k_list' -d $OUT/tw_$MODE -o p -f csv -- python tools/traffic_r03.py $M > $OUT/tw_$MODE.log 2>&1
  python tools/traffic_r03_json.py $OUT $MODE > $OUT/traffic_r03_$MODE.json
done
head -30 $OUT/r03_rocprofv3_summary.txt
