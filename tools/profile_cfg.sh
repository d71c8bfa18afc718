#!/bin/bash
# rocprofv3 of ONE configuration (round tag RND, default r06): kernel-trace stats, the PMC passes (each on its own, under `timeout -k`, nothing reads
# stdin), then FETCH_SIZE / WRITE_SIZE per kernel, calibrated on k_read_bw of the same run and tied to the kernel sources' sha256.
# usage (GPU box): [RND=r06] tools/profile_cfg.sh <M|C2|C3|C5|Ms|C4s|D> [dataset] [--no-pmc]
#   -> gpurun_out/prof_$RND/<tag>/{$RND_<tag>_kernel_stats.csv, $RND_<tag>_summary.txt, traffic_$RND_<tag>.json}
set -u
CFG=$1; DS=${2:--}; NOPMC=${3:-}; RND=${RND:-r06}
TAG=$(echo "$CFG" | tr 'A-Z' 'a-z'); [ "$DS" != "-" ] && TAG=${TAG}_$DS
R=$PWD; OUT=$R/gpurun_out/prof_$RND/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
W="python $R/tools/cfg_run.py $CFG $DS"
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 200 rocprofv3 --kernel-trace --stats -T -d $OUT/kt -o kt -f csv -- $W 50 < /dev/null > $OUT/kt.log 2>&1
INC='k_spline_scan|k_scan_gaps|k_leaf_regs|k_regs_finalize|k_leaf_lanes|k_leaf_search|k_leaf_samples|k_lane_reduce|k_list|k_finalize|k_init|k_verify|k_giant|k_long_regs'
if [ "$NOPMC" != "--no-pmc" ]; then
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
   --kernel-include-regex "$INC" -d $OUT/pmc1 -o p -f csv -- $W 2 < /dev/null > $OUT/pmc1.log 2>&1
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
   --kernel-include-regex "$INC" -d $OUT/pmc2 -o p -f csv -- $W 2 < /dev/null > $OUT/pmc2.log 2>&1
fi
K="k_read_bw|$INC"
RMI_CFG_TRACE=1 timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$K" -d $OUT/tf -o p -f csv -- $W 3 < /dev/null > $OUT/tf.log 2>&1
RMI_CFG_TRACE=1 timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$K" -d $OUT/tw -o p -f csv -- $W 3 < /dev/null > $OUT/tw.log 2>&1
cd $R
python tools/summarize_prof.py $OUT > $OUT/${RND}_${TAG}_summary.txt 2>&1 < /dev/null
F=$(find $OUT/kt -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/${RND}_${TAG}_kernel_stats.csv
python tools/traffic_json.py $OUT $CFG $DS > $OUT/traffic_${RND}_${TAG}.json 2> $OUT/traffic_json.err < /dev/null
head -24 $OUT/${RND}_${TAG}_summary.txt
python - <<PY
import json
j = json.load(open("$OUT/traffic_${RND}_${TAG}.json"))
print({k: j[k] for k in ("algorithmic_bytes", "step_hbm_bytes", "traffic_ratio", "read_correction")})
print({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in j["kernels"].items()})
PY
