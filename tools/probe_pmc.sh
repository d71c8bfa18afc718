#!/bin/bash
# usage (GPU box): tools/probe_pmc.sh  -> the wave-cycle accounting (active / waiting) of tools/probe/wave1_probe's loops: what a lone
# wave's SQ_WAIT_ANY share looks like when nothing but arithmetic is in flight
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout -k 5 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/pp -o p -f csv -- $R/build_ab/wave1_probe 4 < /dev/null > /tmp/pp.log 2>&1
python3 $R/tools/probe_pmc.py $(find /tmp/pp -name "*counter_collection.csv" | head -1) < /dev/null
