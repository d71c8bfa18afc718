#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
RMI_CFG_TRACE=1 RMI_CFG_BW=0 RMI_HIP_LIB=build_ab/librmi_hip_prof.so python tools/cfg_run.py C5 - 1 > $O/prof.log 2>&1
grep "^wave\|^general\|^  tile" $O/prof.log | sort | uniq -c | sort -rn | head -20
