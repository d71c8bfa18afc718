#!/bin/bash
O=gpurun_out/r5w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 200 rocprofv3 --kernel-trace --stats -T -d $OLDPWD/$O/kt -o kt -f csv -- python $OLDPWD/tools/cfg_run.py C5 - 30 > $OLDPWD/$O/kt.log 2>&1
cd $OLDPWD
python tools/summarize_prof.py $O/kt | head -8
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5w/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_spline_scan" in r["Kernel_Name"]]
for r in rows[-6:]:
    print(r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else "", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
