#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel (separate PMC passes), calibrated on k_read_bw, which reads the 1.6 GB of keys exactly once
# per launch with 16-byte loads per lane -- NOT on the kernels under test.  -> gpurun_out/traffic_r02/traffic_r02.json
set -u
OUT=gpurun_out/traffic_r02; mkdir -p $OUT; export TMPDIR=/tmp
INC='k_read_bw|k_sigma2|k_fit_stream|k_err_range|k_finalize|k_fit_long|k_list'
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$INC" -d $OUT/f -o p -f csv -- python tools/traffic_r02.py > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$INC" -d $OUT/w -o p -f csv -- python tools/traffic_r02.py > $OUT/w.log 2>&1
python - <<'PY'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
names = ["k_read_bw", "k_sigma2", "k_fit_stream", "k_err_range", "k_finalize", "k_fit_long", "k_list_tail", "k_list"]
for f in glob.glob("gpurun_out/traffic_r02/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((x for x in names if x in r["Kernel_Name"]), None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
n_bytes = 200_000_000 * 8
corr = n_bytes / (avg["k_read_bw"]["FETCH_SIZE"] * 1024)
out = {"note": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) per launch, separate PMC passes; reads scaled by %.3f = 1.6e9 bytes / FETCH_SIZE(k_read_bw), "
               "the streaming kernel that reads every key byte exactly once with 16-byte loads per lane (the guide's gfx950 factor for that width is 2); "
               "writes as counted" % corr,
       "raw_kb": avg, "read_correction": corr}
for k in avg:
    rd = avg[k].get("FETCH_SIZE", 0.0) * 1024 * corr
    wr = avg[k].get("WRITE_SIZE", 0.0) * 1024
    out[k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr}
json.dump(out, open("gpurun_out/traffic_r02/traffic_r02.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
