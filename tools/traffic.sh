#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel for the bench command (separate PMC passes), then calibrated:
# k_err_range reads exactly N*sizeof(key) bytes with the same 8-B/lane load pattern as k_fit_stream,
# so FETCH_SIZE(k_err_range) -> bytes gives the gfx950 correction factor for this access width.
set -u
OUT=gpurun_out/traffic; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1"
INC='k_fit_stream|k_err_range'
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$INC" -d $OUT/f -o p -f csv -- $B > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$INC" -d $OUT/w -o p -f csv -- $B > $OUT/w.log 2>&1
python - <<'PY'
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/traffic/*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/traffic/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = "k_fit_stream" if "k_fit_stream" in r["Kernel_Name"] else "k_err_range"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
n_bytes = 200_000_000 * 8
corr = n_bytes / (avg["k_err_range"]["FETCH_SIZE"] * 1024)
out = {"note": "rocprofv3 FETCH_SIZE/WRITE_SIZE (KB) per launch, separate PMC passes; reads scaled by %.3f = N*8 bytes / FETCH_SIZE(k_err_range), "
               "the gfx950 under-count for 8-B/lane loads calibrated on a kernel that reads the keys exactly once" % corr,
       "raw_kb": avg, "read_correction": corr}
for k in avg:
    out[k] = {"hbm_bytes_per_launch": avg[k]["FETCH_SIZE"] * 1024 * corr + avg[k].get("WRITE_SIZE", 0.0) * 1024,
              "read_bytes": avg[k]["FETCH_SIZE"] * 1024 * corr, "write_bytes": avg[k].get("WRITE_SIZE", 0.0) * 1024}
json.dump(out, open("gpurun_out/traffic/traffic_calibrated.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
