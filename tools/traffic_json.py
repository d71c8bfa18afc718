"""FETCH_SIZE / WRITE_SIZE CSVs of tools/profile_cfg.sh -> one JSON per configuration: bytes per launch and kernel, the step's sum, its
ratio to the algorithmic bytes.  usage: traffic_json.py <outdir> <cfg> [dataset]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfg_run import CFG  # noqa: E402
import numpy as np  # noqa: E402

out_dir, cfg = sys.argv[1], sys.argv[2]
ds = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
n, L, root, leaf, ds0, dt = CFG[cfg]
names = ["k_read_bw", "k_spline_scan", "k_scan_gaps", "k_leaf_regs", "k_long_regs", "k_regs_finalize", "k_leaf_lanes_listed", "k_leaf_lanes", "k_leaf_search", "k_leaf_samples", "k_lane_reduce",
         "k_verify_listed", "k_giant_scan", "k_finalize_listed", "k_finalize", "k_list_tail", "k_list", "k_init"]
acc = defaultdict(lambda: defaultdict(list))
for d in (f"{out_dir}/tf", f"{out_dir}/tw"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            k = next((x for x in names if x in kn), None)
            if k == "k_spline_scan":                      # two kernels of one template: the short form's (PHASE 0) and the general form's (PHASE 1)
                targs = kn.split("(")[0].rstrip().rstrip(">").split("<", 1)[-1].split(",")     # <ROOT, K, V, PHASE, FAR>
                k = "k_spline_scan<short form>" if len(targs) == 5 and targs[3].strip() == "0" else "k_spline_scan<general form>"
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
key_bytes = n * np.dtype(dt).itemsize
corr = key_bytes / (avg["k_read_bw"]["FETCH_SIZE"] * 1024) if "k_read_bw" in avg and avg["k_read_bw"].get("FETCH_SIZE") else 2.0
b_alg = key_bytes + 24 * L
res = {"config": cfg, "dataset": ds or ds0, "keys": n, "leaves": L, "spec": f"{root},{leaf}",
       "note": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) per launch, separate PMC passes (tools/profile_cfg.sh); reads scaled by %.3f = the key array's bytes / "
               "FETCH_SIZE(k_read_bw), the streaming kernel of the same run that reads every key byte exactly once with 16-byte loads per lane (the guide's "
               "gfx950 factor for that width is 2); Infinity-Cache hits are counted like HBM reads; writes as counted" % corr,
       "raw_kb": avg, "read_correction": corr, "sources_sha256": bench.sources_sha256(), "algorithmic_bytes": b_alg, "kernels": {}}
tot = 0.0
for k in avg:
    if k == "k_read_bw":
        continue
    rd = avg[k].get("FETCH_SIZE", 0.0) * 1024 * corr
    wr = avg[k].get("WRITE_SIZE", 0.0) * 1024
    res["kernels"][k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr}
    tot += rd + wr
res["step_hbm_bytes"] = tot
res["traffic_ratio"] = tot / b_alg
print(json.dumps(res, indent=1))
