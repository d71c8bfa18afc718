"""FETCH_SIZE / WRITE_SIZE CSVs of tools/profile_r04.sh -> one JSON per fit mode.  usage: traffic_r03_json.py <outdir> <mode>"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from collections import defaultdict
out_dir, mode = sys.argv[1], sys.argv[2]
names = ["k_read_bw", "k_leaf_regs", "k_regs_finalize", "k_leaf_lanes_listed", "k_leaf_lanes", "k_leaf_search", "k_leaf_samples", "k_lane_reduce", "k_sigma2", "k_finalize", "k_list_tail", "k_list"]
acc = defaultdict(lambda: defaultdict(list))
for d in (f"{out_dir}/tf_{mode}", f"{out_dir}/tw_{mode}"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((x for x in names if x in r["Kernel_Name"]), None)
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
corr = 200_000_000 * 8 / (avg["k_read_bw"]["FETCH_SIZE"] * 1024) if "k_read_bw" in avg and avg["k_read_bw"].get("FETCH_SIZE") else 2.0
res = {"note": "fit mode %s; rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) per launch, separate PMC passes; reads scaled by %.3f = 1.6e9 bytes / "
               "FETCH_SIZE(k_read_bw), the streaming kernel that reads every key byte exactly once with 16-byte loads per lane (the guide's gfx950 "
               "factor for that width is 2); Infinity-Cache hits are counted like HBM reads; writes as counted" % (mode, corr),
       "raw_kb": avg, "read_correction": corr, "sources_sha256": bench.sources_sha256()}
for k in avg:
    rd = avg[k].get("FETCH_SIZE", 0.0) * 1024 * corr
    wr = avg[k].get("WRITE_SIZE", 0.0) * 1024
    res[k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr}
print(json.dumps(res, indent=1))
