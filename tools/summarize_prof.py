#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel-trace stats + PMC counter CSVs) per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    n = name.split("(")[0]
    if "k_spline_scan" in n:                              # two kernels of one template
        # k_spline_scan<ROOT, K, V, PHASE, FAR>: PHASE 0 = the short form's kernel, 1 = the general form's (the stats file truncates long names: then unnamed)
        a = n.rstrip().rstrip(">").split("<", 1)[-1].split(",") if n.rstrip().endswith(">") else []
        if len(a) == 5:
            return "k_spline_scan<short>" if a[3].strip() == "0" else "k_spline_scan<general>"
        return "k_spline_scan"
    for key in ("k_spline_scan", "k_scan_gaps", "k_long_regs", "k_verify_listed", "k_giant_scan", "k_finalize_listed", "k_leaf_regs", "k_regs_finalize", "k_regs_table", "k_leaf_lanes_listed", "k_leaf_lanes", "k_leaf_search", "k_leaf_samples", "k_lane_reduce", "k_lane_table", "k_sigma2", "k_list_tail", "k_list", "k_err_long", "k_fit_list", "k_err_list", "k_read_bw", "k_init", "k_fit_stream", "k_err_range", "k_fit_long", "k_fill_tilemin", "k_fill_scan_tiles", "k_fill_apply", "k_finalize",
                "k_stats_reduce", "k_generate", "k_boundaries", "k_fit_leaf", "k_err"):
        if key in n:
            return key
    return n[-60:]


def main(root):
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"== {f}")
        for row in csv.DictReader(open(f)):
            print(f"  {short(row['Name']):22s} calls={row['Calls']:>5s} avg_ns={float(row['AverageNs']):12.0f} "
                  f"min={row['MinNs']:>10s} max={row['MaxNs']:>10s} pct={row['Percentage']}")
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        print(f"== {f}")
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        seen = set()
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
        for k in acc:
            print(f"  {k} (dispatches={cnt[k]})")
            for c, v in sorted(acc[k].items()):
                print(f"      {c:28s} {v / max(cnt[k], 1):18.1f} per dispatch")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
