"""Development aid: the start/end of every kernel of the LAST training of a `rocprofv3 --kernel-trace` run, relative to the
step's first kernel.  usage: python tools/timeline.py <dir with *_kernel_trace.csv> [steps back from the end]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?")))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
idx = [i for i, r in enumerate(rows) if "k_init" in r[2] or "k_leaf_samples" in r[2]]
i0 = idx[-back]
i1 = idx[-back + 1] if back > 1 else len(rows)
t0 = rows[i0][0]
for a, b, nm, q in rows[i0:i1]:
    print(f"{(a - t0) / 1e3:9.1f} .. {(b - t0) / 1e3:9.1f} us  ({(b - a) / 1e3:7.1f})  q{q}  {nm}")
