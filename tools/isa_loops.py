"""List the loops (backward branches) of one kernel in an ISA dump with instruction counts by class.
usage: python tools/isa_loops.py file.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(lines):
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        body = [x.strip() for x in lines[labels[m.group(1)]:i + 1] if re.match(r"^\s+[a-z]", x)]
        cnt = {}
        for ins in body:
            op = ins.split()[0]
            cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop") else
                   "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else
                   "wait" if op.startswith("s_waitcnt") else "nop" if op.startswith("s_nop") else "other")
            cnt[cls] = cnt.get(cls, 0) + 1
        f64 = sum(1 for ins in body if re.match(r"v_\w+_f64", ins))
        print(f"{m.group(1)} lines {labels[m.group(1)]}-{i}: {len(body)} instr {cnt} f64={f64}")
