#!/usr/bin/env python3
"""usage: tools/isa_loops.py kernel.s -> the loops of one kernel's ISA listing (backward branches) with what each holds:
waits, scratch traffic, SGPR spills to lanes, f64 ops, LDS / scalar / global memory instructions, AGPR moves."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
lab, ins = {}, []
for l in lines:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        lab[m.group(1)] = len(ins)
        continue
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    ins.append(s)
print(len(ins), 'instructions')
loops = []
for i, s in enumerate(ins):
    m = re.match(r's_c?branch\S*\s+(\.LBB\d+_\d+)', s)
    if m and m.group(1) in lab and lab[m.group(1)] <= i:
        loops.append((lab[m.group(1)], i, m.group(1)))
for a, b, n in sorted(loops):
    body = ins[a:b + 1]
    c = lambda p: sum(1 for x in body if re.search(p, x))
    print(f"{n:12s} {a:5d}-{b:5d} n={b-a+1:5d} wait={c('s_waitcnt'):3d} vm0={c(r'vmcnt.0.'):2d} scratch={c('scratch_'):2d} "
          f"lane={c('v_writelane|v_readlane'):3d} f64={c('_f64'):4d} ds={c('^ds_'):3d} sload={c('^s_load'):3d} glob={c('^global_'):3d} "
          f"acc={c('accvgpr'):3d} br={c('^s_c?branch'):3d}")
