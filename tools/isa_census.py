#!/usr/bin/env python3
"""Instruction census of a range of an ISA listing (line numbers as printed by grep -n), by class.
usage: tools/isa_census.py kernel.s first_line last_line [steps]   -- e.g. one written-out block of 16 steps of k_leaf_regs"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
a, b = int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 16
CLASSES = [
    ("f64 arithmetic (fma / add / mul / min / max)", r"^v_(fma|add|mul|min|max)_f64"),
    ("f64 <- key conversion (cvt, ldexp)", r"^v_(cvt_f64|ldexp_f64)"),
    ("u32 <- f64 conversion (error pass)", r"^v_cvt_u32_f64"),
    ("stash: AGPR writes / reads", r"^v_accvgpr"),
    ("stash / copies: v_mov", r"^v_mov_b(32|64)"),
    ("duplicate test, masks (xor / or / and / min_u32 / cmp / cndmask)", r"^v_(xor|or|and|min_u32|cmp|cndmask|bfe|bfi|lshl|lshr|ashr|not)"),
    ("integer arithmetic (add / sub / mad / sad / max_u32)", r"^v_(add_u32|sub_u32|subrev|mad_|sad_|max_u32|add_co|addc|add3|lshl_add|mul_lo|mul_hi|mul_u32)"),
    ("cross-lane (dpp, readlane, writelane, permute)", r"(dpp|v_readlane|v_writelane|v_readfirstlane|ds_bpermute|ds_swizzle)"),
    ("LDS reads / writes", r"^ds_"),
    ("vector memory (DMA panels, stores)", r"^(global_|buffer_|flat_|scratch_)"),
    ("scalar memory", r"^s_(load|buffer_load)"),
    ("waits / nops", r"^s_(waitcnt|nop|sleep)"),
    ("scalar ALU / branches", r"^s_"),
    ("other vector", r"^v_"),
]
cnt = {n: 0 for n, _ in CLASSES}
tot = 0
for l in lines[a - 1:b]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    tot += 1
    for n, p in CLASSES:
        if re.search(p, s):
            cnt[n] += 1
            break
print(f"{tot} instructions in lines {a}..{b} = {tot / steps:.1f} per step of {steps}")
for n, _ in CLASSES:
    if cnt[n]:
        print(f"  {cnt[n]:5d}  {cnt[n] / steps:5.2f} / step   {n}")
