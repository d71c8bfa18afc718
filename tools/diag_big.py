#!/usr/bin/env python3
"""Diagnostic for key sets beyond 2^32 keys: generation, sparse root fits and the bucketing scan
against the closed form of the generator (no training)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import train, datagen as dg

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 32) + 1_000_003
L = 1 << 22
tr = train.Trainer()
t0 = time.time(); tr.generate_keys("uniform", np.uint64, n); print("generate s", time.time() - t0)
k0 = int(dg.uniform_u64(n, start=0, count=1)[0]); kl = int(dg.uniform_u64(n, start=n - 1, count=1)[0])
print("closed form first/last", k0, kl)
r = tr.fit_root("radix", L); print("radix root", r.ip, "expect prefix", 64 - (k0 ^ kl).bit_length())
ls = tr.fit_root("linear_spline", L); print("linear_spline root", ls.p[:2])
slope = (0.0 - float((n - 1) * L // n)) / (float(k0) - float(kl))
print("expected approx", 0.0 - slope * float(k0), slope)
t0 = time.time(); rt = tr.fit_root("radix18", L); print("radix18 fit s", time.time() - t0, rt.ip)
tab = rt.table
bad = 0
for slot in [1, 2, 1000, 77777, 131072, 200000, (1 << 18) - 1]:
    lo, hi = 0, n                                   # first index whose key >> 46 >= slot
    while lo < hi:
        mid = (lo + hi) // 2
        if (int(dg.uniform_u64(n, start=mid, count=1)[0]) >> 46) < slot: lo = mid + 1
        else: hi = mid
    exp = int(float(lo) * (L / n)) if lo < n else (1 << 18)
    print("slot", slot, "table", int(tab[slot]), "expected", exp)
    bad += int(tab[slot]) != exp
print("bad", bad)
tr.close()
