#!/usr/bin/env python3
"""Resource usage of the kernels of one translation unit: tools/kres.py rmi_amd/csrc/rmi_scan.hip [pattern] [-D...]"""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"] + extra
p = subprocess.run(cmd, capture_output=True, text=True)
if p.returncode:
    sys.stderr.write(p.stderr[-4000:]); sys.exit(1)
cur = None
rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name": cur = v; rows[cur] = {}
    elif cur: rows[cur][k.split()[0] + ("_spill" if "Spill" in k else "")] = v
for name, r in rows.items():
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dn: continue
    print(f"{dn[:110]:110s} sgpr {r.get('TotalSGPRs'):>4} vgpr {r.get('VGPRs'):>4} agpr {r.get('AGPRs'):>4} scratch {r.get('ScratchSize'):>5} occ {r.get('Occupancy'):>2} "
          f"sspill {r.get('SGPRs_spill'):>4} vspill {r.get('VGPRs_spill'):>4} lds {r.get('LDS'):>6}")
