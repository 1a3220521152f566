#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a per-kernel resource table (VGPR/AGPR/scratch/LDS/occupancy)."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
       "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:70]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
keys = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs", "VGPRs Spill", "LDS Size [bytes/block]"]
print(f"{'kernel':70s} " + " ".join(f"{k.split(' ')[0]:>9s}" for k in keys))
for k, v in rows.items():
    print(f"{k:70s} " + " ".join(f"{v.get(kk, -1):9d}" for kk in keys))
if "error" in out:
    print(out)
