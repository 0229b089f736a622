#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, AGPRs, spills, occupancy, LDS) of one .hip file:
python tools/kres.py gan_amd/csrc/conv_thin.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(
    ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
     "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", cur)
        rows[cur] = {}
    else:
        rows[cur][k.split(" [")[0]] = v
print(f"{'kernel':60s} VGPR AGPR vspill sspill occ   LDS")
for k, r in rows.items():
    print(f"{k[:60]:60s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('VGPRs Spill','?'):>6} {r.get('SGPRs Spill','?'):>6} {r.get('Occupancy','?'):>3} {r.get('LDS Size','?'):>6}")
