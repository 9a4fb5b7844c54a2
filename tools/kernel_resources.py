#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of libevac, from hipcc's -Rpass-analysis=kernel-resource-usage
(a from-scratch gfx950 compile, ~30 s).  usage: tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from evacuation_amd import build  # noqa: E402

cmd = [build.hipcc_path()] + build.FLAGS + build.SOURCES + ["-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
text = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    m = re.search(r"remark: +(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
    n = n.replace("evac::", "").replace("void ", "").split("(")[0]
    print(f"{n[:64]:64s} VGPR={r.get('VGPRs'):4d} SGPR={r.get('TotalSGPRs'):4d} scratch={r.get('ScratchSize'):4d} occ={r.get('Occupancy')} lds={r.get('LDS')}")
