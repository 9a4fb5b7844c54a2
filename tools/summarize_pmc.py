#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_pmc.sh: per-kernel mean duration and mean
counter values per dispatch for the evac kernels (dispatches of one launch shape: the most frequent grid/steps)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"k_(?:rollout|step|reset|observe)\w*<", name)
    if not m:
        return None
    depth, k = 1, m.end()
    while k < len(name) and depth:
        depth += {"<": 1, ">": -1}.get(name[k], 0)
        k += 1
    return name[m.start():k].replace("evac::", "")


if os.path.exists(os.path.join(out, "command.txt")):
    print("== command:", open(os.path.join(out, "command.txt")).read().strip())
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (", os.path.relpath(f, out), ")")
    for row in csv.DictReader(open(f)):
        s = short(row["Name"])
        if s:
            print(f"  {s:44s} calls={row['Calls']:>6s} avg_ns={float(row['AverageNs']):12.1f} min_ns={row['MinNs']:>10s} max_ns={row['MaxNs']:>10s} pct={row['Percentage']}")

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        s = short(row.get("Kernel_Name", ""))
        if s:
            agg[s][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== PMC (mean per dispatch over all dispatches of the kernel)")
for k in sorted(agg):
    print(" ", k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:26s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
