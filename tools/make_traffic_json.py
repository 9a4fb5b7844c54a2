#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_pmc.sh output directory.
HBM bytes per launch = FETCH_SIZE[KiB] * 1024 * 2  (gfx950 reports half of a coalesced read stream,
MI355X_MICROARCH.md 'HBM') + WRITE_SIZE[KiB] * 1024; separate --pmc passes, mean over the dispatches
of the timed kernel.  usage: make_traffic_json.py <pmc_dir> <key> [<pmc_dir> <key> ...]"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "traffic.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
args = sys.argv[1:]
for d, key in zip(args[0::2], args[1::2]):
    kern = "k_rollout" if ":rollout:" in key else "k_step"
    vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row.get("Kernel_Name", "") and row["Counter_Name"] in vals:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    fetch = sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
    write = sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
    data[key] = {"kernel": kern, "fetch_size_kib": fetch, "write_size_kib": write,
                 "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
                 "note": "FETCH_SIZE doubled per the gfx950 correction; dispatches averaged: %d" % len(vals["FETCH_SIZE"]),
                 "source": os.path.relpath(d, ROOT)}
    print(key, data[key])
json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
