#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE ...) of `bench.py` for one kernel into
profiles/<name>.json.   usage: pmc_summary.py <dir with one sub-dir per pass> <kernel substring> <out.json>

HBM traffic per launch = FETCH_SIZE * 2 + WRITE_SIZE (KiB): on gfx950 FETCH_SIZE reports exactly half of
the bytes of 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE was checked against the
kernel's known 6.55 MB of output per launch (6400 KiB measured)."""
import collections
import csv
import glob
import json
import sys

root, needle, out = sys.argv[1:4]
vals = collections.defaultdict(list)
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
res = {"kernel": needle, "launches_sampled": {k: len(v) for k, v in vals.items()}, "counters_mean_per_launch": mean}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    res["traffic_bytes_per_launch"] = (mean["FETCH_SIZE"] * 2 + mean["WRITE_SIZE"]) * 1024
    res["fetch_bytes_corrected"] = mean["FETCH_SIZE"] * 2 * 1024
    res["write_bytes"] = mean["WRITE_SIZE"] * 1024
res["workload"] = [5, 512, 640, 1]
res["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 3 --warmup 2 --eager`, FETCH_SIZE x2 per MI355X_MICROARCH.md"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
