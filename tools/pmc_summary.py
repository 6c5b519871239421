#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE counter CSVs into per-launch HBM bytes per kernel.

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 bytes... calibrate: the pure fill
kernel k_fill_ceiling writes a known byte count, so the WRITE_SIZE unit is derived from it rather than
assumed (MI355X_MICROARCH.md: WRITE_SIZE is uncalibrated on gfx950; FETCH_SIZE under-reports wide reads 2x).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d):
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


def main():
    out = sys.argv[1]
    res = {}
    for tag, name in (("pmc_w", "WRITE_SIZE"), ("pmc_r", "FETCH_SIZE")):
        per = collect(os.path.join(out, tag))
        for k, cs in per.items():
            short = "k_synth" if "k_synth" in k else "k_seed" if "k_seed" in k else "k_fill_ceiling" if "k_fill" in k else None
            if short and name in cs:
                v = cs[name]
                res.setdefault(short, {})[name + "_raw_per_launch"] = sum(v) / len(v)
                res[short]["launches_" + name] = len(v)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
