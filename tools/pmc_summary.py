#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: per kernel name, the sum of every counter over all dispatches.

usage: pmc_summary.py <dir> [kernel-substring ...]
"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
filters = sys.argv[2:]
sums = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in root.rglob("*counter_collection.csv"):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            if filters and not any(s in name for s in filters):
                continue
            sums[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row["Dispatch_Id"])
for name in sorted(sums):
    print(f"{name}  dispatches={len(calls[name])}")
    for counter, v in sorted(sums[name].items()):
        print(f"    {counter:32s} {v:.6g}   per dispatch {v / max(1, len(calls[name])):.6g}")
