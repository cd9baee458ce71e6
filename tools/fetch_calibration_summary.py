#!/usr/bin/env python3
"""Table of tools/microbench/fetch_calibration: known bytes per kernel against what rocprofv3's counters report for it.
usage: fetch_calibration_summary.py <dir with fetch_calibration_plain.jsonl and the cal_<counters>/ pass directories>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
plain = [json.loads(line) for line in open(os.path.join(root, "fetch_calibration_plain.jsonl")) if line.startswith("{")]
order = plain[len(plain) // 2:]            # second round of the program = dispatches 10..17 under the profiler (1 = the memset, 2..9 = first round)
counters = defaultdict(dict)               # dispatch id -> counter -> value
for d in sorted(glob.glob(os.path.join(root, "cal_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            counters[int(r["Dispatch_Id"])][r["Counter_Name"]] = counters[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
first = 2 + len(order)
print(f"{'kernel / pattern':96s} {'read MB':>9s} {'2*FETCH_SIZE MB':>16s} {'ratio':>6s} {'RDREQ':>10s} {'B/req':>6s} {'L2 hit':>7s} | {'write MB':>9s} {'WRITE_SIZE MB':>14s} {'ratio':>6s} | {'GB/s':>7s}")
for i, e in enumerate(order):
    c = counters.get(first + i, {})
    fetch = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
    write = c.get("WRITE_SIZE", 0.0) * 1024.0
    rd = c.get("TCC_EA0_RDREQ_sum", 0.0)
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    name = (e["kernel"] + ": " + e["pattern"])[:96]
    print(f"{name:96s} {e['read_bytes'] / 1e6:9.1f} {fetch / 1e6:16.1f} {fetch / e['read_bytes'] if e['read_bytes'] else 0:6.2f} {rd:10.0f} {fetch / rd if rd else 0:6.1f} "
          f"{hit / (hit + miss) if hit + miss else 0:7.3f} | {e['write_bytes'] / 1e6:9.1f} {write / 1e6:14.1f} {write / e['write_bytes'] if e['write_bytes'] else 0:6.2f} | {e['GB_per_s']:7.0f}")
