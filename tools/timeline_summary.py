#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the LAST complete pass in it (a pass starts with k_generate), every dispatch with its duration
and the idle gap on the device before it -- what a bounce of a few thousand paths costs in kernels and in between them.
usage: tools/timeline_summary.py <kernel_trace.csv> [--passes N]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("hipr::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_generate")]
if len(starts) < 3:
    sys.exit("fewer than three passes in the trace")
first, last = starts[-2], starts[-1]
span = rows[last][0] - rows[first][0]
busy = 0
previous_end = {}
print(f"pass of {last - first} dispatches, {span / 1e3:.1f} us from its k_generate to the next one's; gap = idle time of the dispatch's own queue before it")
print(f"{'#':>4} {'queue':>5} {'kernel':28} {'duration us':>12} {'gap before us':>14}")
by_kernel = {}
gaps = 0
for i in range(first, last):
    s, e, name, queue = rows[i]
    gap = max(0, s - previous_end.get(queue, s))
    gaps += gap if queue == rows[first][3] else 0
    busy += e - s
    by_kernel.setdefault(name, [0, 0])
    by_kernel[name][0] += 1
    by_kernel[name][1] += e - s
    print(f"{i - first:4d} {queue:>5} {name:28} {(e - s) / 1e3:12.1f} {gap / 1e3:14.1f}")
    previous_end[queue] = max(previous_end.get(queue, e), e)
print(f"kernels (all queues) {busy / 1e3:.1f} us, idle time of queue {rows[first][3]} (the one that traces and shades) {gaps / 1e3:.1f} us")
for name, (count, total) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:28} {count:4d} dispatches {total / 1e3:10.1f} us")
