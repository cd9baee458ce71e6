#!/usr/bin/env python3
"""Reads bench.py JSON lines from stdin and prints a one line summary each (helper for A/B runs)."""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    k = {n: round(v, 3) for n, v in d["kernel_ms_per_step"].items()}
    print(label, round(d["value"]), "Mrays/s", round(d["ms_per_step"], 3), "ms/step", k, "roofline", round(d["roofline"]["frac"], 3))
