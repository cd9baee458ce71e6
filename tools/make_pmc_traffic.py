#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json from two rocprofv3 --pmc runs of bench.py (one with FETCH_SIZE, one with WRITE_SIZE;
TCC has 4 counter slots, FETCH_SIZE costs 3 and WRITE_SIZE 2, so they cannot share a pass -- MI355X_MICROARCH.md
"rocprofv3 PMC slots").

usage: make_pmc_traffic.py <key> <fetch_dir> <write_dir> <command string> [out.json]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and on gfx950 FETCH_SIZE reports half
of the bytes read (MI355X_MICROARCH.md "HBM"; calibrated there on wide coalesced reads -- the BVH gathers here are 16 B
per lane loads of 64 B nodes and 48 B triangles, so the absolute is the guide's correction applied, not a calibration of
this access pattern). Averages are over ALL dispatches of a kernel in the run (instrumented + warmup + timed passes; the
per-launch work is the same frame every time).
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def bench_name(kernel: str):
    k = kernel.split("(")[0]
    if "k_generate" in k:
        return "generate"
    if "k_shade" in k:
        return "shade"
    if "k_accumulate" in k:
        return "accumulate"
    if "k_trace_shadow" in k:
        return "trace_shadow"
    if "k_trace_closest" in k:
        return "trace_closest"
    if "k_trace_persistent" in k or "k_trace_wide8" in k:   # template arguments <STACK, MODE, INSTRUMENT>; MODE 0 closest, 1 shadow, 2 fused
        args = k[k.index("<") + 1:k.rindex(">")].split(",") if "<" in k else []
        mode = args[1].strip() if len(args) > 1 else "2"
        if len(args) > 2 and args[2].strip() == "true":      # the instrumented build (counting passes before the warm-up): not what the timed region runs
            return None
        return {"0": "trace_closest", "1": "trace_shadow"}.get(mode, "trace")
    return None


def collect(root: Path, counter: str):
    total, launches = defaultdict(float), defaultdict(set)
    for f in root.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = bench_name(row["Kernel_Name"])
                if name is None:
                    continue
                total[name] += float(row["Counter_Value"])
                launches[name].add(row["Dispatch_Id"])
    return {n: (total[n] / max(1, len(launches[n])), len(launches[n])) for n in total}


def main():
    key, fetch_dir, write_dir, command = sys.argv[1:5]
    out = Path(sys.argv[5]) if len(sys.argv) > 5 else Path(__file__).resolve().parent.parent / "profiles" / "pmc_traffic.json"
    fetch, write = collect(Path(fetch_dir), "FETCH_SIZE"), collect(Path(write_dir), "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f_kib, f_n = fetch.get(name, (0.0, 0))
        w_kib, w_n = write.get(name, (0.0, 0))
        kernels[name] = {"FETCH_SIZE_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib, "launches_fetch_pass": f_n, "launches_write_pass": w_n,
                         "traffic_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0}
    table = json.loads(out.read_text()) if out.exists() else {}
    import datetime
    import hashlib
    library = Path(__file__).resolve().parent.parent / "bifrost3d_amd" / "csrc" / "libhiprenderer.so"
    sha16 = hashlib.sha256(library.read_bytes()).hexdigest()[:16] if library.exists() else None
    table[key] = {"command": command, "collected": datetime.date.today().isoformat(), "library_sha16": sha16, "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes, separate --pmc passes", "kernels": kernels}
    out.write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
    for name, v in kernels.items():
        print(f"{name:16s} fetch {v['FETCH_SIZE_KiB_per_launch']:12.1f} KiB  write {v['WRITE_SIZE_KiB_per_launch']:12.1f} KiB  traffic {v['traffic_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
