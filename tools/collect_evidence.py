#!/usr/bin/env python3
"""Copies what tools/gpu_evidence.sh <tag> left under gpurun_out/<tag>/ into the committed evidence under profiles/<tag>_* (names: profiles/README.md).
usage: python tools/collect_evidence.py gpurun_out/r04 r04"""
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src, TAG = Path(sys.argv[1]), sys.argv[2]
profiles = ROOT / "profiles"


def one_line(source, target):
    text = (src / source).read_text().strip().splitlines()
    line = [l for l in text if l.startswith("{")][-1]
    json.loads(line)
    (profiles / target).write_text(line + "\n")
    return json.loads(line)


lines = {}
for source, target in (("bench_default.json", TAG + "_bench_final.json"), ("bench_under_rocprof.json", TAG + "_atrium_bench_under_rocprof.json"), ("bench_material.json", TAG + "_bench_material.json"),
                       ("bench_cornell_diffuse.json", TAG + "_bench_cornell_diffuse.json"), ("bench_atrium_textured.json", TAG + "_bench_atrium_textured.json"), ("bench_atrium10M_4k.json", TAG + "_bench_atrium10M_4k.json"),
                       ("bench_atrium_1spp.json", TAG + "_bench_atrium_1spp.json"), ("bench_2rank_gloo_shared_device.json", TAG + "_bench_2rank_gloo_shared_device.json")):
    if (src / source).exists() and (src / source).stat().st_size:
        lines[target] = one_line(source, target)
# the full records behind the compact lines (bench.py --details), and the line of the driver's own command
for source, target in (("bench_details.json", TAG + "_bench_details.json"), ("bench_driver_command.json", TAG + "_bench_driver_command.json"), ("bench_driver_command_details.json", TAG + "_bench_driver_command_details.json"),
                       ("bench_atrium_textured_details.json", TAG + "_bench_atrium_textured_details.json"), ("bench_material_details.json", TAG + "_bench_material_details.json"),
                       ("bench_cornell_diffuse_details.json", TAG + "_bench_cornell_diffuse_details.json"), ("bench_atrium10M_4k_details.json", TAG + "_bench_atrium10M_4k_details.json"),
                       ("bench_atrium_1spp_details.json", TAG + "_bench_atrium_1spp_details.json"), ("bench_2rank_gloo_shared_device_details.json", TAG + "_bench_2rank_gloo_shared_device_details.json"),
                       ("bench_under_rocprof_details.json", TAG + "_atrium_bench_under_rocprof_details.json"), ("verify_probe_1920x1080.json", TAG + "_verify_probe_1920x1080.json"),
                       ("verify_probe_160x90.json", TAG + "_verify_probe_160x90.json")):
    if (src / source).exists() and (src / source).stat().st_size:
        shutil.copy(src / source, profiles / target)
stats = sorted(src.glob("trace/**/*kernel_stats.csv"), key=lambda f: f.stat().st_mtime, reverse=True)      # the newest run
if stats:
    shutil.copy(stats[0], profiles / (TAG + "_atrium_kernel_stats.csv"))
for name in ("rmse_protocol_480x270.json", "rmse_protocol_160x90.json"):
    if (src / name).exists() and (src / name).stat().st_size:
        shutil.copy(src / name, profiles / (TAG + "_" + name))
if (src / "trace_log.txt").exists():
    shutil.copy(src / "trace_log.txt", profiles / (TAG + "_atrium_trace_log.txt"))
log = (src / "gpu_tests.log").read_text() if (src / "gpu_tests.log").exists() else ""
metrics = [l for l in log.splitlines() if re.search(r"IMAGE-METRIC|DENOISER-METRIC|STATISTICS|atrium: pixels within|^\.*VERIFY |^\.*FASTMATH |^\.*DECISIONS |^\.*EXACT ", l)]
if metrics:
    tail = [l for l in log.splitlines() if re.search(r"\d+ passed", l)]
    (profiles / (TAG + "_image_metrics.txt")).write_text("\n".join(metrics + tail) + "\n")
for target, d in lines.items():
    print(f"{target:44} {d['value']:9.1f} {d['unit']:8} {d['ms_per_step']:8.2f} ms/step  roofline.frac {d.get('roofline', {}).get('frac')}")
