#!/usr/bin/env python3
"""Turns what tools/gpu_evidence.sh <tag> counters left under gpurun_out/<dir>/ into the committed evidence under profiles/:
  <tag>_<scene>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary
  <tag>_<scene>_bench_under_rocprof.json  the bench line of that same profiled run
  pmc_traffic.json                        HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (tools/make_pmc_traffic.py)
  <tag>_atrium_sq_counters.txt            SQ / TCC / TCP counter sums per kernel + derived shares
  sq_limiters.json                        the derived shares bench.py quotes as the observed limiter

usage: python tools/collect_profiles.py gpurun_out/r02p r02
"""
import json
import re
import shutil
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src, tag = Path(sys.argv[1]), sys.argv[2]
profiles = ROOT / "profiles"
KEYS = {"atrium": "atrium:1920x1080:spp64:bounces4:tris260000:wf1", "cornell_diffuse": "cornell_diffuse:1920x1080:spp64:bounces4:wf1",
        "material": "material:1920x1080:spp64:bounces32:wf1"}      # the one-wavefront legs (tools/gpu_evidence.sh counters)

for scene, key in KEYS.items():
    d = src / scene
    if not d.is_dir():
        continue
    stats = sorted(d.glob("trace/**/*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], profiles / f"{tag}_{scene}_kernel_stats.csv")
    line = (d / "bench_trace.json").read_text().strip()
    if line:
        json.loads(line)   # must be the one JSON line
        (profiles / f"{tag}_{scene}_bench_under_rocprof.json").write_text(line + "\n")
    command = f"rocprofv3 --pmc {{FETCH_SIZE|WRITE_SIZE}} -- python3 bench.py --no-cpu-baseline --no-other-workloads --no-rmse --no-plugin --scene {scene} --steps 4 --warmup 1"
    subprocess.run([sys.executable, str(ROOT / "tools" / "make_pmc_traffic.py"), key, str(d / "pmc_fetch"), str(d / "pmc_write"), command], check=True)

# ---- SQ / cache counters of the atrium ---------------------------------------------------------------------------------
sums = defaultdict(dict)
for part in ("sq1", "sq2", "tcc", "tcp"):
    f = src / f"sq_atrium_{part}.txt"
    if not f.exists():
        continue
    kernel = None
    for text in f.read_text().splitlines():
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", text)
        if m:
            kernel = m.group(1)
            continue
        m = re.match(r"^\s+(\S+)\s+([0-9.e+\-]+)\s+per dispatch", text)
        if m and kernel:
            sums[kernel][m.group(1)] = float(m.group(2))


def derived(c):
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    out = {}
    if wave:
        out["valu_active_share"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / wave
        out["waiting_on_memory_share"] = c.get("SQ_WAIT_ANY", 0.0) / wave
        out["issue_stall_share"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wave
    if c.get("SQ_INSTS_VALU"):
        out["lanes_per_valu_instruction"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_INSTS_VALU"]
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVES"):
        if name in c:
            out[{"SQ_INSTS_VALU": "valu_instructions", "SQ_INSTS_SALU": "salu_instructions", "SQ_INSTS_VMEM_RD": "vmem_read_instructions", "SQ_INSTS_LDS": "lds_instructions",
                 "SQ_WAVES": "waves"}[name]] = c[name]
    if c.get("TCC_REQ_sum"):
        out["l2_hit_rate"] = c.get("TCC_HIT_sum", 0.0) / c["TCC_REQ_sum"]
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        out["l1_hit_rate"] = 1.0 - c.get("TCP_TCC_READ_REQ_sum", 0.0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    return out


if sums:
    lines = [f"# SQ / TCC / TCP counters of the atrium bench (bench.py --scene atrium --steps 2 --warmup 1, one wavefront), rocprofv3 --pmc, four separate passes",
             "# (tools/profile_sq.sh, tools/gpu_evidence.sh, tools/collect_profiles.py). Sums over all dispatches of a kernel in the run; SQ_*_CYCLES are quad-cycles summed over waves.",
             "# derived: share of wave time = counter / SQ_WAVE_CYCLES; lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU; L2 hit = TCC_HIT / TCC_REQ; L1 hit = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES",
             ""]
    limiters = {}
    for kernel in sorted(sums):
        c = sums[kernel]
        lines.append(kernel)
        for name in sorted(c):
            lines.append(f"    {name:32s} {c[name]:.6g}")
        dv = derived(c)
        lines.append("    -> " + ", ".join(f"{k} {v:.3g}" for k, v in dv.items()))
        lines.append("")
        limiters[kernel] = dv
    (profiles / f"{tag}_atrium_sq_counters.txt").write_text("\n".join(lines))

    def pick(fragment):
        return next((k for k in limiters if fragment in k), None)

    def entry(kernel, waves):
        d = limiters[kernel]
        return {"valu_active_share_of_wave_time": round(d.get("valu_active_share", 0.0), 3), "waiting_on_memory_share": round(d.get("waiting_on_memory_share", 0.0), 3),
                "issue_stall_share": round(d.get("issue_stall_share", 0.0), 3), "lanes_per_valu_instruction": round(d.get("lanes_per_valu_instruction", 0.0), 1),
                "l1_hit_rate": round(d.get("l1_hit_rate", 0.0), 3), "l2_hit_rate": round(d.get("l2_hit_rate", 0.0), 3), "waves_per_simd": waves}

    table = {"atrium": {"source": f"profiles/{tag}_atrium_sq_counters.txt (rocprofv3 --pmc SQ_*, TCC_*, TCP_* passes of this workload)",
                        "summary": "per-wave latency of dependent gathers and arithmetic at partial lane occupancy, not HBM bytes (DESIGN.md section 5: sensitivity experiments)"}}
    wide8 = pick("k_trace_wide8<12, 2, false")
    fused, shade = wide8 or pick("k_trace_persistent<16, 2, false") or pick("k_trace_persistent<32, 2, false"), pick("k_shade<1, false")
    if fused and wide8:
        table["atrium"]["k_trace_wide8<12, TRACE_FUSED>"] = entry(fused, 6)
        table["atrium"]["summary"] = ("VALU issue (wave time the pipe issues x 6 waves per SIMD) at 37 of 64 lanes per instruction and the address pipeline, not HBM bytes "
                                      "(DESIGN.md section 5: issue-rate table, TA_TA_BUSY, sensitivity experiments)")
    elif fused:
        table["atrium"]["k_trace_persistent<16, TRACE_FUSED, overflow to scratch>" if "<16" in fused else "k_trace_persistent<32, TRACE_FUSED>"] = entry(fused, 6 if "<16" in fused else 5)
    if shade:
        table["atrium"]["k_shade<1, false>"] = entry(shade, 3)
    (profiles / "sq_limiters.json").write_text(json.dumps(table, indent=1) + "\n")
    print(json.dumps(table, indent=1))
