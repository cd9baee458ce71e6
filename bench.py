#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HIP path tracer: Mrays/s and ms per 256 spp frame at 1080p, RMSE vs the oracle.

  python bench.py --gpus N --steps K --warmup W
N > 1: one process per GPU over RCCL. Started by torch.distributed.run (RANK / WORLD_SIZE in the environment) the process is
one rank; started plainly (`python bench.py --gpus 8`) it spawns its N rank processes itself -- before it touches a GPU, a
fresh child per rank, never a re-exec -- and forwards rank 0's line.

Workload (the north-star configuration, BASELINE.json configs[3] on ONE GPU at N = 1): the 251 k-triangle procedural atrium
(the Sponza-class stand-in of SURVEY.md 8d: the reference ships no Sponza and there is no network), DefaultShading materials,
1920 x 1080, max_bounce_count 4, next event estimation over 3 RIS candidates. A "step" is one pass of the hot path over one batch:
64 accumulations of the frame traced together (132 710 400 camera paths followed to completion, as two co-running wavefronts), folded one
by one into the f64 running mean and written as half4; four steps are the 256 spp of the metric (`ms_per_256spp_frame`), the default 8 steps two
such frames. Round 3 took 32 per step; 64 give 2 % more rays per second and 128 nothing more (profiles/r04_ab_spp_per_pass.txt). The reference traces
one accumulation per launch; batching is a property of the wavefront design (bit-identical image for any batch size, tested), `--spp-per-pass 1`
reproduces launch-per-accumulation. Inputs (scene, BVH, tables) are resident in HBM before the timed region, the frame stays in HBM.

N > 1 (round 5: the metric's job is the default): tiles of 8 x 8 pixels are dealt round-robin to the ranks and the SAME steps -- 64 accumulations of the
whole 1080p frame each, four of them the 256 spp frame of the metric -- are split over them: every rank traces its tiles of every step ("strong" scaling,
no data-path collective; the line at N = 1 and at N = 8 names the same config.workload), batching up to N steps' worth of accumulations into a pass so that
its wavefront stays as large as the single GPU's (the scaling proxy of round 4 says why). The timed region ends with the RCCL gather of the half4 tiles to
rank 0 and the scatter kernel that assembles the frame. `--weak` is the opt-in of rounds 1-4: N x 64 accumulations per step, per-GPU work fixed.

Rank 0 prints ONE JSON line of under 4 KB (`compact_line`: the contract keys, config, roofline, roofline_valu, cpu_baseline and nothing else -- round 4's
25 KB line could not be parsed by the driver) and writes everything measured to `bench_details.json` (--details) and, as one line, to stderr. The details carry the contract keys plus
  roofline          the kernel with the largest total time. `traffic` = bytes that crossed the L2's memory side per launch, from rocprofv3 counters:
                    (2 * FETCH_SIZE + WRITE_SIZE) * 1024, measured by child runs of this workload under `rocprofv3 --pmc` before the timed run (fallback:
                    profiles/pmc_traffic.json), calibrated on this renderer's access patterns (profiles/r03_fetch_calibration.txt); launches of the
                    INSTRUMENTED build of the kernel (the counting passes) are not in the average (round 3 had them in: 0.38; without: 0.25).
                    `achieved` = traffic / HIP-event launch duration, `frac` = achieved / 8 TB/s -- always <= 1. The timed region runs two co-running
                    wavefronts, whose launches overlap: the durations the rooflines use are those of a ONE-wavefront leg of two steps right after it
                    (`duration_source`; the timed region's own are under `timed_region`). `achieved_model` / `frac_model` price SURVEY.md 8d's
                    ALGORITHMIC bytes (every node visit 64 B, ...) the same way: an upper bound that counts cache hits and may pass 1.
                    `traffic_useful` = the bytes a launch cannot avoid (results written once, queue records read once), `traffic_over_useful`,
                    `write_amplification` = WRITE_SIZE / useful writes;
  roofline_valu     the VALU roof of the same kernel, which is what bounds it: wave64 VALU instructions per second (SQ_INSTS_VALU of a third live
                    counter pass, calibrated on a rate kernel of known instruction count in the same pass) against the v_fma_f32 issue rate measured
                    in this process (hipr_debug_valu_issue_rates), lanes per instruction, and `valu_busy` = SQ_ACTIVE_INST_VALU per second relative
                    to a kernel that does nothing but issue VALU;
  scaling_proxy     the per-GPU shape of the N-way tile split, timed on the one GPU: every phase of a tile_stride-N split rendered in turn, the slowest
                    standing for the step of an N-GPU node; strong / weak / one accumulation per pass, the rate by wavefront size and the smallest
                    wavefront that keeps 90 % of the full rate;
  cpu_baseline      the SmallPT restatement (BASELINE config 1) on the host cores, plus `c2`: the oracle's render of config 2 (Cornell,
                    all Diffuse) at reduced size next to the device's, equal ray counters, like-for-like Mrays/s (N = 1 only);
  other_workloads   BASELINE configs[1] (Cornell, all Diffuse) and configs[2] (material scene, 32 bounces) measured the same way after
                    the main timed region, each with value / ms per 256 spp / roofline / rmse (N = 1 only; --no-other-workloads skips);
  retrace_mode      rays are BVH queries actually made. The 8-wide traversal steps over closest hits the hit program would refuse (back of a one-sided surface)
                    instead of having them refused and retraced as the reference does (hipr_set_backface_culling, default on; same frames): this block is two
                    steps of the same workload with that off -- the ray count and time of the reference's way, beside the line's own (N = 1, outside the timed region);
  config.rmse_vs_oracle   RMSE against the CPU oracle at equal spp and seed, 8 and 256 spp on a 160 x 90 frame, both definitions of SURVEY.md 8d, the bias
                    statistics of the difference, and each image's distance to a converged (16 x spp, disjoint accumulations) oracle image.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
METRIC = "Mrays/sec + ms/frame at 1080p/256spp; per-pixel RMSE vs OptiXRenderer"
# What the headline workload IS, so that lines of different rounds can be told apart: bump when the default step, the stand-in geometry or the
# shade kernel's math flags change. r3: 32 spp per step; r4: 64 spp per step, layered shader ball in the material scene, -freciprocal-math -fapprox-func.
WORKLOAD_VERSION = "r4:spp64:atrium-seed1:shade-approx-rcp"
SCENES = ["atrium", "atrium_textured", "cornell_diffuse", "cornell", "material", "material_coat", "opacity"]


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scene", default="atrium", choices=SCENES)
    p.add_argument("--scene-file", default=None, help="render a model file (.gltf / .glb / .obj) set up the way SimpleViewer sets up a scene from its command line; "
                   "not the headline workload: the line's config.workload names the file")
    p.add_argument("--atrium-triangles", type=int, default=260000)
    p.add_argument("--bounces", type=int, default=None, help="max_bounce_count; default 4, and 32 for the viewer's built-in scenes (apps/SimpleViewer/main.cpp:353)")
    p.add_argument("--spp-per-pass", type=int, default=64, help="accumulations traced together per step and GPU (HiprFrameDesc::samples_per_pass)")
    p.add_argument("--wavefronts", type=int, default=0, choices=[0, 1, 2, 3, 4],
                   help="0 (default): the library's choice -- two half-frame wavefronts on two streams for the small-scene kernels (Cornell box: one shades while the other "
                        "traces) and, since round 4, for the persistent wide-BVH kernels from 2^24 paths per pass on (where one wavefront's launch drains the other's blocks "
                        "move in: +4 %%), one below that. Per-kernel durations of co-running wavefronts overlap; the rooflines are priced on a one-wavefront leg")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --share-device runs the N > 1 code path with every rank on GPU 0 (functional test of tiling / gather / scatter on a 1-GPU box)")
    p.add_argument("--share-device", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-other-workloads", action="store_true", help="skip BASELINE configs[1] and configs[2] after the main timed region")
    p.add_argument("--no-rmse", action="store_true")
    p.add_argument("--no-scaling-proxy", action="store_true", help="skip the one-GPU proxy of the N-GPU tile split (scaling_proxy key)")
    p.add_argument("--no-plugin", action="store_true", help="skip the run through the C++ HIPRenderer::Renderer class (plugin_renderer key)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    p.add_argument("--master-port", type=int, default=0, help="rendezvous port when this process spawns the ranks itself (0: pick a free one)")
    p.add_argument("--pmc-traffic", default="auto", choices=["auto", "file", "off"],
                   help="where roofline.traffic (bytes that crossed the L2's memory side per launch) comes from: auto = two rocprofv3 --pmc child runs of this workload "
                        "(FETCH_SIZE, then WRITE_SIZE; 2 steps each) started BEFORE this process touches the GPU, falling back to the committed profiles/pmc_traffic.json; "
                        "file = that file only; off = none")
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # this process is one of the counter passes: timed region only, no extras
    p.add_argument("--spawn-deadline", type=float, default=900.0, help="seconds the self-spawned ranks get before they are stopped (well under the driver's own timeout)")
    p.add_argument("--fixed-frame", action="store_true", help="the default since round 5 (kept so that older commands still parse): N > 1 splits the SAME job over the ranks -- a step is "
                   "spp-per-pass accumulations of the whole frame, every rank traces its tiles of them: 1/N of the paths per GPU and step, 'strong' scaling")
    p.add_argument("--weak", action="store_true", help="N > 1: N x spp-per-pass accumulations per step instead, so that every GPU keeps the single GPU's paths per step ('weak' scaling; "
                   "the default of rounds 1-4). Not the metric's job: the line says so in `scaling` and config.workload")
    p.add_argument("--details", default=None, help="where the full record goes (default: bench_details.json next to this script; the compact line on stdout names it)")
    p.add_argument("--no-exact-mode", action="store_true", help="skip the exact arithmetic leg (config.exact_mode)")
    p.add_argument("--no-textured", action="store_true", help="skip the textured / cut-out atrium leg (config.workload_textured)")
    return p.parse_args(argv)


# --------------------------------------------------------------------------------------------------------------------------------
# N > 1 started plainly: spawn the ranks (nothing in this function imports torch or touches a GPU)
# --------------------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    import socket
    port = args.master_port
    if port == 0:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    children = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
        children.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                         stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    # Fail fast: every child is polled; the first one that exits non-zero (a rank that died at init would leave the others in the rendezvous until the
    # driver's timeout) ends the rest, and so does the overall deadline. Rank 0's stdout is drained by a thread so that its pipe never fills.
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.append(children[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + args.spawn_deadline
    failed = None
    while failed is None and any(c.poll() is None for c in children):
        for rank, c in enumerate(children):
            code = c.poll()
            if code is not None and code != 0:
                failed = (rank, code)
                break
        if failed is None and time.monotonic() > deadline:
            failed = (-1, 124)
        if failed is None:
            time.sleep(0.2)
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with {failed[1]}; stopping the other ranks\n" if failed[0] >= 0 else
                         f"bench.py: the ranks did not finish within {args.spawn_deadline:.0f} s; stopping them\n")
        for c in children:
            if c.poll() is None:
                c.terminate()
        for c in children:
            try:
                c.wait(timeout=10)
            except subprocess.TimeoutExpired:
                c.kill()
                c.wait()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(captured).decode())
    sys.stdout.flush()
    if failed is not None:
        return abs(failed[1]) or 1
    return max(abs(c.returncode) for c in children)


# --------------------------------------------------------------------------------------------------------------------------------
def make_scene(name, args):
    from bifrost3d_amd.host import Scene
    if name.startswith("file:"):
        scene = Scene(name)
        return scene, f"{os.path.basename(name[5:])} ({scene.desc.triangle_count} triangles) with the SimpleViewer defaults (camera from the scene bounds, one directional light)", 4
    if name == "cornell_diffuse":
        return Scene("cornell", diffuse_only=True), "SimpleViewer Cornell box (34 triangles, 1 sphere light), all materials Diffuse (BASELINE configs[1])", 4
    if name == "cornell":
        return Scene("cornell"), "SimpleViewer Cornell box (34 triangles, 1 sphere light), reference materials", 4
    if name in ("material", "material_coat"):
        return (Scene("material", coat=name == "material_coat"),
                "SimpleViewer material scene (BASELINE configs[2]): 7 shader balls (procedural stand-in for Shaderball.gltf, 179 k triangles) blending dielectric to gold"
                + (", coat 1 / coat roughness 0.7" if name == "material_coat" else "") + ", checkered textured floor, directional light", 32)
    if name == "opacity":
        return Scene("opacity"), "SimpleViewer opacity scene (cut-out box, coverage 0.75 planes, 24 triangles)", 32
    if name == "atrium_textured":     # the stand-in with what the real Sponza brings and the plain one does not: a texture on every material, cut-out banners (host/AtriumScene.h)
        scene = Scene("atrium", param0=args.atrium_triangles, param1=1, textured=True)
        return scene, (f"procedural atrium, {scene.desc.triangle_count} triangles, every material textured (8 tint / roughness textures), cut-out cloth banners "
                       f"(30 % of the triangles): the full kernels with texture samplers and coverage lookups"), 4
    scene = Scene("atrium", param0=args.atrium_triangles, param1=1)
    return scene, f"procedural atrium, {scene.desc.triangle_count} triangles (Sponza-class stand-in, BASELINE configs[3]), DefaultShading", 4


def library_sha16() -> str:
    from bifrost3d_amd import capi
    return hashlib.sha256(Path(capi.LIB_PATH).read_bytes()).hexdigest()[:16]


# the SQ pass of the live counter runs (roofline_valu): wave64 VALU instructions issued, lane-cycles spent in them, quad-cycles the VALU was executing
VALU_COUNTERS = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")
RATE_KERNEL_ITERATIONS = 4096      # hipr_debug_valu_issue_rates: 8 blocks of 4 waves per CU, iterations x 8 instructions per wave


def kernel_bench_name(kernel: str):
    """Bench key of a kernel name as rocprofv3 prints it (the keys of kernel_ms_per_step / roofline_by_kernel)."""
    k = kernel.split("(")[0]
    for needle, name in (("k_generate", "generate"), ("k_shade", "shade"), ("k_accumulate", "accumulate"), ("k_trace_shadow", "trace_shadow"), ("k_trace_closest", "trace_closest")):
        if needle in k:
            return name
    if "k_trace_persistent" in k or "k_trace_wide8" in k:   # template arguments <STACK, MODE, INSTRUMENT, ...>; MODE 0 closest, 1 shadow, 2 fused
        args = k[k.index("<") + 1:k.rindex(">")].split(",") if "<" in k else []
        mode = args[1].strip() if len(args) > 1 else "2"
        if len(args) > 2 and args[2].strip() == "true":
            # the INSTRUMENTED build of the kernel (the two counting passes before the warm-up; it keeps ten counters per lane and spills): not the kernel the
            # timed region runs. Round 3 averaged its launches into the traffic per launch -- 11 GB written per launch against 1.8 GB -- which is where the
            # "7x write amplification" of that round's line came from.
            return None
        return {"0": "trace_closest", "1": "trace_shadow"}.get(mode, "trace")
    return None


def measure_traffic_live(argv):
    """roofline.traffic measured by THIS run: two child processes, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (TCC has four counter slots: the two
    cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), each running this script's workload for 2 steps, started before this process has
    initialised the GPU (fresh children; nothing is exec'ed over a process that touched the device). Per kernel and launch:
        bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
    FETCH_SIZE counts the L2's memory-side read requests (TCC_EA0_RDREQ) at 64 B each while every such request on gfx950 moves a 128 B line: calibrated
    for this renderer's access patterns (16 B per lane gathers of 64 B nodes and 48 B triangles, dword-per-lane spills, 16 B per lane streams) by
    tools/microbench/fetch_calibration.hip -> profiles/r03_fetch_calibration.txt: 2 * FETCH_SIZE * 1024 = RDREQ * 128 B for every pattern, WRITE_SIZE exact.
    The figure is FABRIC-side: Infinity Cache hits are counted (TCC_EA0_RDREQ_DRAM reads the same), no DRAM-side counter is exposed.
    Returns ({bench kernel name: bytes per launch}, source description) or ({}, reason)."""
    import csv
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}, {"error": "rocprofv3 not found"}
    passthrough = []
    skip = 0
    for i, a in enumerate(argv):        # the workload arguments, without the ones that shape the parent run
        if skip:
            skip -= 1
            continue
        if a in ("--steps", "--warmup", "--gpus", "--pmc-traffic", "--cpu-baseline-seconds", "--master-port", "--spawn-deadline", "--wavefronts"):
            skip = 1
            continue
        if a.split("=")[0] in ("--steps", "--warmup", "--gpus", "--pmc-traffic", "--cpu-baseline-seconds", "--master-port", "--spawn-deadline", "--wavefronts"):
            continue
        passthrough.append(a)
    totals = {}
    rate_kernels = {}
    t0 = time.perf_counter()
    # three passes: the TCC has four counter slots (FETCH_SIZE and WRITE_SIZE cannot share one), the SQ counters of the VALU roof go together
    passes = (("FETCH_SIZE",), ("WRITE_SIZE",), VALU_COUNTERS)
    for counters in passes:
        tmp = tempfile.mkdtemp(prefix="hipr_pmc_")
        cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", tmp, "--", sys.executable, str(Path(__file__).resolve())] + passthrough + \
              ["--pmc-child", "--steps", "2", "--warmup", "1", "--gpus", "1", "--wavefronts", "1"]
        label = " ".join(counters)
        try:
            # a pass takes seconds (the first import of torch on a fresh box up to two minutes); its own process group, so that a pass that hangs is ended whole
            child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), start_new_session=True)
            try:
                _, err = child.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)      # exactly the group this call started
                child.communicate()
                shutil.rmtree(tmp, ignore_errors=True)
                return {}, {"error": f"rocprofv3 --pmc {label}: no result after 240 s"}
            r = subprocess.CompletedProcess(cmd, child.returncode, None, err)
        except OSError as e:
            shutil.rmtree(tmp, ignore_errors=True)
            return {}, {"error": f"rocprofv3 --pmc {label}: {e}"}
        sums, launches, durations = {c: {} for c in counters}, {c: {} for c in counters}, {}
        for f in Path(tmp).rglob("*counter_collection.csv"):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    counter = row["Counter_Name"]
                    if counter not in sums:
                        continue
                    raw = row["Kernel_Name"].split("(")[0]
                    name = kernel_bench_name(row["Kernel_Name"])
                    if name is None and "k_rate_" in raw:      # the calibration kernels of hipr_debug_valu_issue_rates: instruction count known by construction
                        name = "rate_" + raw.split("k_rate_")[1].split("<")[0].strip()
                    if name is None:
                        continue
                    sums[counter][name] = sums[counter].get(name, 0.0) + float(row["Counter_Value"])
                    launches[counter].setdefault(name, set()).add(row["Dispatch_Id"])
                    if counter == counters[0] and row.get("End_Timestamp") and row.get("Start_Timestamp"):      # the dispatch's duration under the profiler, once per dispatch
                        durations[name] = durations.get(name, 0.0) + (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9
        shutil.rmtree(tmp, ignore_errors=True)
        if r.returncode != 0 or not any(sums.values()):
            if counters is VALU_COUNTERS:      # the traffic passes stand without the VALU pass
                totals["valu_error"] = f"rocprofv3 --pmc {label} exited with {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"
                continue
            return {}, {"error": f"rocprofv3 --pmc {label} exited with {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"}
        for counter in counters:
            totals[counter] = {n: (sums[counter][n] / len(launches[counter][n]), len(launches[counter][n])) for n in sums[counter]}
        if counters is VALU_COUNTERS:
            totals["valu_seconds"] = {n: (durations[n] / len(launches[counters[0]][n]), len(launches[counters[0]][n])) for n in durations}
    traffic, detail = {}, {}
    for name in set(totals["FETCH_SIZE"]) | set(totals["WRITE_SIZE"]):
        if name.startswith("rate_"):
            continue
        f_kib, f_n = totals["FETCH_SIZE"].get(name, (0.0, 0))
        w_kib, w_n = totals["WRITE_SIZE"].get(name, (0.0, 0))
        traffic[name] = (2.0 * f_kib + w_kib) * 1024.0
        detail[name] = {"FETCH_SIZE_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib, "launches_fetch_pass": f_n, "launches_write_pass": w_n}
    valu = {"error": totals["valu_error"]} if "valu_error" in totals else {c: {n: v[0] for n, v in totals.get(c, {}).items()} for c in VALU_COUNTERS + ("valu_seconds",)}
    return traffic, {"measured": "live: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc " + " ".join(VALU_COUNTERS) + " child runs of this workload (2 steps each) before the timed run",
                     "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; calibration profiles/r03_fetch_calibration.txt", "side": "fabric (L2 memory side: Infinity Cache + HBM)",
                     "seconds": time.perf_counter() - t0, "counters": detail, "valu_counters_per_launch": valu, "stale": False}


def load_measured_traffic(key):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
    runs of this command: tools/profile_round.sh + tools/make_pmc_traffic.py; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950
    correction of MI355X_MICROARCH.md "HBM"). NOT measured by this run: the entry records the command, the date and the hash of the
    library it was collected with, and `stale` says whether that is the library running now."""
    path = ROOT / "profiles" / "pmc_traffic.json"
    if not path.exists():
        return {}, None
    try:
        table = json.loads(path.read_text())
    except ValueError:
        return {}, None
    entry = table.get(key)
    if not entry:
        return {}, None
    source = {"file": "profiles/pmc_traffic.json", "key": key, "command": entry.get("command"), "collected": entry.get("collected"),
              "library_sha16": entry.get("library_sha16"), "stale": entry.get("library_sha16") != library_sha16(),
              "note": "PMC passes of an earlier run of this command (rocprofv3 cannot run inside the timed process); not measured by this run"}
    return {k: v["traffic_bytes_per_launch"] for k, v in entry.get("kernels", {}).items()}, source


def load_observed_limiter(scene_name):
    path = ROOT / "profiles" / "sq_limiters.json"
    if path.exists():
        try:
            return json.loads(path.read_text()).get(scene_name)
        except ValueError:
            pass
    return None


def rmse_against_oracle(ctx, scene, bounces, spps=(8, 256), width=160, height=90, converged_name=None):
    """BASELINE.json's third figure, per-pixel RMSE at equal spp and seed, as SURVEY.md 8(d) writes the protocol. No OptiX image can exist here, so the
    comparand is the CPU oracle (same scene, camera, accumulations 0..spp-1, the search the GPU uses) on a frame small enough for the CPU:
    (i) sqrt(mean over pixels and channels of (a - b)^2), (ii) the reference's ImageOperations::Compare::rms (extensions/ImageOperations/ImageOperations/
    Compare.h:23-43): sqrt(mean(luminance(|a - b|)^2)); plus, to separate bias from noise, the distance of EACH of the two images to a converged oracle
    image of 16 x the spp from disjoint accumulations (profiles/converged/<scene>_<w>x<h>_acc<spp>_<17 spp>.npy, tools/converged_reference.py: the oracle is
    deterministic, the file is its output on host cores) and the mean signed difference with its standard error. The full protocol with the per-sample
    breakdown of the worst pixels is tools/rmse_protocol.py -> profiles/r03_rmse_protocol_*.json. Runs after the timed region."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)   # unorm16 tables, as uploaded to the device
    ctx.set_wavefront_count(1)

    def both(a, b):
        diff = np.abs(a - b)
        luminance = 0.2126 * diff[..., 0] + 0.7152 * diff[..., 1] + 0.0722 * diff[..., 2]   # BF/Math/Color.h luminance()
        return {"rmse_rgb": float(np.sqrt(np.mean(diff ** 2))), "rmse_reference_compare_rms": float(np.sqrt(np.mean(luminance ** 2)))}

    out = {"frame": [width, height], "comparand": "CPU oracle (oracle/integrator.cpp), same seed and search", "north_star_bound": 1e-3}
    for spp in spps:
        batch = min(spp, 32)
        ctx.set_frame(width, height, 0, 1, batch)
        for a in range(0, spp, batch):
            ctx.render_pass(scene.camera(width, height, accumulations=a, max_bounce_count=bounces))
        ctx.synchronize()
        gpu = ctx.read_accumulation()[..., :3]
        cpu, _, seconds = oracle.render(scene.desc, scene.state, scene.camera(width, height, max_bounce_count=bounces), width, height, spp, use_bvh=ctx.oracle_search())
        cpu = cpu[..., :3]
        entry = dict(both(gpu, cpu), mean_radiance=float(cpu.mean()), oracle_seconds=float(seconds))
        d = (gpu - cpu).reshape(-1, 3)
        worst = np.sort((d ** 2).sum(axis=-1))[::-1][:8].sum()
        entry["bias"] = {"mean_signed_difference_rgb": [float(v) for v in d.mean(axis=0)], "standard_error_rgb": [float(v) for v in d.std(axis=0) / np.sqrt(len(d))],
                         "share_of_squared_error_in_8_worst_pixels": float(worst / max((d ** 2).sum(), 1e-300)),
                         "rmse_rgb_without_8_worst_pixels": float(np.sqrt(max((d ** 2).sum() - worst, 0.0) / (3.0 * (len(d) - 8))))}
        stem = ROOT / "profiles" / "converged" / f"{converged_name}_{width}x{height}_acc{spp}_{17 * spp}" if converged_name else None
        if stem is not None and not Path(str(stem) + ".npy").exists() or (stem is not None and Path(str(stem) + ".json").exists() and
                                                                           json.loads(Path(str(stem) + ".json").read_text()).get("triangles") != int(scene.desc.triangle_count)):
            # atriums other than the headline one are filed under their triangle count (tools/converged_reference.py)
            stem = ROOT / "profiles" / "converged" / f"{converged_name}{int(scene.desc.triangle_count)}_{width}x{height}_acc{spp}_{17 * spp}"
        if stem is not None and Path(str(stem) + ".npy").exists() and Path(str(stem) + ".json").exists():
            meta = json.loads(Path(str(stem) + ".json").read_text())
            if meta.get("triangles") == int(scene.desc.triangle_count) and meta.get("bounces") == bounces:      # any of the oracle's searches converges to the same image
                converged = np.load(str(stem) + ".npy").astype(np.float64)
                entry["converged_leg"] = {"reference": f"profiles/converged/{stem.name}.npy: oracle, accumulations [{spp}, {17 * spp}), disjoint from the compared ones",
                                          "device_vs_converged": both(gpu, converged), "oracle_vs_converged": both(cpu, converged)}
        out[f"spp{spp}"] = entry
    return out


def verify_build_leg(ctx, scene, bounces, spp=256, width=160, height=90):
    """Parity of the CODE, exact: the renderer in its EXACT arithmetic mode (hipr_set_arithmetic: the second build of the shade unit inside libhiprenderer.so --
    correctly rounded division / square root, no contraction, sin / cos / pow as the specified f64 sequences of csrc/spec_math.h) renders the rmse_vs_oracle frame
    and is compared with the oracle evaluating the same specified functions: the share of pixels whose f64 running mean is bit-identical (expected: 1.0) -- and the
    fast mode against the exact mode ON THE DEVICE, which is what the fast arithmetic does to the image (tests/test_gpu_verify_build.py).
    The oracle is the checker here, outside the timed region."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    from bifrost3d_amd import capi
    from bifrost3d_amd.renderer import Context
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)

    def render(context):
        batch = min(spp, 32)
        context.set_frame(width, height, 0, 1, batch)
        for a in range(0, spp, batch):
            context.render_pass(scene.camera(width, height, accumulations=a, max_bounce_count=bounces))
        context.synchronize()
        return context.read_accumulation()[..., :3]

    ctx.set_wavefront_count(1)
    product = render(ctx)
    verify = Context(ctx_device(ctx), arithmetic="exact")
    try:
        verify.upload_scene(scene)
        exact = render(verify)
        search = verify.oracle_search()
    finally:
        verify.close()
    before = oracle.lib.oracle_set_f64_transcendentals(1)
    try:
        cpu, _, seconds = oracle.render(scene.desc, scene.state, scene.camera(width, height, max_bounce_count=bounces), width, height, spp, use_bvh=search)
    finally:
        oracle.lib.oracle_set_f64_transcendentals(before)
    cpu = cpu[..., :3]
    identical = (exact == cpu).all(axis=-1)
    d = np.abs(product - exact)
    luminance = 0.2126 * d[..., 0] + 0.7152 * d[..., 1] + 0.0722 * d[..., 2]
    return {"frame": [width, height], "spp": spp, "pixels_bit_identical_to_oracle": float(identical.mean()), "rmse_vs_oracle": float(np.sqrt(np.mean((exact - cpu) ** 2))),
            "product_vs_verify_rmse": float(np.sqrt(np.mean(d ** 2))), "product_vs_verify_compare_rms": float(np.sqrt(np.mean(luminance ** 2))), "oracle_seconds": float(seconds),
            "what": "the renderer's exact arithmetic mode (hipr_set_arithmetic) vs the oracle with the same specified transcendentals; fast mode vs exact mode on the device"}


def ctx_device(ctx) -> int:
    return getattr(ctx, "device_id", 0)


def cpu_baseline_smallpt(seconds: float):
    """SmallPT restatement (oracle/smallpt.cpp, follows apps/SmallPT/smallpt.h:22-147) on the host cores: 256x256, as many
    accumulations as fit the time budget (at most 64, BASELINE.json config 1)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import ctypes as C
    import numpy as np
    from oracle_bindings import get_oracle
    o = get_oracle(False)
    w = h = 256
    fp = C.POINTER(C.c_float)
    # The reference's loop is `#pragma omp parallel for schedule(dynamic, 16)` over the rows (apps/SmallPT/smallpt.h:129): 256 rows are 16 chunks, so at most 16 threads
    # ever have work, whatever the box offers. Round 5's line said "cores: 128" and moved +- 40 % between runs with where the 16 busy threads happened to sit; now the
    # team IS 16 threads (one per chunk), bound to neighbouring cores (OMP_PROC_BIND=close / OMP_PLACES=cores, set in main() before the OpenMP runtime starts), the
    # 64-accumulation job of BASELINE config 1 is repeated for the time budget, and the MEDIAN job is reported.
    available = int(o.lib.oracle_max_threads())
    threads = max(1, min(available, (h + 15) // 16))
    o.lib.oracle_set_threads(threads)
    jobs = []
    t_start = time.perf_counter()
    try:
        while True:
            buf = np.zeros((h, w, 3), np.float32)
            acc = C.c_int(0)
            rays = 0
            t0 = time.perf_counter()
            while acc.value < 64:
                rays += o.lib.oracle_smallpt_accumulate(w, h, buf.ctypes.data_as(fp), C.byref(acc))
            jobs.append((rays / (time.perf_counter() - t0) / 1e6, rays))
            if time.perf_counter() - t_start > seconds or len(jobs) >= 64:
                break
    finally:
        o.lib.oracle_set_threads(available)
    rates = sorted(r for r, _ in jobs)
    median = rates[len(rates) // 2]
    return {"value": median, "unit": "Mrays/s", "cores": threads, "kind": "port", "host_threads_available": available, "jobs": len(jobs), "min": rates[0], "max": rates[-1],
            "thread_binding": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
            "sample": f"SmallPT 9-sphere scene, 256x256 x 64 accumulations ({jobs[0][1]} radiance() rays), {len(jobs)} times in {time.perf_counter() - t_start:.1f} s, median job "
                      f"(min {rates[0]:.1f}, max {rates[-1]:.1f}); OpenMP dynamic,16 over 256 rows = 16 chunks: {threads} threads, bound close"}


def cpu_baseline_c2(ctx, seconds: float):
    """the CPU restatement of the path tracer itself on BASELINE config 2 (Cornell box, all Diffuse, 4 bounces) at reduced
    size, next to the device on the same frame and accumulations: rays counted the same way on both sides (closest-hit traces incl.
    retraces + shadow rays), so the two Mrays/s are like for like."""
    sys.path.insert(0, str(ROOT / "tests"))
    from bifrost3d_amd.host import Scene
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)
    scene = Scene("cornell", diffuse_only=True)
    w, h = 480, 270
    cam = scene.camera(w, h, max_bounce_count=4)
    _, c1, s1 = oracle.render(scene.desc, scene.state, cam, w, h, 1, use_bvh=0)
    spp = int(max(2, min(64, seconds / max(s1, 1e-3))))
    _, cc, cpu_seconds = oracle.render(scene.desc, scene.state, cam, w, h, spp, use_bvh=0)
    ctx.upload_scene(scene)
    ctx.set_wavefront_count(1)
    ctx.set_frame(w, h, 0, 1, spp)
    ctx.render_pass(scene.camera(w, h, accumulations=0, max_bounce_count=4), synchronize=True)      # warm
    ctx.reset_counters()
    t0 = time.perf_counter()
    ctx.render_pass(scene.camera(w, h, accumulations=0, max_bounce_count=4), synchronize=True)
    gpu_seconds = time.perf_counter() - t0
    gc = ctx.counters()
    cpu_rays, gpu_rays = cc["closest_rays"] + cc["shadow_rays"], gc["closest_rays"] + gc["shadow_rays"]
    return {"value": cpu_rays / cpu_seconds / 1e6, "unit": "Mrays/s", "cores": int(oracle.lib.oracle_max_threads()), "kind": "port",
            "sample": f"oracle render of BASELINE config 2 (Cornell box, all Diffuse), {w}x{h}, {spp} accumulations, 4 bounces: {cpu_rays} rays in {cpu_seconds:.1f} s",
            "device_same_sample": {"value": gpu_rays / gpu_seconds / 1e6, "unit": "Mrays/s", "rays": gpu_rays, "seconds": gpu_seconds},
            "counters_equal": {k: [int(cc[k]), int(gc[k])] for k in ("camera_rays", "closest_rays", "shadow_rays", "shaded_hits")},
            "ray_count_relative_difference": abs(cpu_rays - gpu_rays) / max(1, cpu_rays)}


def plugin_renderer_figures(ctx, args, main_figures):
    """The same workload through the plugin class a Bifrost application holds: the atrium built in the Bifrost managers, pulled by
    HIPRenderer::Renderer::handle_updates, one blocking Renderer::render() per accumulation into a device render target (host/host_api.cpp
    hiprh_renderer_bench). render() traces ahead in batches of up to 64 accumulations and folds one per call (bit-identical frames, tested);
    `one_launch_per_accumulation` is the same loop at the reference's launch granularity. Rays per accumulation are those of the main
    measurement (same scene, camera and frame; the ray counts are deterministic)."""
    from bifrost3d_amd.host import renderer_bench
    ctx.set_frame(8, 8)            # give the main context's queues back before the renderer allocates its own
    rays_per_accumulation = main_figures["rays_per_step"] / max(1, main_figures["spp_per_step"])
    out = {}
    import ctypes
    libc = ctypes.CDLL(None)
    for key, max_batch, warmup, calls in (("batched", 64, 320, 128), ("one_launch_per_accumulation", 1, 4, 16)):
        r = renderer_bench(args.atrium_triangles, args.width, args.height, warmup, calls, max_batch)
        libc.fflush(None)      # the renderer announces its device with printf like the reference does (OR/Renderer.cpp:300); stdout is stderr here (main)
        ms = r["milliseconds"] / r["calls"]
        out[key] = {"ms_per_render_call": ms, "Mrays_per_s": rays_per_accumulation / ms / 1e3, "calls_timed": r["calls"], "accumulations_reached": r["accumulations"],
                    "max_batch": max_batch, "warmup_calls": warmup, "triangles": r["triangles"]}
    out["batched"]["fraction_of_c_abi_batched_rate"] = out["batched"]["Mrays_per_s"] / main_figures["value"]
    out["note"] = ("HIPRenderer::Renderer::render(), blocking, one more accumulation in the frame per call; scene through the Bifrost managers. The batches grow with the accumulation "
                   "count (1, 1, 1, 1, 2, 3, ... up to max_batch, reached at 128 accumulations) and the queues with them: the warm-up calls take the renderer past that point, so that no "
                   "queue is re-allocated inside the timed calls (a 64-accumulation pass allocates 29 GB, once)")
    return out


def measured_copy_bandwidth(device):
    """next to the nominal 8 TB/s, what a plain device-to-device copy reaches on this box (bytes read + bytes written per second):
    the practical ceiling of a streaming kernel. 1 GiB buffers, the best of five copies."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.zero_()
    b.copy_(a)
    torch.cuda.synchronize(device)
    best = float("inf")
    for _ in range(5):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        b.copy_(a)
        stop.record()
        stop.synchronize()
        best = min(best, start.elapsed_time(stop))
    del a, b
    return 2.0 * n / (best * 1e-3) / 1e9


def denoiser_figures(args, device):
    """The filter stage of Backend::AIDenoisedPathTracing at the bench's frame size (csrc/denoiser.hip, include/hipr_denoiser_c.h): milliseconds
    per hipr_denoiser_process call on synthetic half4 frames already on the device, and the HBM roofline of the image work. Algorithmic bytes
    per pixel: prepare 16 read + 32 written, each of the 5 passes 32 read + 16 written, finish 32 + 16, output 16 + 8 = 360 B."""
    import torch
    from bifrost3d_amd import denoiser
    W, H = args.width, args.height
    generator = torch.Generator(device=device).manual_seed(1)
    noisy = (torch.rand((H, W, 4), generator=generator, device=device) * 2.0).to(torch.float16)
    albedo = torch.rand((H, W, 4), generator=generator, device=device).to(torch.float16)
    out = torch.empty_like(noisy)
    d = denoiser.Denoiser(device.index or 0)
    settings = denoiser.default_settings()
    for _ in range(3):
        d.process(noisy, albedo, out, W, H, settings)
    calls = 20
    t0 = time.perf_counter()
    for _ in range(calls):
        d.process(noisy, albedo, out, W, H, settings)      # synchronises the denoiser's stream
    ms = (time.perf_counter() - t0) / calls * 1e3
    d.close()
    bytes_per_frame = 360.0 * W * H
    achieved = bytes_per_frame / (ms * 1e-3) / 1e9
    return {"ms_per_frame": ms, "frame": [W, H], "iterations": settings.iterations, "bytes_per_pixel_algorithmic": 360,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None},
            "note": "host-timed, launch + synchronise included; 25 taps per pixel per pass come from L1 / L2, the algorithmic bytes count each plane once per pass"}


# --------------------------------------------------------------------------------------------------------------------------------
# Rooflines. Algorithmic HBM bytes per unit of work, DESIGN.md "Kernels" / SURVEY.md 8d:
#   generate       56 B per path        (origin 16 + direction 16 + meta 8 written + the 16 B radiance slot zeroed; rounds 1-3: 80 B, with the throughput word and a 16-byte meta)
#   trace_closest  40 B path state read (origin, direction, 8-byte meta) + 16 B hit written per ray, + 64 B per BVH node visited + 48 B per triangle tested (8-wide tree: 32 B, two triangles share a 64 B leaf record)
#   shade          60 B per queued ray (hit 16, direction 16, throughput 16 -- not for camera rays --, meta 8, listing index 4) + 16 B origin per ray that did not hit a triangle
#                  + 352 B per shaded hit (triangle 48, shading record 96, material 64, 3 RIS light candidates 144) + 56 B per continued path + 48 B per shadow ray queued
#                  + 32 B radiance read-modify-write per queued ray
#   trace_shadow   48 B record + 32 B radiance rmw per shadow ray, + 64 B per node + 48 B per triangle
#   accumulate     16 B radiance per sample + 64 B f64 accumulation rmw + 8 B half4 per owned pixel
# --------------------------------------------------------------------------------------------------------------------------------
def rooflines_of(counters, times, per_ray, small, fused, samples_per_step, traffic, triangle_bytes=48.0):
    n_closest, n_shadow, n_camera, n_hits = (counters[k] for k in ("closest_rays", "shadow_rays", "camera_rays", "shaded_hits"))
    tri_share = 1.0 / 64.0 if small else 1.0   # exhaustive search: the triangle array is read once per 64-ray wave through the scalar cache
    kernel_bytes = {
        "generate": 56.0 * n_camera,
        "trace_closest": n_closest * (40 + 16 + 64 * per_ray["nodes"] + triangle_bytes * per_ray["triangles"] * tri_share),
        "shade": 60.0 * n_closest - 16.0 * n_camera + 16.0 * max(0, n_closest - n_hits) + 352.0 * n_hits + 56.0 * max(0, n_closest - n_camera) + 48.0 * n_shadow + 32.0 * n_closest,
        "trace_shadow": n_shadow * (48 + 32 + 64 * per_ray["shadow_nodes"] + triangle_bytes * per_ray["shadow_triangles"] * tri_share),
        "accumulate": 16.0 * n_camera + (64.0 + 8.0) * n_camera / max(1, samples_per_step),
    }
    kernel_names = {"generate": "k_generate", "trace_closest": "k_trace_closest_small" if small else "k_trace_closest / k_trace_wide8<TRACE_CLOSEST>", "shade": "k_shade",
                    "trace_shadow": "k_trace_shadow_small" if small else "k_trace_shadow", "accumulate": "k_accumulate",
                    "trace": "k_trace_wide8<TRACE_FUSED> (closest-hit rays of bounce k + shadow rays of bounce k-1 over the 8-wide tree with leaf records)"}
    kernel_times = dict(times)
    if fused:   # one launch serves both ray kinds: bytes and time of the two are reported together
        kernel_bytes["trace"] = kernel_bytes.pop("trace_closest") + kernel_bytes.pop("trace_shadow")
        a, b = kernel_times.pop("trace_closest"), kernel_times.pop("trace_shadow")
        kernel_times["trace"] = {"ms": a["ms"] + b["ms"], "launches": a["launches"] + b["launches"]}
    rooflines = {}
    for name, t in kernel_times.items():
        if not t or t["ms"] <= 0 or t["launches"] == 0:
            continue
        nbytes = kernel_bytes.get(name)
        seconds_per_launch = t["ms"] / t["launches"] * 1e-3
        entry = {"bound": "hbm", "kernel": kernel_names.get(name, name), "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": t["ms"] / t["launches"],
                 "launches": t["launches"], "total_ms": t["ms"]}
        measured = traffic.get(name)
        entry["traffic"] = measured
        if nbytes is not None:      # SURVEY.md 8d's byte model: an upper bound on what the algorithm touches, most of it served by L1 / L2
            model_gbs = nbytes / (t["ms"] * 1e-3) / 1e9
            entry.update({"achieved_model": model_gbs, "frac_model": model_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes / t["launches"],
                          "model": "SURVEY.md 8d: every node visit 64 B, every triangle test 48 B, queue records once; counts cache hits, so it can exceed the HBM peak"})
        if measured:                # the headline fraction: bytes that actually crossed the L2's memory side (rocprofv3 counters) / duration / peak
            counter_gbs = measured / seconds_per_launch / 1e9
            entry.update({"achieved": counter_gbs, "frac": counter_gbs / HBM_PEAK_GBS, "frac_basis": "counters: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch / HIP-event launch duration / peak"})
        elif nbytes is not None:    # no counters for this kernel: the model, capped by what the roof allows, and marked as such
            entry.update({"achieved": min(entry["achieved_model"], HBM_PEAK_GBS), "frac": min(entry["frac_model"], 1.0),
                          "frac_basis": "byte model (no counter pass available for this kernel), capped at the peak"})
        rooflines[name] = entry
    return rooflines, kernel_times


def valu_roofline(name, kernel_text, seconds_per_launch, valu, rates, compute_units):
    """The VALU roof beside the HBM one: wave64 VALU instructions per second of the dominant kernel against what the device issues of
    v_fma_f32 chains, both measured by this run -- the counters by the SQ child pass, the peak by hipr_debug_valu_issue_rates in this process. The SQ counters are
    calibrated on the rate kernel of the same pass (instruction count known by construction, VALU busy by construction)."""
    if not valu or "error" in valu or name not in valu.get("SQ_INSTS_VALU", {}):
        return {"bound": "valu", "kernel": kernel_text, "error": (valu or {}).get("error", "no SQ counter pass for this kernel")}
    insts, lane_cycles, active = (valu[c].get(name, 0.0) for c in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"))
    expected_rate_insts = compute_units * 8 * 4 * RATE_KERNEL_ITERATIONS * 8.0
    rate_insts = valu["SQ_INSTS_VALU"].get("rate_fma", 0.0)
    scale = expected_rate_insts / rate_insts if rate_insts > 0 else 1.0          # 1.0 when the counter sees every SIMD
    rate_lanes = valu["SQ_THREAD_CYCLES_VALU"].get("rate_fma", 0.0) / rate_insts if rate_insts > 0 else 64.0
    peak = rates["v_fma_f32"]
    achieved = scale * insts / seconds_per_launch
    entry = {"bound": "valu", "kernel": kernel_text, "achieved": achieved * 1e-9, "peak": peak * 1e-9, "unit": "G wave64 VALU instructions/s", "frac": achieved / peak,
             "instructions_per_launch": scale * insts, "lanes_per_instruction": 64.0 * (lane_cycles / insts) / rate_lanes if insts > 0 and rate_lanes > 0 else None,
             "peak_by_kind": {k: v * 1e-9 for k, v in rates.items()},
             "peak_note": "v_fma_f32 chains, eight waves per SIMD (hipr_debug_valu_issue_rates, measured in this process); v_max_f32 / v_cvt_f32_ubyteN issue at the lower rates "
                          "beside it, so a kernel made of all three saturates the pipe below frac 1",
             "calibration": {"rate_kernel_instructions_expected": expected_rate_insts, "rate_kernel_instructions_counted": rate_insts, "scale": scale,
                             "rate_kernel_lane_cycles_per_instruction": rate_lanes}}
    secs = valu.get("valu_seconds", {})
    rate_active = valu["SQ_ACTIVE_INST_VALU"].get("rate_fma", 0.0)
    if secs.get(name) and secs.get("rate_fma") and rate_active > 0:
        # share of time the VALU executes, relative to a kernel that does nothing else (both under the profiler, same pass)
        entry["valu_busy"] = (active / secs[name]) / (rate_active / secs["rate_fma"])
        entry["valu_busy_basis"] = "SQ_ACTIVE_INST_VALU per second of this kernel / of the v_fma_f32 rate kernel (same counter pass, dispatch timestamps)"
    return entry


def useful_traffic(name, counters, launches, samples_per_step):
    """Bytes a launch of the kernel has to move whatever the caches do: results written once + queue records read once. Set against
    roofline.traffic (what crossed the L2's memory side) it shows the re-reads and the partial-line / spill writes."""
    n_closest, n_shadow, n_camera = counters["closest_rays"], counters["shadow_rays"], counters["camera_rays"]
    if name in ("trace", "trace_closest", "trace_shadow"):
        closest = n_closest if name != "trace_shadow" else 0
        shadow = n_shadow if name != "trace_closest" else 0
        writes = 16.0 * closest + 16.0 * shadow                 # a hit record per closest-hit ray; the radiance slot of a shadow ray's path
        reads = 40.0 * closest + (48.0 + 16.0) * shadow        # ray records (origin, direction, 8-byte slot / id; shadow rays: + radiance carried) + the radiance slot read for the add
        return {"writes": writes / launches, "reads": reads / launches, "bytes": (writes + reads) / launches,
                "what": "per launch: 16 B hit per closest-hit ray + 16 B radiance per shadow ray written; 40 B ray record per closest-hit ray, 48 B per shadow ray + its 16 B radiance slot read; the tree itself "
                        "(10 MB at 251 k triangles) is compulsory once per launch at most and left out"}
    return None


XGMI_LINK_GBS = 153.0      # one xGMI link per peer, MI355X_MICROARCH.md


def fixed_frame_batch(steps: int, world: int) -> int:
    """The fixed frame split over `world` ranks: how many steps' worth of accumulations a rank traces per pass -- the largest g <= world that divides the timed
    step count, so that exactly `steps` steps are timed (a rank of N holds 1 / N of the pixels: N steps per pass make its wavefront the single GPU's). The
    warm-up needs no such divisor: its remainder runs as one shorter pass on the queues the full passes allocated (`measure`)."""
    if world <= 1:
        return 1
    return max(g for g in range(1, world + 1) if steps % g == 0)


def scaling_proxy(ctx, scene, bounces, args, device, t1_ms):
    """The per-GPU shape of north_star's 8-GPU job, timed on the one GPU there is: the frame's 8 x 8 tiles dealt round robin to N
    ranks (tile_stride N), every phase 0 .. N-1 rendered in turn by this device, the slowest phase standing for the step of an N-GPU node.
      strong: the SAME job split N ways (spp_per_pass accumulations of the whole frame per step; `bench.py --gpus N --fixed-frame`);
      weak:   N x the accumulations per step, so that a rank traces as many paths as the single GPU does (`bench.py --gpus N`, the default);
      interactive: one accumulation per pass, split N ways (a camera move on an N-GPU node).
    Not in it: the gather of the half4 tiles to rank 0 (bounded below from the xGMI link rate) and any imbalance between devices."""
    import torch
    from bifrost3d_amd import distributed
    W, H, S0 = args.width, args.height, args.spp_per_pass

    def phases(n, samples, timed, warm=1):
        out, paths = [], 0
        for phase in range(n):
            ctx.set_frame(W, H, tile_phase=phase, tile_stride=n, samples_per_pass=samples)
            # a rank of an N-way split writes its tiles compactly (pitch 0); the single GPU writes the frame
            compact = torch.zeros((distributed.padded_pixels_per_rank(W, H, n), 4) if n > 1 else (H, W, 4), dtype=torch.float16, device=device)
            pitch = 0 if n > 1 else W
            torch.cuda.synchronize(device)
            a = 0
            for _ in range(warm):
                ctx.render_pass(scene.camera(W, H, accumulations=a, max_bounce_count=bounces), compact.data_ptr(), pitch)
                a += samples
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(timed):
                ctx.render_pass(scene.camera(W, H, accumulations=a, max_bounce_count=bounces), compact.data_ptr(), pitch)
                a += samples
            ctx.synchronize()
            out.append((time.perf_counter() - t0) / timed * 1e3)
            paths = max(paths, ctx.owned_pixel_count() * samples)
            del compact
        return out, paths

    frame_bytes = W * H * 8
    proxy = {"strong": {}, "weak": {}, "interactive": {}, "t1_ms_per_step": t1_ms}
    one_spp, _ = phases(1, 1, 16, warm=4)
    for n in (1, 2, 4, 8):
        gather_ms = (frame_bytes / n) / (XGMI_LINK_GBS * 1e9) * 1e3 if n > 1 else 0.0      # every peer sends its 1/N of the frame over its own link to rank 0
        if n == 1:
            proxy["strong"]["1"] = {"ms_per_step": t1_ms, "paths_per_gpu_per_step": W * H * S0, "predicted_speedup": 1.0, "predicted_efficiency": 1.0}
            proxy["interactive"]["1"] = {"ms_per_pass": one_spp[0], "paths_per_gpu_per_pass": W * H, "predicted_speedup": 1.0}
            continue
        strong, strong_paths = phases(n, S0, 2)
        weak, weak_paths = phases(n, S0 * n, 1)
        inter, inter_paths = phases(n, 1, 16, warm=4)
        proxy["strong"][str(n)] = {"ms_per_step": max(strong), "ms_per_step_by_phase": strong, "paths_per_gpu_per_step": strong_paths, "gather_ms_lower_bound": gather_ms,
                                   "predicted_speedup": t1_ms / (max(strong) + gather_ms), "predicted_efficiency": t1_ms / (max(strong) + gather_ms) / n}
        proxy["weak"][str(n)] = {"ms_per_step": max(weak), "ms_per_step_by_phase": weak, "paths_per_gpu_per_step": weak_paths, "gather_ms_lower_bound": gather_ms,
                                 "predicted_speedup": n * t1_ms / (max(weak) + gather_ms), "predicted_efficiency": t1_ms / (max(weak) + gather_ms)}
        proxy["interactive"][str(n)] = {"ms_per_pass": max(inter), "paths_per_gpu_per_pass": inter_paths, "predicted_speedup": one_spp[0] / (max(inter) + gather_ms)}
    ctx.set_frame(W, H, tile_phase=0, tile_stride=1, samples_per_pass=S0)
    # smallest per-GPU wavefront that kept 90 % of the single-GPU rate per path, over everything measured above
    rate1 = W * H * S0 / t1_ms
    points = [(v["paths_per_gpu_per_step"], v["paths_per_gpu_per_step"] / v["ms_per_step"] / rate1) for kind in ("strong", "weak") for v in proxy[kind].values()]
    points += [(v["paths_per_gpu_per_pass"], v["paths_per_gpu_per_pass"] / v["ms_per_pass"] / rate1) for v in proxy["interactive"].values()]
    points.sort()
    proxy["rate_by_wavefront"] = [{"paths_per_gpu": p, "share_of_full_rate": r} for p, r in points]
    enough = [p for p, r in points if r >= 0.9]
    proxy["min_paths_per_gpu_for_90_percent"] = min(enough) if enough else None
    proxy["note"] = ("one device renders every phase of an N-way round-robin tile split in turn; ms_per_step = the slowest phase; predicted_speedup = T(1) / (that + the gather's "
                     f"lower bound, frame / N over one {XGMI_LINK_GBS:.0f} GB/s xGMI link per peer); measured on ONE GPU: no second device, no RCCL in it")
    return proxy


def measure(ctx, scene, scene_name, bounces, args, rank, world, device, steps, warmup, sync):
    """Instrumented passes (per-ray node / triangle counts), warmup, then exactly `steps` timed steps. Returns the figures of this rank."""
    import torch
    import torch.distributed as dist
    from bifrost3d_amd import capi, distributed
    W, H = args.width, args.height
    # Accumulations a rank traces per pass. The default (strong scaling: the same `steps` x spp_per_pass accumulations of the whole frame, every rank its
    # tiles): a rank of N holds 1 / N of the pixels, so it takes up to N steps' worth of accumulations per pass where the step count allows -- the wavefront
    # the scaling proxy shows it needs (a pass of 8 M paths runs at 79 %% of the rate of one of 66 M: scaling_proxy.rate_by_wavefront) -- and makes
    # steps / batch passes. --weak: N x the single GPU's accumulations per step, so that a rank's wavefront is as large as the single GPU's.
    fixed = not args.weak
    batch = fixed_frame_batch(steps, world) if fixed else 1
    S = args.spp_per_pass * (batch if fixed else world)
    steps_run, (warmup_run, warmup_rest) = steps // batch, divmod(warmup, batch)
    on_host = world > 1 and args.dist_backend == "gloo"
    ctx.upload_scene(scene)
    ctx.set_wavefront_count(args.wavefronts)
    ctx.set_frame(W, H, tile_phase=rank, tile_stride=world, samples_per_pass=S)
    n_compact = distributed.padded_pixels_per_rank(W, H, world)
    frame = torch.zeros((H, W, 4), dtype=torch.float16, device=device) if rank == 0 else None
    compact = torch.zeros((n_compact, 4), dtype=torch.float16, device=device) if world > 1 else None
    torch.cuda.synchronize(device)   # the fills above ran on torch's stream; the wavefronts render on their own

    def run_pass(accumulation):
        cam = scene.camera(W, H, accumulations=accumulation, max_bounce_count=bounces)
        if world == 1:
            ctx.render_pass(cam, frame.data_ptr(), W)
        else:
            ctx.render_pass(cam, compact.data_ptr(), 0)

    def finish_frame():
        if world == 1:
            return
        ctx.synchronize()   # the context renders on its own streams, torch.distributed on torch's: order them explicitly
        if getattr(args, "gather_group", None) is None:
            injected = os.environ.get("HIPR_BENCH_TEST_FAIL_GATHER") == "1"      # test hook (tests/test_gpu_bench.py): the first gather raises on every rank
            try:
                if injected:
                    raise RuntimeError("injected by HIPR_BENCH_TEST_FAIL_GATHER")
                gathered = distributed.gather_to_root(compact.cpu() if on_host else compact, world, rank)
            except Exception as e:      # RCCL refused the gather (on every rank alike; a collective that merely stalls ends the job at the process group's timeout instead)
                if on_host and not injected:
                    raise
                sys.stderr.write(f"bench.py: rank {rank}: the RCCL gather failed ({str(e)[:200]}); the tiles go through the host (gloo) from now on\n")
                args.gather_group = dist.new_group(backend="gloo")
        if getattr(args, "gather_group", None) is not None:
            gathered = distributed.gather_to_root(compact.cpu(), world, rank, group=args.gather_group)
            on_host_now = True
        else:
            on_host_now = on_host
        if rank == 0:
            if on_host_now:
                gathered = gathered.to(device)
            torch.cuda.current_stream(device).synchronize()
            ctx.scatter_tiles(gathered.data_ptr(), n_compact, world, W, H, frame.data_ptr(), W)
            ctx.synchronize()

    ctx.set_instrumentation(True)
    ctx.reset_counters()
    for a in (0, S):
        run_pass(a)
    ctx.synchronize()
    ic = ctx.counters()
    ctx.set_instrumentation(False)
    per_ray = {"nodes": ic["closest_nodes"] / max(1, ic["closest_rays"]), "triangles": ic["closest_triangles"] / max(1, ic["closest_rays"]),
               "shadow_nodes": ic["shadow_nodes"] / max(1, ic["shadow_rays"]), "shadow_triangles": ic["shadow_triangles"] / max(1, ic["shadow_rays"])}

    a = 2 * S
    if warmup_rest:     # the warm-up steps that do not fill a pass: one shorter pass on the queues the instrumented passes above allocated at full size (buffers never shrink)
        ctx.set_frame(W, H, tile_phase=rank, tile_stride=world, samples_per_pass=args.spp_per_pass * warmup_rest)
        run_pass(a)
        ctx.synchronize()
        ctx.set_frame(W, H, tile_phase=rank, tile_stride=world, samples_per_pass=S)
        a += S
    for _ in range(warmup_run):
        run_pass(a)
        a += S
    finish_frame()
    sync()
    ctx.reset_counters()
    ctx.reset_timers()

    sync()
    t0 = time.perf_counter()
    for _ in range(steps_run):
        run_pass(a)
        a += S
    t_gather = time.perf_counter()
    finish_frame()
    gather_ms = (time.perf_counter() - t_gather) * 1e3
    sync()
    elapsed = time.perf_counter() - t0

    ctx.synchronize()
    counters, times = ctx.counters(), ctx.kernel_times()
    if args.pmc_child:
        ctx.valu_issue_rates()      # calibration kernels of the SQ pass: their instruction count is known by construction
    stats = torch.tensor([elapsed, counters["closest_rays"], counters["shadow_rays"], counters["camera_rays"]], dtype=torch.float64, device=device)
    if world > 1:
        if on_host:
            stats = stats.cpu()
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        per_rank = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(per_rank, stats)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
    total = {"closest_rays": float(stats[1]), "shadow_rays": float(stats[2]), "camera_rays": float(stats[3])}
    result = {"elapsed": elapsed, "total": total, "counters": counters, "times": times, "per_ray": per_ray, "S": S, "gather_ms": gather_ms, "passes": steps_run, "steps_per_pass": batch,
              "rank_elapsed": [float(t[0]) for t in per_rank] if world > 1 else [elapsed]}
    if rank == 0:
        result["frame_ok"] = bool(torch.isfinite(frame.float()).all().item()) and float(frame[..., :3].float().mean()) > 0
        result["small"] = ctx.trace_variant() == capi.TRACE_EXHAUSTIVE
        result["fused"] = ctx.trace_is_fused()
        result["wide8"] = ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT
    result["wavefronts"] = ctx.wavefront_count()
    if world == 1 and result.get("wide8") and not args.pmc_child and not getattr(args, "under_profiler", False) and not getattr(args, "alone_leg", False) and not getattr(args, "skip_retrace", False):      # kept out of profiled runs: their per-kernel averages are the line's own launches
        # The same frames with every refused closest hit retraced, as the reference does it (hipr_set_backface_culling 0): a short run beside the line's own,
        # for the ray count and the time the stepping saves. Outside the timed region.
        ctx.set_backface_culling(False)
        try:
            run_pass(a)
            ctx.synchronize()
            ctx.reset_counters()
            t1 = time.perf_counter()
            for k in range(2):
                run_pass(a + (k + 1) * S)
            ctx.synchronize()
            seconds = time.perf_counter() - t1
            rc = ctx.counters()
        finally:
            ctx.set_backface_culling(True)
        rays = rc["closest_rays"] + rc["shadow_rays"]
        result["retrace_mode"] = {"Mrays_per_s": rays / seconds * 1e-6, "ms_per_step": seconds / 2 * 1e3, "rays_per_step": rays / 2, "closest_rays_per_step": rc["closest_rays"] / 2,
                                  "steps": 2, "note": "hipr_set_backface_culling(0): closest hits on the back of one-sided surfaces go to the hit program, which refuses them and has the ray "
                                                      "traced again (ORS/MonteCarlo.cu:147-164), one more BVH query each; the line's own figures step over them in the traversal -- "
                                                      "same frames, fewer rays"}
    return result


def ranks_proof(ctx, scene, bounces, args, rank, world, device, width=160, height=90, spp=8):
    """What lets a reader trust an N > 1 line: how many ranks the process group really has -- dist.get_world_size() AND the sum of an
    all_reduce of ones over the group the gather uses --, which devices they sit on (the PCI bus ids, gathered), the collective library's version, and a probe INSIDE
    the run: the ranks render a 160 x 90 frame as their round-robin tiles, the tiles are gathered and assembled on rank 0 exactly as the timed frames are, and rank 0
    renders the same frame alone; the two half4 frames must be equal bit for bit (the RNG is a pure function of pixel, accumulation and bounce): `tile_split_probe.identical`.
    Called by every rank after the timed region; returns the record on rank 0."""
    import torch
    import torch.distributed as dist
    from bifrost3d_amd import distributed
    on_host = args.dist_backend == "gloo" or getattr(args, "gather_group", None) is not None
    group = getattr(args, "gather_group", None)
    ones = torch.ones(1, dtype=torch.int64, device="cpu" if on_host else device)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=group)
    properties = torch.cuda.get_device_properties(device)
    bus = getattr(properties, "pci_bus_id", None)
    mine = {"rank": rank, "device_index": device.index, "name": properties.name,
            "pci": (f"{getattr(properties, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(properties, 'pci_device_id', 0):02x}" if bus is not None else None),
            "uuid": str(getattr(properties, "uuid", "")) or None, "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    try:
        version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:      # a build of torch without it
        version = f"unavailable ({str(e)[:60]})"

    def render(phase, stride, out_ptr, pitch):
        ctx.set_frame(width, height, tile_phase=phase, tile_stride=stride, samples_per_pass=spp)
        ctx.render_pass(scene.camera(width, height, accumulations=0, max_bounce_count=bounces), out_ptr, pitch)
        ctx.synchronize()

    ctx.set_wavefront_count(1)
    n_compact = distributed.padded_pixels_per_rank(width, height, world)
    compact = torch.zeros((n_compact, 4), dtype=torch.float16, device=device)
    torch.cuda.synchronize(device)
    render(rank, world, compact.data_ptr(), 0)
    gathered = distributed.gather_to_root(compact.cpu() if on_host else compact, world, rank, group=group)
    record = None
    if rank == 0:
        assembled = torch.zeros((height, width, 4), dtype=torch.float16, device=device)
        alone = torch.zeros((height, width, 4), dtype=torch.float16, device=device)
        if on_host:
            gathered = gathered.to(device)
        torch.cuda.synchronize(device)
        ctx.scatter_tiles(gathered.data_ptr(), n_compact, world, width, height, assembled.data_ptr(), width)
        ctx.synchronize()
        render(0, 1, alone.data_ptr(), width)
        differing = int((assembled.view(torch.int16) != alone.view(torch.int16)).any(dim=-1).sum().item())
        lit = float(alone[..., :3].float().mean().item())
        record = {"world_seen": {"get_world_size": dist.get_world_size(), "all_reduce_of_ones": int(ones.item())}, "devices": everyone,
                  "distinct_devices": len({(d["pci"], d["uuid"]) for d in everyone}) if all(d["pci"] or d["uuid"] for d in everyone) else None,
                  "backend": dist.get_backend(), "rccl_version": version,
                  "tile_split_probe": {"frame": [width, height], "spp": spp, "pixels_differing_from_one_rank": differing, "identical": differing == 0 and lit > 0.0, "mean_radiance": lit}}
        if differing:      # reported, not fatal: one rank leaving here would strand the others in the closing barrier; the line says `identical: false` and the tests assert on it
            sys.stderr.write(f"bench.py: the {world}-rank frame differs from the 1-rank frame in {differing} pixels of the {width} x {height} probe\n")
    ctx.set_wavefront_count(args.wavefronts)
    return record


def summarise(result, scene_name, scene_text, bounces, args, world, steps, live_traffic=None):
    """The figures of one measured workload as the bench line reports them (rank 0)."""
    W, H, S = args.width, args.height, result["S"] // result.get("steps_per_pass", 1)      # accumulations per STEP (a fixed-frame rank batches several steps into a pass)
    key = f"{scene_name}:{W}x{H}:spp{args.spp_per_pass}:bounces{bounces}"
    if scene_name == "atrium":
        key += f":tris{args.atrium_triangles}"
    key += f":wf{result['wavefronts']}"
    traffic, traffic_source = ({}, None)
    if world == 1:
        if live_traffic and live_traffic[0]:
            traffic, traffic_source = live_traffic
        elif args.pmc_traffic != "off":
            traffic, traffic_source = load_measured_traffic(key)
            if live_traffic and traffic_source is not None:
                traffic_source["live_measurement_failed"] = live_traffic[1]
    rooflines, kernel_times = rooflines_of(result["counters"], result["times"], result["per_ray"], result["small"], result["fused"], result["S"], traffic,
                                           triangle_bytes=32.0 if result.get("wide8") else 48.0)   # a 64 B leaf record holds two triangles
    dominant = max(rooflines, key=lambda n: rooflines[n]["total_ms"])
    roofline = dict(rooflines[dominant])
    roofline["traffic_source"] = traffic_source
    if result["wavefronts"] > 1:
        roofline["co_running"] = "two half-frame wavefronts on two streams: this kernel's launches overlap the other wavefront's kernels, so avg_launch_ms is a co-running duration"
    limiter = load_observed_limiter(scene_name)
    roofline["observed_limiter"] = limiter if limiter else "see DESIGN.md 'What bounds the kernels': the counters put these kernels on VALU issue and gather latency, not on HBM bytes"
    roofline.update({"nodes_per_ray": result["per_ray"]["nodes"], "triangles_per_ray": result["per_ray"]["triangles"], "shadow_nodes_per_ray": result["per_ray"]["shadow_nodes"],
                     "shadow_triangles_per_ray": result["per_ray"]["shadow_triangles"], "selection": "kernel with the largest total time in the timed region"})
    total = result["total"]
    rays = total["closest_rays"] + total["shadow_rays"]
    elapsed = result["elapsed"]
    return {"value": rays / elapsed / 1e6, "ms_per_step": elapsed / steps * 1e3, "ms_per_256spp_frame": elapsed / (steps * S) * 1e3 * 256,
            "workload": f"{scene_text}; {W}x{H}, {S} accumulation(s) per step ({W * H * S} paths per step), max_bounce_count {bounces}, "
                        f"next_event_sample_count 3, path regularisation PDF_scale 0.5; f64 accumulation + half4 output",
            "spp_per_step": S, "steps": steps, "spp_total": S * steps, "rays_per_step": rays / steps, "closest_rays": total["closest_rays"], "shadow_rays": total["shadow_rays"],
            "pixel_samples": total["camera_rays"], "frame_finite_and_lit": result["frame_ok"], "roofline": roofline, "roofline_by_kernel": rooflines,
            "kernel_ms_per_step": {name: v["ms"] / steps for name, v in kernel_times.items()}}



# --------------------------------------------------------------------------------------------------------------------------------
# The line the driver parses. Round 4's single line had grown to 25 KB and the driver's record of it could not be parsed (BENCH_r04.json: parsed null);
# the line is now a fixed, small selection of the full record, which goes to bench_details.json.
# --------------------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _clean(value, digits=6):
    """JSON-safe copy: floats rounded to `digits` significant digits, NaN / infinity as null (json.dumps(allow_nan=False) must succeed)."""
    import math
    if isinstance(value, bool) or value is None or isinstance(value, (int, str)):
        return value
    if isinstance(value, float):
        if not math.isfinite(value):
            return None
        return float(f"{value:.{digits}g}")
    if isinstance(value, dict):
        return {str(k): _clean(v, digits) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_clean(v, digits) for v in value]
    try:
        return _clean(float(value), digits)
    except (TypeError, ValueError):
        return str(value)


def _pick(source, keys):
    return {k: source[k] for k in keys if isinstance(source, dict) and k in source}


def compact_line(full: dict, details_path=None) -> dict:
    """The one line of stdout: the contract keys, config, roofline, roofline_valu and cpu_baseline of `full` (the complete record), and nothing else.
    Text fields are cut to fixed lengths; if the result were still over LINE_LIMIT bytes the optional fields go, least important first."""
    def text(value, limit):
        value = "" if value is None else str(value)
        return value if len(value) <= limit else value[:limit - 3] + "..."

    config = full.get("config", {})
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    c = _pick(config, ("frame", "spp_per_step", "spp_total", "ms_per_256spp_frame", "rays_per_step", "parallelism", "wavefronts", "frame_finite_and_lit", "library_sha16", "workload_version"))
    c["workload"] = text(config.get("workload"), 420)
    rmse = config.get("rmse_vs_oracle") or {}
    if rmse:
        c["rmse_vs_oracle"] = {"frame": rmse.get("frame"), "comparand": "CPU oracle, equal spp and seed", "bound": rmse.get("north_star_bound")}
        for spp in ("spp8", "spp256"):
            if spp in rmse:
                c["rmse_vs_oracle"][spp] = _pick(rmse[spp], ("rmse_rgb", "rmse_reference_compare_rms"))
    for key in ("workload_textured", "rmse_full_size"):
        if key in config:
            c[key] = config[key]
    if isinstance(config.get("exact_mode"), dict):
        c["exact_mode"] = _pick(config["exact_mode"], ("value", "unit", "ms_per_step", "ms_per_256spp_frame", "fraction_of_fast_mode", "rmse_vs_oracle"))
    if isinstance(config.get("verify_build"), dict):
        c["verify_build"] = _pick(config["verify_build"], ("frame", "spp", "pixels_bit_identical_to_oracle", "rmse_vs_oracle", "product_vs_verify_rmse", "product_vs_verify_compare_rms", "error"))
    line["config"] = c
    roofline = full.get("roofline") or {}
    r = _pick(roofline, ("bound", "peak", "unit", "avg_launch_ms", "launches", "traffic", "achieved", "frac", "algorithmic_bytes_per_launch", "achieved_model", "frac_model",
                         "traffic_over_useful", "write_amplification", "limiter"))
    r["kernel"] = text(roofline.get("kernel"), 160)
    r["frac_basis"] = text(roofline.get("frac_basis"), 120)
    if isinstance(roofline.get("traffic_source"), dict):
        source = roofline["traffic_source"]
        r["traffic_source"] = text(source.get("measured") or source.get("file"), 120)
    if isinstance(roofline.get("measured_copy_bandwidth"), dict):
        r["copy_GBps"] = roofline["measured_copy_bandwidth"].get("GB/s")
    line["roofline"] = r
    valu = full.get("roofline_valu")
    if isinstance(valu, dict):
        v = _pick(valu, ("bound", "achieved", "peak", "unit", "frac", "lanes_per_instruction", "valu_busy"))
        if "error" in valu:
            v["error"] = text(valu["error"], 160)
        line["roofline_valu"] = v
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        b = _pick(cpu, ("value", "unit", "cores", "kind", "pinned", "jobs", "min", "max"))
        b["sample"] = text(cpu.get("sample"), 200)
        if isinstance(cpu.get("c2"), dict):
            b["c2"] = _pick(cpu["c2"], ("value", "unit", "cores", "kind"))
            b["c2"]["sample"] = text(cpu["c2"].get("sample"), 160)
        line["cpu_baseline"] = b
    if isinstance(full.get("ranks"), dict):
        line["ranks"] = _pick(full["ranks"], ("ms_per_step", "gather_ms", "gather_transport", "passes", "steps_per_pass", "paths_per_gpu_per_step", "world_seen", "distinct_devices", "backend", "rccl_version", "proof_error"))
        if isinstance(full["ranks"].get("devices"), list):
            line["ranks"]["devices"] = [d.get("pci") or d.get("uuid") for d in full["ranks"]["devices"] if isinstance(d, dict)]
        if isinstance(full["ranks"].get("tile_split_probe"), dict):
            line["ranks"]["tile_split_probe_identical"] = full["ranks"]["tile_split_probe"].get("identical")
    if isinstance(full.get("kernel_ms_per_step_alone") or full.get("kernel_ms_per_step"), dict):
        line["kernel_ms_per_step"] = full.get("kernel_ms_per_step_alone") or full.get("kernel_ms_per_step")
    if details_path:
        line["details"] = str(details_path)
    line = _clean(line)
    # never over the limit: drop what the contract does not ask for, least important first
    for path in (("kernel_ms_per_step",), ("details",), ("ranks",), ("config", "verify_build"), ("config", "rmse_full_size"), ("roofline", "traffic_source"),
                 ("cpu_baseline", "c2", "sample"), ("cpu_baseline", "sample"), ("roofline", "frac_basis"), ("config", "library_sha16"), ("config", "workload_textured")):
        if len(json.dumps(line, allow_nan=False)) < LINE_LIMIT:
            break
        node = line
        for k in path[:-1]:
            node = node.get(k, {}) if isinstance(node, dict) else {}
        if isinstance(node, dict):
            node.pop(path[-1], None)
    if len(json.dumps(line, allow_nan=False)) >= LINE_LIMIT:
        line["config"]["workload"] = text(line["config"].get("workload"), 120)
    return line


def emit(full: dict, result_fd: int, details_arg=None) -> dict:
    """Writes the complete record to the details file (and, as one line, to stderr) and the compact line -- the only thing on stdout -- to `result_fd`."""
    details_path = Path(details_arg) if details_arg else ROOT / "bench_details.json"
    cleaned = _clean(full, digits=9)
    written = None
    for candidate in (details_path, Path(os.environ.get("TMPDIR", "/tmp")) / "bench_details.json"):
        try:
            candidate.parent.mkdir(parents=True, exist_ok=True)
            candidate.write_text(json.dumps(cleaned, allow_nan=False, indent=1) + "\n")
            written = candidate
            break
        except OSError:
            continue
    sys.stderr.write("bench.py details: " + json.dumps(cleaned, allow_nan=False) + "\n")
    sys.stderr.flush()
    shown = None
    if written is not None:
        try:
            shown = written.resolve().relative_to(ROOT)
        except ValueError:
            shown = written
    line = compact_line(full, shown)
    payload = json.dumps(line, allow_nan=False)
    if len(payload) >= LINE_LIMIT:      # cannot happen with the fixed text lengths of compact_line; a result is never lost to it: the contract keys alone
        line = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in line}
        line["config"] = {"workload": str(full.get("config", {}).get("workload", ""))[:200]}
        line["note"] = "the compact line exceeded its limit and was cut to the contract keys; the details file holds everything"
        payload = json.dumps(line, allow_nan=False)
    os.write(result_fd, (payload + "\n").encode())
    return line


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL and device-tensor sharing across processes need it on this pool
    if args.gpus == 1:      # the CPU baseline's threads stay where they start (cpu_baseline_smallpt); read by the OpenMP runtime when it is first loaded
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    # test hooks of tests/test_bench_spawn_cpu.py: a rank that dies at start-up, a rank that never gets anywhere
    if os.environ.get("RANK") is not None and os.environ.get("HIPR_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK"):
        sys.exit(3)
    if os.environ.get("RANK") is not None and os.environ.get("HIPR_BENCH_TEST_HANG_RANK") in (os.environ.get("RANK"), "all"):
        time.sleep(3600)
    if args.pmc_child:      # a counter pass of measure_traffic_live: the timed region only
        args.no_cpu_baseline = args.no_other_workloads = args.no_rmse = args.no_plugin = args.no_scaling_proxy = True
        args.pmc_traffic = "off"
    # roofline.traffic, measured by child processes under rocprofv3 BEFORE this process initialises the GPU (importing torch does not)
    live_traffic = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    args.under_profiler = under_profiler
    if args.gpus == 1 and args.pmc_traffic == "auto" and not args.scene_file and not under_profiler:     # no profiler inside a profiler
        live_traffic = measure_traffic_live(sys.argv[1:])
        if not live_traffic[0]:
            sys.stderr.write(f"bench.py: live counter passes unavailable ({live_traffic[1]}); falling back to profiles/pmc_traffic.json\n")

    # stdout carries ONE line, the result: libraries that chat on stdout (gloo's rendezvous message, the renderer's device announcement) go to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from bifrost3d_amd import distributed
    from bifrost3d_amd.renderer import Context

    rank, world, local_rank = distributed.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        device_index = 0 if args.share_device else local_rank
        torch.cuda.set_device(device_index)
        # a collective that does not complete ends the rank after five minutes (the default is ten; the driver's own limit is far above both), which ends the job
        import datetime
        limit = datetime.timedelta(seconds=300)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index), timeout=limit)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
    else:
        device_index = 0
        torch.cuda.set_device(0)
    device = torch.device("cuda", device_index)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    scene_name = "file:" + args.scene_file if args.scene_file else args.scene
    scene, scene_text, default_bounces = make_scene(scene_name, args)
    bounces = args.bounces if args.bounces is not None else default_bounces
    ctx = Context(device_index)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    result = measure(ctx, scene, scene_name, bounces, args, rank, world, device, args.steps, args.warmup, sync)

    alone = None
    if world == 1 and result["wavefronts"] > 1 and not args.pmc_child:
        # The timed region ran co-running wavefronts: its per-launch HIP-event durations are those of kernels sharing the device. The rooflines price a KERNEL:
        # two more steps with one wavefront -- every kernel alone on the device, HIP events on its launch stream -- right after the timed region, the same
        # shape the counter passes and the committed rocprofv3 --stats profile run in.
        import copy
        alone_args = copy.copy(args)
        alone_args.wavefronts, alone_args.alone_leg = 1, True
        alone = measure(ctx, scene, scene_name, bounces, alone_args, 0, 1, device, 2, 1, sync)
        ctx.set_wavefront_count(args.wavefronts)
    proof = None
    if world > 1:
        try:
            proof = ranks_proof(ctx, scene, bounces, args, rank, world, device)
        except Exception as e:      # the timed line stands without it (a failure here is the same on every rank: library versions, properties torch does not expose)
            proof = {"proof_error": str(e)[:300]} if rank == 0 else None
    if rank == 0:
        main_figures = summarise(result, scene_name, scene_text, bounces, args, world, args.steps, None if alone else live_traffic)
        if alone:
            alone_figures = summarise(alone, scene_name, scene_text, bounces, args, 1, 2, live_traffic)
            timed = main_figures["roofline_by_kernel"]
            main_figures["roofline"], main_figures["roofline_by_kernel"] = alone_figures["roofline"], alone_figures["roofline_by_kernel"]
            dominant = max(alone_figures["roofline_by_kernel"], key=lambda n: alone_figures["roofline_by_kernel"][n]["total_ms"])
            main_figures["roofline"]["duration_source"] = ("2 steps with ONE wavefront right after the timed region: the kernel alone on the device (HIP events on its launch stream), as in the "
                                                           "counter passes and the committed rocprofv3 --stats profile. The timed region runs two co-running half-frame wavefronts; its own "
                                                           "per-launch figures are under timed_region")
            main_figures["roofline"]["timed_region"] = {k: timed[dominant].get(k) for k in ("avg_launch_ms", "launches", "total_ms", "achieved_model", "frac_model")} if dominant in timed else None
            main_figures["roofline"]["alone_ms_per_step"] = alone_figures["ms_per_step"]
            main_figures["kernel_ms_per_step_alone"] = alone_figures["kernel_ms_per_step"]
        out = {
            "metric": METRIC, "value": main_figures["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_figures["ms_per_step"], "higher_is_better": True, "scaling": "weak" if args.weak and world > 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": main_figures["workload"], "frame": [args.width, args.height], "spp_per_step": main_figures["spp_per_step"], "spp_total": main_figures["spp_total"],
                       "parallelism": f"tiles8x8-round-robin-x{world}" if world > 1 else "single-gpu", "wavefronts": result["wavefronts"],
                       "ms_per_256spp_frame": main_figures["ms_per_256spp_frame"], "rays_per_step": main_figures["rays_per_step"],
                       "closest_rays": main_figures["closest_rays"], "shadow_rays": main_figures["shadow_rays"], "pixel_samples": main_figures["pixel_samples"],
                       "frame_finite_and_lit": main_figures["frame_finite_and_lit"], "library_sha16": library_sha16(), "workload_version": WORKLOAD_VERSION,
                       "rmse_note": "no OptiX image exists or can be produced here (DESIGN.md); rmse_vs_oracle compares with the pinned CPU oracle at equal spp and seed"},
            "roofline": main_figures["roofline"], "roofline_by_kernel": main_figures["roofline_by_kernel"], "kernel_ms_per_step": main_figures["kernel_ms_per_step"],
        }
        if "kernel_ms_per_step_alone" in main_figures:
            out["kernel_ms_per_step_alone"] = main_figures["kernel_ms_per_step_alone"]
            out["kernel_ms_per_step_note"] = "kernel_ms_per_step sums HIP-event durations of launches that overlap the other wavefront's (it exceeds ms_per_step); kernel_ms_per_step_alone: one wavefront"
        if result.get("wide8"):
            out["config"]["backface_culling"] = ("hipr_set_backface_culling 1 (the default): the 8-wide traversal steps over closest hits on the back of one-sided surfaces instead of "
                                                  "handing them to the hit program to be refused and retraced; rays counted are the BVH queries actually made")
        if "retrace_mode" in result:
            out["retrace_mode"] = result["retrace_mode"]
        if world > 1:
            out["ranks"] = {"ms_per_step": [e / args.steps * 1e3 for e in result["rank_elapsed"]], "gather_ms": result["gather_ms"],
                            "gather_transport": ("gloo through the host (the first gather failed)" if getattr(args, "gather_group", None) is not None else ("gloo through the host" if args.dist_backend == "gloo" else "RCCL (torch.distributed nccl)")),
                            "passes": result["passes"], "steps_per_pass": result["steps_per_pass"], "accumulations_per_pass_per_rank": result["S"],
                            "paths_per_gpu_per_step": int(main_figures["pixel_samples"] / args.steps / world),
                            "note": "ms_per_step per rank = that rank's own clock over the timed region (the line's ms_per_step is the maximum); gather_ms = rank 0's time in the "
                                    "final gather of the half4 tiles + the scatter kernel, inside the timed region"}
            out["ranks"].update(proof or {})
        if world == 1 and not args.pmc_child:
            # the VALU roof beside the HBM one, and the bytes the dominant kernel could not avoid
            dominant = max(main_figures["roofline_by_kernel"], key=lambda n: main_figures["roofline_by_kernel"][n]["total_ms"])
            entry = main_figures["roofline_by_kernel"][dominant]
            try:
                rates = ctx.valu_issue_rates()
                valu = (live_traffic[1] or {}).get("valu_counters_per_launch") if live_traffic and live_traffic[0] else None
                out["roofline_valu"] = valu_roofline(dominant, entry["kernel"], entry["avg_launch_ms"] * 1e-3, valu, rates, torch.cuda.get_device_properties(device).multi_processor_count)
            except Exception as e:      # the line stands without it
                out["roofline_valu"] = {"bound": "valu", "error": str(e)}
            useful = useful_traffic(dominant, (alone or result)["counters"], entry["launches"], main_figures["spp_per_step"])
            if useful:
                out["roofline"]["traffic_useful"] = useful
                if out["roofline"].get("traffic"):
                    out["roofline"]["traffic_over_useful"] = out["roofline"]["traffic"] / useful["bytes"]
                    counters_detail = (main_figures["roofline"].get("traffic_source") or {}).get("counters", {}).get(dominant)
                    if counters_detail:
                        out["roofline"]["write_amplification"] = counters_detail["WRITE_SIZE_KiB_per_launch"] * 1024.0 / useful["writes"]
            out["roofline"]["limiter"] = ("valu" if out["roofline_valu"].get("valu_busy", 0) >= 0.6 else "see roofline_valu") if "error" not in out["roofline_valu"] else None
            if not args.no_scaling_proxy and not args.scene_file and not under_profiler:
                out["scaling_proxy"] = scaling_proxy(ctx, scene, bounces, args, device, main_figures["ms_per_step"])
        if world == 1:
            copy_gbs = measured_copy_bandwidth(device)
            out["roofline"]["measured_copy_bandwidth"] = {"GB/s": copy_gbs, "what": "1 GiB device-to-device copy, bytes read + written, best of five (torch)",
                                                          "frac_of_copy": out["roofline"]["achieved"] / copy_gbs if out["roofline"].get("achieved") else None}
            if not args.no_rmse:
                out["config"]["rmse_vs_oracle"] = rmse_against_oracle(ctx, scene, bounces, converged_name=scene_name if not args.scene_file else None)
                try:
                    out["config"]["verify_build"] = verify_build_leg(ctx, scene, bounces)
                except Exception as e:      # the line stands without it
                    out["config"]["verify_build"] = {"error": str(e)[:200]}
            if not args.pmc_child and not under_profiler and not getattr(args, "no_exact_mode", False):
                # The SAME renderer in its exact arithmetic mode (hipr_set_arithmetic: IEEE division / sqrt, no contraction, specified sin / cos / pow), same workload,
                # same step shape, right after the line's own run: the rate of the mode whose frames equal the CPU restatement bit for bit (north_star: RMSE < 1e-3 AND
                # the throughput from one renderer). Its rmse is config.verify_build's: every pixel identical at 160 x 90 x 256 spp, RMSE 0.
                import copy
                exact_args = copy.copy(args)
                exact_args.skip_retrace = True
                ctx.set_arithmetic("exact")
                try:
                    r = measure(ctx, scene, scene_name, bounces, exact_args, 0, 1, device, 4, 1, sync)
                    figures = summarise(r, scene_name, scene_text, bounces, args, 1, 4)
                finally:
                    ctx.set_arithmetic("fast")
                vb = out["config"].get("verify_build") if isinstance(out["config"].get("verify_build"), dict) else {}
                out["config"]["exact_mode"] = {"value": figures["value"], "unit": "Mrays/s", "ms_per_step": figures["ms_per_step"], "ms_per_256spp_frame": figures["ms_per_256spp_frame"],
                                               "steps": 4, "kernel_ms_per_step": figures["kernel_ms_per_step"], "fraction_of_fast_mode": figures["value"] / main_figures["value"],
                                               "rmse_vs_oracle": {"frame": vb.get("frame"), "spp": vb.get("spp"), "rmse_rgb": vb.get("rmse_vs_oracle"),
                                                                  "pixels_bit_identical": vb.get("pixels_bit_identical_to_oracle")},
                                               "what": "hipr_set_arithmetic(HIPR_ARITHMETIC_EXACT) on the same context and workload: the shade stage in IEEE arithmetic with specified "
                                                       "transcendentals; frames equal the CPU oracle's bit for bit (rmse_vs_oracle measured in this run, config.verify_build)"}
            if not args.no_other_workloads and not args.scene_file:
                others = {}
                for other in ("cornell_diffuse", "material"):
                    if other == scene_name:
                        continue
                    other_scene, other_text, other_bounces = make_scene(other, args)
                    r = measure(ctx, other_scene, other, other_bounces, args, 0, 1, device, 4, 1, sync)
                    figures = summarise(r, other, other_text, other_bounces, args, 1, 4)
                    if not args.no_rmse:
                        figures["rmse_vs_oracle"] = rmse_against_oracle(ctx, other_scene, other_bounces, spps=(8, 256) if other == "cornell_diffuse" else (8,))
                    figures.pop("roofline_by_kernel")
                    others[other] = figures
                out["other_workloads"] = others
            if scene_name == "atrium" and not args.no_textured and not args.scene_file and not under_profiler:
                # What a real Sponza brings and the plain stand-in does not: a texture on every material and cut-out cloth, i.e. the FULL
                # kernels (coverage lookups for shadow rays, texture samplers in shade). Same frame, same steps' shape; reported in the compact line.
                import copy
                textured_args = copy.copy(args)
                textured_args.skip_retrace = True
                t_scene, t_text, t_bounces = make_scene("atrium_textured", args)
                r = measure(ctx, t_scene, "atrium_textured", t_bounces, textured_args, 0, 1, device, 4, 1, sync)
                figures = summarise(r, "atrium_textured", t_text, t_bounces, args, 1, 4)
                figures.pop("roofline_by_kernel")
                out.setdefault("other_workloads", {})["atrium_textured"] = figures
                out["config"]["workload_textured"] = {"value": figures["value"], "unit": "Mrays/s", "ms_per_step": figures["ms_per_step"], "ms_per_256spp_frame": figures["ms_per_256spp_frame"],
                                                      "what": "same atrium, every material textured + cut-out cloth (30 % of triangles not statically opaque)"}
            if scene_name == "atrium" and not args.no_plugin:
                out["plugin_renderer"] = plugin_renderer_figures(ctx, args, main_figures)
                out["denoiser_stage"] = denoiser_figures(args, device)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_smallpt(args.cpu_baseline_seconds)
                # apps/SmallPT/smallpt.h does not build outside MSVC: the restatement is checked by properties only (tests/test_smallpt_cpu.py), no reference pin
                out["cpu_baseline"]["pinned"] = False
                out["cpu_baseline"]["c2"] = cpu_baseline_c2(ctx, args.cpu_baseline_seconds)
        sys.stdout.flush()
        emit(out, result_fd, args.details)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
