#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HIP path tracer: Mrays/s and ms per accumulation pass at 1080p.

  python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches it under torch.distributed.run, one rank per GPU over RCCL).

A "step" is one pass of the hot path over one batch of synthetic input: 32 accumulations (samples per pixel) of the
1920x1080 frame traced together = 66 355 200 camera paths followed to completion (<= 5 surface interactions, next event
estimation with 3 RIS candidates, shadow rays), folded one by one into the f64 running mean and written as half4. The
reference traces one accumulation per launch; batching is a property of the wavefront design (HiprFrameDesc::samples_per_pass):
the image is bit-identical for any batch size (tests), while the per-bounce launches get 32x the rays and their long-ray tails
and launch gaps amortise (measured, DESIGN.md: 1 -> 8 samples per pass is +27 % on the Cornell box and +66 % on the atrium,
8 -> 32 another +3 % / +11 %; the queues of 32 take 13.6 GB of the 288 GB; `--spp-per-pass 1` reproduces the
one-accumulation-per-pass numbers).
Workload (BASELINE.json configs[1]): SimpleViewer Cornell box, every material forced to the Diffuse shading
model, max_bounce_count 4 (34 triangles: traced by the exhaustive-search kernels; `--scene atrium` is the 251 k-triangle
Sponza-class stand-in, traced by the fused persistent kernel over the compressed wide BVH). Inputs (scene, BVH, tables) are
resident in HBM before the timed region; the output
frame stays in HBM. N > 1: tiles of 8x8 pixels are dealt round-robin to the ranks and a step traces N
x 32 accumulations of the frame, so every GPU keeps the same 66 355 200 paths per step as N grows ("weak" scaling;
no data-path collective). The timed region ends with the RCCL gather of the half4 tiles to rank 0 plus the
scatter kernel that assembles the frame.

Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (the kernel with the largest total time in the
timed region; every kernel's figures are under `roofline_by_kernel`) and
`cpu_baseline` (SmallPT restatement on the host cores, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=16)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scene", default="cornell_diffuse", choices=["cornell_diffuse", "cornell", "atrium", "material", "material_coat"])
    p.add_argument("--scene-file", default=None, help="render a model file (.gltf / .glb / .obj, PNG textures) set up the way SimpleViewer sets up a scene from its command line; "
                   "not the headline workload: the line's config.workload names the file")
    p.add_argument("--atrium-triangles", type=int, default=260000)
    p.add_argument("--bounces", type=int, default=None, help="max_bounce_count; default 4, and 32 for the material scenes (the viewer's setting, apps/SimpleViewer/main.cpp:353)")
    p.add_argument("--spp-per-pass", type=int, default=32, help="accumulations traced together per step and GPU (HiprFrameDesc::samples_per_pass)")
    p.add_argument("--wavefronts", type=int, default=2, choices=[1, 2, 3, 4],
                   help="2 (default): each pass runs as two half-frame wavefronts on two streams, one shades while the other traces (bit-identical image, "
                        "+24 %% Cornell / +6 %% atrium); the per-kernel durations the roofline uses are then those of co-running kernels, and the line also "
                        "carries roofline.alone, the dominant kernel measured with one wavefront after the timed region. 1: one wavefront throughout")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --share-device runs the N > 1 code path with every rank on GPU 0 (a functional test of the tiling / gather / scatter logic on a 1-GPU box; the gather then goes through host memory)")
    p.add_argument("--share-device", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-alone-region", action="store_true", help="skip the one-wavefront region after the timed region (roofline.alone); the profiling runs use it so that "
                                                                  "every dispatch of a kernel in the profile belongs to the same launch shape")
    p.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    return p.parse_args()


def make_scene(args):
    from bifrost3d_amd.host import Scene
    if args.bounces is None:
        args.bounces = 32 if args.scene in ("material", "material_coat") and not args.scene_file else 4
    if args.scene_file:
        scene = Scene("file:" + args.scene_file)
        args.scene = "file:" + os.path.basename(args.scene_file)
        return scene, f"{os.path.basename(args.scene_file)} ({scene.desc.triangle_count} triangles) with the SimpleViewer defaults (camera from the scene bounds, one directional light)"
    if args.scene == "cornell_diffuse":
        return Scene("cornell", diffuse_only=True), "SimpleViewer Cornell box (34 triangles, 1 sphere light), all materials Diffuse"
    if args.scene == "cornell":
        return Scene("cornell"), "SimpleViewer Cornell box (34 triangles, 1 sphere light), reference materials"
    if args.scene in ("material", "material_coat"):
        return (Scene("material", coat=args.scene == "material_coat"),
                "SimpleViewer material scene (BASELINE config 3): 7 shader balls (procedural stand-in for Shaderball.gltf, 179 k triangles) blending dielectric to gold"
                + (", coat 1 / coat roughness 0.7" if args.scene == "material_coat" else "") + ", checkered textured floor, directional light, 32 bounces")
    return Scene("atrium", param0=args.atrium_triangles, param1=1), f"procedural atrium ({args.atrium_triangles} triangles target), DefaultShading"


def load_measured_traffic(args, wavefronts=None):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    separate runs of this same command; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of
    MI355X_MICROARCH.md "HBM"). Only used when the recorded command matches this run's workload; otherwise traffic is null."""
    path = ROOT / "profiles" / "pmc_traffic.json"
    if not path.exists():
        return {}
    try:
        table = json.loads(path.read_text())
    except ValueError:
        return {}
    key = f"{args.scene}:{args.width}x{args.height}:spp{args.spp_per_pass}:bounces{args.bounces}"
    if args.scene == "atrium":
        key += f":tris{args.atrium_triangles}"
    key += f":wf{wavefronts or args.wavefronts}"      # launches of a half-frame wavefront move half the bytes
    entry = table.get(key, {})
    return {k: v["traffic_bytes_per_launch"] for k, v in entry.get("kernels", {}).items()}


def rmse_against_oracle(ctx, scene, args, width=160, height=90, spp=8):
    """BASELINE.json's third figure, per-pixel RMSE at equal spp and seed. No OptiX image can exist here, so the comparand is the
    CPU oracle (same scene, camera, accumulations 0..spp-1, the search the GPU uses), on a frame small enough for the CPU:
    (i) sqrt(mean over pixels and channels of (a - b)^2), (ii) the reference's ImageOperations::Compare::rms
    (extensions/ImageOperations/ImageOperations/Compare.h:23-43): sqrt(mean(luminance(|a - b|)^2)). Runs after the timed region."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)   # unorm16 tables, as uploaded to the device
    ctx.set_wavefront_count(1)
    ctx.set_frame(width, height)
    for a in range(spp):
        ctx.render_pass(scene.camera(width, height, accumulations=a, max_bounce_count=args.bounces))
    ctx.synchronize()
    gpu = ctx.read_accumulation()[..., :3]
    cpu, _, seconds = oracle.render(scene.desc, scene.state, scene.camera(width, height, max_bounce_count=args.bounces), width, height, spp, use_bvh=ctx.oracle_search())
    cpu = cpu[..., :3]
    diff = np.abs(gpu - cpu)
    luminance = 0.2126 * diff[..., 0] + 0.7152 * diff[..., 1] + 0.0722 * diff[..., 2]   # BF/Math/Color.h luminance()
    return {"frame": [width, height], "spp": spp, "comparand": "CPU oracle (oracle/integrator.cpp), same seed", "rmse_rgb": float(np.sqrt(np.mean(diff ** 2))),
            "rmse_reference_compare_rms": float(np.sqrt(np.mean(luminance ** 2))), "mean_radiance": float(cpu.mean()), "oracle_seconds": float(seconds)}


def cpu_baseline(seconds: float):
    """SmallPT restatement (oracle/smallpt.cpp, follows apps/SmallPT/smallpt.h:22-147) on the host cores: 256x256,
    as many accumulations as fit the time budget (at most 64, BASELINE.json config 1)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import ctypes as C
    import numpy as np
    from oracle_bindings import get_oracle
    o = get_oracle(False)
    w = h = 256
    buf = np.zeros((h, w, 3), np.float32)
    acc = C.c_int(0)
    fp = C.POINTER(C.c_float)
    rays = 0
    t0 = time.perf_counter()
    while acc.value < 64:
        rays += o.lib.oracle_smallpt_accumulate(w, h, buf.ctypes.data_as(fp), C.byref(acc))
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": int(o.lib.oracle_smallpt_threads()), "kind": "port",
            "sample": f"SmallPT 9-sphere scene, 256x256, {acc.value} accumulations, {rays} radiance() rays in {dt:.1f} s, OpenMP dynamic,16"}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from bifrost3d_amd import distributed
    from bifrost3d_amd.renderer import Context

    rank, world, local_rank = distributed.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        device_index = 0 if args.share_device else local_rank
        torch.cuda.set_device(device_index)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        device_index = 0
        torch.cuda.set_device(0)
    device = torch.device("cuda", device_index)
    on_host = world > 1 and args.dist_backend == "gloo"   # gloo collectives take host tensors

    W, H = args.width, args.height
    S = args.spp_per_pass * world   # accumulations per step: per-GPU paths per step stay W*H*spp_per_pass for every N
    scene, scene_text = make_scene(args)
    ctx = Context(device_index)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    ctx.upload_scene(scene)
    ctx.set_wavefront_count(args.wavefronts)
    ctx.set_frame(W, H, tile_phase=rank, tile_stride=world, samples_per_pass=S)

    n_compact = distributed.padded_pixels_per_rank(W, H, world)
    frame = torch.zeros((H, W, 4), dtype=torch.float16, device=device) if rank == 0 else None
    compact = torch.zeros((n_compact, 4), dtype=torch.float16, device=device) if world > 1 else None

    def run_pass(accumulation):
        cam = scene.camera(W, H, accumulations=accumulation, max_bounce_count=args.bounces)
        if world == 1:
            ctx.render_pass(cam, frame.data_ptr(), W)
        else:
            ctx.render_pass(cam, compact.data_ptr(), 0)

    def finish_frame():
        if world == 1:
            return
        # The context renders on its own (non-blocking) stream, torch.distributed on torch's: order them explicitly.
        ctx.synchronize()
        gathered = distributed.gather_to_root(compact.cpu() if on_host else compact, world, rank)
        if rank == 0:
            if on_host:
                gathered = gathered.to(device)
            torch.cuda.current_stream(device).synchronize()
            ctx.scatter_tiles(gathered.data_ptr(), n_compact, world, W, H, frame.data_ptr(), W)
            ctx.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- instrumented passes: average BVH nodes / triangles per closest-hit ray (roofline numerator) ----------
    ctx.set_instrumentation(True)
    ctx.reset_counters()
    for a in (0, S):
        run_pass(a)
    ctx.synchronize()
    ic = ctx.counters()
    ctx.set_instrumentation(False)
    nodes_per_ray = ic["closest_nodes"] / max(1, ic["closest_rays"])
    tris_per_ray = ic["closest_triangles"] / max(1, ic["closest_rays"])
    shadow_nodes_per_ray = ic["shadow_nodes"] / max(1, ic["shadow_rays"])
    shadow_tris_per_ray = ic["shadow_triangles"] / max(1, ic["shadow_rays"])

    # ---- warmup -------------------------------------------------------------------------------------------------
    a = 2 * S
    for _ in range(args.warmup):
        run_pass(a)
        a += S
    finish_frame()
    barrier()
    ctx.reset_counters()
    ctx.reset_timers()

    # ---- timed region: exactly K steps ------------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_pass(a)
        a += S
    finish_frame()
    barrier()
    elapsed = time.perf_counter() - t0

    ctx.synchronize()
    counters = ctx.counters()
    times = ctx.kernel_times()
    stats = torch.tensor([elapsed, counters["closest_rays"], counters["shadow_rays"], counters["camera_rays"],
                          times["trace_closest"]["ms"], times["trace_closest"]["launches"]], dtype=torch.float64, device=device)
    if world > 1:
        if on_host:
            stats = stats.cpu()
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
    total_closest, total_shadow, total_camera = float(stats[1]), float(stats[2]), float(stats[3])
    total_rays = total_closest + total_shadow

    if rank == 0:
        ok = bool(torch.isfinite(frame.float()).all().item()) and float(frame[..., :3].float().mean()) > 0
        # Rooflines (rank 0's launches). Algorithmic HBM bytes per unit of work, DESIGN.md "Kernels" / SURVEY.md 8d:
        #   generate       80 B per path        (64 B path state + 16 B radiance slot written)
        #   trace_closest  48 B path state read + 16 B hit written per ray, + 64 B per BVH node visited + 48 B per triangle tested
        #   shade          80 B per queued ray (hit + path state) + 352 B per shaded hit (triangle 48, shading record 96, material 64,
        #                  3 RIS light candidates 144) + 64 B per continued path
        #                  + 48 B per shadow ray queued + 32 B radiance read-modify-write per queued ray
        #   trace_shadow   48 B record + 32 B radiance rmw per shadow ray, + 64 B per node + 48 B per triangle
        #   accumulate     16 B radiance per sample + 64 B f64 accumulation rmw + 8 B half4 per owned pixel
        from bifrost3d_amd import capi
        small = ctx.trace_variant() == capi.TRACE_EXHAUSTIVE
        fused = ctx.trace_is_fused()
        measured_traffic = load_measured_traffic(args)

        def rooflines_of(counters, times):
            n_closest, n_shadow, n_camera, n_hits = (counters[k] for k in ("closest_rays", "shadow_rays", "camera_rays", "shaded_hits"))
            # exhaustive-search kernels (<= 64 triangles): the triangle array is read once per 64-ray wave through the scalar cache
            tri_share = 1.0 / 64.0 if small else 1.0
            kernel_bytes = {
                "generate": 80.0 * n_camera,
                "trace_closest": n_closest * (48 + 16 + 64 * nodes_per_ray + 48 * tris_per_ray * tri_share),
                "shade": 80.0 * n_closest + 352.0 * n_hits + 64.0 * max(0, n_closest - n_camera) + 48.0 * n_shadow + 32.0 * n_closest,
                "trace_shadow": n_shadow * (48 + 32 + 64 * shadow_nodes_per_ray + 48 * shadow_tris_per_ray * tri_share),
                "accumulate": 16.0 * n_camera + (64.0 + 8.0) * n_camera / max(1, S),
            }
            kernel_names = {"generate": "k_generate", "trace_closest": "k_trace_closest_small" if small else "k_trace_closest", "shade": "k_shade",
                            "trace_shadow": "k_trace_shadow_small" if small else "k_trace_shadow",
                            "accumulate": "k_accumulate", "trace": "k_trace_persistent<TRACE_FUSED> (closest-hit rays of bounce k + shadow rays of bounce k-1)"}
            kernel_times = dict(times)
            if fused:   # one launch serves both ray kinds: bytes and time of the two are reported together
                kernel_bytes["trace"] = kernel_bytes.pop("trace_closest") + kernel_bytes.pop("trace_shadow")
                a, b = kernel_times.pop("trace_closest"), kernel_times.pop("trace_shadow")
                kernel_times["trace"] = {"ms": a["ms"] + b["ms"], "launches": a["launches"] + b["launches"]}
            rooflines = {}
            for name, nbytes in kernel_bytes.items():
                t = kernel_times.get(name)
                if not t or t["ms"] <= 0 or t["launches"] == 0:
                    continue
                gbs = nbytes / (t["ms"] * 1e-3) / 1e9
                rooflines[name] = {"bound": "hbm", "kernel": kernel_names[name], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                   "traffic": measured_traffic.get(name), "avg_launch_ms": t["ms"] / t["launches"], "launches": t["launches"],
                                   "algorithmic_bytes_per_launch": nbytes / t["launches"], "total_ms": t["ms"]}
            return rooflines, kernel_times

        rooflines, kernel_times = rooflines_of(counters, times)
        dominant = max(rooflines, key=lambda n: rooflines[n]["total_ms"])
        roofline = dict(rooflines[dominant])
        if args.wavefronts > 1 and not args.no_alone_region:
            # With two wavefronts a kernel shares the machine with the other wavefront's (that is the point: one shades while the other
            # traces), so the durations above -- and `achieved` with them -- are those of co-running kernels. The same kernel alone on
            # the machine: a short extra region with one wavefront, after and outside the timed region.
            ctx.set_wavefront_count(1)
            ctx.set_frame(W, H, tile_phase=rank, tile_stride=world, samples_per_pass=S)
            run_pass(0)
            ctx.synchronize()
            ctx.reset_counters()
            ctx.reset_timers()
            for k in range(4):
                run_pass((k + 1) * S)
            ctx.synchronize()
            alone, _ = rooflines_of(ctx.counters(), ctx.kernel_times())
            if dominant in alone:
                roofline["alone"] = {k: alone[dominant][k] for k in ("achieved", "frac", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch")}
                roofline["alone"]["traffic"] = load_measured_traffic(args, 1).get(dominant)
                roofline["alone"]["note"] = "the same kernel with one wavefront (nothing co-running), 4 steps after the timed region"
        if args.wavefronts > 1:
            roofline["co_running"] = "two half-frame wavefronts on two streams: this kernel's launches overlap the other wavefront's kernels"
        roofline.update({"nodes_per_ray": nodes_per_ray, "triangles_per_ray": tris_per_ray, "shadow_nodes_per_ray": shadow_nodes_per_ray,
                         "shadow_triangles_per_ray": shadow_tris_per_ray, "selection": "kernel with the largest total time in the timed region"})
        out = {
            "metric": "Mrays/sec + ms/frame at 1080p/256spp; per-pixel RMSE vs OptiXRenderer",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{scene_text}; {W}x{H}, {S} accumulation(s) per step ({W * H * args.spp_per_pass} paths per GPU per step), max_bounce_count {args.bounces}, "
                            f"next_event_sample_count 3, path regularisation PDF_scale 0.5; f64 accumulation + half4 output",
                "frame": [W, H], "spp_per_step": S,
                "parallelism": f"tiles8x8-round-robin-x{world}" if world > 1 else "single-gpu", "wavefronts": args.wavefronts,
                "ms_per_256spp_frame": elapsed / (args.steps * S) * 1e3 * 256,
                "rays_per_step": total_rays / args.steps,
                "closest_rays": total_closest, "shadow_rays": total_shadow, "pixel_samples": total_camera,
                "frame_finite_and_lit": ok,
                "rmse_note": "no OptiX image exists or can be produced here (DESIGN.md); rmse_vs_oracle compares with the CPU oracle at equal spp and seed",
            },
            "roofline": roofline,
            "roofline_by_kernel": rooflines,
            "kernel_ms_per_step": {name: v["ms"] / args.steps for name, v in kernel_times.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_seconds)
            if scene.desc.triangle_count <= 300000:
                out["config"]["rmse_vs_oracle"] = rmse_against_oracle(ctx, scene, args)
        print(json.dumps(out))
        sys.stdout.flush()

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
