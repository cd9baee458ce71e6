#!/usr/bin/env python3
"""SURVEY.md 8(d)'s RMSE protocol in full (GPU box; test infrastructure, not the product):

  1. device vs CPU oracle at EQUAL spp and seed (accumulations 0..spp-1), both definitions: (i) sqrt(mean over pixels and channels of
     (a - b)^2) and (ii) the reference's ImageOperations::Compare::rms (extensions/ImageOperations/ImageOperations/Compare.h:23-43);
  2. each of the two against a CONVERGED oracle image of 16 x the spp from DISJOINT accumulations [spp, 17 spp) -- profiles/converged/*.npy
     (tools/converged_reference.py; computed here when the file for this scene and frame is missing). Equal distances to the converged image say
     the equal-seed difference is noise of two estimators of the same integral; a bias would show as one of them sitting farther away;
  3. bias statistics of the equal-seed difference d = device - oracle: mean(d) per channel with its standard error over the pixels, and the share
     of sum(d^2) carried by the eight worst pixels;
  4. the eight worst pixels: per-SAMPLE radiance of both sides (a pass of ONE accumulation a into an empty frame leaves r / (a + 1)), the samples
     that differ, and for the worst sample of each pixel the bounce at which the two sides part (max_bounce_count 0, 1, ...).

usage: tools/rmse_protocol.py [--scene atrium] [--size 480x270] [--spp 256] [--out profiles/r03_rmse_protocol_480x270.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def rms_pair(a, b):
    diff = np.abs(a - b)
    luminance = 0.2126 * diff[..., 0] + 0.7152 * diff[..., 1] + 0.0722 * diff[..., 2]   # BF/Math/Color.h luminance()
    return {"rmse_rgb": float(np.sqrt(np.mean(diff ** 2))), "rmse_reference_compare_rms": float(np.sqrt(np.mean(luminance ** 2)))}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scene", default="atrium")
    p.add_argument("--triangles", type=int, default=260000)
    p.add_argument("--size", default="480x270")
    p.add_argument("--spp", type=int, default=256)
    p.add_argument("--factor", type=int, default=16)
    p.add_argument("--bounces", type=int, default=4)
    p.add_argument("--worst", type=int, default=8)
    p.add_argument("--decay-to", type=int, default=1024, help="largest sample count of the decay table (64, 256, 1024, ...); 0: none")
    p.add_argument("--out", default=None)
    args = p.parse_args()
    from bifrost3d_amd.host import Scene
    from bifrost3d_amd.renderer import Context
    from oracle_bindings import get_oracle
    w, h = (int(v) for v in args.size.split("x"))
    scene = Scene("atrium", param0=args.triangles, param1=1) if args.scene == "atrium" else (Scene("cornell", diffuse_only=True) if args.scene == "cornell_diffuse" else Scene(args.scene))
    oracle = get_oracle(True)
    ctx = Context(0)
    ctx.upload_scene(scene)
    ctx.set_wavefront_count(1)
    search = ctx.oracle_search()
    spp, bounces = args.spp, args.bounces
    cam = lambda a, b=bounces: scene.camera(w, h, accumulations=a, max_bounce_count=b)

    def device_image(first, count, max_bounces=bounces):
        batch = min(count, 32)
        ctx.set_frame(w, h, 0, 1, batch)
        for a in range(first, first + count, batch):
            ctx.render_pass(cam(a, max_bounces))
        ctx.synchronize()
        return ctx.read_accumulation()[..., :3].copy()

    def oracle_image(first, count, max_bounces=bounces, accum=None):
        image, _, seconds = oracle.render(scene.desc, scene.state, cam(first, max_bounces), w, h, count, use_bvh=search, accum=accum)
        return image, seconds

    out = {"scene": args.scene, "triangles": int(scene.desc.triangle_count), "frame": [w, h], "spp": spp, "bounces": bounces, "oracle_search": search,
           "host_threads": int(oracle.lib.oracle_max_threads())}
    gpu = device_image(0, spp)
    cpu_accum, seconds = oracle_image(0, spp)
    cpu = cpu_accum[..., :3].copy()
    out["oracle_seconds_equal_seed"] = seconds
    out["equal_seed"] = dict(rms_pair(gpu, cpu), mean_radiance=float(cpu.mean()))

    # ---- the converged leg (--factor 0: none -- the full-size frame, where the oracle's equal-seed image alone takes minutes)
    first, last = spp, spp + args.factor * spp
    converged = None
    if args.factor > 0:
        stem = ROOT / "profiles" / "converged" / f"{args.scene}_{w}x{h}_acc{first}_{last}"
        meta_ok = False
        if Path(str(stem) + ".npy").exists() and Path(str(stem) + ".json").exists():
            meta = json.loads(Path(str(stem) + ".json").read_text())
            meta_ok = meta.get("triangles") == int(scene.desc.triangle_count) and meta.get("bounces") == bounces      # any of the oracle's searches converges to the same image
        if meta_ok:
            converged = np.load(str(stem) + ".npy").astype(np.float64)
            out["converged_source"] = f"profiles/converged/{stem.name}.npy (tools/converged_reference.py, {meta['seconds']:.0f} s on {meta['threads']} host threads)"
        else:
            t0 = time.time()
            accum = cpu_accum.copy()
            for a in range(first, last, 256):
                accum, _ = oracle_image(a, min(256, last - a), accum=accum)
            converged = (accum[..., :3] * last - cpu * first) / (last - first)
            out["converged_source"] = f"computed here by the oracle in {time.time() - t0:.0f} s"
        # an independent device image of the same disjoint accumulations: the two converged images should agree far better than either 256 spp image does
        gpu_converged = (device_image(0, last) * last - gpu * first) / (last - first)
        out["converged"] = {"accumulations": [first, last], "device_vs_converged_oracle": rms_pair(gpu, converged), "oracle_vs_converged_oracle": rms_pair(cpu, converged),
                            "device_vs_converged_device": rms_pair(gpu, gpu_converged), "oracle_vs_converged_device": rms_pair(cpu, gpu_converged),
                            "converged_device_vs_converged_oracle": rms_pair(gpu_converged, converged)}

    # ---- bias statistics
    d = gpu - cpu
    n = d.shape[0] * d.shape[1]
    worst_flat = np.argsort((d ** 2).sum(axis=-1).ravel())[::-1][:args.worst]
    out["bias"] = {"mean_signed_difference_rgb": [float(v) for v in d.reshape(-1, 3).mean(axis=0)],
                   "standard_error_rgb": [float(v) for v in d.reshape(-1, 3).std(axis=0) / np.sqrt(n)],
                   "share_of_squared_error_in_worst_pixels": float((d ** 2).sum(axis=-1).ravel()[worst_flat].sum() / (d ** 2).sum()),
                   "rmse_rgb_without_worst_pixels": float(np.sqrt(((d ** 2).sum() - (d ** 2).sum(axis=-1).ravel()[worst_flat].sum()) / (3.0 * (n - len(worst_flat))))),
                   "pixels_beyond_1e-3_relative": int(((np.abs(d) / (np.abs(cpu) + 1e-3)).max(axis=-1) > 1e-3).sum()), "pixels": int(n)}

    # ---- how the equal-seed difference decays with the sample count, and how heavy its tails are. Zero-mean noise of finite variance falls as 1 / sqrt(spp) and
    # its per-pixel distribution tends to a Gaussian (kurtosis 3); rare large per-sample differences (a path that takes another discrete decision under
    # the shade kernel's approximate arithmetic: one sample of a pixel differs by O(1)) make it heavy-tailed -- a few pixels carry the sum of squares, the
    # RMSE falls more slowly than 1 / sqrt(spp) until every pixel has had its share of such samples, and the MEAN stays where it is.
    if args.decay_to > spp:
        ctx.set_frame(w, h, 0, 1, 32)           # start over: the running means are read at the checkpoints
        accum = None
        done = 0
        checkpoints = []
        c = 64
        while c <= args.decay_to:
            checkpoints.append(c)
            c *= 4
        decay = []
        for checkpoint in checkpoints:
            for a in range(done, checkpoint, 32):
                ctx.render_pass(cam(a))
            ctx.synchronize()
            g = ctx.read_accumulation()[..., :3].astype(np.float64)
            accum, _ = oracle_image(done, checkpoint - done, accum=accum)
            o = accum[..., :3].astype(np.float64)
            done = checkpoint
            dd = (g - o)
            per_pixel = (dd ** 2).sum(axis=-1).ravel()
            top = np.sort(per_pixel)[::-1]
            flat = dd.ravel()
            decay.append({"spp": checkpoint, **rms_pair(g, o), "kurtosis": float(np.mean(flat ** 4) / max(np.mean(flat ** 2) ** 2, 1e-300)),
                          "share_of_squared_error_in_the_worst_1_percent_of_pixels": float(top[:max(1, len(top) // 100)].sum() / max(per_pixel.sum(), 1e-300)),
                          "median_absolute_difference": float(np.median(np.abs(flat))), "mean_signed_difference": float(flat.mean()),
                          "standard_error_of_the_mean": float(flat.std() / np.sqrt(len(flat)))})
        for k in range(1, len(decay)):
            decay[k]["rmse_ratio_to_previous"] = decay[k - 1]["rmse_rgb"] / decay[k]["rmse_rgb"]      # 2.0 for 1 / sqrt(spp) at 4 x the samples
        out["decay"] = decay

    # ---- the worst pixels, sample by sample
    ys, xs = np.unravel_index(worst_flat, (h, w))
    per_sample_gpu = np.zeros((spp, len(xs), 3))
    per_sample_cpu = np.zeros((spp, len(xs), 3))
    ctx.set_frame(w, h, 0, 1, 1)
    for a in range(spp):
        ctx.set_frame(w, h, 0, 1, 1)     # empties the running mean: one pass of accumulation a leaves r / (a + 1)
        ctx.render_pass(cam(a), synchronize=True)
        per_sample_gpu[a] = ctx.read_accumulation()[ys, xs, :3] * (a + 1.0 if a else 1.0)
        image, _ = oracle_image(a, 1)
        per_sample_cpu[a] = image[ys, xs, :3] * (a + 1.0 if a else 1.0)
    pixels = []
    for k, (x, y) in enumerate(zip(xs, ys)):
        delta = per_sample_gpu[:, k] - per_sample_cpu[:, k]
        magnitude = np.abs(delta).max(axis=-1)
        differing = [int(a) for a in np.argsort(magnitude)[::-1][:4] if magnitude[a] > 1e-3 * (np.abs(per_sample_cpu[a, k]).max() + 1e-3)]
        entry = {"pixel": [int(x), int(y)], "device_mean": [float(v) for v in gpu[y, x]], "oracle_mean": [float(v) for v in cpu[y, x]], "converged_oracle": [float(v) for v in converged[y, x]] if converged is not None else None,
                 "samples_that_differ": int((magnitude > 1e-3 * (np.abs(per_sample_cpu[:, k]).max(axis=-1) + 1e-3)).sum()),
                 "largest": [{"accumulation": a, "device": [float(v) for v in per_sample_gpu[a, k]], "oracle": [float(v) for v in per_sample_cpu[a, k]]} for a in differing]}
        if differing:    # where along the path do the two sides part: radiance of the worst sample with the path cut after 0, 1, ... bounces
            a = differing[0]
            by_bounce = []
            for b in range(bounces + 1):
                ctx.set_frame(w, h, 0, 1, 1)
                ctx.render_pass(cam(a, b), synchronize=True)
                g = ctx.read_accumulation()[y, x, :3] * (a + 1.0 if a else 1.0)
                image, _ = oracle_image(a, 1, b)
                c = image[y, x, :3] * (a + 1.0 if a else 1.0)
                by_bounce.append({"max_bounce_count": b, "device": [float(v) for v in g], "oracle": [float(v) for v in c]})
            entry["worst_sample_by_max_bounce_count"] = by_bounce
        pixels.append(entry)
    out["worst_pixels"] = pixels
    ctx.close()
    # one line per top-level key: the per-sample tables of the worst pixels are data, not prose (thousands of lines per file bloat every diff)
    text = "{\n" + ",\n".join(f" {json.dumps(k)}: {json.dumps(v, separators=(', ', ': '))}" for k, v in out.items()) + "\n}"
    if args.out:
        Path(args.out).write_text(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
