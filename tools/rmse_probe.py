#!/usr/bin/env python3
"""RMSE of the device image against the CPU oracle at equal spp and seed (GPU box; test infrastructure, not the product).

usage: tools/rmse_probe.py [--scene atrium|cornell|cornell_diffuse|material|opacity] [--spp 8,256] [--size 160x90] [--bounces N]
Set HIPR_LIBRARY to probe another build of libhiprenderer.so (A/B of compile flags). Prints one JSON line per spp with the two
RMSE definitions of SURVEY.md 8(d), the count of pixels that differ by more than 1e-3 relative and the largest differences.
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


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scene", default="atrium")
    p.add_argument("--spp", default="8,256")
    p.add_argument("--size", default="160x90")
    p.add_argument("--bounces", type=int, default=None)
    p.add_argument("--triangles", type=int, default=260000)
    args = p.parse_args()
    from bifrost3d_amd.host import Scene
    from bifrost3d_amd.renderer import Context
    from oracle_bindings import get_oracle
    w, h = (int(v) for v in args.size.split("x"))
    if args.scene == "atrium":
        scene = Scene("atrium", param0=args.triangles, param1=1)
    elif args.scene == "cornell_diffuse":
        scene = Scene("cornell", diffuse_only=True)
    else:
        scene = Scene(args.scene)
    bounces = args.bounces if args.bounces is not None else (32 if args.scene in ("material", "glass", "opacity") else 4)
    oracle = get_oracle(True)
    ctx = Context(0)
    ctx.upload_scene(scene)
    for spp in (int(v) for v in args.spp.split(",")):
        batch = min(spp, 32)
        ctx.set_frame(w, h, 0, 1, batch)
        t0 = time.perf_counter()
        for a in range(0, spp, batch):
            ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=bounces))
        ctx.synchronize()
        gpu_seconds = time.perf_counter() - t0
        gpu = ctx.read_accumulation()[..., :3]
        cpu, _, seconds = oracle.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=bounces), w, h, spp, use_bvh=ctx.oracle_search())
        cpu = cpu[..., :3]
        diff = np.abs(gpu - cpu)
        luminance = 0.2126 * diff[..., 0] + 0.7152 * diff[..., 1] + 0.0722 * diff[..., 2]
        rel = diff / (np.abs(cpu) + 1e-3)
        worst = np.sort(diff.max(axis=-1).ravel())[::-1][:8]
        print(json.dumps({"scene": args.scene, "frame": [w, h], "spp": spp, "bounces": bounces, "rmse_rgb": float(np.sqrt(np.mean(diff ** 2))),
                          "rmse_reference_compare_rms": float(np.sqrt(np.mean(luminance ** 2))), "mean_radiance": float(cpu.mean()),
                          "pixels_beyond_1e-3_relative": int((rel.max(axis=-1) > 1e-3).sum()), "pixels": w * h,
                          "largest_pixel_differences": [float(v) for v in worst], "oracle_seconds": float(seconds), "gpu_seconds": gpu_seconds}))
        sys.stdout.flush()
    ctx.close()


if __name__ == "__main__":
    main()
