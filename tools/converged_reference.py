#!/usr/bin/env python3
"""The converged leg of SURVEY.md 8(d)'s RMSE protocol, computed on host cores only (no GPU): the CPU oracle's image of the bench
workload at 16 x the spp of the equal-seed comparison, from accumulations DISJOINT from it.

  python tools/converged_reference.py [--scene atrium] [--width 160 --height 90] [--spp 256] [--factor 16] [--threads 6]

Writes profiles/converged/<scene>_<w>x<h>_acc<first>_<last>.npy (float32 RGB mean of accumulations [spp, spp + factor * spp)) and a
.json next to it (scene parameters, oracle search, seconds). bench.py's rmse leg loads it when scene, frame and spp match. The oracle is
deterministic, so the file can be produced in the build container and used on the GPU box.
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
    p.add_argument("--atrium-triangles", type=int, default=260000)
    p.add_argument("--width", type=int, default=160)
    p.add_argument("--height", type=int, default=90)
    p.add_argument("--spp", type=int, default=256)
    p.add_argument("--factor", type=int, default=16)
    p.add_argument("--bounces", type=int, default=4)
    p.add_argument("--threads", type=int, default=6)
    p.add_argument("--search", type=int, default=2, help="0 exhaustive, 1 BVH2, 2 wide BVH (what the device uses for the scene)")
    args = p.parse_args()
    from bifrost3d_amd.host import Scene
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)
    oracle.lib.oracle_set_threads(args.threads)
    if args.scene == "atrium":
        scene = Scene("atrium", param0=args.atrium_triangles, param1=1)
    elif args.scene == "cornell_diffuse":
        scene = Scene("cornell", diffuse_only=True)
    else:
        scene = Scene(args.scene)
    w, h = args.width, args.height
    first, last = args.spp, args.spp + args.factor * args.spp
    accum = np.zeros((h, w, 4), np.float64)
    t0 = time.time()
    accum, _, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, accumulations=0, max_bounce_count=args.bounces), w, h, first, use_bvh=args.search, accum=accum)
    head = accum[..., :3].copy()
    chunk = 64
    for a in range(first, last, chunk):
        n = min(chunk, last - a)
        accum, _, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, accumulations=a, max_bounce_count=args.bounces), w, h, n, use_bvh=args.search, accum=accum)
        print(f"accumulations [0, {a + n}) after {time.time() - t0:.0f} s", flush=True)
    # mean over [first, last) from the two running means (f64)
    tail = (accum[..., :3] * last - head * first) / (last - first)
    name = args.scene if args.scene != "atrium" or args.atrium_triangles == 260000 else f"atrium{int(scene.desc.triangle_count)}"      # the headline atrium keeps its plain name
    stem = ROOT / "profiles" / "converged" / f"{name}_{w}x{h}_acc{first}_{last}"
    np.save(str(stem) + ".npy", tail.astype(np.float32))
    json.dump({"scene": args.scene, "triangles": int(scene.desc.triangle_count), "frame": [w, h], "accumulations": [first, last], "bounces": args.bounces, "search": args.search,
               "quantized_tables": True, "seconds": time.time() - t0, "threads": args.threads, "mean_radiance": float(tail.mean())}, open(str(stem) + ".json", "w"), indent=1)
    print("wrote", stem)


if __name__ == "__main__":
    main()
