#!/usr/bin/env python3
"""Where do the verification build and the oracle (f64 transcendentals) part? One accumulation at a time, paths cut after 0, 1, ... bounces: the number of pixels
whose single-sample radiance differs at all, and the first few of them. usage: tools/verify_bisect.py <scene> [accumulation] [size]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from verify_probe import make
from bifrost3d_amd import capi
from bifrost3d_amd.renderer import Context
from oracle_bindings import get_oracle

name = sys.argv[1]
accumulation = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "160x90").split("x"))
scene, bounces = make(name)
oracle = get_oracle(True)
oracle.lib.oracle_set_f64_transcendentals(1)
verify = Context(0, arithmetic="exact")
verify.upload_scene(scene)
for cut in range(0, min(bounces, 6) + 1):
    verify.set_frame(w, h, 0, 1, 1)
    verify.render_pass(scene.camera(w, h, accumulations=accumulation, max_bounce_count=cut), synchronize=True)
    gpu = verify.read_accumulation()[..., :3]
    accum = np.zeros((h, w, 4), np.float64)
    cpu, _, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, accumulations=accumulation, max_bounce_count=cut), w, h, 1, use_bvh=verify.oracle_search())
    cpu = cpu[..., :3]
    differ = (gpu != cpu).any(axis=-1)
    rel = np.abs(gpu - cpu).max(axis=-1) / (np.abs(cpu).max(axis=-1) + 1e-12)
    big = (rel > 1e-4) & differ
    print(f"{name} acc {accumulation} cut {cut}: {int(differ.sum())} of {w * h} pixels differ ({int(big.sum())} by more than 1e-4 relative)", flush=True)
    ys, xs = np.where(differ)
    for y, x in list(zip(ys, xs))[:4]:
        print("     pixel", x, y, "device", gpu[y, x], "oracle", cpu[y, x], "rel", rel[y, x])
verify.close()
