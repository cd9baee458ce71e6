#!/usr/bin/env python3
"""Renders the same frames with the product, the verification build and the one-approximation variants of tools/fast_math_attribution.sh and reports each one's
equal-seed distance to the VERIFICATION build (which equals the oracle with f64 transcendentals bit for bit: tests/test_gpu_verify_build.py).
usage: tools/fast_math_attribution.py [--scenes atrium,material_coat,glass] [--spp 64] [--size 160x90] [--out file.json]"""
import argparse, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from verify_probe import make, render, rmse, compare_rms


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scenes", default="atrium,material_coat,glass")
    p.add_argument("--spp", type=int, default=64)
    p.add_argument("--size", default="160x90")
    p.add_argument("--out", default=None)
    args = p.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    from bifrost3d_amd import capi
    from bifrost3d_amd.renderer import Context
    csrc = ROOT / "bifrost3d_amd" / "csrc"
    tags = ["product", "verify"] + sorted(q.stem.replace("libhiprenderer_", "") for q in csrc.glob("libhiprenderer_[PV]_*.so"))
    report = {"size": [w, h], "spp": args.spp, "scenes": {}}
    for name in args.scenes.split(","):
        scene, bounces = make(name)
        images = {}
        for tag in tags:
            path = None if tag in ("product", "verify") else csrc / f"libhiprenderer_{tag}.so"      # "verify": the product library in its exact arithmetic mode
            # "verify" and the V_* variants (one approximation made fast in the EXACT shade unit) render in the exact arithmetic mode; "product" and P_* in the fast one
            ctx = Context(0, library=path, arithmetic="exact" if tag == "verify" or tag.startswith("V_") else "fast")
            images[tag], _ = render(ctx, scene, w, h, args.spp, bounces)
            ctx.close()
        ref = images["verify"]
        rows = {}
        for tag in tags:
            d = images[tag] - ref
            differ = (d != 0).any(axis=-1)
            rel = np.abs(d).max(axis=-1) / (np.abs(ref).max(axis=-1) + 1e-3)
            rows[tag] = {"rmse_rgb": rmse(images[tag], ref), "compare_rms": compare_rms(images[tag], ref), "pixels_differing": float(differ.mean()),
                         "pixels_beyond_1e-3_relative": float((rel > 1e-3).mean()), "pixels_beyond_1e-2_relative": float((rel > 1e-2).mean())}
            print(f"{name:16s} {tag:30s} rmse {rows[tag]['rmse_rgb']:.3e}  Compare::rms {rows[tag]['compare_rms']:.3e}  pixels differing {rows[tag]['pixels_differing']:.4f}  "
                  f"beyond 1e-3 {rows[tag]['pixels_beyond_1e-3_relative']:.4f}  beyond 1e-2 {rows[tag]['pixels_beyond_1e-2_relative']:.4f}", flush=True)
        report["scenes"][name] = rows
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(report, indent=1) + "\n")


if __name__ == "__main__":
    main()
