#!/usr/bin/env python3
"""Times the camera effects stages (include/hipr_camera_effects_c.h) on a synthetic frame and prices each against the HBM roofline.

    python tools/bench_camera_effects.py [--width 1920 --height 1080 --iterations 20 --bloom-threshold 4]

Algorithmic bytes per pixel: exposure reads the half4 frame once (8 B); each bloom pass reads 8 B and writes 8 B (perfect
reuse of the taps' neighbours); the tonemap pass reads frame + bloom (16 B) and writes the target (4 B for RGBA8).
Stage times come from HIP events on the effects' stream (hipr_camera_effects_set_instrumentation)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bifrost3d_amd import camera_effects  # noqa: E402
from bifrost3d_amd.camera_effects import Settings  # noqa: E402

PEAK_GBS = 8000.0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--iterations", type=int, default=20)
    p.add_argument("--bloom-threshold", type=float, default=4.0)
    p.add_argument("--exposure", choices=["histogram", "log_average", "fixed"], default="histogram")
    p.add_argument("--tonemapping", choices=["linear", "filmic", "agx", "khronos"], default="filmic")
    args = p.parse_args()

    rng = np.random.default_rng(1)
    y, x = np.mgrid[0:args.height, 0:args.width]
    pixels = np.ones((args.height, args.width, 4), dtype=np.float16)
    pixels[..., :3] = (np.exp2(4.0 * np.sin(x / 37.0) * np.cos(y / 23.0))[..., None] * rng.uniform(0.25, 1.0, (args.height, args.width, 3))).astype(np.float16)

    fx = camera_effects.CameraEffects(0)
    frame = fx.upload(pixels)
    s = Settings.preset()
    s.bloom_threshold = args.bloom_threshold
    s.exposure_mode = {"fixed": 0, "log_average": 1, "histogram": 2}[args.exposure]
    s.tonemapping_mode = {"linear": 0, "filmic": 1, "agx": 2, "khronos": 3}[args.tonemapping]
    target = fx.process(s, 1 / 60.0, frame, target_format=camera_effects.TARGET_RGBA8_SRGB)      # warm-up, allocates the intermediates
    fx.synchronize()
    fx.set_instrumentation(True)
    fx.reset_timers()
    for _ in range(args.iterations):
        fx.process(s, 1 / 60.0, frame, target_format=camera_effects.TARGET_RGBA8_SRGB, target=target)
    fx.synchronize()
    t = fx.times()
    pixel_count = args.width * args.height
    support = int(s.bloom_support * args.height)
    stages = {"exposure": (t.exposure_ms, t.exposure_launches, 8), "bloom_horizontal": (t.bloom_horizontal_ms, t.bloom_horizontal_launches, 16),
              "bloom_vertical": (t.bloom_vertical_ms, t.bloom_vertical_launches, 16), "tonemap": (t.tonemap_ms, t.tonemap_launches, 20)}
    out = {"frame": [args.width, args.height], "bloom_support_pixels": support, "exposure": args.exposure, "tonemapping": args.tonemapping, "stages": {}}
    total = 0.0
    for name, (ms, launches, bytes_per_pixel) in stages.items():
        if not launches:
            continue
        average = ms / launches
        total += average
        achieved = bytes_per_pixel * pixel_count / (average * 1e-3) / 1e9
        out["stages"][name] = {"avg_ms": round(average, 4), "algorithmic_bytes": bytes_per_pixel * pixel_count, "achieved_GBs": round(achieved, 1), "frac_of_8TBs": round(achieved / PEAK_GBS, 3)}
    out["total_ms_per_frame"] = round(total, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
