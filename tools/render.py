#!/usr/bin/env python3
"""Headless render to a PNG: scene -> path tracer -> camera effects -> file. What `SimpleViewer --scene X` shows, without a window.

    python tools/render.py --scene material --spp 256 --size 1280x720 --out material.png
    python tools/render.py --scene-file model.glb --spp 64 --effects linear --out model.png

Scenes: cornell, cornell_diffuse, atrium, material, material_coat, glass, or a model file (.gltf / .glb / .obj with PNG, JPEG or TGA textures)
set up the way the viewer sets up its command-line scene. `--effects preset` applies the camera's default post-process
(histogram exposure, vignette, filmic tonemapping, film grain; eye adaptation off so that a single frame is fully adapted),
`linear` only converts to sRGB. `--denoise` runs the denoising backend's data flow before the effects: an albedo feature pass of the
same accumulations (HIPR_ENTRY_DENOISER_ALBEDO into the second running mean) and the filter of hipr_denoiser_c.h. Python here is
plumbing around the C-ABIs (hiprenderer_c.h, hipr_camera_effects_c.h, hipr_denoiser_c.h).
"""
import argparse
import struct
import sys
import time
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bifrost3d_amd import camera_effects, capi, denoiser  # noqa: E402
from bifrost3d_amd.host import Scene  # noqa: E402
from bifrost3d_amd.renderer import Context  # noqa: E402


def write_png(path: str, rgba8: np.ndarray):
    """8 bit RGBA, rows top-down. Stdlib only (zlib), filter type 0."""
    height, width, _ = rgba8.shape
    raw = b"".join(b"\x00" + rgba8[y].tobytes() for y in range(height))

    def chunk(kind: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scene", default="cornell", choices=["cornell", "cornell_diffuse", "atrium", "material", "material_coat", "glass"])
    p.add_argument("--scene-file", default=None)
    p.add_argument("--size", default="1280x720")
    p.add_argument("--spp", type=int, default=64)
    p.add_argument("--spp-per-pass", type=int, default=8)
    p.add_argument("--bounces", type=int, default=-1, help="max_bounce_count; -1 keeps the scene's own (4, or 32 for the viewer's test scenes)")
    p.add_argument("--effects", choices=["preset", "linear"], default="preset")
    p.add_argument("--denoise", action="store_true", help="filter the frame with the denoising backend's filter (albedo feature pass + edge-avoiding a-trous wavelets)")
    p.add_argument("--arithmetic", choices=["fast", "exact"], default="fast", help="hipr_set_arithmetic: exact = IEEE shading arithmetic with specified sin / cos / pow, frames equal to the CPU oracle's bit for bit")
    p.add_argument("--out", default="render.png")
    args = p.parse_args()
    width, height = (int(v) for v in args.size.lower().split("x"))

    import torch
    if args.scene_file:
        scene = Scene("file:" + args.scene_file)
    elif args.scene == "cornell_diffuse":
        scene = Scene("cornell", diffuse_only=True)
    elif args.scene == "material_coat":
        scene = Scene("material", coat=True)
    elif args.scene == "atrium":
        scene = Scene("atrium", param0=260000, param1=1)
    else:
        scene = Scene(args.scene)

    ctx = Context(0, arithmetic=args.arithmetic)
    ctx.upload_scene(scene)
    batch = max(1, min(args.spp_per_pass, args.spp))
    ctx.set_frame(width, height, samples_per_pass=batch)
    frame = torch.zeros((height, width, 4), dtype=torch.float16, device="cuda:0")
    started = time.time()
    done = 0
    while done < args.spp:
        n = min(batch, args.spp - done)
        if n != batch:      # samples_per_pass is a property of the frame: the remainder cannot be traced at another batch size
            raise SystemExit(f"--spp {args.spp} must be a multiple of --spp-per-pass {batch}")
        ctx.render_pass(scene.camera(width, height, accumulations=done, max_bounce_count=args.bounces), frame.data_ptr(), width, synchronize=True)
        done += n
    seconds = time.time() - started
    counters = ctx.counters()

    if args.denoise:
        albedo = torch.zeros_like(frame)
        ctx.set_entry_point(capi.ENTRY_DENOISER_ALBEDO)
        ctx.use_scratch_accumulation(2)
        for first in range(0, args.spp, batch):
            ctx.render_pass(scene.camera(width, height, accumulations=first, max_bounce_count=args.bounces), albedo.data_ptr(), width, synchronize=True)
        ctx.use_scratch_accumulation(0)
        ctx.set_entry_point(capi.ENTRY_PATH_TRACING)
        dn = denoiser.Denoiser(0)
        dn.process(frame, albedo, frame, width, height)      # in place: the output kernel reads the filtered plane, not the noisy frame
        dn.close()

    fx = camera_effects.CameraEffects(0)
    settings = camera_effects.Settings.preset() if args.effects == "preset" else camera_effects.Settings.linear()
    settings.eye_adaptation_enabled = 0
    # The renderer's frame has row 0 at the bottom (the adaptor flips it when presenting): flip here, then post-process.
    flipped = torch.flip(frame, dims=[0]).contiguous()
    target = fx.process(settings, 1 / 60.0, flipped, target_format=camera_effects.TARGET_RGBA8_SRGB)
    fx.synchronize()
    write_png(args.out, target.cpu().numpy())
    rays = counters["closest_rays"] + counters["shadow_rays"]
    print(f"{args.out}: {width}x{height}, {args.spp} spp in {seconds:.2f} s ({rays / seconds / 1e6:.0f} Mrays/s incl. host loop), exposure {fx.linear_exposure:.3f}")


if __name__ == "__main__":
    main()
