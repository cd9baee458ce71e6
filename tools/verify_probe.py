#!/usr/bin/env python3
"""The verification build against the oracle, stage by stage and image by image (round 5).

  python tools/verify_probe.py [--scenes cornell,atrium20k,...] [--spp 64] [--size 160x90] [--out profiles/r05_verify_probe.json]

Stage level: hipr_debug_shading / hipr_debug_light of libhiprenderer_verify.so against the oracle with f64 transcendentals, counted BIT for bit.
Image level, per scene at equal seed: verify build vs oracle (f64 transcendentals) -- RMSE, share of bit-identical pixels, pixels that differ at all; product vs verify build
on the device (= what the fast arithmetic does to the image); product vs oracle (the figure of the bench line)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


def compare_rms(a, b):
    d = np.abs(a - b)
    return float(np.sqrt(np.mean((0.2126 * d[..., 0] + 0.7152 * d[..., 1] + 0.0722 * d[..., 2]) ** 2)))


def make(name):
    from bifrost3d_amd.host import Scene
    if name == "cornell": return Scene("cornell"), 4
    if name == "cornell_diffuse": return Scene("cornell", diffuse_only=True), 4
    if name == "cornell_spot": return Scene("cornell", spot=True), 4
    if name == "material": return Scene("material"), 32
    if name == "material_coat": return Scene("material", coat=True), 32
    if name == "glass": return Scene("glass"), 32
    if name == "opacity": return Scene("opacity"), 32
    if name == "atrium20k": return Scene("atrium", param0=20000, param1=1), 4
    if name == "atrium": return Scene("atrium", param0=260000, param1=1), 4
    if name == "atrium_textured": return Scene("atrium", param0=260000, param1=1, textured=True), 4
    if name == "atrium1m": return Scene("atrium", param0=1000000, param1=1), 4
    raise SystemExit(f"unknown scene {name}")


def render(ctx, scene, w, h, spp, bounces):
    ctx.upload_scene(scene)
    batch = min(spp, 32)
    ctx.set_frame(w, h, 0, 1, batch)
    for a in range(0, spp, batch):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=bounces))
    ctx.synchronize()
    return ctx.read_accumulation()[..., :3].astype(np.float64), ctx.oracle_search()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scenes", default="cornell_diffuse,cornell,cornell_spot,opacity,material,glass,atrium20k,atrium")
    p.add_argument("--spp", type=int, default=64)
    p.add_argument("--size", default="160x90")
    p.add_argument("--out", default=None)
    p.add_argument("--no-stages", action="store_true")
    p.add_argument("--no-libm", action="store_true", help="skip the oracle's render with glibc's f32 transcendentals (its default mode): halves the host time of a large frame")
    args = p.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    from bifrost3d_amd import capi
    from bifrost3d_amd.renderer import Context
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)
    product, verify = Context(0), Context(0, arithmetic="exact")
    report = {"size": [w, h], "spp": args.spp, "stages": {}, "images": {}}

    if not args.no_stages:
        from test_oracle_goldens import shading_params, normalize
        goldens = json.loads((ROOT / "tests" / "golden" / "reference_goldens.json").read_text())
        rng = np.random.default_rng(5)
        oracle.lib.oracle_set_f64_transcendentals(1)
        for model, oracle_model, names in ((0, 4, ("gold", "plastic", "coated_plastic")), (1, 6, ("gold", "plastic")), (2, 5, ("frosted_glass",))):
            for name in names:
                if name not in goldens["materials"]:
                    continue
                for trial in range(4):
                    wo = normalize([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.02, 1.0)])
                    params = shading_params(goldens["materials"][name])
                    if trial >= 2:
                        params = np.array(params, np.float32); params[3] = rng.uniform(0.02, 1.0); params[9] = rng.uniform(0.1, 20.0)      # roughness, a PDF hint (path regularisation)
                    n = 20000
                    u = rng.uniform(0, 1, (n, 3)).astype(np.float32)
                    gpu, cpu = verify.debug_shading(model, params, wo, u, mode=0), oracle.bsdf_sample(oracle_model, params, wo, u)
                    same = (gpu.view(np.uint32) == cpu.view(np.uint32)) | (np.isnan(gpu) & np.isnan(cpu))
                    wi = rng.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
                    if model != 2: wi[:, 2] = np.abs(wi[:, 2])
                    gpu_e, cpu_e = np.ascontiguousarray(verify.debug_shading(model, params, wo, wi, mode=1)[:, :4]), oracle.bsdf_eval(oracle_model, params, wo, wi)
                    same_e = (gpu_e.view(np.uint32) == cpu_e.view(np.uint32)) | (np.isnan(gpu_e) & np.isnan(cpu_e))
                    key = f"model{model}:{name}:{trial}"
                    report["stages"][key] = {"sample_bit_identical": float(same.all(axis=1).mean()), "evaluate_bit_identical": float(same_e.all(axis=1).mean()),
                                             "sample_max_rel": float(np.nanmax(np.abs(gpu - cpu) / (np.abs(cpu) + 1e-6))), "evaluate_max_rel": float(np.nanmax(np.abs(gpu_e - cpu_e) / (np.abs(cpu_e) + 1e-6)))}
                    print("STAGE", key, report["stages"][key], flush=True)
                    if not same.all():
                        bad = np.where(~same.all(axis=1))[0][:3]
                        for b in bad: print("   sample mismatch u", u[b], "gpu", gpu[b], "cpu", cpu[b])
                    if not same_e.all():
                        bad = np.where(~same_e.all(axis=1))[0][:3]
                        for b in bad: print("   evaluate mismatch wi", wi[b], "gpu", gpu_e[b], "cpu", cpu_e[b])
        from test_oracle_lights import samples02, sphere_light, spot_light
        u1024 = samples02(oracle, 1024)
        for i, light in enumerate((sphere_light((0.3, 2.0, -0.4), 0.5, 7.0), sphere_light((0.0, 1.0, 0.0), 0.0, 3.0), spot_light((0.1, 3.0, 0.2), (0.0, -1.0, 0.0), 0.7, 5.0, 0.6),
                                   spot_light((0.0, 2.0, 0.0), (0.6, -0.8, 0.0), 0.0, 5.0, 0.8))):
            position = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
            gpu, cpu = verify.debug_light(light, position, u1024), oracle.light_sample(light, position, u1024)
            same = (gpu.view(np.uint32) == cpu.view(np.uint32)) | (np.isnan(gpu) & np.isnan(cpu))
            report["stages"][f"light{i}"] = {"bit_identical": float(same.all(axis=1).mean())}
            print("STAGE light", i, report["stages"][f"light{i}"], flush=True)
            if not same.all():
                for b in np.where(~same.all(axis=1))[0][:3]: print("   light mismatch u", u1024[b], "gpu", gpu[b], "cpu", cpu[b])

    for name in args.scenes.split(","):
        scene, bounces = make(name)
        t0 = time.time()
        img_product, search = render(product, scene, w, h, args.spp, bounces)
        img_verify, _ = render(verify, scene, w, h, args.spp, bounces)
        cam = scene.camera(w, h, max_bounce_count=bounces)
        oracle.lib.oracle_set_f64_transcendentals(1)
        img_exact, _, seconds_exact = oracle.render(scene.desc, scene.state, cam, w, h, args.spp, use_bvh=search)
        oracle.lib.oracle_set_f64_transcendentals(0)
        if args.no_libm:
            img_libm, seconds = img_exact, 0.0
        else:
            img_libm, _, seconds = oracle.render(scene.desc, scene.state, cam, w, h, args.spp, use_bvh=search)
        img_exact, img_libm = img_exact[..., :3].astype(np.float64), img_libm[..., :3].astype(np.float64)
        identical = (img_verify == img_exact).all(axis=-1)
        rel = np.abs(img_verify - img_exact) / (np.abs(img_exact) + 1e-3)
        entry = {"triangles": int(scene.desc.triangle_count), "bounces": bounces, "mean_radiance": float(img_exact.mean()),
                 "verify_vs_oracle_f64": {"rmse_rgb": rmse(img_verify, img_exact), "compare_rms": compare_rms(img_verify, img_exact), "pixels_bit_identical": float(identical.mean()),
                                          "pixels_differing": int((~identical).sum()), "pixels_beyond_1e-3_relative": int((rel.max(axis=-1) > 1e-3).sum())},
                 "product_vs_verify_on_device": {"rmse_rgb": rmse(img_product, img_verify), "compare_rms": compare_rms(img_product, img_verify),
                                                 "pixels_bit_identical": float((img_product == img_verify).all(axis=-1).mean())},
                 "product_vs_oracle_libm": {"rmse_rgb": rmse(img_product, img_libm), "compare_rms": compare_rms(img_product, img_libm)},
                 "oracle_f64_vs_oracle_libm": {"rmse_rgb": rmse(img_exact, img_libm), "pixels_bit_identical": float((img_exact == img_libm).all(axis=-1).mean())},
                 "oracle_seconds": [float(seconds_exact), float(seconds)], "seconds": time.time() - t0}
        if args.no_libm:      # the oracle rendered once (f64 transcendentals): no figure against its default mode
            entry.pop("product_vs_oracle_libm"); entry.pop("oracle_f64_vs_oracle_libm")
        report["images"][name] = entry
        print("IMAGE", name, json.dumps(entry), flush=True)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(report, indent=1) + "\n")
    product.close(); verify.close()


if __name__ == "__main__":
    main()
