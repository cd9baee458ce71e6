#!/usr/bin/env python3
"""Writes tests/golden/images: small JPEG / Radiance HDR / TGA files and, in expected.npz, the pixels the REFERENCE's loader
(StbImageLoader::load, built from /root/reference into oracle/_ref) makes of them. Run in the build container; the files and
vectors are data, committed so that the codec test also runs where the reference tree is absent."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import test_image_codecs_cpu as codecs  # noqa: E402

out = ROOT / "tests" / "golden" / "images"
out.mkdir(parents=True, exist_ok=True)
expected = {}
for name, size, mode, arguments in codecs.JPEG_CASES:
    if name in ("wide_strip", "restart_rows_444", "q75_420", "four_components_transform_1", "cmyk_progressive"):
        continue      # keep the committed set small
    path = out / f"{name}.jpg"
    codecs.write_jpeg(path, size, mode, arguments, seed=len(name))
    expected[path.name] = codecs.reference_load(path)
for name, size, rle, magic in codecs.HDR_CASES[:3]:
    path = out / f"{name}.hdr"
    codecs.write_hdr(path, codecs.hdr_picture(size[0], size[1], len(name)), rle, magic)
    expected[path.name] = codecs.reference_load(path)
for name, size, arguments, channels in codecs.TGA_CASES:
    if name in ("colour24_bottom_up", "colour32_rle_top_down", "colour15_rle", "grey16_alpha", "mapped8_32_rle", "mapped16_16_skip", "grey8_rle_top_down"):
        path = out / f"{name}.tga"
        codecs.write_tga(path, size, seed=len(name), **arguments)
        expected[path.name] = codecs.reference_load(path)
np.savez_compressed(out / "expected.npz", **expected)
print({k: v.shape for k, v in expected.items()})
