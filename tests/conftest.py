"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "culling: the oracle's 8-wide search steps over the back of one-sided triangles, as by default (tests/test_wide8_cpu.py turns it off otherwise)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_bindings import get_oracle
    return get_oracle(False)


@pytest.fixture(scope="session")
def goldens():
    import json
    return json.loads((ROOT / "tests" / "golden" / "reference_goldens.json").read_text())


@pytest.fixture(scope="session")
def verify_ctx():
    """A context of the product library in its EXACT arithmetic mode (hipr_set_arithmetic: the shade unit built with correctly rounded division / square root, no
    contraction, sin / cos / pow as the specified f64 sequences of csrc/spec_math.h) -- the mode whose images equal the oracle's bit for bit
    (tests/test_gpu_verify_build.py)."""
    from bifrost3d_amd.renderer import Context
    c = Context(0, arithmetic="exact")
    yield c
    c.close()


def verification_build_equals_oracle(verify_ctx, oracle, scene, w, h, spp, bounces, name=""):
    """Leg A of an image test: the frame of the exact arithmetic mode against the oracle's with the specified transcendentals, pixel by pixel, bit for bit (the
    environment lookup's atan2 / asin come from two f64 libraries, which may round apart with probability ~2^-26 per call: at most a pixel or two of a frame with
    an environment map). Returns the exact image (h, w, 3), f64."""
    import numpy as np
    verify_ctx.upload_scene(scene)
    verify_ctx.set_frame(w, h, 0, 1, 1)
    for a in range(spp):
        verify_ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=bounces))
    verify_ctx.synchronize()
    ours = verify_ctx.read_accumulation()[..., :3]
    before = oracle.lib.oracle_set_f64_transcendentals(1)
    try:
        theirs, _, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=bounces), w, h, spp, use_bvh=verify_ctx.oracle_search())
    finally:
        oracle.lib.oracle_set_f64_transcendentals(before)
    identical = (ours == theirs[..., :3]).all(axis=-1)
    print(f"EXACT {name}: exact arithmetic mode vs oracle (specified transcendentals): {identical.mean():.6f} of the {w}x{h} pixels bit-identical at {spp} spp")
    assert identical.mean() >= 0.9999, (name, float(identical.mean()))
    return ours
