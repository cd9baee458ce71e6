"""BASELINE.json config 1: the reference's CPU SmallPT integrator (apps/SmallPT/smallpt.h:22-147), restated in oracle/smallpt.cpp
and used by bench.py as the `cpu_baseline`. No reference test pins its output (and smallpt.h does not build outside MSVC, so
oracle/_ref cannot hold it): checked here through what the algorithm guarantees -- a pure function of (pixel, accumulation), so
independent of threads and of how the accumulations are batched; a running mean; the scene's known layout."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from oracle_bindings import fptr, get_oracle


@pytest.fixture(scope="module")
def oracle():
    return get_oracle(False)


def accumulate(oracle, width, height, count, buf=None, start=0):
    buf = np.zeros((height, width, 3), np.float32) if buf is None else buf
    acc, rays = C.c_int(start), []
    for _ in range(count):
        rays.append(int(oracle.lib.oracle_smallpt_accumulate(width, height, fptr(buf), C.byref(acc))))
    return buf, rays, acc.value


def test_smallpt_is_deterministic_and_thread_independent(oracle):
    """The per-pixel LCG is seeded from jenkins_hash(subpixel index) ^ reverse_bits(accumulation) (smallpt.h:133-136): the image and
    the ray count cannot depend on the OpenMP schedule."""
    a, rays_a, _ = accumulate(oracle, 96, 64, 6)
    threads = oracle.lib.oracle_smallpt_threads()
    oracle.lib.oracle_set_threads(1)
    try:
        b, rays_b, _ = accumulate(oracle, 96, 64, 6)
    finally:
        oracle.lib.oracle_set_threads(threads)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and rays_a == rays_b


def test_smallpt_accumulates_a_running_mean(oracle):
    """backbuffer = lerp(backbuffer, radiance, 1 / accumulations) (smallpt.h:141-144): resuming from a saved buffer and counter
    continues the same sequence."""
    full, _, count = accumulate(oracle, 64, 64, 8)
    assert count == 8
    half, _, count = accumulate(oracle, 64, 64, 5)
    resumed, _, count = accumulate(oracle, 64, 64, 3, buf=half, start=count)
    assert count == 8 and np.array_equal(full, resumed)


def test_smallpt_renders_the_cornell_box_of_spheres(oracle):
    """Config 1 at a quarter of its size (128x128, 64 accumulations): every camera ray hits a sphere, so each accumulation traces at
    least one ray per pixel; the emitter (radiance 12) is visible at the top; the left wall is red, the right wall blue
    (smallpt.h:47-57; the image is stored bottom row first like the viewer's texture); noise falls as accumulations grow."""
    width = height = 128
    image, rays, _ = accumulate(oracle, width, height, 64)
    assert np.isfinite(image).all() and image.min() >= 0.0
    assert all(r >= width * height for r in rays) and 3.0 < np.mean(rays) / (width * height) < 12.0
    assert image[-1, 54:74].mean() > 6.0 and image[-1].max() <= 12.0       # a sliver of the light (radiance 12) along the top edge
    left, right = image[40:80, 4:12].mean(axis=(0, 1)), image[40:80, -12:-4].mean(axis=(0, 1))
    assert left[0] > 1.8 * left[2] and right[2] > 1.8 * right[0]
    assert 0.3 < image[40:80, 30:100].mean() < 1.2                         # the lit interior, linear radiance
    rough, _, _ = accumulate(oracle, width, height, 8)
    reference, _, _ = accumulate(oracle, width, height, 256)
    def error(candidate):
        return float(np.sqrt(np.mean((np.minimum(candidate, 2.0) - np.minimum(reference, 2.0)) ** 2)))
    assert error(image) < 0.6 * error(rough)


def test_bench_cpu_baseline_reports_the_port(oracle):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    line = bench.cpu_baseline_smallpt(0.5)
    assert line["kind"] == "port" and line["unit"] == "Mrays/s" and line["value"] > 0 and line["cores"] == oracle.lib.oracle_smallpt_threads()
    assert "256x256" in line["sample"]
