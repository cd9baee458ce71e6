"""The SPECIFIED transcendentals of the exact arithmetic mode (csrc/spec_math.h; oracle/vecmath.h spec:: restates them) -- no GPU needed.

sin, cos and pow of the exact mode are fixed sequences of correctly rounded binary64 operations, so two implementations agree in every bit by construction. Here:
  * the device's code compiled for the host (tests/native/libdevice_shade_host.so: the same header hipcc compiles for gfx950) against the oracle's restatement, bit for
    bit, on the ranges the renderer uses and far outside them, including every special case the specification names;
  * both against glibc's f64 functions rounded once: the specified functions are CORRECTLY ROUNDED on these samples (the truncated Taylor sums are good to ~1e-16
    relative; an argument would have to sit within ~1e-14 of a rounding boundary to come out one ulp off) -- so the exact mode is also the accurate one;
  * the call sites' own arguments: 2 pi u for u in [0, 1) (the sampled azimuths), pow(x, 0.25), pow(x, 0.1), pow(x, 2.4) on [0, 1].
The GPU counterpart (the exact shade unit's kernel against the oracle on the device) is tests/test_gpu_verify_build.py::test_specified_transcendentals_on_the_device."""
import ctypes as C

import numpy as np
import pytest

import device_host_bindings
from oracle_bindings import get_oracle

_fp = C.POINTER(C.c_float)
SIN, COS, POW = 0, 1, 2


def _run(fn, function, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.zeros_like(x)
    fn(function, x.size, x.ctypes.data_as(_fp), y.ctypes.data_as(_fp), out.ctypes.data_as(_fp))
    return out


@pytest.fixture(scope="module")
def both():
    oracle = get_oracle(False).lib
    device = device_host_bindings.library()
    oracle.oracle_spec_math.argtypes = device.dsh_spec_math.argtypes = [C.c_int, C.c_int, _fp, _fp, _fp]
    return (lambda f, x, y=None: _run(oracle.oracle_spec_math, f, x, y)), (lambda f, x, y=None: _run(device.dsh_spec_math, f, x, y))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ulp_distance(got, reference64):
    """|got - reference| in units of the last place of the correctly rounded f32 result."""
    rounded = reference64.astype(np.float32)
    return np.abs(got.astype(np.float64) - reference64) / np.abs(np.spacing(rounded)).astype(np.float64), rounded


@pytest.mark.parametrize("low,high", [(0.0, 2 * np.pi), (-50.0, 50.0), (-1.0e5, 1.0e5)])
def test_sin_cos_are_one_function_on_both_sides_and_correctly_rounded(both, low, high):
    oracle, device = both
    x = np.random.default_rng(int(high) + 7).uniform(low, high, 1_000_000).astype(np.float32)
    for function, reference in ((SIN, np.sin), (COS, np.cos)):
        a, b = oracle(function, x), device(function, x)
        assert np.array_equal(bits(a), bits(b))
        distance, rounded = ulp_distance(a, reference(x.astype(np.float64)))
        assert distance.max() <= 0.5 + 1e-6
        assert np.array_equal(bits(a), bits(rounded))


def test_the_sampled_azimuths(both):
    """sincos(2 pi u) as the samplers call it (device_shading.h: 2.0f * HIPR_PI * u.y in f32), every u = k / 2^20 and the last floats below 1."""
    oracle, device = both
    u = np.concatenate([np.arange(1 << 20, dtype=np.float32) / np.float32(1 << 20), np.float32(1) - np.float32(2.0) ** -np.arange(1, 25, dtype=np.float32)])
    x = (np.float32(2.0) * np.float32(np.pi)) * u
    for function, reference in ((SIN, np.sin), (COS, np.cos)):
        a, b = oracle(function, x), device(function, x)
        assert np.array_equal(bits(a), bits(b))
        assert np.array_equal(bits(a), bits(reference(x.astype(np.float64)).astype(np.float32)))
    s, c = oracle(SIN, x).astype(np.float64), oracle(COS, x).astype(np.float64)
    assert np.abs(s * s + c * c - 1.0).max() < 2e-7


@pytest.mark.parametrize("exponent", [0.25, 0.1, 2.4, 1.0, -1.5, 7.0])
def test_pow_at_the_renderers_exponents(both, exponent):
    """modulate_roughness_under_coat (x^0.25), the Oren-Nayar fit (roughness^0.1), the sRGB decode (x^2.4)."""
    oracle, device = both
    x = np.random.default_rng(3).uniform(0.0, 1.0, 1_000_000).astype(np.float32)
    x[:4] = [0.0, 1.0, np.float32(1e-38), np.float32(1e-45)]
    y = np.full_like(x, exponent)
    a, b = oracle(POW, x, y), device(POW, x, y)
    assert np.array_equal(bits(a), bits(b))
    with np.errstate(all="ignore"):
        reference = np.power(x.astype(np.float64), np.float64(np.float32(exponent))).astype(np.float32)
    assert np.array_equal(bits(a), bits(reference))


def test_pow_over_the_whole_float_range(both):
    oracle, device = both
    rng = np.random.default_rng(11)
    x = np.exp(rng.uniform(-87.0, 88.0, 1_000_000)).astype(np.float32)
    y = rng.uniform(-4.0, 4.0, x.size).astype(np.float32)
    a, b = oracle(POW, x, y), device(POW, x, y)
    assert np.array_equal(bits(a), bits(b))
    with np.errstate(all="ignore"):
        reference = np.power(x.astype(np.float64), y.astype(np.float64))
        rounded = reference.astype(np.float32)      # overflows to inf, underflows through the denormals to 0: one rounding, like the specification's last step
    assert np.array_equal(bits(a), bits(rounded))


def test_special_cases_follow_the_specification(both):
    oracle, device = both
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    cases = [  # x, y, expected (None: NaN)
        (0.0, 2.0, 0.0), (0.0, 0.0, 1.0), (0.0, -1.0, inf), (-0.0, 0.5, 0.0), (-1.0, 2.0, None), (-1.0, 0.5, None), (nan, 1.0, None), (1.0, nan, None),
        (inf, 2.0, inf), (inf, 0.0, 1.0), (inf, -2.0, 0.0), (1.0, 123.0, 1.0), (2.0, 127.0, np.float32(2.0) ** 127), (2.0, 128.0, inf), (2.0, -149.0, np.float32(1e-45)),
        (2.0, -151.0, 0.0), (np.float32(1e-45), 0.5, np.float32(np.sqrt(np.float64(np.float32(1e-45))))), (10.0, 38.0, np.float32(1e38)), (10.0, 39.0, inf),
    ]
    x = np.array([c[0] for c in cases], np.float32); y = np.array([c[1] for c in cases], np.float32)
    a, b = oracle(POW, x, y), device(POW, x, y)
    for (cx, cy, expected), got_a, got_b in zip(cases, a, b):
        if expected is None:
            assert np.isnan(got_a) and np.isnan(got_b), (cx, cy, got_a, got_b)
        else:
            assert got_a == np.float32(expected) and bits(got_a) == bits(got_b), (cx, cy, got_a, got_b, expected)
    # sin / cos outside the specified domain (never a sampled azimuth): NaN on both sides, never a wrong number
    far = np.array([1.0e6, -3.0e7, inf, -inf, nan], np.float32)
    for function in (SIN, COS):
        assert np.isnan(oracle(function, far)).all() and np.isnan(device(function, far)).all()
    near = np.array([0.0, -0.0, np.float32(1e-45), np.float32(-1e-30), 999999.0], np.float32)
    assert np.array_equal(bits(oracle(SIN, near)), bits(device(SIN, near))) and np.array_equal(bits(oracle(COS, near)), bits(device(COS, near)))
    assert oracle(SIN, near)[1] == 0.0 and oracle(COS, near)[0] == 1.0      # (the reduction turns -0 into +0: sin(-0) = +0 here, on both sides)
