"""The restatements against the REAL reference code, where that code builds here: oracle/_ref/libbifrost_ref.so is compiled from
the reference's own Bifrost core sources (oracle/Makefile `_ref`). Covers what the golden vectors elsewhere only pin at points:
the rho / alpha tables and their lookups, the camera matrices and rays, the tonemapping operators, the bloom taps, octahedral
normals and the importance sampled environment light. Skipped where the library was not built (no /root/reference, no prebuilt)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import pytest

import reference_bindings as ref
from bifrost3d_amd import capi

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libbifrost_ref.so is not built (needs /root/reference)")


@pytest.fixture(scope="module")
def oracle():
    from oracle_bindings import get_oracle
    return get_oracle(False)


# ---- shading tables -----------------------------------------------------------------------------------------------------------------

def test_shipped_tables_are_the_reference_arrays():
    """data/HIPRenderer/shading_tables.bin (tools/extract_tables.py read the numeric initialiser lists) against the arrays as the
    reference's compiler sees them, bit for bit."""
    base, full, light, dense, alphas = capi.load_tables()
    for ours, which in ((base, 0), (full, 1), (light, 2), (dense, 3), (alphas, 4)):
        theirs = ref.table(which)
        assert ours.shape == theirs.shape
        assert np.array_equal(ours.view(np.uint32), theirs.view(np.uint32)), which


def test_rho_lookups_match_the_reference_functions(oracle):
    rng = np.random.default_rng(11)
    lib, out = ref.lib(), np.empty(2, np.float32)
    for cos_theta, roughness in np.vstack([rng.random((400, 2)), [[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5]]]).astype(np.float32):
        oracle.lib.oracle_specular_rho(float(cos_theta), float(roughness), ref.fptr(out))
        # SpecularRho: base = the table with the Fresnel term at specularity 0, full = without Fresnel (specularity 1)
        assert out[0] == lib.ref_sample_GGX_with_fresnel(float(cos_theta), float(roughness))
        assert out[1] == lib.ref_sample_GGX(float(cos_theta), float(roughness))


def test_dielectric_rho_lookups_match_the_reference_function(oracle):
    rng = np.random.default_rng(12)
    lib, ours, theirs, ranges = ref.lib(), np.empty(2, np.float32), np.empty(2, np.float32), np.empty(4, np.float32)
    lib.ref_dielectric_ior_ranges(ref.fptr(ranges))
    iors = np.concatenate([rng.uniform(ranges[0], ranges[1], 200), rng.uniform(ranges[2], ranges[3], 200), ranges, [1.0 / 1.5, 1.5, 0.2, 4.0]])
    for ior in iors.astype(np.float32):
        cos_theta, roughness = rng.random(2).astype(np.float32)
        oracle.lib.oracle_dielectric_rho(float(cos_theta), float(roughness), float(ior), ref.fptr(ours))
        lib.ref_sample_dielectric_GGX(float(cos_theta), float(roughness), float(ior), ref.fptr(theirs))
        assert np.array_equal(ours, theirs), (cos_theta, roughness, ior, ours, theirs)


def test_alpha_estimation_matches_the_reference_function(oracle):
    rng = np.random.default_rng(13)
    lib = ref.lib()
    oracle.lib.oracle_estimate_alpha.argtypes = [C.c_float] * 2; oracle.lib.oracle_estimate_alpha.restype = C.c_float
    for cos_theta, pdf in zip(rng.random(400).astype(np.float32), np.exp(rng.uniform(-4, 12, 400)).astype(np.float32)):
        assert oracle.lib.oracle_estimate_alpha(float(cos_theta), float(pdf)) == lib.ref_estimate_alpha(float(cos_theta), float(pdf)), (cos_theta, pdf)


def test_PDF_encoding_and_sRGB_transfer_match_the_reference():
    """encode_PDF as the alpha table was fitted with (EstimateGGXBoundedVNDFAlpha.cpp:88-100; the oracle and the kernels index the
    table with the same expression) and Color.h's sRGB transfer functions, which the RGBA8_SRGB camera-effects target and the
    8-bit sRGB texel decode follow."""
    lib = ref.lib()
    for pdf in np.exp(np.random.default_rng(6).uniform(-6, 14, 200)).astype(np.float32):
        expected = (pdf / (np.float32(1.0) + pdf) - np.float32(0.13)) / np.float32(0.87)
        assert math.isclose(lib.ref_encode_PDF(float(pdf)), float(expected), rel_tol=1e-6, abs_tol=1e-7)
    for v in np.linspace(0.0, 1.0, 257):
        linear = v / 12.92 if v < 0.04045 else ((v + 0.055) / 1.055) ** 2.4
        assert math.isclose(lib.ref_sRGB_to_linear(float(v)), linear, rel_tol=2e-6, abs_tol=1e-7)
        encoded = v * 12.92 if v < 0.0031308 else 1.055 * v ** (1 / 2.4) - 0.055
        assert math.isclose(lib.ref_linear_to_sRGB(float(v)), encoded, rel_tol=2e-6, abs_tol=2e-7)


# ---- random numbers ------------------------------------------------------------------------------------------------------------------

def test_rng_header_functions_match_the_reference(oracle):
    lib = ref.lib()
    values = np.concatenate([np.arange(0, 70), np.random.default_rng(2).integers(0, 2**32, 500), [2**32 - 1, 2**31]]).astype(np.uint32)
    ours, theirs = np.empty(2, np.float32), np.empty(2, np.float32)
    for v in values:
        assert oracle.lib.oracle_reverse_bits(int(v)) == lib.ref_reverse_bits(int(v))
        assert oracle.lib.oracle_jenkins_hash(int(v)) == lib.ref_jenkins_hash(int(v))
        oracle.lib.oracle_sample02(int(v), ref.fptr(ours)); lib.ref_sample02(int(v), ref.fptr(theirs))
        assert np.array_equal(ours, theirs), v
    oracle.lib.oracle_power_heuristic.argtypes = [C.c_float] * 2; oracle.lib.oracle_power_heuristic.restype = C.c_float
    for a, b in np.exp(np.random.default_rng(3).uniform(-8, 8, (300, 2))).astype(np.float32):
        assert math.isclose(oracle.lib.oracle_power_heuristic(float(a), float(b)), lib.ref_power_heuristic(float(a), float(b)), rel_tol=2e-6)
    assert lib.ref_power_heuristic(float("inf"), 1.0) == oracle.lib.oracle_power_heuristic(float("inf"), 1.0) == 1.0


# ---- camera -------------------------------------------------------------------------------------------------------------------------

def rotation_cases():
    axis = np.array([1.0, 2.0, 3.0]) / math.sqrt(14.0)
    half = math.radians(30.0) / 2
    return [(0.0, 0.0, 0.0, 1.0), tuple(axis * math.sin(half)) + (math.cos(half),), (0.0, math.sin(0.6), 0.0, math.cos(0.6))]


@pytest.mark.parametrize("rotation", rotation_cases())
@pytest.mark.parametrize("position", [(0.0, 0.0, 0.0), (100.0, 10.0, -30.0)])
def test_camera_state_and_oracle_rays_match_the_reference_camera(oracle, position, rotation):
    """Cameras::create + set_transform + CameraUtils::ray_from_viewport_point of the reference (Camera.cpp:215-263) against the host's
    camera state and the oracle's ray generation at pixel centres (accumulation 0 has no jitter)."""
    from bifrost3d_amd.host import make_camera
    width, height, fov, near, far = 160, 120, math.pi / 4, 0.5, 200.0
    cam = make_camera(width, height, position, rotation, fov, near, far)
    _, inverse_projection = ref.perspective(near, far, fov, width / height)
    assert np.allclose(np.array(cam.inverse_projection_matrix).reshape(4, 4), inverse_projection, rtol=1e-6, atol=1e-7)

    rng = np.random.default_rng(5)
    pixels = np.vstack([[[0, 0], [width - 1, height - 1], [width // 2, height // 2]], rng.integers(0, [width, height], (61, 2))]).astype(np.uint32)
    points = (pixels.astype(np.float32) + np.float32(0.5)) / np.array([width, height], np.float32)
    theirs = ref.rays(position, rotation, near, far, fov, width / height, points)
    origins, directions = oracle.generate_rays(cam, width, height, 0, pixels)
    # The reference's CPU helper takes the direction as far point - near point in world space, which cancels digits when the camera
    # is far from the origin; the renderer's ray generation (and so the oracle and K1) rotates the view space direction instead.
    assert np.allclose(directions[:, :3], theirs[:, 3:], atol=2e-6 if max(map(abs, position)) == 0 else 2e-5)
    assert np.allclose(origins[:, :3], theirs[:, :3], rtol=1e-6, atol=2e-5)


def test_orthographic_projection_matches_the_reference():
    from bifrost3d_amd.host import make_camera
    cam = make_camera(64, 32, orthographic=(8.0, 4.0, 50.0))
    _, inverse_projection = ref.orthographic(8.0, 4.0, 50.0)
    assert np.allclose(np.array(cam.inverse_projection_matrix).reshape(4, 4), inverse_projection, rtol=1e-6, atol=1e-7)


# ---- camera effects -------------------------------------------------------------------------------------------------------------------

def levels():
    rng = np.random.default_rng(3)
    return np.vstack([np.exp2(rng.uniform(-10, 6, (2000, 3))), np.zeros((1, 3)), np.full((1, 3), 1e-4), np.full((1, 3), 1.0), [[4.0, 0.01, 0.3]]]).astype(np.float32)


@pytest.mark.parametrize("mode", ["filmic", "agx", "khronos"])
def test_oracle_tonemappers_match_the_reference_cpu_operators(mode):
    """Bifrost/Math/CameraEffects.h's operators. The renderer's shaders (which the oracle and the kernels follow where the two differ)
    compute the filmic curve in another arrangement: 3e-4; AgX and Khronos neutral are the same arithmetic."""
    import camera_effects_oracle
    from bifrost3d_amd import camera_effects
    settings = camera_effects.Settings.preset()
    rgb = levels()
    if mode == "filmic":
        settings.tonemapping_mode = camera_effects.TONEMAPPING_FILMIC
        s = [settings.tonemapping_black_clip, settings.tonemapping_toe, settings.tonemapping_slope, settings.tonemapping_shoulder, settings.tonemapping_white_clip]
        theirs, tolerance = ref.tonemap(1, s, rgb), 3e-4
    elif mode == "agx":
        settings.tonemapping_mode = camera_effects.TONEMAPPING_AGX
        theirs, tolerance = ref.tonemap(2, [0] * 5, rgb), 2e-5
    else:
        settings.tonemapping_mode = camera_effects.TONEMAPPING_KHRONOS_NEUTRAL
        theirs, tolerance = ref.tonemap(3, [0] * 5, rgb), 2e-6
    ours = camera_effects_oracle.tonemap(settings, rgb)
    finite = np.isfinite(theirs).all(axis=1)
    if mode == "agx":   # the CPU operator raises the sigmoid's small negative values near black to 2.2 -> NaN; the shader takes pow(abs(c), 2.2)
        assert np.all(rgb[~finite].max(axis=1) < 2e-3) and np.all(np.abs(ours[~finite]) < 1e-4) and finite.mean() > 0.7
    else:
        assert finite.all()
    assert np.allclose(ours[finite], theirs[finite], rtol=tolerance, atol=tolerance), float(np.abs(ours[finite] - theirs[finite]).max())


@pytest.mark.parametrize("std_dev, count", [(1.0, 2), (2.5, 5), (6.0, 11), (13.3, 20), (40.0, 48)])
def test_bloom_taps_match_the_reference(std_dev, count):
    import camera_effects_oracle
    offsets, weights = camera_effects_oracle.gaussian_taps(std_dev, count)
    ref_offsets, ref_weights = ref.gaussian_taps(std_dev, count)
    assert np.allclose(offsets, ref_offsets, rtol=2e-6) and np.allclose(weights, ref_weights, rtol=2e-6, atol=1e-9)


# ---- octahedral normals -----------------------------------------------------------------------------------------------------------------

def test_octahedral_normals_match_the_reference(oracle):
    from bifrost3d_amd.host import load_host_library
    rng = np.random.default_rng(9)
    normals = rng.normal(size=(4000, 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    normals = np.vstack([normals, np.eye(3, dtype=np.float32), -np.eye(3, dtype=np.float32)])
    theirs = ref.octahedral_encode(normals)
    ours = np.empty_like(theirs)
    load_host_library().hiprh_encode_octahedral(ref.fptr(np.ascontiguousarray(normals)), len(normals), ours.ctypes.data_as(C.POINTER(C.c_int16)))
    assert np.array_equal(ours, theirs)
    decoded = np.empty((len(theirs), 3), np.float32)
    oracle.lib.oracle_decode_octahedral(theirs.ctypes.data_as(C.POINTER(C.c_int16)), len(theirs), ref.fptr(decoded))
    assert np.allclose(decoded, ref.octahedral_decode(theirs), atol=1e-6)


# ---- software textures -------------------------------------------------------------------------------------------------------------------

PIXEL_FORMATS = {"Alpha8": (1, np.uint8, 1), "Intensity8": (2, np.uint8, 1), "RGB24": (3, np.uint8, 3), "RGBA32": (4, np.uint8, 4), "Intensity_Float": (5, np.float32, 1),
                 "RGB_Float": (6, np.float32, 3), "RGBA_Float": (7, np.float32, 4)}


@pytest.mark.parametrize("name", list(PIXEL_FORMATS))
def test_host_texture_lookup_matches_the_reference(name):
    """Assets::sample2D over Images::get_pixel (Texture.cpp:114-176, Image.cpp:221-286) against the host's, for every pixel format,
    sRGB decoding of the 8-bit colour formats, nearest / bilinear filtering and clamp / repeat wrapping, texcoords beyond [0, 1]."""
    from bifrost3d_amd.host import load_host_library
    host = load_host_library()
    host.hiprh_sample2D.argtypes = ref.lib().ref_sample2D.argtypes
    value, dtype, channels = PIXEL_FORMATS[name]
    rng = np.random.default_rng(value)
    width, height = 7, 5
    shape = (height, width) if channels == 1 else (height, width, channels)
    pixels = rng.integers(0, 256, shape).astype(np.uint8) if dtype == np.uint8 else (rng.random(shape) * 4).astype(np.float32)
    uv = np.vstack([rng.uniform(-1.5, 2.5, (300, 2)), [[0, 0], [1, 1], [0.5, 0.5], [1.0, 0.0], [0.999999, 0.999999]], (np.mgrid[0:7, 0:5].reshape(2, -1).T + 0.5) / [7, 5]])
    for is_sRGB in ([False, True] if name in ("RGB24", "RGBA32") else [False]):
        for magnification, minification in ((0, 0), (1, 1)):
            for wrap_U, wrap_V in ((0, 0), (1, 1), (1, 0)):
                theirs = ref.sample2D(ref.lib().ref_sample2D, value, is_sRGB, pixels, magnification, minification, wrap_U, wrap_V, uv)
                ours = ref.sample2D(host.hiprh_sample2D, value, is_sRGB, pixels, magnification, minification, wrap_U, wrap_V, uv)
                assert np.allclose(ours, theirs, rtol=1e-6, atol=1e-6), (name, is_sRGB, magnification, wrap_U, wrap_V, float(np.abs(ours - theirs).max()))


# ---- environment light ------------------------------------------------------------------------------------------------------------------

def sky(width, height, seed):
    rng = np.random.default_rng(seed)
    rgba = np.ones((height, width, 4), np.float32)
    rgba[..., :3] = rng.random((height, width, 3)) * 0.5
    for _ in range(3):                                 # a few suns, some next to black texels
        x, y = rng.integers(0, width), rng.integers(1, height - 1)
        rgba[y, x, :3] = rng.uniform(50, 400)
        rgba[y, (x + 1) % width, :3] = 0.0
    return rgba


@pytest.mark.parametrize("width, height", [(64, 32), (32, 16), (96, 160)])
def test_host_environment_light_matches_the_reference_class(width, height):
    """Assets::InfiniteAreaLight of the reference (InfiniteAreaLight.cpp, Distribution2D.h, Texture.cpp sample2D) against the host's:
    the PDF image (images lower than 128 rows are upsampled), importance samples, PDF(direction) and the per pixel PDF upload."""
    from bifrost3d_amd.host import load_host_library
    host = load_host_library()
    host.hiprh_infinite_area_light.argtypes = ref.lib().ref_infinite_area_light.argtypes
    rgba = sky(width, height, width + height)
    u = np.random.default_rng(21).random((500, 2)).astype(np.float32)
    theirs = ref.infinite_area_light(ref.lib().ref_infinite_area_light, rgba, u)
    ours = ref.infinite_area_light(host.hiprh_infinite_area_light, rgba, u)
    assert ours[2] == theirs[2]
    assert np.allclose(ours[3], theirs[3], rtol=2e-5, atol=1e-9)
    # A sample that falls the other side of a CDF entry by rounding picks the neighbouring texel: allow a handful.
    same = np.all(np.isclose(ours[0][:, 4:7], theirs[0][:, 4:7], atol=1e-4), axis=1)
    assert same.mean() >= 0.99, same.mean()
    assert np.allclose(ours[0][same, :4], theirs[0][same, :4], rtol=2e-4, atol=1e-6)
    assert np.allclose(ours[1][same], theirs[1][same], rtol=2e-4, atol=1e-7)
