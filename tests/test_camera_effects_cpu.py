"""Pins the camera effects oracle (oracle/camera_effects.cpp) with what the reference's own tests expect of these stages:
tests/DX11RendererTests/{ExposureHistogramTest,LogAverageLuminanceTest,BloomTest}.h and tests/BifrostTests/Math/UtilsTest.h:69-123.
The reference runs those against its DX11 shaders; here the same inputs and expectations run against the CPU restatement,
which the GPU tests (tests/test_gpu_camera_effects.py) then hold the HIP kernels to. Also checks the C-ABI surface."""
import ctypes as C
import math

import numpy as np
import pytest

import camera_effects_oracle as oracle
from bifrost3d_amd import camera_effects, capi
from bifrost3d_amd.camera_effects import Settings


def grey_image(values: np.ndarray) -> np.ndarray:
    """(rows, pitch) luminances -> (rows, pitch, 4) half pixels with alpha 1."""
    values = np.asarray(values, dtype=np.float32)
    pixels = np.ones(values.shape + (4,), dtype=np.float16)
    pixels[..., :3] = values[..., None].astype(np.float16)
    return pixels


def histogram_settings(min_log_luminance=-8.0, max_log_luminance=4.0, min_percentage=0.8, max_percentage=0.95) -> Settings:
    # create_camera_effects_constants, tests/DX11RendererTests/Utils.h:36-56: bias 0, eye adaptation off, bloom off, 1/60 s
    s = Settings.linear()
    s.exposure_mode = camera_effects.EXPOSURE_HISTOGRAM
    s.min_log_luminance, s.max_log_luminance = min_log_luminance, max_log_luminance
    s.min_histogram_percentage, s.max_histogram_percentage = min_percentage, max_percentage
    return s


def lerp(a, b, t):
    return a + t * (b - a)


# ---- ExposureHistogramTest.h ------------------------------------------------------------------------------------------------

def tiny_histogram_image(bin_count=64, lo=-8.0, hi=4.0):
    return grey_image(np.exp2(lerp(lo, hi, (np.arange(bin_count) + 0.49) / bin_count))[None, :])


def small_histogram_image(bin_count=64, lo=-8.0, hi=4.0):
    x = np.arange(bin_count)
    g1 = np.exp2(lerp(lo, hi, (x + 0.49) / bin_count))
    g2 = np.exp2(lerp(lo, hi, 1.0 - (x + 0.51) / bin_count))
    rows = [np.full(bin_count, 2.0 ** lo * 0.5), g1, g2, g1, g2, np.full(bin_count, 2.0 ** hi * 2.0)]
    return grey_image(np.stack(rows))


def test_histogram_tiny_image():
    bins = oracle.histogram(histogram_settings(), tiny_histogram_image())
    assert np.array_equal(bins, np.ones(64, dtype=np.uint32))


def test_histogram_small_image():
    bins = oracle.histogram(histogram_settings(), small_histogram_image())
    assert bins[0] == 4 + 64 and bins[63] == 4 + 64
    assert np.all(bins[1:63] == 4)


def average_luminance_without_outlier(histogram, min_percentage, max_percentage, min_log_luminance, max_log_luminance):
    """ExposureHistogramTest.h:31-60 compute_average_luminance_without_outlier, the reference's expectation for the exposure stage."""
    size = len(histogram)
    pixel_count = int(np.sum(histogram))
    min_pixel_count = np.float32(pixel_count * np.float32(min_percentage))
    max_pixel_count = np.float32(pixel_count * np.float32(max_percentage))
    weighted, counted = 0.0, 0.0
    for i in range(size):
        bucket_count = np.float32(histogram[i])
        sub = min(bucket_count, min_pixel_count)
        bucket_count -= sub
        min_pixel_count -= sub
        max_pixel_count -= sub
        bucket_count = min(bucket_count, max_pixel_count)
        max_pixel_count -= bucket_count
        luminance_at_bucket = 2.0 ** lerp(min_log_luminance, max_log_luminance, (i + 0.5) / size)
        weighted += luminance_at_bucket * float(bucket_count)
        counted += float(bucket_count)
    return weighted / max(0.0001, counted)


def shuffled_histogram():
    return np.random.default_rng(1234567799).permutation(64).astype(np.uint32)      # the reference shuffles 0..63 with a fixed seed


@pytest.mark.parametrize("bins", [np.ones(64, dtype=np.uint32), shuffled_histogram()], ids=["constant", "shuffled"])
def test_exposure_from_histogram(bins):
    s = histogram_settings()
    expected = 1.0 / average_luminance_without_outlier(bins, 0.8, 0.95, -8.0, 4.0)
    assert oracle.exposure_from_histogram(s, 1 / 60.0, bins) == pytest.approx(expected, rel=2e-6)


def test_eye_adaptation_moves_towards_the_target():
    s = histogram_settings()
    bins = shuffled_histogram()
    target = oracle.exposure_from_histogram(s, 1 / 60.0, bins)
    s.eye_adaptation_enabled, s.eye_adaptation_brightness, s.eye_adaptation_darkness = 1, 3.0, 1.0
    # CameraEffects/Utils.hlsl:42-47: current + (target - current) * (1 - 2^(-dt * speed)), brightening with the brightness speed, darkening with the darkness speed
    brighter = oracle.exposure_from_histogram(s, 0.5, bins, 0.0)
    assert brighter == pytest.approx(target * (1 - 2 ** (-0.5 * 3.0)), rel=1e-6)
    darker = oracle.exposure_from_histogram(s, 0.5, bins, 2 * target)
    assert darker == pytest.approx(2 * target - target * (1 - 2 ** (-0.5 * 1.0)), rel=1e-6)


# ---- LogAverageLuminanceTest.h ------------------------------------------------------------------------------------------------

def log_average_images():
    width = 128 * 8 + 17       # LogAverageLuminance::max_groups_dispatched * group_width + 17
    height = 21
    large = np.arange(width * height, dtype=np.float32)
    np.random.default_rng(1234567799).shuffle(large)
    return {"tiny": grey_image(np.arange(64, dtype=np.float32)[None, :]), "large": grey_image(large.reshape(height, width)), "black": grey_image(np.zeros((height, width)))}


def expected_log_average(pixels):
    rgb = pixels[..., :3].astype(np.float32)
    luminance = (rgb[..., 0] * np.float32(0.2126) + rgb[..., 1] * np.float32(0.7152) + rgb[..., 2] * np.float32(0.0722)).astype(np.float32)
    return 2.0 ** np.mean(np.log2(np.maximum(luminance, np.float32(0.0001)).astype(np.float64)))


def geometric_mean_linear_exposure(log_average_luminance):      # LogAverageLuminanceTest.h:30-33
    key_value = 1.03 - (2.0 / (2 + math.log10(log_average_luminance + 1)))
    return key_value / log_average_luminance


@pytest.mark.parametrize("name", ["tiny", "large", "black"])
def test_log_average_luminance(name):
    pixels = log_average_images()[name]
    log_average = oracle.log_average(pixels)
    assert log_average == pytest.approx(expected_log_average(pixels), rel=1e-5)
    s = histogram_settings(-24.0, 24.0)
    s.exposure_mode = camera_effects.EXPOSURE_LOG_AVERAGE
    linear_exposure = geometric_mean_linear_exposure(log_average)
    assert oracle.exposure_from_log_average(s, 1 / 60.0, pixels, linear_exposure) == pytest.approx(linear_exposure, rel=1e-5)


# ---- Math/UtilsTest.h:69-123 ----------------------------------------------------------------------------------------------------

def test_bilinear_gaussian_samples():
    values = np.array([0] * 7 + [1] * 7 + [0] * 7, dtype=np.float32)
    support, sample_count = 4, 2
    for std_dev in (0.1, 0.5, 1.0):
        offsets, weights = oracle.gaussian_taps(std_dev, sample_count)
        assert float(weights.sum()) == pytest.approx(0.5, rel=5e-7)        # one half of the bell curve
        for i in (5, 7, 10):
            k = np.arange(-support, support + 1)
            w = np.exp(-(k * k) / (2.0 * std_dev * std_dev))
            gaussian = float((values[i + k] * w).sum() / w.sum())
            sampled = 0.0
            for s in range(sample_count - 1, -1, -1):
                index = int(offsets[s])
                frac = float(offsets[s]) - index
                lower = lerp(values[i - index], values[i - index - 1], frac)
                upper = lerp(values[i + index], values[i + index + 1], frac)
                sampled += (lower + upper) * float(weights[s])
            assert sampled == pytest.approx(gaussian, rel=0.0025, abs=1e-12)


# ---- BloomTest.h (the Gaussian filter, the one CameraEffects::process uses) -------------------------------------------------------

def test_bloom_energy_conservation():
    pixels = np.ones((64, 64, 4), dtype=np.float16)
    filtered = oracle.bloom(0.0, 11, pixels)
    assert np.allclose(filtered.sum(axis=(0, 1), dtype=np.float64), pixels[..., :3].sum(axis=(0, 1), dtype=np.float64), rtol=0.002)


def threshold_image():
    y, x = np.mgrid[0:64, 0:64]
    pixels = np.ones((64, 64, 4), dtype=np.float16)
    pixels[..., 0], pixels[..., 1], pixels[..., 2] = (x + y * 64).astype(np.float16), x.astype(np.float16), (y * y).astype(np.float16)
    return pixels


def test_bloom_thresholding():
    pixels = threshold_image()
    filtered = oracle.bloom(5.0, 11, pixels)
    expected = np.maximum(pixels[..., :3].astype(np.float64) - 5.0, 0.0).sum(axis=(0, 1))
    assert np.allclose(filtered.sum(axis=(0, 1), dtype=np.float64), expected, rtol=0.01)


def test_bloom_mirroring():
    """BloomTest.h test_mirroring, as it was meant: filtering the point-mirrored image gives the point-mirrored result."""
    pixels = np.zeros((64, 64, 4), dtype=np.float16)
    pixels[..., 3] = 1
    pixels[:32, :32, 0] = 1; pixels[32:, :32, 1] = 1; pixels[:32, 32:, 2] = 1
    mirrored = np.ascontiguousarray(pixels[::-1, ::-1])
    assert np.allclose(oracle.bloom(0.0, 11, pixels), oracle.bloom(0.0, 11, mirrored)[::-1, ::-1], atol=1e-3)


# ---- BloomTest.h:218-245, the dual Kawase filter (tested by the reference, not used by its CameraEffects::process) --------------------------

def test_dual_kawase_energy_conservation():
    pixels = np.ones((64, 64, 4), dtype=np.float16)
    filtered = oracle.dual_kawase_bloom(0.0, 1, pixels).astype(np.float64)
    assert np.allclose(filtered[..., :3].sum(axis=(0, 1)), pixels[..., :3].astype(np.float64).sum(axis=(0, 1)), rtol=1e-6)


def test_dual_kawase_mirroring():
    """BloomTest.h test_mirroring with four half passes, as it was meant: filtering the point-mirrored image gives the point-mirrored result."""
    pixels = np.zeros((64, 64, 4), dtype=np.float16)
    pixels[..., 3] = 1
    pixels[:32, :32, 0] = 1; pixels[32:, :32, 1] = 1; pixels[:32, 32:, 2] = 1
    mirrored = np.ascontiguousarray(pixels[::-1, ::-1])
    a, b = oracle.dual_kawase_bloom(0.0, 4, pixels).astype(np.float32), oracle.dual_kawase_bloom(0.0, 4, mirrored).astype(np.float32)[::-1, ::-1]
    assert np.allclose(a, b, atol=1e-3)
    assert 0.05 < float(a[31, 31, 0]) < 0.95          # the colours have bled across the quadrant borders


def test_dual_kawase_thresholding():
    pixels = threshold_image()
    extracted = oracle.dual_kawase_bloom(5.0, 0, pixels).astype(np.float64)
    expected = np.maximum(pixels[..., :3].astype(np.float64) - 5.0, 0.0)
    assert np.allclose(extracted[..., :3].sum(axis=(0, 1)), expected.sum(axis=(0, 1)), rtol=1e-3)          # the bar of BloomTest.h:238-245; the level is stored as half
    assert np.array_equal(extracted[..., 3], pixels[..., 3].astype(np.float64))       # alpha is carried


def test_dual_kawase_levels_and_viewport():
    """More half passes than the image has levels are clamped (CameraEffects.cpp:211); a viewport is filtered on its own pixels only."""
    rng = np.random.default_rng(5)
    pixels = rng.uniform(0, 4, (20, 37, 4)).astype(np.float16)
    assert np.array_equal(oracle.dual_kawase_bloom(1.0, 40, pixels), oracle.dual_kawase_bloom(1.0, 6, pixels))      # 37 -> 6 levels
    inside = oracle.dual_kawase_bloom(1.0, 2, pixels, viewport=(5, 3, 16, 8))
    alone = oracle.dual_kawase_bloom(1.0, 2, np.ascontiguousarray(pixels[3:11, 5:21]))
    assert inside.shape == (8, 16, 4) and np.array_equal(inside, alone)


def test_bloom_reads_the_frame_around_the_viewport_horizontally():
    pixels = np.zeros((8, 32, 4), dtype=np.float16)
    pixels[:, 10, :3] = 8.0        # a bright column just left of the viewport
    inside = oracle.bloom(1.0, 8, pixels, viewport=(12, 2, 16, 4))
    assert inside.shape == (4, 16, 3) and inside[:, 0, 0].min() > 0.1 and inside[:, 8:, 0].max() == 0.0


# ---- tonemapping operators against the reference's CPU versions (Bifrost/Math/CameraEffects.h:135-283) in float64 ----------------

AP1_RGB2Y = np.array([0.2722287168, 0.6740817658, 0.0536895174])
D65_TO_D60 = np.array([[1.01303, 0.00610531, -0.014971], [0.00769823, 0.998165, -0.00503203], [-0.00284131, 0.00468516, 0.924507]])
SRGB_TO_XYZ = np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]])
XYZ_TO_AP1 = np.array([[1.6410233797, -0.3248032942, -0.2364246952], [-0.6636628587, 1.6153315917, 0.0167563477], [0.0117218943, -0.0082844420, 0.9883948585]])
SRGB_TO_AP1 = XYZ_TO_AP1 @ D65_TO_D60 @ SRGB_TO_XYZ


def filmic_reference(color, slope=0.91, toe=0.53, shoulder=0.23, black_clip=0.0, white_clip=0.035):
    working = np.maximum(SRGB_TO_AP1 @ color, 0.0)
    working = lerp(np.full(3, working @ AP1_RGB2Y), working, 0.96)
    toe_scale, shoulder_scale = 1.0 + black_clip - toe, 1.0 + white_clip - shoulder
    in_match = out_match = 0.18
    if toe > 0.8:
        toe_match = (1.0 - toe - out_match) / slope + math.log10(in_match)
    else:
        bt = (out_match + black_clip) / toe_scale - 1.0
        toe_match = math.log10(in_match) - 0.5 * math.log((1.0 + bt) / (1.0 - bt)) * (toe_scale / slope)
    straight_match = (1.0 - toe) / slope - toe_match
    shoulder_match = shoulder / slope - straight_match
    with np.errstate(divide="ignore", over="ignore"):
        log_color = np.log10(working)
        straight = (log_color + straight_match) * slope
        toe_color = -black_clip + (2.0 * toe_scale) / (1.0 + np.exp((log_color - toe_match) * (-2 * slope / toe_scale)))
        toe_color = np.where(log_color < toe_match, toe_color, straight)
        shoulder_color = (1.0 + white_clip) - (2.0 * shoulder_scale) / (1.0 + np.exp((log_color - shoulder_match) * (2 * slope / shoulder_scale)))
        shoulder_color = np.where(log_color > shoulder_match, shoulder_color, straight)
    t = np.clip((log_color - toe_match) / (shoulder_match - toe_match), 0.0, 1.0)
    t = 1.0 - t if shoulder_match < toe_match else t
    t = (3.0 - t * 2.0) * t * t
    tone = lerp(toe_color, shoulder_color, t)
    tone = lerp(np.full(3, tone @ AP1_RGB2Y), tone, 0.93)
    return np.linalg.inv(SRGB_TO_AP1) @ np.maximum(tone, 0.0)


def agx_reference(color):
    to_agx = np.array([[0.842479062253094, 0.0784335999999992, 0.0792237451477643], [0.0423282422610123, 0.878468636469772, 0.0791661274605434], [0.0423756549057051, 0.0784336, 0.879142973793104]])
    c = np.log2(to_agx @ color)
    c = np.clip((c - -12.47393) / (4.026069 - -12.47393), 0.0, 1.0)
    c = -0.00232 + c * (0.1191 + c * (0.4298 + c * (-6.868 + c * (31.96 + c * (-40.14 + c * 15.5)))))
    from_agx = np.array([[1.19687900512017, -0.0980208811401368, -0.0990297440797205], [-0.0528968517574562, 1.15190312990417, -0.0989611768448433],
                         [-0.0529716355144438, -0.0980434501171241, 1.15107367264116]])
    return np.abs(from_agx @ c) ** 2.2


def khronos_reference(color):
    start_compression, desaturation = 0.8 - 0.04, 0.15
    x = color.min()
    color = color - (x - 6.25 * x * x if x < 0.08 else 0.04)
    peak = color.max()
    if peak < start_compression:
        return color
    d = 1.0 - start_compression
    new_peak = 1.0 - d * d / (peak + d - start_compression)
    color = color * (new_peak / peak)
    g = 1.0 - 1.0 / (desaturation * (peak - new_peak) + 1.0)
    return lerp(color, np.full(3, new_peak), g)


def tonemapping_inputs():
    rng = np.random.default_rng(5)
    colours = np.exp2(rng.uniform(-10, 6, (400, 3)))
    colours[:20] = np.exp2(rng.uniform(-10, 6, (20, 1)))      # greys
    return colours.astype(np.float32)


@pytest.mark.parametrize("preset", ["ACES", "uncharted2", "HP", "legacy"])
def test_filmic_tonemapping_follows_the_reference_cpu_operator(preset):
    s = Settings.preset().set_tonemapping(preset)
    black_clip, toe, slope, shoulder, white_clip = Settings.TONEMAPPING_PRESETS[preset]
    colours = tonemapping_inputs()
    expected = np.array([filmic_reference(c.astype(np.float64), slope, toe, shoulder, black_clip, white_clip) for c in colours])
    # The shader carries AP1 -> sRGB as a rounded literal matrix, the CPU header inverts the forward matrix: a few 1e-6 apart.
    # With a straight segment between toe and shoulder (HP: toe_match < shoulder_match) the shader blends the two sigmoids
    # where the CPU header switches to the straight line; the oracle follows the shader, the two are 1.2e-4 apart there.
    assert np.allclose(oracle.tonemap(s, colours), expected, rtol=2e-4, atol=3e-4 if preset == "HP" else 2e-5)
    # 0.18 grey maps close to 0.18 by construction (in_match / out_match), apart from the two desaturation steps
    grey = oracle.tonemap(s, np.array([[0.18, 0.18, 0.18]], dtype=np.float32))[0]
    assert abs(float(grey.mean()) - 0.18) < 0.02


def test_agx_and_khronos_tonemapping_follow_the_reference_cpu_operators():
    colours = tonemapping_inputs()
    s = Settings.preset()
    s.tonemapping_mode = camera_effects.TONEMAPPING_AGX
    assert np.allclose(oracle.tonemap(s, colours), np.array([agx_reference(c.astype(np.float64)) for c in colours]), rtol=3e-4, atol=1e-6)
    s.tonemapping_mode = camera_effects.TONEMAPPING_KHRONOS_NEUTRAL
    assert np.allclose(oracle.tonemap(s, colours), np.array([khronos_reference(c.astype(np.float64)) for c in colours]), rtol=1e-5, atol=1e-6)
    s.tonemapping_mode = camera_effects.TONEMAPPING_LINEAR
    assert np.array_equal(oracle.tonemap(s, colours), colours)


def test_vignette_and_film_grain():
    assert oracle.vignette(0.5, 0.5, 0.63) == 1.0                       # the centre is untouched
    assert oracle.vignette(0.0, 0.0, 0.63) < oracle.vignette(0.25, 0.25, 0.63) < 1.0
    assert oracle.vignette(0.0, 0.0, 0.0) == 1.0                        # strength 0 switches it off
    grain = np.array([oracle.film_grain(x / 64.0, y / 36.0, 1 / 60.0, 1 / 255.0) for y in range(36) for x in range(64)])
    assert np.abs(grain).max() <= 0.5 / 255.0 + 1e-9 and abs(grain.mean()) < 0.2 / 255.0 and grain.std() > 0.2 / 255.0


def test_process_with_linear_settings_is_the_identity():
    rng = np.random.default_rng(3)
    pixels = np.ones((9, 16, 4), dtype=np.float16)
    pixels[..., :3] = rng.uniform(0, 4, (9, 16, 3)).astype(np.float16)
    image, exposure = oracle.process(Settings.linear(), 1 / 60.0, pixels)
    assert exposure == 1.0                                               # 2^bias, taken at once with eye adaptation off
    assert np.array_equal(image[..., :3], pixels[..., :3].astype(np.float32)) and np.all(image[..., 3] == 1.0)


def test_process_preset_exposes_a_dim_frame_up_and_a_bright_frame_down():
    s = Settings.preset()
    s.eye_adaptation_enabled = 0
    s.film_grain = 0.0
    for level in (0.1, 10.0):       # inside the preset's range of 2^-4 .. 2^4; luminances beyond it fall into the end bins
        pixels = grey_image(np.full((18, 32), level))
        image, exposure = oracle.process(s, 1 / 60.0, pixels)
        # histogram exposure: the average luminance lands in one bin whose centre is within half a bin (1/16 stop... 8 stops / 64 bins) of the level
        assert exposure == pytest.approx(1.0 / level, rel=0.1)
        centre = image[9, 16, :3]
        assert 0.6 < float(centre.mean()) < 0.85                        # exposed to about 1.0, which the filmic curve takes to about 0.72


# ---- the C-ABI surface ------------------------------------------------------------------------------------------------------------

def test_camera_effects_c_abi_is_exported_and_declared():
    import re
    from pathlib import Path
    header = (Path(__file__).resolve().parent.parent / "include" / "hipr_camera_effects_c.h").read_text()
    declared = sorted(set(re.findall(r"\b(hipr_camera_effects_\w+)\s*\(", header)))
    assert declared == sorted(camera_effects.C_ABI_SYMBOLS)
    lib = C.CDLL(str(capi.LIB_PATH))
    for name in camera_effects.C_ABI_SYMBOLS:
        assert hasattr(lib, name), name
    assert C.sizeof(camera_effects.Settings) == 19 * 4 and C.sizeof(camera_effects.FrameView) == 8 + 8 + 16
