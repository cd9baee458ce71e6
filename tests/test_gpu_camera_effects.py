"""GPU tests of the camera effects (csrc/camera_effects.hip) through the C-ABI of include/hipr_camera_effects_c.h.

The reference's own stage tests (tests/DX11RendererTests/{ExposureHistogram,LogAverageLuminance,Bloom}Test.h) with their
inputs and expectations, plus parity with the CPU oracle (oracle/camera_effects.cpp) on seeded random frames. Integer results
(histograms) are exact up to pixels whose log2 luminance lands on a bin boundary in one libm and not the other; float results
carry the stated tolerances (device and host transcendentals differ in the last ulp)."""
import math

import numpy as np
import pytest

import camera_effects_oracle as oracle
from bifrost3d_amd import camera_effects
from bifrost3d_amd.camera_effects import Settings
from test_camera_effects_cpu import (average_luminance_without_outlier, expected_log_average, geometric_mean_linear_exposure, grey_image, histogram_settings, log_average_images,
                                     shuffled_histogram, small_histogram_image, threshold_image, tiny_histogram_image)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    effects = camera_effects.CameraEffects(0)
    yield effects
    effects.close()


def random_frame(rows, pitch, seed, stops=6.0):
    """HDR-ish content: smooth gradients with texture, luminances over about 2^-stops .. 2^stops."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:pitch]
    base = np.exp2(stops * np.sin(x / 17.0 + seed) * np.cos(y / 11.0))[..., None]
    pixels = np.ones((rows, pitch, 4), dtype=np.float16)
    pixels[..., :3] = (base * rng.uniform(0.25, 1.0, (rows, pitch, 3))).astype(np.float16)
    return pixels


# ---- ExposureHistogramTest.h ------------------------------------------------------------------------------------------------

def test_histogram_tiny_image(fx):
    bins = fx.reduce_histogram(histogram_settings(), fx.upload(tiny_histogram_image()))
    assert np.array_equal(bins, np.ones(64, dtype=np.uint32))


def test_histogram_small_image(fx):
    bins = fx.reduce_histogram(histogram_settings(), fx.upload(small_histogram_image()))
    assert bins[0] == 4 + 64 and bins[63] == 4 + 64 and np.all(bins[1:63] == 4)


@pytest.mark.parametrize("viewport", [None, (5, 3, 200, 97)])
def test_histogram_matches_oracle(fx, viewport):
    pixels = random_frame(131, 257, 1)
    s = histogram_settings(-4.0, 4.0)
    gpu, cpu = fx.reduce_histogram(s, fx.upload(pixels), viewport), oracle.histogram(s, pixels, viewport)
    count = (viewport[2] * viewport[3]) if viewport else 131 * 257
    assert int(gpu.sum()) == count == int(cpu.sum())
    assert int(np.abs(gpu.astype(np.int64) - cpu.astype(np.int64)).sum()) <= 2 * max(1, count // 5000)      # bin-boundary pixels only


def test_histogram_of_a_full_frame(fx):
    pixels = random_frame(1080, 1920, 2)
    bins = fx.reduce_histogram(Settings.preset(), fx.upload(pixels))
    assert int(bins.sum()) == 1920 * 1080
    cpu = oracle.histogram(Settings.preset(), pixels)
    assert int(np.abs(bins.astype(np.int64) - cpu.astype(np.int64)).sum()) <= 400


@pytest.mark.parametrize("bins", [np.ones(64, dtype=np.uint32), shuffled_histogram()], ids=["constant", "shuffled"])
def test_exposure_from_histogram(fx, bins):
    s = histogram_settings()
    expected = 1.0 / average_luminance_without_outlier(bins, 0.8, 0.95, -8.0, 4.0)
    gpu = fx.exposure_from_histogram(s, 1 / 60.0, bins)
    assert gpu == pytest.approx(expected, rel=2e-6)
    assert gpu == pytest.approx(oracle.exposure_from_histogram(s, 1 / 60.0, bins), rel=1e-6)


def test_eye_adaptation(fx):
    s = histogram_settings()
    bins = shuffled_histogram()
    target = fx.exposure_from_histogram(s, 1 / 60.0, bins)
    s.eye_adaptation_enabled, s.eye_adaptation_brightness, s.eye_adaptation_darkness = 1, 3.0, 1.0
    assert fx.exposure_from_histogram(s, 0.5, bins, 0.0) == pytest.approx(target * (1 - 2 ** -1.5), rel=1e-6)
    assert fx.exposure_from_histogram(s, 0.5, bins, 2 * target) == pytest.approx(2 * target - target * (1 - 2 ** -0.5), rel=1e-6)


# ---- LogAverageLuminanceTest.h ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["tiny", "large", "black"])
def test_log_average_luminance(fx, name):
    pixels = log_average_images()[name]
    frame = fx.upload(pixels)
    log_average = fx.log_average(frame)
    assert log_average == pytest.approx(expected_log_average(pixels), rel=1e-5)
    assert log_average == pytest.approx(oracle.log_average(pixels), rel=1e-5)
    s = histogram_settings(-24.0, 24.0)
    s.exposure_mode = camera_effects.EXPOSURE_LOG_AVERAGE
    linear_exposure = geometric_mean_linear_exposure(log_average)
    assert fx.exposure_from_log_average(s, 1 / 60.0, frame, linear_exposure) == pytest.approx(linear_exposure, rel=1e-5)


def test_log_average_of_a_viewport_matches_oracle(fx):
    pixels = random_frame(300, 500, 4)
    viewport = (17, 9, 401, 233)
    s = Settings.preset()
    s.exposure_mode = camera_effects.EXPOSURE_LOG_AVERAGE
    assert fx.log_average(fx.upload(pixels), viewport) == pytest.approx(oracle.log_average(pixels, viewport), rel=1e-5)
    assert fx.exposure_from_log_average(s, 0.1, fx.upload(pixels), 0.25, viewport) == pytest.approx(oracle.exposure_from_log_average(s, 0.1, pixels, 0.25, viewport), rel=1e-5)


# ---- BloomTest.h, the Gaussian filter ------------------------------------------------------------------------------------------

def test_bloom_energy_conservation(fx):
    pixels = np.ones((64, 64, 4), dtype=np.float16)
    filtered = fx.bloom(0.0, 11, fx.upload(pixels)).astype(np.float64)
    assert np.allclose(filtered[..., :3].sum(axis=(0, 1)), pixels[..., :3].sum(axis=(0, 1), dtype=np.float64), rtol=0.002)


def test_bloom_thresholding(fx):
    pixels = threshold_image()
    filtered = fx.bloom(5.0, 11, fx.upload(pixels)).astype(np.float64)
    expected = np.maximum(pixels.astype(np.float64) - np.array([5.0, 5.0, 5.0, 0.0]), 0.0).sum(axis=(0, 1))
    assert np.allclose(filtered.sum(axis=(0, 1)), expected, rtol=0.01)      # all four channels, alpha is written as one


def test_bloom_mirroring(fx):
    pixels = np.zeros((64, 64, 4), dtype=np.float16)
    pixels[..., 3] = 1
    pixels[:32, :32, 0] = 1; pixels[32:, :32, 1] = 1; pixels[:32, 32:, 2] = 1
    mirrored = np.ascontiguousarray(pixels[::-1, ::-1])
    assert np.allclose(fx.bloom(0.0, 11, fx.upload(pixels)), fx.bloom(0.0, 11, fx.upload(mirrored))[::-1, ::-1], atol=1e-3)


@pytest.mark.parametrize("support,viewport", [(11, None), (54, None), (8, (12, 5, 100, 40)), (1, None), (0, None)])
def test_bloom_matches_oracle(fx, support, viewport):
    pixels = random_frame(72, 160, 6, stops=3.0)
    gpu = fx.bloom(1.5, support, fx.upload(pixels), viewport).astype(np.float32)
    cpu = oracle.bloom(1.5, support, pixels, viewport)
    assert gpu.shape[:2] == cpu.shape[:2]
    # both sides round the intermediate and the result to half: one half ulp (2^-11 relative) each way
    assert np.allclose(gpu[..., :3], cpu, rtol=2e-3, atol=1e-4)
    assert np.all(gpu[..., 3] == 1.0)


# ---- the dual Kawase filter (BloomTest.h:218-245; Bloom.hlsl:69-119) ------------------------------------------------------------------

@pytest.mark.parametrize("half_passes,viewport", [(0, None), (1, None), (3, None), (4, (12, 5, 100, 40)), (2, (7, 3, 33, 21)), (40, None)])
def test_dual_kawase_matches_oracle(fx, half_passes, viewport):
    """Odd level sizes (the half-sized grid does not sit on texel pairs), viewports inside a larger frame, more passes than levels."""
    pixels = random_frame(72, 161, 9, stops=3.0)
    pixels[..., 3] = np.random.default_rng(2).uniform(0, 1, pixels.shape[:2]).astype(np.float16)
    gpu = fx.dual_kawase_bloom(1.5, half_passes, fx.upload(pixels), viewport).astype(np.float32)
    cpu = oracle.dual_kawase_bloom(1.5, half_passes, pixels, viewport).astype(np.float32)
    assert gpu.shape == cpu.shape
    # every level is rounded to half on both sides: a half ulp per level each way
    assert np.allclose(gpu, cpu, rtol=4e-3, atol=2e-4)
    assert float(np.abs(gpu - cpu).max()) <= 4e-3 * float(np.abs(cpu).max())


def test_dual_kawase_reference_cases_on_the_device(fx):
    white = np.ones((64, 64, 4), dtype=np.float16)
    filtered = fx.dual_kawase_bloom(0.0, 1, fx.upload(white)).astype(np.float64)
    assert np.allclose(filtered[..., :3].sum(axis=(0, 1)), 64.0 * 64.0, rtol=1e-6)                                   # energy conservation
    quadrants = np.zeros((64, 64, 4), dtype=np.float16)
    quadrants[..., 3] = 1
    quadrants[:32, :32, 0] = 1; quadrants[32:, :32, 1] = 1; quadrants[:32, 32:, 2] = 1
    a = fx.dual_kawase_bloom(0.0, 4, fx.upload(quadrants)).astype(np.float32)
    b = fx.dual_kawase_bloom(0.0, 4, fx.upload(np.ascontiguousarray(quadrants[::-1, ::-1]))).astype(np.float32)[::-1, ::-1]
    assert np.allclose(a, b, atol=1e-3)                                                                               # mirroring
    ramp = threshold_image()
    extracted = fx.dual_kawase_bloom(5.0, 0, fx.upload(ramp)).astype(np.float64)
    expected = np.maximum(ramp[..., :3].astype(np.float64) - 5.0, 0.0)
    assert np.allclose(extracted[..., :3].sum(axis=(0, 1)), expected.sum(axis=(0, 1)), rtol=1e-3)                    # thresholding


# ---- process: exposure -> bloom -> tonemapping (CameraEffects.cpp:412-507) ----------------------------------------------------------

def settings_for(mode, exposure_mode=camera_effects.EXPOSURE_HISTOGRAM, bloom_threshold=math.inf, film_grain=0.0):
    s = Settings.preset()
    s.tonemapping_mode, s.exposure_mode, s.bloom_threshold, s.film_grain = mode, exposure_mode, bloom_threshold, film_grain
    s.eye_adaptation_enabled = 0
    return s


@pytest.mark.parametrize("mode", [camera_effects.TONEMAPPING_LINEAR, camera_effects.TONEMAPPING_FILMIC, camera_effects.TONEMAPPING_AGX, camera_effects.TONEMAPPING_KHRONOS_NEUTRAL])
@pytest.mark.parametrize("exposure_mode", [camera_effects.EXPOSURE_FIXED, camera_effects.EXPOSURE_LOG_AVERAGE, camera_effects.EXPOSURE_HISTOGRAM])
def test_process_matches_oracle(fx, mode, exposure_mode):
    pixels = random_frame(90, 160, 8, stops=4.0)
    s = settings_for(mode, exposure_mode)
    fx.linear_exposure = 0.0
    gpu = fx.process(s, 1 / 60.0, fx.upload(pixels)).cpu().numpy()
    cpu, cpu_exposure = oracle.process(s, 1 / 60.0, pixels)
    assert fx.linear_exposure == pytest.approx(cpu_exposure, rel=2e-5)
    assert np.isfinite(gpu).all()
    assert np.allclose(gpu, cpu, rtol=3e-4, atol=3e-5)


def test_process_with_bloom_viewports_and_target_offset(fx):
    pixels = random_frame(120, 200, 9, stops=4.0)
    viewport, offset = (10, 6, 160, 90), (7, 3)
    s = settings_for(camera_effects.TONEMAPPING_FILMIC, bloom_threshold=2.0)
    s.bloom_support = 0.1
    fx.linear_exposure = 0.0
    target = fx.process(s, 1 / 60.0, fx.upload(pixels), viewport, target_offset=offset).cpu().numpy()
    cpu, _ = oracle.process(s, 1 / 60.0, pixels, viewport=viewport)
    assert target.shape == (90 + 3, 160 + 7, 4)
    assert np.all(target[:3] == 0) and np.all(target[:, :7] == 0)             # outside the target viewport nothing is written
    assert np.allclose(target[3:, 7:], cpu, rtol=2e-3, atol=2e-4)             # bloom intermediates are halfs
    no_bloom = settings_for(camera_effects.TONEMAPPING_FILMIC)
    fx.linear_exposure = 0.0
    plain = fx.process(no_bloom, 1 / 60.0, fx.upload(pixels), viewport, target_offset=offset).cpu().numpy()
    assert np.abs(plain[3:, 7:] - cpu).max() > 0.01                            # the bloom is there


def test_process_linear_settings_are_the_identity(fx):
    pixels = random_frame(36, 64, 10)
    fx.linear_exposure = 0.0
    gpu = fx.process(Settings.linear(), 1 / 60.0, fx.upload(pixels), target_format=camera_effects.TARGET_RGBA16F).cpu().numpy()
    assert fx.linear_exposure == 1.0
    assert np.array_equal(gpu[..., :3], pixels[..., :3]) and np.all(gpu[..., 3] == 1.0)


def test_process_srgb8_target(fx):
    pixels = random_frame(45, 80, 11, stops=3.0)
    s = settings_for(camera_effects.TONEMAPPING_FILMIC)
    fx.linear_exposure = 0.0
    gpu = fx.process(s, 1 / 60.0, fx.upload(pixels), target_format=camera_effects.TARGET_RGBA8_SRGB).cpu().numpy()
    cpu, _ = oracle.process(s, 1 / 60.0, pixels)
    linear = np.clip(cpu[..., :3].astype(np.float64), 0.0, 1.0)
    encoded = np.where(linear < 0.0031308, linear * 12.92, 1.055 * linear ** (1 / 2.4) - 0.055)
    expected = np.floor(np.clip(encoded, 0.0, 1.0) * 255.0 + 0.5)
    assert gpu.dtype == np.uint8 and np.all(gpu[..., 3] == 255)
    assert np.abs(gpu[..., :3].astype(np.int32) - expected.astype(np.int32)).max() <= 1


def test_film_grain_is_bounded_and_mostly_reproduces(fx):
    pixels = grey_image(np.full((36, 64), 0.18))
    s = settings_for(camera_effects.TONEMAPPING_LINEAR, camera_effects.EXPOSURE_FIXED, film_grain=1 / 255.0)
    s.vignette = 0.0
    fx.linear_exposure = 0.0
    gpu = fx.process(s, 1 / 60.0, fx.upload(pixels)).cpu().numpy()
    cpu, _ = oracle.process(s, 1 / 60.0, pixels)
    grain = gpu[..., 0] - np.float32(np.float16(0.18))
    assert np.abs(grain).max() <= 0.5 / 255.0 + 1e-6 and grain.std() > 0.2 / 255.0
    # sin(x) * 43758.5 turns the last ulp of sin into 3e-3 of the noise: most pixels agree, a few wrap around the fraction
    assert (np.abs(gpu - cpu).max(axis=-1) < 2e-4).mean() > 0.97


def test_exposure_adapts_over_frames(fx):
    """m_linear_exposure is carried from frame to frame (CameraEffects.h:222-223): a dark frame brightens at the eye's speed."""
    pixels = grey_image(np.full((36, 64), 0.125))
    s = Settings.preset()
    s.film_grain = 0.0
    frame = fx.upload(pixels)
    fx.linear_exposure = 0.0
    exposures, cpu_exposure = [], 0.0
    for _ in range(4):
        fx.process(s, 0.25, frame)
        exposures.append(fx.linear_exposure)
        _, cpu_exposure = oracle.process(s, 0.25, pixels, cpu_exposure)
        assert exposures[-1] == pytest.approx(cpu_exposure, rel=1e-5)
    assert exposures[0] < exposures[1] < exposures[2] < exposures[3] < 8.5      # towards about 1 / 0.125
    assert exposures[0] == pytest.approx(exposures[3] / (1 - (2 ** -0.75) ** 4) * (1 - 2 ** -0.75), rel=1e-3)


def test_full_frame_and_stage_timers(fx):
    pixels = random_frame(1080, 1920, 12, stops=4.0)
    s = Settings.preset()
    s.bloom_threshold = 4.0
    frame = fx.upload(pixels)
    fx.set_instrumentation(True)
    fx.reset_timers()
    fx.linear_exposure = 0.0
    target = fx.process(s, 1 / 60.0, frame, target_format=camera_effects.TARGET_RGBA8_SRGB)
    fx.synchronize()
    times = fx.times()
    fx.set_instrumentation(False)
    assert times.exposure_launches == 1 and times.bloom_horizontal_launches == 1 and times.bloom_vertical_launches == 1 and times.tonemap_launches == 1
    assert 0 < times.tonemap_ms < 50 and 0 < times.exposure_ms < 50
    image = target.cpu().numpy()
    assert image.shape == (1080, 1920, 4) and image[..., :3].std() > 10


def test_errors_are_reported(fx):
    from bifrost3d_amd import capi
    frame = fx.upload(random_frame(16, 16, 13))
    with pytest.raises(capi.HiprError):
        fx.reduce_histogram(Settings.preset(), frame, viewport=(8, 8, 16, 16))      # viewport outside the frame
    s = Settings.preset()
    s.tonemapping_mode = 9
    with pytest.raises(capi.HiprError):
        fx.process(s, 1 / 60.0, frame)
    with pytest.raises(capi.HiprError):
        fx.bloom(1.0, -3, frame)


def test_tiled_bloom_is_bit_identical_to_the_direct_kernels(fx, monkeypatch):
    """The LDS-staged bloom passes do the direct kernels' operations in the same order: identical halfs, for viewports, odd sizes
    and filters from 2 to 108 pixels of support."""
    monkeypatch.setenv("HIPR_BLOOM_DIRECT", "1")
    direct = camera_effects.CameraEffects(0)
    monkeypatch.delenv("HIPR_BLOOM_DIRECT")
    try:
        pixels = random_frame(211, 333, 21, stops=3.0)
        frame, direct_frame = fx.upload(pixels), direct.upload(pixels)
        for support, viewport in ((2, None), (11, None), (54, None), (108, None), (54, (7, 5, 301, 190)), (9, (300, 200, 33, 11))):
            assert np.array_equal(fx.bloom(1.0, support, frame, viewport).view(np.uint16), direct.bloom(1.0, support, direct_frame, viewport).view(np.uint16)), (support, viewport)
    finally:
        direct.close()


def test_the_effects_follow_the_callers_current_stream(fx):
    """the effects used to capture torch's current stream once, at construction; a caller working under `with torch.cuda.stream(s)` then had its upload and
    its zero fill on `s` and the effects' kernels on the old stream, unordered. Now every call moves the effects to the stream that is current: a frame produced on a
    side stream by a long chain of kernels (so that it is certainly not finished when process() is called) is processed after it, and the result equals the one of the
    default stream bit for bit; back on the default stream the object follows again."""
    import torch
    settings = Settings.preset()
    settings.eye_adaptation_enabled, settings.film_grain = 0, 0.0
    pixels = random_frame(270, 480, 31)
    fx.linear_exposure = 0.0
    expected = fx.process(settings, 1 / 60.0, fx.upload(pixels)).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        frame = torch.from_numpy(pixels).to(fx.device, non_blocking=True).float()
        for _ in range(200):      # the frame becomes final at the end of a chain of dependent kernels on the side stream (exact: powers of two)
            frame = (frame * 2.0) * 0.5
        frame = frame.half()
        fx.linear_exposure = 0.0
        on_side = fx.process(settings, 1 / 60.0, frame)
        assert fx._stream_in_use == side.cuda_stream
        result = on_side.clone()
    side.synchronize()
    assert torch.equal(frame.cpu(), torch.from_numpy(pixels))      # exact powers of two up and down: the same half pixels
    assert torch.equal(result.cpu(), expected.cpu())
    fx.linear_exposure = 0.0
    again = fx.process(settings, 1 / 60.0, fx.upload(pixels))
    assert fx._stream_in_use == torch.cuda.current_stream(fx.device).cuda_stream
    torch.cuda.synchronize()
    assert torch.equal(again.cpu(), expected.cpu())
