"""The denoising stage without a GPU: the C-ABI surface of include/hipr_denoiser_c.h and what the CPU restatement of the filter
(oracle/denoiser.cpp) does to images -- the properties the GPU tests then hold the HIP kernels to, pixel by pixel."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from bifrost3d_amd import capi, denoiser
import denoiser_oracle

ROOT = Path(__file__).resolve().parent.parent


def settings(**changes):
    s = denoiser.Settings(5, 0.1, 1.0, 0.001)
    for k, v in changes.items():
        setattr(s, k, v)
    return s


def test_denoiser_c_abi_is_exported_and_declared():
    header = (ROOT / "include" / "hipr_denoiser_c.h").read_text()
    declared = sorted(set(re.findall(r"\b(hipr_denoiser_\w+)\s*\(", header)))
    assert declared == sorted(denoiser.C_ABI_SYMBOLS)
    lib = C.CDLL(str(capi.LIB_PATH))
    for name in denoiser.C_ABI_SYMBOLS:
        assert hasattr(lib, name), name
    assert C.sizeof(denoiser.Settings) == 16
    s = denoiser.default_settings(capi.load_library())      # no device needed
    assert (s.iterations, round(s.sigma_albedo, 6), s.sigma_luminance, round(s.albedo_floor, 6)) == (5, 0.1, 1.0, 0.001)


def test_a_constant_image_is_a_fixed_point():
    noisy = np.zeros((20, 33, 4), np.float32); noisy[..., :3] = [0.7, 1.9, 0.05]; noisy[..., 3] = 1
    albedo = np.zeros_like(noisy); albedo[..., :3] = [0.5, 0.9, 0.0]
    out = denoiser_oracle.denoise(noisy, albedo, settings())
    assert np.allclose(out[..., :3], noisy[..., :3], rtol=2e-6, atol=0)
    assert np.all(out[..., 3] == 1)


def test_noise_drops_and_the_mean_stays():
    noisy, albedo, clean, left, emitter = denoiser_oracle.test_frames()
    out = denoiser_oracle.denoise(noisy, albedo, settings())
    interior = ~emitter
    err_in = np.sqrt(np.mean((noisy[interior][:, :3] - clean[interior][:, :3]) ** 2))
    err_out = np.sqrt(np.mean((out[interior][:, :3] - clean[interior][:, :3]) ** 2))
    assert err_out < 0.45 * err_in, (err_in, err_out)
    # an average of neighbours neither creates nor loses light on the whole
    assert np.mean(out[..., :3]) == pytest.approx(np.mean(noisy[..., :3]), rel=0.03)
    assert np.all(np.isfinite(out)) and np.all(out[..., :3] >= 0)


def test_albedo_edges_and_emitters_survive():
    noisy, albedo, clean, left, emitter = denoiser_oracle.test_frames(noise=0.2)
    out = denoiser_oracle.denoise(noisy, albedo, settings())
    # the two regions keep their colour ratio: red dominates left of the edge, blue right of it, right up to the edge
    assert np.all(out[left & ~emitter][:, 0] > out[left & ~emitter][:, 2])
    assert np.all(out[~left & ~emitter][:, 2] > out[~left & ~emitter][:, 0])
    # the emitter (no albedo, 10x brighter than its surroundings) is not smeared into them, nor they into it
    assert np.allclose(out[emitter][:, :3], clean[emitter][:, :3], rtol=0.05)
    ring = np.zeros_like(emitter)
    ys, xs = np.where(emitter)
    ring[ys.min() - 3:ys.max() + 4, xs.min() - 3:xs.max() + 4] = True
    ring &= ~emitter
    assert np.all(out[ring][:, :3].max(axis=-1) < 4.0)


def test_passes_widen_the_filter_and_sigmas_gate_the_taps():
    noisy, albedo, clean, left, emitter = denoiser_oracle.test_frames()
    interior = ~emitter
    rms = lambda a: float(np.sqrt(np.mean((a[interior][:, :3] - clean[interior][:, :3]) ** 2)))
    one, three, five = (rms(denoiser_oracle.denoise(noisy, albedo, settings(iterations=n))) for n in (1, 3, 5))
    # one pass (5 x 5 pixels) leaves most of the noise, three (29 x 29) remove most of it; the last two trade residual noise against
    # blur of the smooth highlight in this image, so they are only held to the bar of the whole filter
    assert three < 0.7 * one < 0.7 * rms(noisy)
    assert five < 0.45 * rms(noisy)
    # a luminance sigma near zero admits only taps of (by chance) equal luminance: the image passes through
    sharp = denoiser_oracle.denoise(noisy, albedo, settings(sigma_luminance=1e-6))
    assert np.mean(np.isclose(sharp[..., :3], noisy[..., :3], rtol=1e-3, atol=1e-6)) > 0.995


def test_the_oracle_rejects_what_the_product_rejects():
    noisy, albedo, *_ = denoiser_oracle.test_frames(16, 8)
    for bad in (settings(iterations=0), settings(iterations=13), settings(sigma_albedo=0.0), settings(sigma_luminance=-1.0), settings(albedo_floor=-0.1)):
        with pytest.raises(ValueError):
            denoiser_oracle.denoise(noisy, albedo, bad)
