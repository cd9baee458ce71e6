"""ctypes binding of the denoiser part of the CPU oracle (oracle/denoiser.cpp). Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from bifrost3d_amd.denoiser import Settings
from oracle_bindings import get_oracle


def denoise(noisy: np.ndarray, albedo: np.ndarray, settings: Settings) -> np.ndarray:
    """oracle_denoise on float32 (height, width, 4) frames."""
    lib = get_oracle(False).lib
    fp = C.POINTER(C.c_float)
    lib.oracle_denoise.argtypes = [fp, fp, C.c_int, C.c_int, C.POINTER(Settings), fp]
    noisy, albedo = np.ascontiguousarray(noisy, dtype=np.float32), np.ascontiguousarray(albedo, dtype=np.float32)
    assert noisy.shape == albedo.shape and noisy.ndim == 3 and noisy.shape[2] == 4
    out = np.empty_like(noisy)
    status = lib.oracle_denoise(noisy.ctypes.data_as(fp), albedo.ctypes.data_as(fp), noisy.shape[1], noisy.shape[0], C.byref(settings), out.ctypes.data_as(fp))
    if status != 0:
        raise ValueError("oracle_denoise rejected its arguments")
    return out


def test_frames(width=96, height=64, seed=7, noise=0.35):
    """A synthetic {noisy, albedo, clean} triple: two albedo regions split by a slanted edge, smooth HDR irradiance with a bright
    spot, multiplicative noise; an emitter patch without albedo."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    left = (x + 0.3 * y) < 0.55 * width
    albedo = np.where(left[..., None], np.float32([0.8, 0.3, 0.2]), np.float32([0.15, 0.5, 0.7])).astype(np.float32)
    irradiance = (0.4 + 0.6 * x / width + 6.0 * np.exp(-((x - 0.7 * width) ** 2 + (y - 0.4 * height) ** 2) / (0.02 * width * width))).astype(np.float32)
    clean = albedo * irradiance[..., None]
    emitter = (x > 0.1 * width) & (x < 0.2 * width) & (y > 0.7 * height) & (y < 0.85 * height)
    albedo[emitter] = 0.0
    clean[emitter] = np.float32([12.0, 10.0, 7.0])
    noisy = clean * rng.gamma(shape=1.0 / noise ** 2, scale=noise ** 2, size=clean.shape).astype(np.float32)
    noisy[emitter] = clean[emitter]
    pad = lambda a, w: np.concatenate([a, np.full(a.shape[:2] + (1,), w, np.float32)], axis=-1)
    return pad(noisy, 1.0), pad(albedo, 1.0), pad(clean, 1.0), left, emitter
