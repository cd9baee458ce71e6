"""ctypes bindings of the camera effects part of the CPU oracle (oracle/camera_effects.cpp). Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from bifrost3d_amd.camera_effects import FrameView, Rect, Settings
from oracle_bindings import get_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = get_oracle(False).lib
        SP, FP, fp, up = C.POINTER(Settings), C.POINTER(FrameView), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        _lib.oracle_ce_histogram.argtypes = [SP, FP, up]
        _lib.oracle_ce_exposure_from_histogram.argtypes = [SP, C.c_float, up, C.c_float]; _lib.oracle_ce_exposure_from_histogram.restype = C.c_float
        _lib.oracle_ce_log_average.argtypes = [FP]; _lib.oracle_ce_log_average.restype = C.c_double
        _lib.oracle_ce_exposure_from_log_average.argtypes = [SP, C.c_float, FP, C.c_float]; _lib.oracle_ce_exposure_from_log_average.restype = C.c_float
        _lib.oracle_ce_gaussian_taps.argtypes = [C.c_float, C.c_int, fp, fp]
        _lib.oracle_ce_bloom.argtypes = [C.c_float, C.c_int, FP, fp]
        _lib.oracle_ce_dual_kawase_bloom.argtypes = [C.c_float, C.c_uint, FP, C.POINTER(C.c_uint16)]
        _lib.oracle_ce_tonemap.argtypes = [SP, fp, C.c_int, fp]
        _lib.oracle_ce_vignette.argtypes = [C.c_float] * 3; _lib.oracle_ce_vignette.restype = C.c_float
        _lib.oracle_ce_film_grain.argtypes = [C.c_float] * 4; _lib.oracle_ce_film_grain.restype = C.c_float
        _lib.oracle_ce_process.argtypes = [SP, C.c_float, FP, fp, fp]
    return _lib


def view_of(half_pixels: np.ndarray, viewport=None) -> FrameView:
    """FrameView over a host (rows, pitch, 4) float16 array. The caller keeps the array alive."""
    assert half_pixels.dtype == np.float16 and half_pixels.flags["C_CONTIGUOUS"] and half_pixels.shape[2] == 4
    rows, pitch = half_pixels.shape[:2]
    x, y, w, h = viewport if viewport is not None else (0, 0, pitch, rows)
    return FrameView(half_pixels.ctypes.data, pitch, rows, Rect(x, y, w, h))


def histogram(settings: Settings, half_pixels, viewport=None) -> np.ndarray:
    out = np.zeros(64, dtype=np.uint32)
    view = view_of(half_pixels, viewport)
    lib().oracle_ce_histogram(C.byref(settings), C.byref(view), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def exposure_from_histogram(settings: Settings, delta_time: float, bins, current_exposure: float = 0.0) -> float:
    bins = np.ascontiguousarray(bins, dtype=np.uint32)
    return lib().oracle_ce_exposure_from_histogram(C.byref(settings), delta_time, bins.ctypes.data_as(C.POINTER(C.c_uint32)), current_exposure)


def log_average(half_pixels, viewport=None) -> float:
    view = view_of(half_pixels, viewport)
    return lib().oracle_ce_log_average(C.byref(view))


def exposure_from_log_average(settings: Settings, delta_time: float, half_pixels, current_exposure: float = 0.0, viewport=None) -> float:
    view = view_of(half_pixels, viewport)
    return lib().oracle_ce_exposure_from_log_average(C.byref(settings), delta_time, C.byref(view), current_exposure)


def gaussian_taps(std_dev: float, count: int):
    offsets, weights = np.zeros(count, np.float32), np.zeros(count, np.float32)
    lib().oracle_ce_gaussian_taps(std_dev, count, offsets.ctypes.data_as(C.POINTER(C.c_float)), weights.ctypes.data_as(C.POINTER(C.c_float)))
    return offsets, weights


def bloom(threshold: float, support: int, half_pixels, viewport=None) -> np.ndarray:
    view = view_of(half_pixels, viewport)
    out = np.zeros((view.viewport.height, view.viewport.width, 3), dtype=np.float32)
    lib().oracle_ce_bloom(threshold, support, C.byref(view), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def dual_kawase_bloom(threshold: float, half_passes: int, half_pixels, viewport=None) -> np.ndarray:
    """(height, width, 4) float16, the viewport-sized level 0 of the filter."""
    view = view_of(half_pixels, viewport)
    out = np.zeros((view.viewport.height, view.viewport.width, 4), dtype=np.float16)
    lib().oracle_ce_dual_kawase_bloom(threshold, half_passes, C.byref(view), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def tonemap(settings: Settings, rgb: np.ndarray) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(rgb)
    lib().oracle_ce_tonemap(C.byref(settings), rgb.ctypes.data_as(C.POINTER(C.c_float)), len(rgb), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def vignette(u: float, v: float, scale: float) -> float:
    return lib().oracle_ce_vignette(u, v, scale)


def film_grain(u: float, v: float, delta_time: float, scale: float) -> float:
    return lib().oracle_ce_film_grain(u, v, delta_time, scale)


def process(settings: Settings, delta_time: float, half_pixels, linear_exposure: float = 0.0, viewport=None):
    """Returns (viewport-sized RGBA float32 image, the new linear exposure)."""
    view = view_of(half_pixels, viewport)
    out = np.zeros((view.viewport.height, view.viewport.width, 4), dtype=np.float32)
    exposure = C.c_float(linear_exposure)
    lib().oracle_ce_process(C.byref(settings), delta_time, C.byref(view), C.byref(exposure), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out, exposure.value
