"""ctypes plumbing for the denoiser C-ABI (include/hipr_denoiser_c.h) -- tests and benchmarks only.

Device memory is held by torch tensors (plumbing, as everywhere in this package); the filter is the HIP kernels of
csrc/denoiser.hip behind `hipr_denoiser_*`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

SHOW_FILTERED, SHOW_NOISE, SHOW_ALBEDO = 0, 1, 2


class Settings(C.Structure):
    """HiprDenoiserSettings."""
    _fields_ = [("iterations", C.c_uint32), ("sigma_albedo", C.c_float), ("sigma_luminance", C.c_float), ("albedo_floor", C.c_float)]


C_ABI_SYMBOLS = (
    "hipr_denoiser_create", "hipr_denoiser_destroy", "hipr_denoiser_last_error", "hipr_denoiser_set_stream", "hipr_denoiser_synchronize",
    "hipr_denoiser_default_settings", "hipr_denoiser_process", "hipr_denoiser_filter_host",
)


def declare(lib: C.CDLL):
    vp, SP, FP, u32 = C.c_void_p, C.POINTER(Settings), C.POINTER(C.c_float), C.c_uint32
    lib.hipr_denoiser_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.hipr_denoiser_destroy.argtypes = [vp]; lib.hipr_denoiser_destroy.restype = None
    lib.hipr_denoiser_last_error.argtypes = [vp]; lib.hipr_denoiser_last_error.restype = C.c_char_p
    lib.hipr_denoiser_set_stream.argtypes = [vp, vp]
    lib.hipr_denoiser_synchronize.argtypes = [vp]
    lib.hipr_denoiser_default_settings.argtypes = [SP]
    lib.hipr_denoiser_process.argtypes = [vp, SP, vp, u32, vp, u32, u32, u32, C.c_int, C.c_int, vp, u32]
    lib.hipr_denoiser_filter_host.argtypes = [vp, SP, FP, FP, u32, u32, FP]


def default_settings(lib: C.CDLL | None = None) -> Settings:
    lib = lib or capi.load_library()
    declare(lib)
    s = Settings()
    if lib.hipr_denoiser_default_settings(C.byref(s)) != 0:
        raise capi.HiprError("hipr_denoiser_default_settings failed")
    return s


class Denoiser:
    """One HiprDenoiser object. Device frames are torch CUDA tensors of shape (rows, pitch, 4), dtype float16."""

    def __init__(self, device_index: int = 0):
        self.lib = capi.load_library()
        declare(self.lib)
        self.handle = C.c_void_p()
        status = self.lib.hipr_denoiser_create(device_index, C.byref(self.handle))
        if status != 0:
            raise capi.HiprError(f"hipr_denoiser_create failed with {status}")
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device_index)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hipr_denoiser_destroy(self.handle)
            self.handle = None

    __del__ = close

    def _check(self, status: int, what: str):
        if status != 0:
            message = self.lib.hipr_denoiser_last_error(self.handle)
            raise capi.HiprError(f"{what} failed with {status}: {message.decode() if message else ''}")

    def filter_host(self, noisy: np.ndarray, albedo: np.ndarray, settings: Settings | None = None) -> np.ndarray:
        """The filter on float32 (height, width, 4) host frames."""
        noisy, albedo = np.ascontiguousarray(noisy, dtype=np.float32), np.ascontiguousarray(albedo, dtype=np.float32)
        assert noisy.shape == albedo.shape and noisy.ndim == 3 and noisy.shape[2] == 4
        settings = settings or default_settings(self.lib)
        out = np.empty_like(noisy)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self._check(self.lib.hipr_denoiser_filter_host(self.handle, C.byref(settings), fp(noisy), fp(albedo), noisy.shape[1], noisy.shape[0], fp(out)), "hipr_denoiser_filter_host")
        return out

    def process(self, noisy, albedo, out, width: int, height: int, settings: Settings | None = None, update_filtered: bool = True, show: int = SHOW_FILTERED):
        """One frame of the backend: torch half4 frames (rows, pitch, 4) on the device; writes `out` in place."""
        settings = settings or default_settings(self.lib)
        self.torch.cuda.current_stream(self.device).synchronize()      # the frames were filled on torch's stream; the backend runs on its own (hipStreamNonBlocking)
        self._check(self.lib.hipr_denoiser_process(self.handle, C.byref(settings), noisy.data_ptr(), noisy.shape[1], albedo.data_ptr(), albedo.shape[1], width, height,
                                                   int(update_filtered), show, out.data_ptr(), out.shape[1]), "hipr_denoiser_process")
        self._check(self.lib.hipr_denoiser_synchronize(self.handle), "hipr_denoiser_synchronize")
        return out
