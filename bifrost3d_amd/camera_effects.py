"""ctypes plumbing for the camera effects C-ABI (include/hipr_camera_effects_c.h) -- tests and benchmarks only.

Device memory is held by torch tensors (plumbing, as everywhere in this package); the effects themselves are the HIP
kernels of csrc/camera_effects.hip behind `hipr_camera_effects_*`.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import capi

EXPOSURE_FIXED, EXPOSURE_LOG_AVERAGE, EXPOSURE_HISTOGRAM = 0, 1, 2
TONEMAPPING_LINEAR, TONEMAPPING_FILMIC, TONEMAPPING_AGX, TONEMAPPING_KHRONOS_NEUTRAL = 0, 1, 2, 3
TARGET_RGBA16F, TARGET_RGBA32F, TARGET_RGBA8_SRGB = 0, 1, 2
HISTOGRAM_BINS = 64


class Settings(C.Structure):
    """HiprCameraEffectsSettings; preset() and linear() are Bifrost::Math::CameraEffects::Settings::preset / linear (BF/Math/CameraEffects.h:65-118)."""
    _fields_ = [("exposure_mode", C.c_int32), ("min_log_luminance", C.c_float), ("max_log_luminance", C.c_float),
                ("min_histogram_percentage", C.c_float), ("max_histogram_percentage", C.c_float), ("log_luminance_bias", C.c_float),
                ("eye_adaptation_enabled", C.c_int32), ("eye_adaptation_brightness", C.c_float), ("eye_adaptation_darkness", C.c_float),
                ("bloom_threshold", C.c_float), ("bloom_support", C.c_float), ("vignette", C.c_float),
                ("tonemapping_mode", C.c_int32), ("tonemapping_black_clip", C.c_float), ("tonemapping_toe", C.c_float), ("tonemapping_slope", C.c_float),
                ("tonemapping_shoulder", C.c_float), ("tonemapping_white_clip", C.c_float), ("film_grain", C.c_float)]

    TONEMAPPING_PRESETS = {"ACES": (0.0, 0.53, 0.91, 0.23, 0.035), "uncharted2": (0.0, 0.55, 0.63, 0.47, 0.01), "HP": (0.0, 0.63, 0.65, 0.45, 0.0),
                           "legacy": (0.0, 0.3, 0.98, 0.22, 0.025)}     # black_clip, toe, slope, shoulder, white_clip (BF/Math/CameraEffects.h:27-30)

    def set_tonemapping(self, name: str) -> "Settings":
        (self.tonemapping_black_clip, self.tonemapping_toe, self.tonemapping_slope, self.tonemapping_shoulder, self.tonemapping_white_clip) = self.TONEMAPPING_PRESETS[name]
        return self

    @classmethod
    def preset(cls) -> "Settings":
        s = cls(exposure_mode=EXPOSURE_HISTOGRAM, min_log_luminance=-4, max_log_luminance=4, min_histogram_percentage=0.7, max_histogram_percentage=0.95,
                log_luminance_bias=0, eye_adaptation_enabled=1, eye_adaptation_brightness=3.0, eye_adaptation_darkness=1.0,
                bloom_threshold=math.inf, bloom_support=0.05, vignette=0.63, tonemapping_mode=TONEMAPPING_FILMIC, film_grain=1 / 255.0)
        return s.set_tonemapping("ACES")

    @classmethod
    def linear(cls) -> "Settings":
        s = cls(exposure_mode=EXPOSURE_FIXED, min_log_luminance=-4, max_log_luminance=4, min_histogram_percentage=0.7, max_histogram_percentage=0.95,
                log_luminance_bias=0, eye_adaptation_enabled=0, eye_adaptation_brightness=math.inf, eye_adaptation_darkness=math.inf,
                bloom_threshold=math.inf, bloom_support=0.0, vignette=0.0, tonemapping_mode=TONEMAPPING_LINEAR, film_grain=0.0)
        return s.set_tonemapping("ACES")


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("pitch", C.c_uint32), ("rows", C.c_uint32), ("viewport", Rect)]


class Times(C.Structure):
    _fields_ = [("exposure_ms", C.c_float), ("bloom_horizontal_ms", C.c_float), ("bloom_vertical_ms", C.c_float), ("tonemap_ms", C.c_float),
                ("exposure_launches", C.c_uint32), ("bloom_horizontal_launches", C.c_uint32), ("bloom_vertical_launches", C.c_uint32), ("tonemap_launches", C.c_uint32)]


C_ABI_SYMBOLS = [
    "hipr_camera_effects_create", "hipr_camera_effects_destroy", "hipr_camera_effects_last_error", "hipr_camera_effects_set_stream", "hipr_camera_effects_synchronize",
    "hipr_camera_effects_process", "hipr_camera_effects_get_linear_exposure", "hipr_camera_effects_set_linear_exposure", "hipr_camera_effects_reduce_histogram",
    "hipr_camera_effects_exposure_from_histogram", "hipr_camera_effects_log_average", "hipr_camera_effects_exposure_from_log_average", "hipr_camera_effects_bloom", "hipr_camera_effects_dual_kawase_bloom",
    "hipr_camera_effects_set_instrumentation", "hipr_camera_effects_reset_timers", "hipr_camera_effects_get_times",
]


def declare(lib) -> None:
    vp, SP, FP = C.c_void_p, C.POINTER(Settings), C.POINTER(FrameView)
    lib.hipr_camera_effects_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.hipr_camera_effects_destroy.argtypes = [vp]; lib.hipr_camera_effects_destroy.restype = None
    lib.hipr_camera_effects_last_error.argtypes = [vp]; lib.hipr_camera_effects_last_error.restype = C.c_char_p
    lib.hipr_camera_effects_set_stream.argtypes = [vp, vp]
    lib.hipr_camera_effects_synchronize.argtypes = [vp]
    lib.hipr_camera_effects_process.argtypes = [vp, SP, C.c_float, FP, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
    lib.hipr_camera_effects_get_linear_exposure.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hipr_camera_effects_set_linear_exposure.argtypes = [vp, C.c_float]
    lib.hipr_camera_effects_reduce_histogram.argtypes = [vp, SP, FP, C.POINTER(C.c_uint32)]
    lib.hipr_camera_effects_exposure_from_histogram.argtypes = [vp, SP, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    lib.hipr_camera_effects_log_average.argtypes = [vp, FP, C.POINTER(C.c_float)]
    lib.hipr_camera_effects_exposure_from_log_average.argtypes = [vp, SP, C.c_float, FP, C.POINTER(C.c_float)]
    lib.hipr_camera_effects_bloom.argtypes = [vp, C.c_float, C.c_int32, FP, vp]
    lib.hipr_camera_effects_dual_kawase_bloom.argtypes = [vp, C.c_float, C.c_uint32, FP, vp]
    lib.hipr_camera_effects_set_instrumentation.argtypes = [vp, C.c_int]
    lib.hipr_camera_effects_reset_timers.argtypes = [vp]
    lib.hipr_camera_effects_get_times.argtypes = [vp, C.POINTER(Times)]


def frame_view(pixels_ptr: int, pitch: int, rows: int, viewport=None) -> FrameView:
    x, y, w, h = viewport if viewport is not None else (0, 0, pitch, rows)
    return FrameView(pixels_ptr, pitch, rows, Rect(x, y, w, h))


class CameraEffects:
    """One HiprCameraEffects object. Frames are torch CUDA tensors of shape (rows, pitch, 4), dtype float16."""

    def __init__(self, device_index: int = 0):
        self.lib = capi.load_library()
        declare(self.lib)
        self.handle = C.c_void_p()
        status = self.lib.hipr_camera_effects_create(device_index, C.byref(self.handle))
        if status != 0:
            raise capi.HiprError(f"hipr_camera_effects_create failed with {status}")
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        # The effects run on torch's current stream (hipr_camera_effects_set_stream): what torch queued before a call -- an upload, the fill of a fresh target -- and what it
        # queues after it are then ordered with the effects' kernels by the stream itself, and no call blocks the host (the hand-over used to be two
        # blocking synchronisations per processed frame). Should the stream not be taken, the blocking hand-over stays.
        # The stream is followed, not captured: a caller that later works under `with torch.cuda.stream(s)` has the effects moved to `s` at its next call
        # (_follow_torch_stream), so the ordering argument above holds for whatever stream is current then.
        self._stream_in_use = None
        self.shares_torch_stream = False
        self._follow_torch_stream()

    def _follow_torch_stream(self):
        """Puts the effects on torch's CURRENT stream if they are not on it already. Returns with shares_torch_stream saying whether they are (False: the blocking hand-over)."""
        current = self.torch.cuda.current_stream(self.device).cuda_stream
        if current != self._stream_in_use:
            if self._stream_in_use is not None:
                self.lib.hipr_camera_effects_synchronize(self.handle)      # what the effects still run on the stream they leave is not ordered with the new one
            self.shares_torch_stream = self.lib.hipr_camera_effects_set_stream(self.handle, C.c_void_p(current)) == 0
            self._stream_in_use = current if self.shares_torch_stream else None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hipr_camera_effects_destroy(self.handle)
            self.handle = None

    __del__ = close

    def _check(self, status: int, what: str):
        if status != 0:
            message = self.lib.hipr_camera_effects_last_error(self.handle)
            raise capi.HiprError(f"{what} failed with {status}: {message.decode() if message else ''}")

    def upload(self, half_pixels: np.ndarray):
        assert half_pixels.dtype == np.float16 and half_pixels.ndim == 3 and half_pixels.shape[2] == 4
        frame = self.torch.from_numpy(np.ascontiguousarray(half_pixels)).to(self.device)
        self._torch_stream_done()
        return frame

    def _torch_stream_done(self):
        """The effects run on a stream of their own (hipStreamNonBlocking): what torch queued on ITS stream -- an upload, the fill of a fresh target -- has to be
        complete before a kernel of that stream reads or overwrites the memory. With the effects on torch's current stream (the normal case) the stream orders them."""
        self._follow_torch_stream()
        if not self.shares_torch_stream:
            self.torch.cuda.current_stream(self.device).synchronize()

    def view(self, frame, viewport=None) -> FrameView:
        rows, pitch = frame.shape[0], frame.shape[1]
        return frame_view(frame.data_ptr(), pitch, rows, viewport)

    def reduce_histogram(self, settings: Settings, frame, viewport=None) -> np.ndarray:
        out = np.zeros(HISTOGRAM_BINS, dtype=np.uint32)
        self._torch_stream_done()
        view = self.view(frame, viewport)
        self._check(self.lib.hipr_camera_effects_reduce_histogram(self.handle, C.byref(settings), C.byref(view), out.ctypes.data_as(C.POINTER(C.c_uint32))), "reduce_histogram")
        return out

    def exposure_from_histogram(self, settings: Settings, delta_time: float, histogram: np.ndarray, current_exposure: float = 0.0) -> float:
        histogram = np.ascontiguousarray(histogram, dtype=np.uint32)
        exposure = C.c_float(current_exposure)
        self._check(self.lib.hipr_camera_effects_exposure_from_histogram(self.handle, C.byref(settings), delta_time, histogram.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(exposure)),
                    "exposure_from_histogram")
        return exposure.value

    def log_average(self, frame, viewport=None) -> float:
        out = C.c_float()
        self._torch_stream_done()
        view = self.view(frame, viewport)
        self._check(self.lib.hipr_camera_effects_log_average(self.handle, C.byref(view), C.byref(out)), "log_average")
        return out.value

    def exposure_from_log_average(self, settings: Settings, delta_time: float, frame, current_exposure: float = 0.0, viewport=None) -> float:
        exposure = C.c_float(current_exposure)
        self._torch_stream_done()
        view = self.view(frame, viewport)
        self._check(self.lib.hipr_camera_effects_exposure_from_log_average(self.handle, C.byref(settings), delta_time, C.byref(view), C.byref(exposure)), "exposure_from_log_average")
        return exposure.value

    def bloom(self, threshold: float, support: int, frame, viewport=None) -> np.ndarray:
        view = self.view(frame, viewport)
        out = self.torch.empty((view.viewport.height, view.viewport.width, 4), dtype=self.torch.float16, device=self.device)
        self._torch_stream_done()
        self._check(self.lib.hipr_camera_effects_bloom(self.handle, threshold, support, C.byref(view), out.data_ptr()), "bloom")
        self._check(self.lib.hipr_camera_effects_synchronize(self.handle), "synchronize")
        return out.cpu().numpy()

    def dual_kawase_bloom(self, threshold: float, half_passes: int, frame, viewport=None) -> np.ndarray:
        view = self.view(frame, viewport)
        out = self.torch.empty((view.viewport.height, view.viewport.width, 4), dtype=self.torch.float16, device=self.device)
        self._torch_stream_done()
        self._check(self.lib.hipr_camera_effects_dual_kawase_bloom(self.handle, threshold, half_passes, C.byref(view), out.data_ptr()), "dual_kawase_bloom")
        self._check(self.lib.hipr_camera_effects_synchronize(self.handle), "synchronize")
        return out.cpu().numpy()

    def process(self, settings: Settings, delta_time: float, frame, viewport=None, target_format: int = TARGET_RGBA32F, target=None, target_offset=(0, 0)):
        """Returns the target as a torch tensor ((rows, pitch, 4); float32, float16 or uint8 by format)."""
        view = self.view(frame, viewport)
        dtype = {TARGET_RGBA16F: self.torch.float16, TARGET_RGBA32F: self.torch.float32, TARGET_RGBA8_SRGB: self.torch.uint8}[target_format]
        if target is None:
            target = self.torch.zeros((view.viewport.height + target_offset[1], view.viewport.width + target_offset[0], 4), dtype=dtype, device=self.device)
        self._torch_stream_done()       # the zero fill above must not land on top of the result
        self._check(self.lib.hipr_camera_effects_process(self.handle, C.byref(settings), delta_time, C.byref(view), target.data_ptr(), target_format,
                                                         target.shape[1], target.shape[0], target_offset[0], target_offset[1]), "process")
        if not self.shares_torch_stream:
            self._check(self.lib.hipr_camera_effects_synchronize(self.handle), "synchronize")       # the caller goes on with the tensor on torch's stream
        return target

    def synchronize(self):
        self._check(self.lib.hipr_camera_effects_synchronize(self.handle), "synchronize")

    @property
    def linear_exposure(self) -> float:
        out = C.c_float()
        self._check(self.lib.hipr_camera_effects_get_linear_exposure(self.handle, C.byref(out)), "get_linear_exposure")
        return out.value

    @linear_exposure.setter
    def linear_exposure(self, value: float):
        self._check(self.lib.hipr_camera_effects_set_linear_exposure(self.handle, value), "set_linear_exposure")

    def set_instrumentation(self, on: bool):
        self._check(self.lib.hipr_camera_effects_set_instrumentation(self.handle, int(on)), "set_instrumentation")

    def reset_timers(self):
        self._check(self.lib.hipr_camera_effects_reset_timers(self.handle), "reset_timers")

    def times(self) -> Times:
        t = Times()
        self._check(self.lib.hipr_camera_effects_get_times(self.handle, C.byref(t)), "get_times")
        return t
