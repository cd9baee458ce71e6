"""ctypes view of oracle/_ref/libbifrost_ref.so: the slice of the REAL reference that builds here (oracle/Makefile `_ref`,
oracle/ref/reference_api.cpp). Test infrastructure only; absent where neither /root/reference nor a prebuilt library is."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

REF_LIB_PATH = Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "libbifrost_ref.so"
_fp = C.POINTER(C.c_float)
_sp = C.POINTER(C.c_int16)
_lib = None


def fptr(a):
    return a.ctypes.data_as(_fp)


def available() -> bool:
    return REF_LIB_PATH.exists()


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(REF_LIB_PATH))
        f, i = C.c_float, C.c_int
        _lib.ref_table.argtypes = [i, C.POINTER(_fp), C.POINTER(i)]
        _lib.ref_dielectric_ior_ranges.argtypes = [_fp]
        for name, n in (("ref_sample_GGX", 2), ("ref_sample_GGX_with_fresnel", 2), ("ref_estimate_alpha", 2), ("ref_encode_PDF", 1), ("ref_sRGB_to_linear", 1),
                        ("ref_linear_to_sRGB", 1)):
            fn = getattr(_lib, name); fn.argtypes = [f] * n; fn.restype = f
        _lib.ref_sample_dielectric_GGX.argtypes = [f, f, f, _fp]
        _lib.ref_tonemap.argtypes = [i, _fp, _fp, i, _fp]
        _lib.ref_gaussian_taps.argtypes = [f, i, _fp, _fp]
        _lib.ref_octahedral_encode_precise.argtypes = [_fp, i, _sp]
        _lib.ref_octahedral_decode.argtypes = [_sp, i, _fp]
        _lib.ref_perspective_projection.argtypes = [f, f, f, f, _fp, _fp]
        _lib.ref_orthographic_projection.argtypes = [f, f, f, _fp, _fp]
        _lib.ref_rays_from_viewport_points.argtypes = [_fp, _fp, f, f, f, f, _fp, i, _fp]
        _lib.ref_reverse_bits.argtypes = [C.c_uint32]; _lib.ref_reverse_bits.restype = C.c_uint32
        _lib.ref_jenkins_hash.argtypes = [C.c_uint32]; _lib.ref_jenkins_hash.restype = C.c_uint32
        _lib.ref_sample02.argtypes = [C.c_uint32, _fp]
        _lib.ref_power_heuristic.argtypes = [f, f]; _lib.ref_power_heuristic.restype = f
        _lib.ref_sample2D.argtypes = [i, i, i, i, C.c_void_p, i, i, i, i, i, _fp, i, _fp]
        _lib.ref_infinite_area_light.argtypes = [i, i, _fp, _fp, i, _fp, _fp, C.POINTER(i), _fp, i]
    return _lib


def table(which: int) -> np.ndarray:
    data, count = _fp(), C.c_int()
    assert lib().ref_table(which, C.byref(data), C.byref(count)) == 0
    return np.ctypeslib.as_array(data, shape=(count.value,)).copy()


def tonemap(mode: int, settings5, rgb: np.ndarray) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, np.float32)
    s = np.asarray(settings5, np.float32)
    out = np.empty_like(rgb)
    lib().ref_tonemap(mode, fptr(s), fptr(rgb), len(rgb), fptr(out))
    return out


def gaussian_taps(std_dev: float, count: int):
    offsets, weights = np.empty(count, np.float32), np.empty(count, np.float32)
    lib().ref_gaussian_taps(std_dev, count, fptr(offsets), fptr(weights))
    return offsets, weights


def octahedral_encode(normals: np.ndarray) -> np.ndarray:
    normals = np.ascontiguousarray(normals, np.float32)
    out = np.empty((len(normals), 2), np.int16)
    lib().ref_octahedral_encode_precise(fptr(normals), len(normals), out.ctypes.data_as(_sp))
    return out


def octahedral_decode(encoded: np.ndarray) -> np.ndarray:
    encoded = np.ascontiguousarray(encoded, np.int16)
    out = np.empty((len(encoded), 3), np.float32)
    lib().ref_octahedral_decode(encoded.ctypes.data_as(_sp), len(encoded), fptr(out))
    return out


def perspective(near, far, fov, aspect):
    p, ip = np.empty(16, np.float32), np.empty(16, np.float32)
    lib().ref_perspective_projection(near, far, fov, aspect, fptr(p), fptr(ip))
    return p.reshape(4, 4), ip.reshape(4, 4)


def orthographic(width, height, depth):
    p, ip = np.empty(16, np.float32), np.empty(16, np.float32)
    lib().ref_orthographic_projection(width, height, depth, fptr(p), fptr(ip))
    return p.reshape(4, 4), ip.reshape(4, 4)


def rays(position, rotation, near, far, fov, aspect, viewport_points: np.ndarray) -> np.ndarray:
    position, rotation = np.asarray(position, np.float32), np.asarray(rotation, np.float32)
    points = np.ascontiguousarray(viewport_points, np.float32)
    out = np.empty((len(points), 6), np.float32)
    lib().ref_rays_from_viewport_points(fptr(position), fptr(rotation), near, far, fov, aspect, fptr(points), len(points), fptr(out))
    return out


def sample2D(fn, pixel_format: int, is_sRGB: bool, pixels: np.ndarray, magnification: int, minification: int, wrap_U: int, wrap_V: int, uv: np.ndarray) -> np.ndarray:
    """fn: ref_sample2D or the host library's hiprh_sample2D. pixels: [height, width(, channels)] u8 or f32."""
    pixels, uv = np.ascontiguousarray(pixels), np.ascontiguousarray(uv, np.float32)
    out = np.empty((len(uv), 4), np.float32)
    fn(pixel_format, int(is_sRGB), pixels.shape[1], pixels.shape[0], pixels.ctypes.data_as(C.c_void_p), pixels.nbytes, magnification, minification, wrap_U, wrap_V, fptr(uv),
       len(uv), fptr(out))
    return out


def infinite_area_light(fn, rgba: np.ndarray, u: np.ndarray):
    """fn: ref_infinite_area_light or the host library's hiprh_infinite_area_light (same argument list)."""
    rgba, u = np.ascontiguousarray(rgba, np.float32), np.ascontiguousarray(u, np.float32)
    height, width = rgba.shape[:2]
    samples, pdfs, size = np.empty((len(u), 8), np.float32), np.empty(len(u), np.float32), (C.c_int * 2)()
    capacity = max(width, 1) * max(height, 128) * 4
    per_pixel = np.zeros(capacity, np.float32)
    assert fn(width, height, fptr(rgba), fptr(u), len(u), fptr(samples), fptr(pdfs), size, fptr(per_pixel), capacity) == 0
    return samples, pdfs, (size[0], size[1]), per_pixel[:size[0] * size[1]].reshape(size[1], size[0])
