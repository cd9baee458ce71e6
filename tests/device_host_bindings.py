"""ctypes handle on tests/native/libdevice_shade_host.so: the device code of the shade stage (csrc/shade_kernel.h shade_path and everything below it) compiled for the
host, with the verification build's arithmetic (tests/native/DeviceShadeHost.hip). Test infrastructure: built by bifrost3d_amd/Makefile, loaded by tests only."""
import ctypes as C
from pathlib import Path

import numpy as np

from bifrost3d_amd import capi

LIB_PATH = Path(__file__).resolve().parent / "native" / "libdevice_shade_host.so"
_fp, _up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
_lib = None


def library():
    global _lib
    if _lib is None:
        lib = C.CDLL(str(LIB_PATH))
        lib.dsh_scene_create.restype = C.c_void_p
        lib.dsh_scene_create.argtypes = [C.POINTER(capi.HiprSceneDesc), C.POINTER(capi.HiprSceneState)] + [_fp] * 5
        lib.dsh_scene_destroy.argtypes = [C.c_void_p]
        lib.dsh_shade.argtypes = [C.c_void_p, C.POINTER(capi.HiprCameraState), C.c_uint32, _fp, _fp, _fp, _up, _up, _up, C.c_int, C.c_int, _fp]
        lib.dsh_shading.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]
        lib.dsh_light.argtypes = [C.POINTER(capi.HiprLight), _fp, _fp, C.c_int, C.c_int, _fp]
        _lib = lib
    return _lib


class DeviceShadeOnHost:
    """The scene as the uploads leave it on the device, and shade_path over arrays of queue entries."""

    def __init__(self, scene):
        self.lib = library()
        self._scene = scene      # keeps the host arrays the description points into alive
        self._tables = capi.load_tables()
        desc, state = scene.desc, scene.state
        self.handle = self.lib.dsh_scene_create(C.byref(desc), C.byref(state), *[t.ctypes.data_as(_fp) for t in self._tables])

    def close(self):
        if self.handle:
            self.lib.dsh_scene_destroy(self.handle)
            self.handle = None

    def shade(self, cam, rays, throughput_bounces, hits, last_triangle, pixel_hash, accumulation, models=7, textures=2):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = len(rays)
        throughput_bounces = np.ascontiguousarray(throughput_bounces, np.float32).reshape(n, 4)
        hits = np.ascontiguousarray(hits, np.float32).reshape(n, 4)
        words = [np.ascontiguousarray(a, np.uint32).reshape(n) for a in (last_triangle, pixel_hash, accumulation)]
        out = np.zeros((n, 32), np.float32)
        status = self.lib.dsh_shade(self.handle, C.byref(cam), n, rays.ctypes.data_as(_fp), throughput_bounces.ctypes.data_as(_fp), hits.ctypes.data_as(_fp),
                                    *[a.ctypes.data_as(_up) for a in words], int(models), int(textures), out.ctypes.data_as(_fp))
        if status != 0:
            raise ValueError(f"dsh_shade: the instantiation <MODELS {models}, TEXTURES {textures}> is not built")
        return out


    def shading(self, model, params10, wo, inputs, mode=0, terms=True):
        """hipr_debug_shading's kernel on the host: mode 0 sample(wo, u) -> (n, 7) f, pdf, direction; mode 1 evaluate_with_PDF(wo, wi) -> (n, 7) f, pdf, 0, 0, 0."""
        inputs = np.ascontiguousarray(inputs, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(np.broadcast_to(np.asarray(wo, np.float32), inputs.shape))
        params = np.ascontiguousarray(params10, np.float32)
        out = np.zeros((len(inputs), 7), np.float32)
        self.lib.dsh_shading(self.handle, int(model), params.ctypes.data_as(_fp), wo.ctypes.data_as(_fp), inputs.ctypes.data_as(_fp), len(inputs), int(mode), int(bool(terms)), out.ctypes.data_as(_fp))
        return out

    def light(self, light, position, inputs, mode=0):
        """hipr_debug_light's kernel on the host: mode 0 sample_radiance -> (n, 8) radiance, PDF, direction, distance; mode 1 (spot) evaluate + pdf."""
        inputs = np.ascontiguousarray(inputs, np.float32)
        if inputs.ndim == 2 and inputs.shape[1] == 2:
            inputs = np.concatenate([inputs, np.zeros((len(inputs), 1), np.float32)], axis=1)
        inputs = np.ascontiguousarray(inputs.reshape(-1, 3))
        position = np.ascontiguousarray(position, np.float32)
        out = np.zeros((len(inputs), 8), np.float32)
        self.lib.dsh_light(C.byref(light), position.ctypes.data_as(_fp), inputs.ctypes.data_as(_fp), len(inputs), int(mode), out.ctypes.data_as(_fp))
        return out


RECORD_WORDS = {0: "flags", 1: "radiance.x", 2: "radiance.y", 3: "radiance.z", 4: "origin.x", 5: "origin.y", 6: "origin.z", 7: "tmin", 8: "direction.x", 9: "direction.y",
                10: "direction.z", 11: "bsdf_pdf", 12: "throughput.x", 13: "throughput.y", 14: "throughput.z", 15: "bounces", 16: "last_triangle", 17: "shadow_origin.x",
                18: "shadow_origin.y", 19: "shadow_origin.z", 20: "shadow_tmax", 21: "light_direction.x", 22: "light_direction.y", 23: "light_direction.z",
                24: "shadow_radiance.x", 25: "shadow_radiance.y", 26: "shadow_radiance.z"}


def oracle_shade(oracle, scene, cam, rays, throughput_bounces, hits, last_triangle, pixel_hash, accumulation):
    """oracle_debug_shade: the oracle's hit programs for the same entries, same record."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    n = len(rays)
    throughput_bounces = np.ascontiguousarray(throughput_bounces, np.float32).reshape(n, 4)
    hits = np.ascontiguousarray(hits, np.float32).reshape(n, 4)
    words = [np.ascontiguousarray(a, np.uint32).reshape(n) for a in (last_triangle, pixel_hash, accumulation)]
    out = np.zeros((n, 32), np.float32)
    desc, state = scene.desc, scene.state
    oracle.lib.oracle_debug_shade(C.byref(desc), C.byref(state), C.byref(cam), n, rays.ctypes.data_as(_fp), throughput_bounces.ctypes.data_as(_fp), hits.ctypes.data_as(_fp),
                                  *[a.ctypes.data_as(_up) for a in words], out.ctypes.data_as(_fp))
    return out


def camera_paths(oracle, cam, width, height, accumulation):
    """The queue entries of the camera rays of one accumulation: rays (n, 8), throughput + bounces (n, 4), last triangle, pixel hash, accumulation."""
    ys, xs = np.mgrid[0:height, 0:width]
    pixels = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.uint32)
    o, d = oracle.generate_rays(cam, width, height, accumulation, pixels)
    n = len(pixels)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o[:, 0:3]; rays[:, 3] = 0.0; rays[:, 4:7] = d[:, 0:3]; rays[:, 7] = -1.0      # tmin 0, bsdf_PDF = delta_dirac(1)
    throughput = np.zeros((n, 4), np.float32); throughput[:, 0:3] = 1.0      # bounces 0 = bits 0
    hashes = np.array([oracle.pcg2d(int(x), int(y))[0] for x, y in pixels], np.uint32)
    return rays, throughput, np.full(n, 0xFFFFFFFF, np.uint32), hashes, np.full(n, accumulation, np.uint32)
