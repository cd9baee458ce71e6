"""ctypes bindings of the CPU oracle (oracle/_build/liboracle.so). Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from bifrost3d_amd import capi

ROOT = Path(__file__).resolve().parent.parent
ORACLE_LIB = ROOT / "oracle" / "_build" / "liboracle.so"

MODEL_OREN_NAYAR, MODEL_GGX_R, MODEL_GGX_T, MODEL_GGX, MODEL_DEFAULT, MODEL_TRANSMISSIVE, MODEL_DIFFUSE = range(7)

_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def uptr(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_up)


class Oracle:
    def __init__(self, quantize_tables: bool = False):
        if not ORACLE_LIB.exists():
            subprocess.check_call(["make", "-C", str(ROOT / "oracle")])
        self.lib = lib = C.CDLL(str(ORACLE_LIB))
        f, u32, i = C.c_float, C.c_uint32, C.c_int
        lib.oracle_set_tables.argtypes = [_fp] * 5 + [i]
        lib.oracle_pcg2d.argtypes = [u32, u32, _up]
        lib.oracle_sobol4ui.argtypes = [_up, u32, _up]
        lib.oracle_sobol4f.argtypes = [u32, u32, u32, _fp]
        lib.oracle_sample_offsets.argtypes = [_fp, i]
        lib.oracle_sample02.argtypes = [u32, _fp]
        lib.oracle_reverse_bits.argtypes = [u32]; lib.oracle_reverse_bits.restype = u32
        lib.oracle_jenkins_hash.argtypes = [u32]; lib.oracle_jenkins_hash.restype = u32
        lib.oracle_bsdf_sample.argtypes = [i, _fp, _fp, _fp, i, _fp]
        lib.oracle_bsdf_eval.argtypes = [i, _fp, _fp, _fp, i, i, _fp]
        lib.oracle_default_shading_info.argtypes = [_fp, f, _fp]
        lib.oracle_transmissive_rho.argtypes = [_fp, f, _fp]
        lib.oracle_thin_sheet.argtypes = [f, f, f, _fp, _fp]
        lib.oracle_specular_rho.argtypes = [f, f, _fp]
        lib.oracle_dielectric_rho.argtypes = [f, f, f, _fp]
        for name, n in (("oracle_estimate_alpha", 2), ("oracle_min_roughness_from_PDF", 2), ("oracle_dielectric_specularity", 2),
                        ("oracle_dielectric_ior_from_specularity", 1), ("oracle_adjust_dielectric_specularity", 2),
                        ("oracle_balance_heuristic", 2), ("oracle_power_heuristic", 2)):
            fn = getattr(lib, name); fn.argtypes = [f] * n; fn.restype = f
        lib.oracle_E_FON.argtypes = [f, f, i]; lib.oracle_E_FON.restype = f
        lib.oracle_conductor_specularity.argtypes = [_fp] * 4
        lib.oracle_conductor_ior_from_specularity.argtypes = [_fp] * 3
        lib.oracle_adjust_conductor_specularity.argtypes = [_fp] * 4
        lib.oracle_refract.argtypes = [_fp, _fp, f, _fp]
        lib.oracle_refract_z.argtypes = [_fp, f, _fp]
        lib.oracle_refract_cos.argtypes = [f, f, _fp]
        lib.oracle_uniform_hemisphere.argtypes = [_fp, _fp]
        lib.oracle_uniform_sphere.argtypes = [_fp, _fp]
        LP = C.POINTER(capi.HiprLight)
        lib.oracle_light_sample.argtypes = [LP, _fp, _fp, i, _fp]
        lib.oracle_light_pdf.argtypes = [LP, _fp, _fp]; lib.oracle_light_pdf.restype = f
        lib.oracle_light_evaluate.argtypes = [LP, _fp, _fp, _fp]
        lib.oracle_light_evaluate_intersection.argtypes = [LP, _fp, _fp, f, _fp]
        lib.oracle_decode_octahedral.argtypes = [C.POINTER(C.c_int16), i, _fp]
        lib.oracle_fix_backfacing_shading_normal.argtypes = [_fp, _fp, f, _fp]
        CP = C.POINTER(capi.HiprCameraState)
        SP = C.POINTER(capi.HiprSceneDesc)
        lib.oracle_generate_rays.argtypes = [CP, i, i, u32, _up, u32, _fp, _fp]
        lib.oracle_trace_closest.argtypes = [SP, _fp, _up, u32, i, i, _fp, C.POINTER(C.c_uint64)]
        lib.oracle_trace_shadow.argtypes = [SP, _fp, u32, i, _fp, C.POINTER(C.c_uint64)]
        lib.oracle_render.argtypes = [SP, C.POINTER(capi.HiprSceneState), CP, i, i, u32, i, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.oracle_render.restype = C.c_double
        lib.oracle_render_entry.argtypes = [SP, C.POINTER(capi.HiprSceneState), CP, i, i, u32, i, i, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.oracle_render_entry.restype = C.c_double
        lib.oracle_max_threads.restype = i
        lib.oracle_set_threads.argtypes = [i]
        lib.oracle_smallpt_accumulate.argtypes = [i, i, _fp, C.POINTER(i)]
        lib.oracle_smallpt_accumulate.restype = C.c_uint64
        lib.oracle_smallpt_threads.restype = i
        lib.oracle_pmjbn_samples.argtypes = [_fp, C.c_uint, C.c_uint]
        lib.oracle_integrate_thin_sheet.argtypes = [_fp, f, f, f, _fp, C.c_uint, C.c_uint, _fp]
        self.set_tables(quantize_tables)

    # ------------------------------------------------------------------ tables
    def set_tables(self, quantize: bool):
        self._tables = capi.load_tables()
        self.lib.oracle_set_tables(*[fptr(t) for t in self._tables], int(quantize))

    # ------------------------------------------------------------------ rng
    def pcg2d(self, x, y):
        out = np.zeros(2, np.uint32)
        self.lib.oracle_pcg2d(x, y, uptr(out))
        return out

    def sobol4ui(self, triples: np.ndarray) -> np.ndarray:
        triples = np.ascontiguousarray(triples, np.uint32).reshape(-1, 3)
        out = np.zeros((len(triples), 4), np.uint32)
        self.lib.oracle_sobol4ui(uptr(triples), len(triples), uptr(out))
        return out

    def sobol4f(self, accumulation, pixel_hash, dimension):
        out = np.zeros(4, np.float32)
        self.lib.oracle_sobol4f(accumulation, pixel_hash, dimension, fptr(out))
        return out

    def sample_offsets(self, n=256):
        out = np.zeros((n, 4), np.float32)
        self.lib.oracle_sample_offsets(fptr(out), n)
        return out

    def sample02(self, n):
        out = np.zeros(2, np.float32)
        self.lib.oracle_sample02(n, fptr(out))
        return out

    # ------------------------------------------------------------------ bsdf
    def bsdf_sample(self, model, params, wo, u):
        """wo: (3,) or (n,3); u: (n,3). Returns (n,7): reflectance, raw pdf, direction."""
        params = np.ascontiguousarray(params, np.float32)
        u = np.ascontiguousarray(u, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(np.broadcast_to(np.asarray(wo, np.float32), u.shape))
        out = np.zeros((len(u), 7), np.float32)
        self.lib.oracle_bsdf_sample(model, fptr(params), fptr(wo), fptr(u), len(u), fptr(out))
        return out

    def bsdf_eval(self, model, params, wo, wi, which=0):
        params = np.ascontiguousarray(params, np.float32)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(np.broadcast_to(np.asarray(wo, np.float32), wi.shape))
        out = np.zeros((len(wi), 4), np.float32)
        self.lib.oracle_bsdf_eval(model, fptr(params), fptr(wo), fptr(wi), len(wi), which, fptr(out))
        return out

    def default_shading_info(self, params, cos_theta):
        params = np.ascontiguousarray(params, np.float32)
        out = np.zeros(10, np.float32)
        self.lib.oracle_default_shading_info(fptr(params), cos_theta, fptr(out))
        return dict(rho=out[0:3], diffuse_probability=out[3], specular_probability=out[4], coat_probability=out[5],
                    roughness=out[6], specularity=out[7:10])

    def vec3_call(self, name, *args):
        out = np.zeros(3, np.float32)
        cargs = [fptr(np.ascontiguousarray(a, np.float32)) if isinstance(a, (list, tuple, np.ndarray)) else a for a in args]
        getattr(self.lib, name)(*cargs, fptr(out))
        return out

    # ------------------------------------------------------------------ lights
    def light_sample(self, light, position, u):
        u = np.ascontiguousarray(u, np.float32).reshape(-1, 2)
        pos = np.ascontiguousarray(position, np.float32)
        out = np.zeros((len(u), 8), np.float32)
        self.lib.oracle_light_sample(C.byref(light), fptr(pos), fptr(u), len(u), fptr(out))
        return out

    def light_pdf(self, light, position, direction):
        return self.lib.oracle_light_pdf(C.byref(light), fptr(np.ascontiguousarray(position, np.float32)), fptr(np.ascontiguousarray(direction, np.float32)))

    def light_evaluate(self, light, position, direction):
        out = np.zeros(3, np.float32)
        self.lib.oracle_light_evaluate(C.byref(light), fptr(np.ascontiguousarray(position, np.float32)), fptr(np.ascontiguousarray(direction, np.float32)), fptr(out))
        return out

    # ------------------------------------------------------------------ integrator
    def generate_rays(self, cam, width, height, accumulation, pixels_xy):
        pixels_xy = np.ascontiguousarray(pixels_xy, np.uint32).reshape(-1, 2)
        n = len(pixels_xy)
        o = np.zeros((n, 4), np.float32)
        d = np.zeros((n, 4), np.float32)
        self.lib.oracle_generate_rays(C.byref(cam), width, height, accumulation, uptr(pixels_xy), n, fptr(o), fptr(d))
        return o, d

    def trace_closest(self, scene, rays, skip=None, use_bvh=True, with_lights=True):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = len(rays)
        hits = np.zeros((n, 4), np.float32)
        counters = (C.c_uint64 * 2)()
        sk = uptr(np.ascontiguousarray(skip, np.uint32)) if skip is not None else None
        self.lib.oracle_trace_closest(C.byref(scene), fptr(rays), sk, n, int(use_bvh), int(with_lights), fptr(hits), counters)
        return hits, (counters[0], counters[1])

    def trace_shadow(self, scene, rays, use_bvh=True):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.float32)
        counters = (C.c_uint64 * 2)()
        self.lib.oracle_trace_shadow(C.byref(scene), fptr(rays), len(rays), int(use_bvh), fptr(out), counters)
        return out, (counters[0], counters[1])

    def render(self, scene, state, cam, width, height, accumulation_count, use_bvh=True, accum=None, entry=0):
        if accum is None:
            accum = np.zeros((height, width, 4), np.float64)
        counters = (C.c_uint64 * 9)()
        seconds = self.lib.oracle_render_entry(C.byref(scene), C.byref(state), C.byref(cam), width, height, accumulation_count, int(use_bvh), entry,
                                               accum.ctypes.data_as(C.POINTER(C.c_double)), counters)
        names = [f[0] for f in capi.HiprCounters._fields_]
        result = dict(zip(names, list(counters)))
        result["rejected_hits"] = result.pop("iterations")      # the oracle has no launch iterations; its ninth counter is the hits its hit program refused
        result["iterations"] = 0
        return accum, result, seconds

    def set_backface_culling(self, enable: bool):
        """The oracle's side of hipr_set_backface_culling: its 8-wide search (use_bvh=3) steps over closest hits on the back of one-sided triangles."""
        self.lib.oracle_set_backface_culling(int(enable))

    def pmjbn(self, count=16384, candidates=8):
        out = np.zeros((count, 2), np.float32)
        self.lib.oracle_pmjbn_samples(fptr(out), count, candidates)
        return out

    def smallpt(self, width, height, accumulations):
        buf = np.zeros((height, width, 3), np.float32)
        acc = C.c_int(0)
        rays = 0
        for _ in range(accumulations):
            rays += self.lib.oracle_smallpt_accumulate(width, height, fptr(buf), C.byref(acc))
        return buf, rays


_oracles = {}


def get_oracle(quantize_tables: bool = False) -> Oracle:
    """One handle per table mode; the library has one global table set, so re-apply the mode on each fetch."""
    key = bool(quantize_tables)
    if key not in _oracles:
        _oracles[key] = Oracle(quantize_tables)
    _oracles[key].set_tables(key)
    return _oracles[key]
