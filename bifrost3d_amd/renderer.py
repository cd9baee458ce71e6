"""Thin Python handle on a HiprContext (include/hiprenderer_c.h). Plumbing only; fails loudly without the HIP library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class Context:
    def __init__(self, device_id: int = 0, library=None, arithmetic=None):
        """`library`: another build of libhiprenderer.so (A/B builds of the kernels); the product library by default.
        `arithmetic`: "fast" / "exact" (or capi.HIPR_ARITHMETIC_*): hipr_set_arithmetic; None keeps the library's default (fast, or what HIPR_ARITHMETIC says)."""
        self.lib = capi.load_library(library)
        self.device_id = device_id
        self.handle = C.c_void_p()
        capi.check(self.lib, self.lib.hipr_create(device_id, C.byref(self.handle)), "hipr_create")
        if arithmetic is not None:
            self.set_arithmetic(arithmetic)
        self._tables = capi.load_tables()
        t = capi.HiprTables(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in self._tables])
        capi.check(self.lib, self.lib.hipr_upload_tables(self.handle, C.byref(t)), "hipr_upload_tables")
        self.frame = None

    def close(self):
        if self.handle:
            self.lib.hipr_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status, what):
        capi.check(self.lib, status, what)

    def set_arithmetic(self, arithmetic):
        mode = {"fast": capi.HIPR_ARITHMETIC_FAST, "exact": capi.HIPR_ARITHMETIC_EXACT}.get(arithmetic, arithmetic)
        self._check(self.lib.hipr_set_arithmetic(self.handle, int(mode)), "hipr_set_arithmetic")

    @property
    def arithmetic(self) -> str:
        mode = self.lib.hipr_get_arithmetic(self.handle)
        if mode < 0:
            self._check(mode, "hipr_get_arithmetic")
        return "exact" if mode == capi.HIPR_ARITHMETIC_EXACT else "fast"

    def debug_math(self, function: int, x, y=None):
        """sin (0), cos (1) or pow(x, y) (2) as the shade unit of the context's arithmetic mode evaluates them."""
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
        out = np.zeros_like(x)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.hipr_debug_math(self.handle, function, x.size, x.ctypes.data_as(fp), y.ctypes.data_as(fp), out.ctypes.data_as(fp)), "hipr_debug_math")
        return out

    def set_stream(self, stream_ptr: int | None):
        self._check(self.lib.hipr_set_stream(self.handle, C.c_void_p(stream_ptr or 0)), "hipr_set_stream")

    def upload_scene(self, scene):
        self._scene = scene   # keep the host arrays alive until uploaded (copy happens inside)
        self._check(self.lib.hipr_upload_scene(self.handle, C.byref(scene.desc)), "hipr_upload_scene")
        state = scene.state
        self._check(self.lib.hipr_set_scene_state(self.handle, C.byref(state)), "hipr_set_scene_state")

    def update_scene_geometry(self, scene):
        self._scene = scene
        self._check(self.lib.hipr_update_scene_geometry(self.handle, C.byref(scene.desc)), "hipr_update_scene_geometry")

    def set_scene_state(self, state: capi.HiprSceneState):
        self._check(self.lib.hipr_set_scene_state(self.handle, C.byref(state)), "hipr_set_scene_state")

    def set_frame(self, width, height, tile_phase=0, tile_stride=1, samples_per_pass=1):
        f = capi.HiprFrameDesc(width, height, tile_phase, tile_stride, samples_per_pass)
        self._check(self.lib.hipr_set_frame(self.handle, C.byref(f)), "hipr_set_frame")
        self.frame = f

    def set_entry_point(self, entry: int):
        self._check(self.lib.hipr_set_entry_point(self.handle, entry), "hipr_set_entry_point")

    def use_scratch_accumulation(self, on: bool):
        self._check(self.lib.hipr_use_scratch_accumulation(self.handle, int(on)), "hipr_use_scratch_accumulation")

    def owned_pixel_count(self) -> int:
        n = C.c_uint32()
        self._check(self.lib.hipr_owned_pixel_count(self.handle, C.byref(n)), "hipr_owned_pixel_count")
        return n.value

    def render_pass(self, camera: capi.HiprCameraState, out_ptr: int = 0, out_pitch: int = 0, synchronize: bool = False):
        self._check(self.lib.hipr_render_pass(self.handle, C.byref(camera), C.c_void_p(out_ptr), out_pitch, int(synchronize)), "hipr_render_pass")

    def set_samples_per_pass(self, samples: int):
        self._check(self.lib.hipr_set_samples_per_pass(self.handle, samples), "hipr_set_samples_per_pass")
        self.frame.samples_per_pass = samples

    def trace_pass(self, camera: capi.HiprCameraState):
        self._check(self.lib.hipr_trace_pass(self.handle, C.byref(camera)), "hipr_trace_pass")

    def accumulate_samples(self, first_sample: int, sample_count: int, first_accumulation: int, out_ptr: int = 0, out_pitch: int = 0, synchronize: bool = False):
        self._check(self.lib.hipr_accumulate_samples(self.handle, first_sample, sample_count, first_accumulation, C.c_void_p(out_ptr), out_pitch, int(synchronize)),
                    "hipr_accumulate_samples")

    def synchronize(self):
        self._check(self.lib.hipr_synchronize(self.handle), "hipr_synchronize")

    def read_accumulation(self) -> np.ndarray:
        f = self.frame
        if f.tile_stride == 1:
            out = np.zeros((f.height, f.width, 4), np.float64)
        else:
            out = np.zeros((self.owned_pixel_count(), 4), np.float64)
        self._check(self.lib.hipr_read_accumulation(self.handle, out.ctypes.data_as(C.POINTER(C.c_double)), out.size // 4), "hipr_read_accumulation")
        return out

    def counters(self) -> dict:
        c = capi.HiprCounters()
        self._check(self.lib.hipr_get_counters(self.handle, C.byref(c)), "hipr_get_counters")
        return {name: int(getattr(c, name)) for name, _ in capi.HiprCounters._fields_}

    def reset_counters(self):
        self._check(self.lib.hipr_reset_counters(self.handle), "hipr_reset_counters")

    def set_instrumentation(self, on: bool):
        self._check(self.lib.hipr_set_instrumentation(self.handle, int(on)), "hipr_set_instrumentation")

    def reset_timers(self):
        self._check(self.lib.hipr_reset_timers(self.handle), "hipr_reset_timers")

    def wavefront_count(self) -> int:
        n = C.c_int()
        self._check(self.lib.hipr_get_wavefront_count(self.handle, C.byref(n)), "hipr_get_wavefront_count")
        return n.value

    def set_wavefront_count(self, count: int):
        self._check(self.lib.hipr_set_wavefront_count(self.handle, count), "hipr_set_wavefront_count")

    def trace_variant(self) -> int:
        """capi.TRACE_BVH2 / TRACE_WIDE_PERSISTENT / TRACE_EXHAUSTIVE for the uploaded scene."""
        v = C.c_int(0)
        self._check(self.lib.hipr_get_trace_variant(self.handle, C.byref(v)), "hipr_get_trace_variant")
        return v.value

    def set_trace_variant(self, variant: int):
        """Forces a search for the scenes uploaded after the call (-1: by scene size)."""
        self._check(self.lib.hipr_set_trace_variant(self.handle, int(variant)), "hipr_set_trace_variant")

    def set_backface_culling(self, enable: bool):
        self._check(self.lib.hipr_set_backface_culling(self.handle, int(enable)), "hipr_set_backface_culling")

    def set_pass_pipelining(self, enable: bool):
        self._check(self.lib.hipr_set_pass_pipelining(self.handle, int(enable)), "hipr_set_pass_pipelining")

    def trace_is_fused(self) -> bool:
        return self.trace_variant() in (capi.TRACE_WIDE_PERSISTENT, capi.TRACE_WIDE8_PERSISTENT)

    def oracle_search(self) -> int:
        """The `use_bvh` mode of the oracle that states the same search: 0 exhaustive, 1 BVH2, 2 compressed 4-wide BVH, 3 compressed 8-wide BVH with leaf records."""
        return {capi.TRACE_BVH2: 1, capi.TRACE_WIDE_PERSISTENT: 2, capi.TRACE_EXHAUSTIVE: 0, capi.TRACE_WIDE8_PERSISTENT: 3}[self.trace_variant()]

    def kernel_times(self) -> dict:
        t = capi.HiprKernelTimes()
        self._check(self.lib.hipr_get_kernel_times(self.handle, C.byref(t)), "hipr_get_kernel_times")
        return {name: dict(ms=t.milliseconds[i], launches=int(t.launches[i])) for i, name in enumerate(capi.HIPR_KERNEL_NAMES)}

    def valu_issue_rates(self) -> dict:
        """Wave64 instructions per second device-wide of v_fma_f32 / v_max_f32 / v_cvt_f32_ubyte1 chains (hipr_debug_valu_issue_rates)."""
        out = (C.c_double * 3)()
        self._check(self.lib.hipr_debug_valu_issue_rates(self.handle, out), "hipr_debug_valu_issue_rates")
        return {"v_fma_f32": out[0], "v_max_f32": out[1], "v_cvt_f32_ubyte1": out[2]}

    def scatter_tiles(self, compact_ptr, rank_stride, rank_count, width, height, out_ptr, out_pitch):
        self._check(self.lib.hipr_scatter_tiles(self.handle, C.c_void_p(compact_ptr), rank_stride, rank_count, width, height, C.c_void_p(out_ptr), out_pitch),
                    "hipr_scatter_tiles")

    # ---- stage-level parity entry points -------------------------------------------------------
    def debug_generate(self, camera, accumulation):
        n = self.owned_pixel_count()
        o = np.zeros((n, 4), np.float32); d = np.zeros((n, 4), np.float32); px = np.zeros(n, np.uint32)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        self._check(self.lib.hipr_debug_generate(self.handle, C.byref(camera), accumulation, o.ctypes.data_as(fp), d.ctypes.data_as(fp), px.ctypes.data_as(up)),
                    "hipr_debug_generate")
        return o, d, px

    def debug_shading(self, shading_model, params10, wo, inputs, mode=0):
        """mode 0: sample(wo, u) -> (n, 7) f, pdf, direction; mode 1: evaluate_with_PDF(wo, wi) -> (n, 7) f, pdf, 0, 0, 0."""
        inputs = np.ascontiguousarray(inputs, np.float32).reshape(-1, 3)
        wo = np.ascontiguousarray(np.broadcast_to(np.asarray(wo, np.float32), inputs.shape))
        params = np.ascontiguousarray(params10, np.float32)
        out = np.zeros((len(inputs), 7), np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.hipr_debug_shading(self.handle, int(shading_model), params.ctypes.data_as(fp), wo.ctypes.data_as(fp), inputs.ctypes.data_as(fp), len(inputs),
                                                int(mode), out.ctypes.data_as(fp)), "hipr_debug_shading")
        return out

    def debug_shade(self, camera, rays, throughput_bounces, hits, last_triangle, pixel_hash, accumulation):
        """hipr_debug_shade: shade_path for n queue entries of the uploaded scene -> (n, 32) records (include/hiprenderer_c.h)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = len(rays)
        throughput_bounces = np.ascontiguousarray(throughput_bounces, np.float32).reshape(n, 4)
        hits = np.ascontiguousarray(hits, np.float32).reshape(n, 4)
        words = [np.ascontiguousarray(a, np.uint32).reshape(n) for a in (last_triangle, pixel_hash, accumulation)]
        out = np.zeros((n, 32), np.float32)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        self._check(self.lib.hipr_debug_shade(self.handle, C.byref(camera), n, rays.ctypes.data_as(fp), throughput_bounces.ctypes.data_as(fp), hits.ctypes.data_as(fp),
                                              *[a.ctypes.data_as(up) for a in words], out.ctypes.data_as(fp)), "hipr_debug_shade")
        return out

    def debug_light(self, light: capi.HiprLight, position, inputs, mode=0):
        """mode 0: sample_radiance(light, position, u = inputs[:, :2]) -> (n, 8) radiance, PDF, direction, distance;
        mode 1 (spot lights): evaluate and pdf for the directions in `inputs` -> (n, 8) radiance, pdf, 0, 0, 0, 0."""
        inputs = np.ascontiguousarray(inputs, np.float32)
        if inputs.ndim == 2 and inputs.shape[1] == 2:
            inputs = np.concatenate([inputs, np.zeros((len(inputs), 1), np.float32)], axis=1)
        inputs = np.ascontiguousarray(inputs.reshape(-1, 3))
        position = np.ascontiguousarray(position, np.float32)
        out = np.zeros((len(inputs), 8), np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.hipr_debug_light(self.handle, C.byref(light), position.ctypes.data_as(fp), inputs.ctypes.data_as(fp), len(inputs), int(mode), out.ctypes.data_as(fp)),
                    "hipr_debug_light")
        return out

    def debug_sobol(self, triples):
        triples = np.ascontiguousarray(triples, np.uint32).reshape(-1, 3)
        out = np.zeros((len(triples), 4), np.uint32)
        up = C.POINTER(C.c_uint32)
        self._check(self.lib.hipr_debug_sobol(self.handle, triples.ctypes.data_as(up), len(triples), out.ctypes.data_as(up)), "hipr_debug_sobol")
        return out

    def debug_sample_offsets(self):
        out = np.zeros((256, 4), np.float32)
        self._check(self.lib.hipr_debug_sample_offsets(self.handle, out.ctypes.data_as(C.POINTER(C.c_float))), "hipr_debug_sample_offsets")
        return out

    def debug_trace_closest(self, rays, skip=None):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros((len(rays), 4), np.float32)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        sk = np.ascontiguousarray(skip, np.uint32).ctypes.data_as(up) if skip is not None else None
        self._check(self.lib.hipr_debug_trace_closest(self.handle, rays.ctypes.data_as(fp), sk, len(rays), hits.ctypes.data_as(fp)), "hipr_debug_trace_closest")
        return hits

    def debug_trace_shadow(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.hipr_debug_trace_shadow(self.handle, rays.ctypes.data_as(fp), len(rays), out.ctypes.data_as(fp)), "hipr_debug_trace_shadow")
        return out
