"""ctypes mirror of include/hiprenderer_c.h and loader for the C-ABI shared library.

Python here is plumbing for tests, bench.py and the multi-GPU driver only: it holds no
rendering logic. The library must have been built (``__graft_entry__.build()``); loading
fails loudly otherwise -- there is no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("HIPR_LIBRARY", PKG_DIR / "csrc" / "libhiprenderer.so"))   # HIPR_LIBRARY: A/B builds of the kernels
HIPR_ARITHMETIC_FAST, HIPR_ARITHMETIC_EXACT = 0, 1      # hipr_set_arithmetic: the two builds of the shade unit inside the library
HOST_LIB_PATH = Path(os.environ.get("HIPR_HOST_LIBRARY", PKG_DIR / "host" / "libhiprenderer_host.so"))   # HIPR_HOST_LIBRARY: A/B builds of the host side (BVH builder)
TABLES_PATH = PKG_DIR / "data" / "HIPRenderer" / "shading_tables.bin"

HIPR_OK = 0

c_f = C.c_float
c_u32 = C.c_uint32
c_i32 = C.c_int32
c_u16 = C.c_uint16
c_u8 = C.c_uint8
c_u64 = C.c_uint64


class HiprMaterial(C.Structure):
    _fields_ = [("flags", c_u16), ("shading_model", c_u16), ("tint", c_f * 3),
                ("roughness", c_f), ("tint_roughness_texture_ID", c_i32), ("roughness_texture_ID", c_i32), ("specularity", c_f),
                ("metallic", c_f), ("metallic_texture_ID", c_i32), ("coverage", c_f), ("coverage_texture_ID", c_i32),
                ("emission", c_f * 3), ("coat", c_u16), ("coat_roughness", c_u16)]


class HiprLight(C.Structure):
    _fields_ = [("data", c_f * 11), ("flags", c_u32)]


class HiprVertexGeometry(C.Structure):
    _fields_ = [("position", c_f * 3), ("oct_normal", C.c_int16 * 2)]


class HiprInstance(C.Structure):
    _fields_ = [("object_to_world", c_f * 12), ("index_offset", c_u32), ("vertex_offset", c_u32),
                ("instance_id", c_i32), ("material_index", c_i32), ("mesh_flags", c_u32), ("_pad", c_u32 * 3)]


class HiprTriangle(C.Structure):
    _fields_ = [("v0", c_f * 3), ("v1", c_f * 3), ("v2", c_f * 3),
                ("instance_index", c_u32), ("primitive_index", c_u32), ("flags", c_u32)]


class HiprBvhNode(C.Structure):
    _fields_ = [("c0xy", c_f * 4), ("c1xy", c_f * 4), ("cz", c_f * 4), ("child", c_i32 * 2), ("_pad", c_u32 * 2)]


class HiprTexture(C.Structure):
    _fields_ = [("width", c_u32), ("height", c_u32), ("texel_offset", c_u64), ("format", c_u8),
                ("wrap_u", c_u8), ("wrap_v", c_u8), ("filter", c_u8), ("is_sRGB", c_u8), ("_pad", c_u8 * 3)]


class HiprWideNode(C.Structure):
    _fields_ = [("origin", c_f * 3), ("exponents", c_u32), ("qlo", c_u32 * 3), ("qhi", c_u32 * 3), ("_pad", c_u32 * 2), ("child", c_i32 * 4)]


class HiprLightSample(C.Structure):
    _fields_ = [("radiance", c_f * 3), ("PDF", c_f), ("direction_to_light", c_f * 3), ("distance", c_f)]


class HiprEnvironment(C.Structure):
    _fields_ = [("environment_map_ID", c_i32), ("pdf_width", c_u32), ("pdf_height", c_u32), ("per_pixel_PDF", C.POINTER(c_f)),
                ("samples", C.POINTER(HiprLightSample)), ("sample_count", c_u32)]


class HiprSceneDesc(C.Structure):
    _fields_ = [("nodes", C.POINTER(HiprBvhNode)), ("node_count", c_u32),
                ("triangles", C.POINTER(HiprTriangle)), ("triangle_count", c_u32),
                ("instances", C.POINTER(HiprInstance)), ("instance_count", c_u32),
                ("indices", C.POINTER(c_u32)), ("index_count", c_u32),
                ("geometry", C.POINTER(HiprVertexGeometry)), ("vertex_count", c_u32),
                ("texcoords", C.POINTER(c_f)), ("tints", C.POINTER(c_u32)), ("emissions", C.POINTER(c_f)),
                ("materials", C.POINTER(HiprMaterial)), ("material_count", c_u32),
                ("lights", C.POINTER(HiprLight)), ("light_count", c_u32),
                ("textures", C.POINTER(HiprTexture)), ("texture_count", c_u32),
                ("texels", C.POINTER(c_u8)), ("texel_bytes", c_u64),
                ("bvh_max_depth", c_u32),
                ("wide_nodes", C.POINTER(HiprWideNode)), ("wide_node_count", c_u32), ("wide_stack_entries", c_u32),
                ("environment", C.POINTER(HiprEnvironment)),
                ("wide8_slots", C.POINTER(c_u32 * 16)), ("wide8_slot_count", c_u32), ("wide8_height", c_u32),
                ("wide8_grid_min", c_f * 3), ("wide8_grid_cell", c_f * 3)]


class HiprSceneState(C.Structure):
    _fields_ = [("environment_tint", c_f * 3), ("next_event_sample_count", c_i32)]


class HiprCameraState(C.Structure):
    _fields_ = [("view_to_world_rotation", c_f * 9), ("inverse_projection_matrix", c_f * 16),
                ("inverse_view_projection_matrix", c_f * 16), ("accumulations", c_u32),
                ("max_bounce_count", c_u32), ("path_regularization_PDF_scale", c_f), ("path_regularization_scale_decay", c_f)]


class HiprTables(C.Structure):
    _fields_ = [("ggx_with_fresnel_rho", C.POINTER(c_f)), ("ggx_rho", C.POINTER(c_f)),
                ("dielectric_light_rho", C.POINTER(c_f)), ("dielectric_dense_rho", C.POINTER(c_f)),
                ("ggx_alpha_from_max_PDF", C.POINTER(c_f))]


class HiprFrameDesc(C.Structure):
    _fields_ = [("width", c_u32), ("height", c_u32), ("tile_phase", c_u32), ("tile_stride", c_u32), ("samples_per_pass", c_u32)]


class HiprCounters(C.Structure):
    _fields_ = [("camera_rays", c_u64), ("closest_rays", c_u64), ("shadow_rays", c_u64), ("shaded_hits", c_u64),
                ("closest_nodes", c_u64), ("closest_triangles", c_u64), ("shadow_nodes", c_u64), ("shadow_triangles", c_u64),
                ("iterations", c_u64)]


TRACE_BVH2, TRACE_WIDE_PERSISTENT, TRACE_EXHAUSTIVE, TRACE_WIDE8_PERSISTENT = 0, 1, 2, 3
SHADING_DEFAULT, SHADING_DIFFUSE, SHADING_TRANSMISSIVE = 0, 1, 2
MATERIAL_THIN_WALLED, MATERIAL_CUTOUT = 1, 2
TRIANGLE_OPAQUE, TRIANGLE_ONE_SIDED = 1, 4
ENTRY_PATH_TRACING, ENTRY_DEPTH, ENTRY_ALBEDO, ENTRY_TINT, ENTRY_ROUGHNESS, ENTRY_SHADING_NORMAL, ENTRY_PRIMITIVE_ID, ENTRY_DENOISER_ALBEDO = 0, 3, 4, 5, 6, 7, 8, 9
HIPR_KERNEL_NAMES = ("generate", "trace_closest", "shade", "trace_shadow", "accumulate")


class HiprKernelTimes(C.Structure):
    _fields_ = [("milliseconds", C.c_double * 5), ("launches", c_u64 * 5)]


assert C.sizeof(HiprMaterial) == 64 and C.sizeof(HiprLight) == 48 and C.sizeof(HiprVertexGeometry) == 16
assert C.sizeof(HiprTexture) == 24 and C.sizeof(HiprCameraState) == 180 and C.sizeof(HiprTriangle) == 48 and C.sizeof(HiprBvhNode) == 64 and C.sizeof(HiprInstance) == 80

# Every symbol include/hiprenderer_c.h declares; tests check the library exports all of them.
C_ABI_SYMBOLS = (
    "hipr_create", "hipr_destroy", "hipr_last_error", "hipr_device_count", "hipr_set_stream",
    "hipr_upload_tables", "hipr_upload_scene", "hipr_validate_scene", "hipr_update_scene_geometry", "hipr_group_update_scene_geometry", "hipr_set_scene_state", "hipr_set_entry_point", "hipr_use_scratch_accumulation",
    "hipr_set_frame", "hipr_owned_pixel_count",
    "hipr_render_pass", "hipr_set_samples_per_pass", "hipr_trace_pass", "hipr_accumulate_samples", "hipr_read_accumulation", "hipr_scatter_tiles", "hipr_synchronize", "hipr_get_counters",
    "hipr_device_malloc", "hipr_device_free", "hipr_device_memset", "hipr_copy_to_host", "hipr_present_flipped",
    "hipr_reset_counters", "hipr_set_wavefront_count", "hipr_get_wavefront_count", "hipr_get_trace_variant", "hipr_set_trace_variant", "hipr_set_pass_pipelining", "hipr_set_backface_culling", "hipr_set_arithmetic", "hipr_get_arithmetic", "hipr_set_instrumentation", "hipr_reset_timers", "hipr_get_kernel_times",
    "hipr_group_create", "hipr_group_destroy", "hipr_group_size", "hipr_group_context", "hipr_group_gather_description", "hipr_group_upload_tables", "hipr_group_upload_scene",
    "hipr_group_set_scene_state", "hipr_group_set_entry_point", "hipr_group_use_scratch_accumulation", "hipr_group_set_frame", "hipr_group_set_samples_per_pass",
    "hipr_group_trace_pass", "hipr_group_accumulate_samples", "hipr_group_read_accumulation", "hipr_group_get_counters",
    "hipr_debug_shading", "hipr_debug_shade", "hipr_debug_light", "hipr_debug_math", "hipr_debug_generate", "hipr_debug_sobol", "hipr_debug_valu_issue_rates", "hipr_debug_sample_offsets", "hipr_debug_trace_closest", "hipr_debug_trace_shadow",
)


class HiprError(RuntimeError):
    pass


_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Loads libhiprenderer.so. Raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise HiprError(f"{p} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    try:
        # PyTorch-ROCm bundles its own libamdhip64; a process that loads the system runtime first and torch second
        # ends up with two HIP runtimes and torch then reports "No HIP GPUs are available". Loading torch first makes
        # this library bind to the runtime torch uses (device memory, streams and RCCL come from torch).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(p))
    vp = C.c_void_p
    lib.hipr_last_error.restype = C.c_char_p
    lib.hipr_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.hipr_destroy.argtypes = [vp]
    lib.hipr_set_stream.argtypes = [vp, vp]
    lib.hipr_upload_tables.argtypes = [vp, C.POINTER(HiprTables)]
    lib.hipr_upload_scene.argtypes = [vp, C.POINTER(HiprSceneDesc)]
    lib.hipr_validate_scene.argtypes = [C.POINTER(HiprSceneDesc)]
    lib.hipr_update_scene_geometry.argtypes = [vp, C.POINTER(HiprSceneDesc)]
    lib.hipr_group_update_scene_geometry.argtypes = [vp, C.POINTER(HiprSceneDesc)]
    lib.hipr_set_scene_state.argtypes = [vp, C.POINTER(HiprSceneState)]
    lib.hipr_set_entry_point.argtypes = [vp, C.c_int]
    lib.hipr_use_scratch_accumulation.argtypes = [vp, C.c_int]
    lib.hipr_set_frame.argtypes = [vp, C.POINTER(HiprFrameDesc)]
    lib.hipr_owned_pixel_count.argtypes = [vp, C.POINTER(c_u32)]
    lib.hipr_render_pass.argtypes = [vp, C.POINTER(HiprCameraState), vp, c_u32, C.c_int]
    lib.hipr_set_samples_per_pass.argtypes = [vp, c_u32]
    lib.hipr_trace_pass.argtypes = [vp, C.POINTER(HiprCameraState)]
    lib.hipr_accumulate_samples.argtypes = [vp, c_u32, c_u32, c_u32, vp, c_u32, C.c_int]
    lib.hipr_read_accumulation.argtypes = [vp, C.POINTER(C.c_double), c_u64]
    lib.hipr_scatter_tiles.argtypes = [vp, vp, c_u64, c_u32, c_u32, c_u32, vp, c_u32]
    lib.hipr_synchronize.argtypes = [vp]
    lib.hipr_get_counters.argtypes = [vp, C.POINTER(HiprCounters)]
    lib.hipr_reset_counters.argtypes = [vp]
    lib.hipr_set_instrumentation.argtypes = [vp, C.c_int]
    lib.hipr_set_arithmetic.argtypes = [vp, C.c_int]
    lib.hipr_get_arithmetic.argtypes = [vp]
    lib.hipr_debug_math.argtypes = [vp, C.c_int, c_u32, C.POINTER(c_f), C.POINTER(c_f), C.POINTER(c_f)]
    lib.hipr_reset_timers.argtypes = [vp]
    lib.hipr_get_kernel_times.argtypes = [vp, C.POINTER(HiprKernelTimes)]
    lib.hipr_debug_generate.argtypes = [vp, C.POINTER(HiprCameraState), c_u32, C.POINTER(c_f), C.POINTER(c_f), C.POINTER(c_u32)]
    lib.hipr_debug_sobol.argtypes = [vp, C.POINTER(c_u32), c_u32, C.POINTER(c_u32)]
    lib.hipr_debug_valu_issue_rates.argtypes = [vp, C.POINTER(C.c_double)]
    lib.hipr_group_create.argtypes = [C.POINTER(C.c_int), c_u32, C.POINTER(vp)]
    lib.hipr_group_destroy.argtypes = [vp]
    lib.hipr_group_size.argtypes = [vp]; lib.hipr_group_size.restype = c_u32
    lib.hipr_group_context.argtypes = [vp, c_u32]; lib.hipr_group_context.restype = vp
    lib.hipr_group_gather_description.argtypes = [vp]; lib.hipr_group_gather_description.restype = C.c_char_p
    lib.hipr_group_upload_tables.argtypes = [vp, C.POINTER(HiprTables)]
    lib.hipr_group_upload_scene.argtypes = [vp, C.POINTER(HiprSceneDesc)]
    lib.hipr_group_set_scene_state.argtypes = [vp, C.POINTER(HiprSceneState)]
    lib.hipr_group_set_entry_point.argtypes = [vp, C.c_int]
    lib.hipr_group_use_scratch_accumulation.argtypes = [vp, C.c_int]
    lib.hipr_group_set_frame.argtypes = [vp, c_u32, c_u32, c_u32]
    lib.hipr_group_set_samples_per_pass.argtypes = [vp, c_u32]
    lib.hipr_group_trace_pass.argtypes = [vp, C.POINTER(HiprCameraState)]
    lib.hipr_group_accumulate_samples.argtypes = [vp, c_u32, c_u32, c_u32, vp, c_u32, C.c_int]
    lib.hipr_group_read_accumulation.argtypes = [vp, C.POINTER(C.c_double), c_u64]
    lib.hipr_group_get_counters.argtypes = [vp, C.POINTER(HiprCounters)]
    lib.hipr_debug_sample_offsets.argtypes = [vp, C.POINTER(c_f)]
    lib.hipr_debug_trace_closest.argtypes = [vp, C.POINTER(c_f), C.POINTER(c_u32), c_u32, C.POINTER(c_f)]
    lib.hipr_debug_trace_shadow.argtypes = [vp, C.POINTER(c_f), c_u32, C.POINTER(c_f)]
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, status: int, what: str = "") -> None:
    if status != HIPR_OK:
        msg = lib.hipr_last_error()
        raise HiprError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def load_tables():
    """Returns the five f32 tables of data/HIPRenderer/shading_tables.bin as numpy arrays (base, full, light, dense, alpha)."""
    import numpy as np
    raw = TABLES_PATH.read_bytes()
    if raw[:8] != b"HIPRTBL1":
        raise HiprError("bad shading table file")
    counts = np.frombuffer(raw[8:28], dtype="<u4")
    data = np.frombuffer(raw[28:], dtype="<f4")
    out, o = [], 0
    for n in counts:
        out.append(np.ascontiguousarray(data[o:o + int(n)]))
        o += int(n)
    return tuple(out)
