"""ctypes wrapper of host/libhiprenderer_host.so (scene construction, BVH build). These entry points need no GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

_lib = None


def load_host_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not capi.HOST_LIB_PATH.exists():
        raise capi.HiprError(f"{capi.HOST_LIB_PATH} is missing: run __graft_entry__.build()")
    capi.load_library()   # the host library links libhiprenderer.so (HIPRenderer::Renderer drives the C-ABI); keeps the torch-first load order
    lib = C.CDLL(str(capi.HOST_LIB_PATH))
    vp = C.c_void_p
    lib.hiprh_scene_create.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint]
    lib.hiprh_scene_create.restype = vp
    lib.hiprh_scene_load.argtypes = [C.c_char_p, C.c_uint]
    lib.hiprh_scene_load.restype = vp
    lib.hiprh_png_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_ubyte), C.c_size_t]
    lib.hiprh_png_load.restype = C.c_size_t
    lib.hiprh_make_camera.argtypes = [C.POINTER(C.c_float), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(capi.HiprCameraState)]
    lib.hiprh_scene_move_model.argtypes = [vp, C.c_uint, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_double]
    lib.hiprh_scene_destroy.argtypes = [vp]
    lib.hiprh_scene_desc.argtypes = [vp]
    lib.hiprh_scene_desc.restype = C.POINTER(capi.HiprSceneDesc)
    lib.hiprh_scene_state.argtypes = [vp, C.POINTER(capi.HiprSceneState)]
    lib.hiprh_scene_camera.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_float, C.POINTER(capi.HiprCameraState)]
    lib.hiprh_bvh_build.argtypes = [C.POINTER(capi.HiprTriangle), C.c_uint, C.c_uint]
    lib.hiprh_bvh_build.restype = vp
    lib.hiprh_bvh_node_count.argtypes = [vp]; lib.hiprh_bvh_node_count.restype = C.c_uint
    lib.hiprh_bvh_max_depth.argtypes = [vp]; lib.hiprh_bvh_max_depth.restype = C.c_uint
    lib.hiprh_bvh_nodes.argtypes = [vp]; lib.hiprh_bvh_nodes.restype = C.POINTER(capi.HiprBvhNode)
    lib.hiprh_bvh_order.argtypes = [vp]; lib.hiprh_bvh_order.restype = C.POINTER(C.c_uint)
    lib.hiprh_bvh_wide_node_count.argtypes = [vp]; lib.hiprh_bvh_wide_node_count.restype = C.c_uint
    lib.hiprh_bvh_wide_stack_entries.argtypes = [vp]; lib.hiprh_bvh_wide_stack_entries.restype = C.c_uint
    lib.hiprh_bvh_wide_nodes.argtypes = [vp]; lib.hiprh_bvh_wide_nodes.restype = C.POINTER(capi.HiprWideNode)
    lib.hiprh_pmjbn_samples.argtypes = [C.POINTER(C.c_float), C.c_uint, C.c_uint]
    lib.hiprh_bvh_destroy.argtypes = [vp]
    lib.hiprh_renderer_bench.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]
    lib.hiprh_encode_octahedral.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_short)]
    _lib = lib
    return lib


def load_png(path: str, flip: bool = False) -> np.ndarray:
    """Decode a PNG with the host library's decoder into (height, width, channels) uint8; rows top-down unless `flip`."""
    lib = load_host_library()
    w, h, c = C.c_uint(), C.c_uint(), C.c_uint()
    size = lib.hiprh_png_load(path.encode(), int(flip), C.byref(w), C.byref(h), C.byref(c), None, 0)
    if size == 0:
        raise capi.HiprError(f"could not decode '{path}'")
    out = np.empty(size, dtype=np.uint8)
    lib.hiprh_png_load(path.encode(), int(flip), None, None, None, out.ctypes.data_as(C.POINTER(C.c_ubyte)), size)
    return out.reshape(h.value, w.value, c.value)


def make_camera(width: int, height: int, position=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0), field_of_view=np.pi / 4, near=0.1, far=100.0, orthographic=None,
                accumulations: int = 0, max_bounce_count: int = 4) -> capi.HiprCameraState:
    """Camera state for a free camera: perspective (field of view, near, far) or orthographic=(width, height, depth)."""
    lib = load_host_library()
    ortho = orthographic if orthographic is not None else (1.0, 1.0, 1000.0)
    params = np.array(list(position) + list(rotation) + [field_of_view, near, far, 1.0 if orthographic is not None else 0.0] + list(ortho), np.float32)
    cam = capi.HiprCameraState()
    if lib.hiprh_make_camera(params.ctypes.data_as(C.POINTER(C.c_float)), width, height, accumulations, max_bounce_count, C.byref(cam)) != 0:
        raise capi.HiprError("hiprh_make_camera failed")
    return cam


class Scene:
    """A flattened scene owned by the C++ host library (what handle_updates() would hand to hipr_upload_scene)."""

    def __init__(self, name: str, diffuse_only: bool = False, param0: int = 0, param1: int = 0, environment: bool = False, coat: bool = False, spot: bool = False, textured: bool = False):
        self.lib = load_host_library()
        if name.startswith("file:"):    # a model file set up the way SimpleViewer sets up its command-line scene
            self.handle = self.lib.hiprh_scene_load(name[5:].encode(), 1 if diffuse_only else 0)
            if not self.handle:
                raise capi.HiprError(f"could not load '{name[5:]}'")
        else:
            self.handle = self.lib.hiprh_scene_create(name.encode(), (1 if diffuse_only else 0) | (2 if environment else 0) | (4 if coat else 0) | (8 if spot else 0) | (16 if textured else 0), param0, param1)
            if not self.handle:
                raise capi.HiprError(f"unknown scene '{name}'")
        self.name = name

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.hiprh_scene_destroy(self.handle)
            self.handle = None

    @property
    def desc(self) -> capi.HiprSceneDesc:
        return self.lib.hiprh_scene_desc(self.handle).contents

    @property
    def state(self) -> capi.HiprSceneState:
        s = capi.HiprSceneState()
        self.lib.hiprh_scene_state(self.handle, C.byref(s))
        return s

    def camera(self, width: int, height: int, accumulations: int = 0, max_bounce_count: int = -1, pdf_scale: float = 0.5, scale_decay: float = 0.0) -> capi.HiprCameraState:
        cam = capi.HiprCameraState()
        if self.lib.hiprh_scene_camera(self.handle, width, height, accumulations, max_bounce_count, pdf_scale, C.byref(cam)) != 0:
            raise capi.HiprError("hiprh_scene_camera failed")
        cam.path_regularization_scale_decay = scale_decay
        return cam

    def move_model(self, model_index: int, translation, rotation=(0.0, 0.0, 0.0, 1.0), scale: float = 1.0, rebuild_threshold: float = 0.0) -> bool:
        """Transform-only update (BVH refit). True when the topology was kept, False when the scene builder rebuilt the tree."""
        t = (C.c_float * 3)(*translation)
        r = (C.c_float * 4)(*rotation)
        status = self.lib.hiprh_scene_move_model(self.handle, model_index, t, r, scale, rebuild_threshold)
        if status < 0:
            raise capi.HiprError("hiprh_scene_move_model failed")
        return status == 1

    def triangles(self) -> np.ndarray:
        d = self.desc
        return np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(C.c_uint32)), shape=(d.triangle_count, 12)).copy()

    def nodes(self) -> np.ndarray:
        d = self.desc
        return np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_uint32)), shape=(d.node_count, 16)).copy()


def renderer_bench(target_triangles: int, width: int, height: int, warmup_calls: int, calls: int, max_batch: int) -> dict:
    """Times `calls` HIPRenderer::Renderer::render() calls on the atrium built in the Bifrost managers (the plugin path end to end)."""
    lib = load_host_library()
    out = (C.c_double * 3)()
    status = lib.hiprh_renderer_bench(str(capi.TABLES_PATH.parent.parent).encode(), target_triangles, width, height, warmup_calls, calls, max_batch, out)
    if status != 0:
        raise capi.HiprError(f"hiprh_renderer_bench failed with status {status}")
    return {"milliseconds": out[0], "accumulations": int(out[1]), "triangles": int(out[2]), "calls": calls, "max_batch": max_batch}
