"""bifrost3d_amd -- MI355X-native path tracing renderer behind the Bifrost renderer surface.

The product is the pair of native libraries built in-tree:
  csrc/libhiprenderer.so        hand-written HIP kernels + the C-ABI of include/hiprenderer_c.h
  host/libhiprenderer_host.so   C++ host side (Bifrost math mirror, BVH builder, scene flattening)
The Python modules are ctypes plumbing for tests, bench.py and the multi-GPU driver.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "host", "renderer"]
