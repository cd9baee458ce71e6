#!/usr/bin/env python3
"""Prints the per-bounce traversal diagnostics of the persistent trace kernels (HIPR_TRACE_LOG=1, instrumented build) for one pass.
usage: HIPR_TRACE_LOG=1 python tools/trace_log_probe.py [scene] [spp_per_pass] [wavefronts]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bifrost3d_amd.host import Scene
from bifrost3d_amd.renderer import Context

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wavefronts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
scene = Scene("atrium", param0=260000, param1=1) if name == "atrium" else Scene(name)
ctx = Context(0)
ctx.upload_scene(scene)
ctx.set_wavefront_count(wavefronts)
ctx.set_frame(1920, 1080, 0, 1, spp)
bounces = 32 if name in ("material", "glass", "opacity") else 4
ctx.render_pass(scene.camera(1920, 1080, accumulations=0, max_bounce_count=bounces), synchronize=True)
ctx.set_instrumentation(True)
ctx.reset_counters()
ctx.render_pass(scene.camera(1920, 1080, accumulations=spp, max_bounce_count=bounces), synchronize=True)
print(ctx.counters())
ctx.close()
