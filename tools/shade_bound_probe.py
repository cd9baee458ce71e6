#!/usr/bin/env python3
"""What bounds k_shade on the atrium? The same frame with 3, 1 and 0 light candidates per hit (next_event_sample_count): the queue traffic of the shade kernel
barely changes (no candidates: no shadow rays either), its arithmetic drops by a third per candidate.
usage: python tools/shade_bound_probe.py [spp_per_pass]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bifrost3d_amd.host import Scene
from bifrost3d_amd.renderer import Context

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = Scene("atrium", param0=260000, param1=1)
ctx = Context(0)
ctx.upload_scene(scene)
ctx.set_wavefront_count(1)
ctx.set_frame(1920, 1080, 0, 1, spp)
for candidates in (3, 1, 0, 3):
    state = scene.state
    state.next_event_sample_count = candidates
    ctx.set_scene_state(state)
    a = 0
    for _ in range(2):
        ctx.render_pass(scene.camera(1920, 1080, accumulations=a, max_bounce_count=4), synchronize=True); a += spp
    ctx.reset_counters(); ctx.reset_timers()
    for _ in range(3):
        ctx.render_pass(scene.camera(1920, 1080, accumulations=a, max_bounce_count=4), synchronize=True); a += spp
    ctx.synchronize()
    t, c = ctx.kernel_times(), ctx.counters()
    print(f"candidates {candidates}: " + ", ".join(f"{k} {v['ms'] / 3:.2f} ms" for k, v in t.items() if v["launches"]) + f"; shaded hits {c['shaded_hits'] / 3:.3e}, shadow rays {c['shadow_rays'] / 3:.3e}, closest rays {c['closest_rays'] / 3:.3e}")
ctx.close()
