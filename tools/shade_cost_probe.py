#!/usr/bin/env python3
"""Shade-kernel cost against the next-event sample count on the atrium: what one light candidate costs next to the rest of a hit.
usage: python tools/shade_cost_probe.py [spp_per_pass]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bifrost3d_amd import capi
from bifrost3d_amd.host import Scene
from bifrost3d_amd.renderer import Context

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene = Scene("atrium", param0=260000, param1=1)
ctx = Context(0)
ctx.upload_scene(scene)
ctx.set_frame(1920, 1080, 0, 1, spp)
for count in (0, 1, 2, 3, 8):
    state = capi.HiprSceneState()
    for i in range(3):
        state.environment_tint[i] = scene.state.environment_tint[i]
    state.next_event_sample_count = count
    ctx.set_scene_state(state)
    ctx.render_pass(scene.camera(1920, 1080, accumulations=0, max_bounce_count=4), synchronize=True)
    ctx.reset_counters(); ctx.reset_timers()
    for a in (1, 2):
        ctx.render_pass(scene.camera(1920, 1080, accumulations=a * spp, max_bounce_count=4), synchronize=True)
    c, t = ctx.counters(), ctx.kernel_times()
    shade = t["shade"]["ms"] / 2
    trace = sum(v["ms"] for k, v in t.items() if k.startswith("trace")) / 2
    print(f"NEE candidates {count}: shade {shade:.2f} ms / pass, {c['shaded_hits'] / 2 / 1e6:.1f} M hits -> {shade * 1e6 / (c['shaded_hits'] / 2):.4f} ns / hit; "
          f"trace {trace:.2f} ms, closest {c['closest_rays'] / 2 / 1e6:.1f} M shadow {c['shadow_rays'] / 2 / 1e6:.1f} M rays")
ctx.close()
