"""GPU parity tests added in round 2 (through the C-ABI, against the pinned CPU oracle):

  * partial coverage / cut-outs: shadow any-hit transmittance bit-exact on all three searches with non-opaque triangles and a
    coverage texture, closest hits bit-exact, and the image of the viewer's opacity scene (stochastic coverage rejection)
    -- ORS/MonteCarlo.cu:152-164,278-285, OR/Types.h:405-414, apps/SimpleViewer/Scenes/Opacity.h:27-104;
  * next_event_sample_count in {1, 8, 64, 256} (clamp to 256, OR/Renderer.cpp:1390) and the 256 sample offsets themselves (:323-336);
  * path regularisation with scale_decay != 0 (OR/PublicTypes.h:38-45);
  * a tree that needs more than the 32 entry LDS stack (the OVERFLOW kernels) and a 1 M-triangle scene (BASELINE config 5's shape);
  * light arrays on both sides of the 32 lights the shade kernel keeps in LDS.
"""
import ctypes as C

import numpy as np
import pytest

from bifrost3d_amd import capi
from bifrost3d_amd.host import Scene
from test_coverage_cpu import deep_chain_rays, opacity_rays, write_deep_chain_obj

pytestmark = pytest.mark.gpu

OPACITY_VARIANT = {1: capi.TRACE_EXHAUSTIVE, 2: capi.TRACE_BVH2, 8: capi.TRACE_WIDE8_PERSISTENT}


@pytest.fixture(scope="module")
def ctx():
    from bifrost3d_amd.renderer import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle_q():
    from oracle_bindings import get_oracle
    return get_oracle(True)


def render_gpu(ctx, scene, w, h, spp, max_bounce, samples_per_pass=1, state=None, **camera_arguments):
    ctx.upload_scene(scene)
    if state is not None:
        ctx.set_scene_state(state)
    ctx.set_frame(w, h, 0, 1, samples_per_pass)
    ctx.reset_counters()
    for a in range(0, spp, samples_per_pass):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=max_bounce, **camera_arguments), synchronize=True)
    return ctx.read_accumulation(), ctx.counters()


def rmse_without_worst(gpu, cpu, outliers):
    """RMSE over all but the `outliers` pixels of largest error: a path that takes another discrete decision under the shade kernel's approximate
    arithmetic is one firefly in a small low-spp frame and the whole RMSE (profiles/r03_rmse_protocol_*.json: zero-mean)."""
    squared = ((gpu[..., :3] - cpu[..., :3]) ** 2).sum(axis=-1).ravel()
    kept = np.sort(squared)[:len(squared) - outliers]
    return float(np.sqrt(kept.sum() / (3 * len(kept))))


def image_metrics(gpu, cpu):
    diff = gpu[..., :3] - cpu[..., :3]
    rel = np.abs(diff) / (np.abs(cpu[..., :3]) + 1e-3)
    return float((rel.max(axis=-1) <= 1e-3).mean()), float(np.sqrt(np.mean(diff ** 2)))


@pytest.mark.parametrize("quads", [1, 2, 8])
def test_shadow_rays_through_partial_coverage_bit_exact(ctx, oracle_q, quads):
    """Non-opaque triangles on every search: the cut-out box (coverage texture, nearest lookup, threshold) and the coverage 0.75
    planes. Transmittance = product of (1 - coverage) over ALL hits in (0, tmax), early out below 1e-7; counters equal."""
    scene = Scene("opacity", param0=quads)
    ctx.upload_scene(scene)
    assert ctx.trace_variant() == OPACITY_VARIANT[quads]
    ctx.set_instrumentation(True)
    rays = opacity_rays(60000, 23 + quads, tmax=True)
    gpu = ctx.debug_trace_shadow(rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu, (nodes, tris) = oracle_q.trace_shadow(scene.desc, rays, use_bvh=ctx.oracle_search())
    assert np.array_equal(gpu, cpu)
    assert counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris
    values = set(np.unique(gpu).tolist())
    assert values <= {0.0, 0.0625, 0.25, 1.0} and values >= {0.0, 0.0625, 0.25, 1.0}, values


@pytest.mark.parametrize("quads", [1, 2, 8])
def test_closest_hits_in_the_opacity_scene_bit_exact(ctx, oracle_q, quads):
    scene = Scene("opacity", param0=quads)
    ctx.upload_scene(scene)
    ctx.set_instrumentation(True)
    rays = opacity_rays(40000, 5)
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris


@pytest.mark.parametrize("quads", [1, 2, 8])
def test_opacity_scene_image_matches_oracle(ctx, oracle_q, quads):
    """apps/SimpleViewer/Scenes/Opacity.h on the device: closest hits on the cut-out box and the 0.75 planes are rejected
    stochastically against the BSDF sample's fourth dimension and retraced with tmin = nextafter(t) (MonteCarlo.cu:152-164);
    shadow rays accumulate (1 - coverage). Closest-hit ray counts include the retraces and must agree with the oracle's."""
    scene = Scene("opacity", param0=quads)
    w, h, spp = 96, 54, 8
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 8)
    cpu, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=8), w, h, spp, use_bvh=ctx.oracle_search())
    assert np.isfinite(gpu).all()
    close, rmse = image_metrics(gpu, cpu)
    print(f"opacity q={quads}: pixels within 1e-3: {close:.4f}, RMSE {rmse:.3e}, mean {float(cpu[..., :3].mean()):.3f}")
    assert close >= 0.999, close
    assert rmse <= 1e-4, rmse               # measured 3.5e-7
    assert cc["closest_rays"] > cc["shaded_hits"] + 0.02 * cc["camera_rays"]          # rejected hits were retraced
    for key in ("closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(gc[key] - cc[key]) <= 0.002 * cc[key], (key, gc[key], cc[key])


def test_sample_offsets_on_the_device_bit_exact(ctx, oracle_q):
    """The 256 reverse-Halton offsets as uploaded (hipr_create) against the oracle's restatement of OR/Renderer.cpp:323-336, bit for bit."""
    assert np.array_equal(ctx.debug_sample_offsets().view(np.uint32), oracle_q.sample_offsets(256).view(np.uint32))


@pytest.mark.parametrize("scene_name", ["cornell", "atrium"])
@pytest.mark.parametrize("count", [1, 8, 64, 256, 1000])
def test_next_event_sample_count(ctx, oracle_q, scene_name, count):
    """RIS over `count` light candidates per hit, drawn with offsets 0 .. count - 1 (MonteCarlo.cu:91-123); 1000 is clamped to the
    256 offsets that exist (OR/Renderer.cpp:1390-1392). The oracle takes the same count; image and ray counters agree."""
    scene = Scene("cornell") if scene_name == "cornell" else Scene("atrium", param0=20000, param1=3)
    state = scene.state
    state.next_event_sample_count = count
    w, h, spp = 48, 27, 4
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 4, state=state)
    oracle_state = scene.state
    oracle_state.next_event_sample_count = min(count, 256)
    cpu, cc, _ = oracle_q.render(scene.desc, oracle_state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    close, rmse = image_metrics(gpu, cpu)
    print(f"{scene_name} NEE x{count}: pixels within 1e-3: {close:.4f}, RMSE {rmse:.3e}")
    assert np.isfinite(gpu).all() and close >= (0.999 if scene_name == "cornell" else 0.99), close
    assert rmse <= (1e-4 if scene_name == "cornell" else 2e-3)      # measured 7e-6 / 1.3e-4
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])
    ctx.set_scene_state(scene.state)


def test_more_candidates_change_the_estimate_not_its_mean(ctx):
    """1 vs 64 candidates: different images (the estimator changed) of the same brightness (both unbiased up to the reference's clamps)."""
    scene = Scene("cornell")
    images = {}
    for count in (1, 64):
        state = scene.state
        state.next_event_sample_count = count
        images[count], _ = render_gpu(ctx, scene, 96, 54, 32, 4, samples_per_pass=8, state=state)
    ctx.set_scene_state(scene.state)
    assert not np.array_equal(images[1], images[64])
    a, b = float(images[1][..., :3].mean()), float(images[64][..., :3].mean())
    assert abs(a - b) < 0.03 * b, (a, b)


@pytest.mark.parametrize("samples_per_pass", [1, 4])
def test_path_regularization_scale_decay(ctx, oracle_q, samples_per_pass):
    """PDF_scale_at_accumulation(a) = PDF_scale * (1 + scale_decay * a) per path (OR/PublicTypes.h:44): batched passes give the image of
    one-by-one passes bit for bit, the image matches the oracle's, and it differs from the decay-free one."""
    scene = Scene("cornell")
    w, h, spp = 64, 36, 8
    decayed, _ = render_gpu(ctx, scene, w, h, spp, 4, samples_per_pass=samples_per_pass, pdf_scale=0.25, scale_decay=0.75)
    one_by_one, _ = render_gpu(ctx, scene, w, h, spp, 4, pdf_scale=0.25, scale_decay=0.75)
    plain, _ = render_gpu(ctx, scene, w, h, spp, 4, pdf_scale=0.25)
    assert np.array_equal(decayed, one_by_one)
    assert not np.array_equal(decayed, plain)
    cpu, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4, pdf_scale=0.25, scale_decay=0.75), w, h, spp, use_bvh=ctx.oracle_search())
    close, rmse = image_metrics(decayed, cpu)
    assert close >= 0.999 and rmse <= 1e-4, (close, rmse)


def test_overflow_stack_kernels_bit_exact(ctx, oracle_q, tmp_path):
    """A BVH whose worst-case traversal needs more than the 32 LDS stack entries: k_trace_persistent<32, *, *, true> keeps the
    rest in a per-lane scratch array. Closest hits, shadow transmittance and counters equal the oracle's; the oracle's own stack
    high-water mark proves these rays go past 32 entries."""
    scene = Scene("file:" + write_deep_chain_obj(tmp_path / "chain.obj"))
    assert scene.desc.wide_stack_entries > 32
    ctx.set_trace_variant(capi.TRACE_WIDE_PERSISTENT)      # the 4-wide tree's kernels (the default for this scene is the 8-wide tree, next test)
    try:
        ctx.upload_scene(scene)
    finally:
        ctx.set_trace_variant(-1)
    assert ctx.trace_variant() == capi.TRACE_WIDE_PERSISTENT
    ctx.set_instrumentation(True)
    rays = deep_chain_rays(30000, 9)
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    skip[::9] = np.random.default_rng(4).integers(0, 300, len(skip[::9]))
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    oracle_q.lib.oracle_wide_stack_high_water(1)
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=2, with_lights=True)
    assert oracle_q.lib.oracle_wide_stack_high_water(1) > 40
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
    shadow_rays = deep_chain_rays(30000, 10, tmax=True)
    gpu_s = ctx.debug_trace_shadow(shadow_rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu_s, (nodes, tris) = oracle_q.trace_shadow(scene.desc, shadow_rays, use_bvh=2)
    assert np.array_equal(gpu_s, cpu_s)
    assert counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris
    # and a full render through the fused launch (closest + shadow rays in one index space) of the same scene
    w, h, spp = 48, 27, 4
    image, gc = render_gpu(ctx, scene, w, h, spp, 4)
    ref, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=2)
    close, rmse = image_metrics(image, ref)
    assert close >= 0.995 and rmse_without_worst(image, ref, 2) <= 1e-4 * max(1.0, float(ref[..., :3].mean())), (close, rmse)     # measured: every pixel within 1e-3, RMSE 1e-6
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 0.002 * cc["closest_rays"]


@pytest.mark.parametrize("count,lowest,highest", [(125, 14, 17), (200, 18, 33)])
def test_deep_tree_on_the_wide8_kernels_bit_exact(ctx, oracle_q, tmp_path, count, lowest, highest):
    """Shorter chains of the same degenerate kind on the 8-wide tree: heights beyond the 12 groups the default LDS stack holds, so the launch picks the
    kernels with the 16- and 32-entry stacks (trees higher than 33 stay with the 4-wide kernels and their scratch-backed stack, previous test); hits,
    transmittance and counters equal the oracle's, whose stack high-water mark proves the rays go that deep."""
    scene = Scene("file:" + write_deep_chain_obj(tmp_path / "chain.obj", count=count))
    assert lowest <= scene.desc.wide8_height <= highest, scene.desc.wide8_height
    ctx.set_trace_variant(capi.TRACE_WIDE8_PERSISTENT)      # the short chains have fewer than 64 BVH2 nodes: by size they would go to the BVH2 kernels
    try:
        ctx.upload_scene(scene)
    finally:
        ctx.set_trace_variant(-1)
    assert ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT
    ctx.set_instrumentation(True)
    rays = deep_chain_rays(30000, 9)
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    skip[::9] = np.random.default_rng(4).integers(0, 300, len(skip[::9]))
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    oracle_q.lib.oracle_wide8_stack_high_water(1)
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=3, with_lights=True)
    assert oracle_q.lib.oracle_wide8_stack_high_water(1) > 12
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
    shadow_rays = deep_chain_rays(30000, 10, tmax=True)
    gpu_s = ctx.debug_trace_shadow(shadow_rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu_s, (nodes, tris) = oracle_q.trace_shadow(scene.desc, shadow_rays, use_bvh=3)
    assert np.array_equal(gpu_s, cpu_s)
    assert counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris


@pytest.mark.parametrize("name,kwargs,bounces", [("material", dict(), 32), ("atrium", dict(param0=20000, param1=3), 4), ("opacity", dict(param0=8), 32)])
def test_pipelined_passes_render_the_unpipelined_frames(ctx, name, kwargs, bounces):
    """hipr_set_pass_pipelining: consecutive passes on alternating queue sets and streams, a pass's last bounces queued blindly while the next pass starts --
    the running mean, the half4 frame and the ray counters are those of passes that run to their end one after the other, bit for bit; also when the reserve
    of blind bounces runs out (the opacity scene's coverage 0.75 planes reject hits without counting a bounce) and the pass is finished when its slot comes up."""
    import torch
    scene = Scene(name, **kwargs)
    w, h, batch, passes = 160, 96, 4, 5
    results = []
    for pipelined in (False, True):
        ctx.set_pass_pipelining(pipelined)
        try:
            ctx.upload_scene(scene)
            assert ctx.trace_is_fused()
            ctx.set_frame(w, h, 0, 1, batch)
            ctx.reset_counters()
            frame = torch.zeros((h, w, 4), dtype=torch.float16, device="cuda")
            for p in range(passes):
                ctx.render_pass(scene.camera(w, h, accumulations=p * batch, max_bounce_count=bounces), frame.data_ptr(), w)
            ctx.synchronize()
            results.append((ctx.read_accumulation(), frame.cpu(), ctx.counters()))
        finally:
            ctx.set_pass_pipelining(False)
    (mean_a, frame_a, counters_a), (mean_b, frame_b, counters_b) = results
    assert np.array_equal(mean_a, mean_b) and torch.equal(frame_a, frame_b)
    for key in ("camera_rays", "closest_rays", "shadow_rays", "shaded_hits"):
        assert counters_a[key] == counters_b[key], (key, counters_a[key], counters_b[key])
    assert counters_a["closest_rays"] > counters_a["camera_rays"]


@pytest.mark.parametrize("name,kwargs,bounces", [("material", dict(), 32), ("atrium", dict(param0=20000, param1=3), 4), ("cornell", dict(), 8), ("opacity", dict(param0=8), 32)])
def test_split_shade_kernel_renders_the_same_frames(ctx, monkeypatch, name, kwargs, bounces):
    """HIPR_SHADE_SPLIT=1 (an experiment kept in the tree): next event estimation and the rest of the shade kernel as two launches over the
    same queue, one flag byte per accepted hit between them. Running mean and ray counters are those of the whole kernel, bit for bit."""
    from bifrost3d_amd.renderer import Context
    scene = Scene(name, **kwargs)
    w, h, batch, passes = 160, 96, 4, 3
    results = []
    for split in (False, True):
        monkeypatch.setenv("HIPR_SHADE_SPLIT", "1" if split else "0")
        c = Context(0)
        try:
            c.upload_scene(scene)
            c.set_frame(w, h, 0, 1, batch)
            c.reset_counters()
            for p in range(passes):
                c.render_pass(scene.camera(w, h, accumulations=p * batch, max_bounce_count=bounces))
            c.synchronize()
            results.append((c.read_accumulation(), c.counters()))
        finally:
            c.close()
    (mean_a, counters_a), (mean_b, counters_b) = results
    assert np.array_equal(mean_a, mean_b)
    for key in ("camera_rays", "closest_rays", "shadow_rays", "shaded_hits"):
        assert counters_a[key] == counters_b[key], (key, counters_a[key], counters_b[key])
    assert counters_a["shadow_rays"] > 0


@pytest.mark.parametrize("name,kwargs,bounces", [("atrium", dict(param0=20000, param1=3), 4), ("material", dict(), 8), ("opacity", dict(param0=8), 16)])
def test_frames_do_not_depend_on_the_backface_culling(ctx, oracle_q, name, kwargs, bounces):
    """hipr_set_backface_culling: the 8-wide traversal steps over closest hits on the back of one-sided surfaces (default) or hands them to the hit program to be
    refused and retraced (0, the reference's way). Same running mean bit for bit; one closest-hit query less per hit stepped over -- and the oracle's counters."""
    scene = Scene(name, **kwargs)
    w, h, batch, passes = 160, 96, 4, 3
    results = []
    ctx.set_trace_variant(capi.TRACE_WIDE8_PERSISTENT)
    try:
        ctx.upload_scene(scene)
        assert ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT
        for culling in (True, False):
            ctx.set_backface_culling(culling)
            ctx.set_frame(w, h, 0, 1, batch)
            ctx.reset_counters()
            for p in range(passes):
                ctx.render_pass(scene.camera(w, h, accumulations=p * batch, max_bounce_count=bounces))
            ctx.synchronize()
            results.append((ctx.read_accumulation(), ctx.counters()))
    finally:
        ctx.set_backface_culling(True)
        ctx.set_trace_variant(-1)
    (stepping, c_on), (retracing, c_off) = results
    assert np.array_equal(stepping, retracing)
    assert c_on["shaded_hits"] == c_off["shaded_hits"] and c_on["shadow_rays"] == c_off["shadow_rays"] and c_on["camera_rays"] == c_off["camera_rays"]
    assert c_on["closest_rays"] <= c_off["closest_rays"]
    if name == "atrium":
        assert c_on["closest_rays"] < 0.92 * c_off["closest_rays"]
    if name == "material":
        assert c_off["closest_rays"] - c_on["closest_rays"] <= 0.02 * c_off["closest_rays"]      # little is reached from behind: the shell through its openings, the grooved layers of the inner ball (measured 0.7 %)
    _, oracle_counters, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, accumulations=0, max_bounce_count=bounces), w, h, batch * passes, use_bvh=3)
    for key in ("closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(c_on[key] - oracle_counters[key]) <= max(2, oracle_counters[key] // 2000), (key, c_on[key], oracle_counters[key])


def test_frames_do_not_depend_on_the_listing_of_the_hits(oracle_q):
    """k_classify_hits lists a bounce's rays as [plain surface hits | coated surface hits | everything else] before k_shade takes them (round 4: the coated
    class, csrc/kernels.h); which rays are shaded and what each yields must not change: the frame is the same bit for bit with the classes on (HIPR_SHADE_CLASSES=1),
    with one class of surface hits (the default: the classes were measured and did not pay) and with no listing at all (HIPR_SHADE_ORDERED=0). The listing starts at 2^18 rays per bounce;
    HIPR_SHADE_ORDERED_FROM brings it down to the test's frame."""
    import os
    from bifrost3d_amd.renderer import Context
    scene = Scene("atrium", param0=20000, param1=3)        # 4 of its 25 materials are coated
    w, h, batch, passes = 160, 96, 4, 2
    frames = []
    for settings in (dict(HIPR_SHADE_ORDERED_FROM="1024", HIPR_SHADE_CLASSES="1"), dict(HIPR_SHADE_ORDERED_FROM="1024", HIPR_SHADE_CLASSES="0"), dict(HIPR_SHADE_ORDERED="0")):
        saved = {k: os.environ.get(k) for k in ("HIPR_SHADE_ORDERED_FROM", "HIPR_SHADE_CLASSES", "HIPR_SHADE_ORDERED")}
        os.environ.update(settings)
        try:
            c = Context(0)       # the switches are read when the context is created
        finally:
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        try:
            c.upload_scene(scene)
            c.set_frame(w, h, 0, 1, batch)
            c.reset_counters()
            for p in range(passes):
                c.render_pass(scene.camera(w, h, accumulations=p * batch, max_bounce_count=4))
            c.synchronize()
            frames.append((c.read_accumulation(), c.counters()))
        finally:
            c.close()
    (classes, c0), (one_class, c1), (unlisted, c2) = frames
    assert np.isfinite(classes).all() and float(classes[..., :3].mean()) > 0
    assert np.array_equal(classes, one_class) and np.array_equal(classes, unlisted)
    for key in ("closest_rays", "shadow_rays", "shaded_hits", "camera_rays"):
        assert c0[key] == c1[key] == c2[key], key


def test_the_kernel_instantiations_without_unreachable_code_render_the_same(oracle_q):
    """Round 4: scenes whose triangles are all statically opaque are traced by k_trace_wide8<..., COVERAGE = false> (no texture-coverage code), scenes without a texture or an
    environment map are shaded by k_shade<..., TEXTURES = false> (no samplers). The traversal's results do not depend on the instantiation: with the full trace kernel forced
    (HIPR_LEAN_TRACE=0) the frame is the same bit for bit and so are the counters. The two shade instantiations are the same source compiled twice with fast-math contraction, so
    their frames agree to rounding, not to the bit: RMSE far below the equal-seed difference to the oracle; and the scene that does bring a texture (the material scene's floor)
    must not take the instantiation without samplers -- its image bar against the oracle is test_material_scene_image_matches_oracle's."""
    import os
    from bifrost3d_amd.renderer import Context
    scene = Scene("atrium", param0=20000, param1=3)
    w, h, batch, passes = 160, 96, 4, 2
    frames = []
    for settings in (dict(), dict(HIPR_LEAN_TRACE="0"), dict(HIPR_LEAN_SHADE="0")):
        saved = {k: os.environ.get(k) for k in ("HIPR_LEAN_TRACE", "HIPR_LEAN_SHADE")}
        os.environ.update(settings)
        try:
            c = Context(0)
        finally:
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        try:
            c.upload_scene(scene)
            c.set_frame(w, h, 0, 1, batch)
            c.reset_counters()
            for p in range(passes):
                c.render_pass(scene.camera(w, h, accumulations=p * batch, max_bounce_count=4))
            c.synchronize()
            frames.append((c.read_accumulation(), c.counters()))
        finally:
            c.close()
    (lean, c0), (full_trace, c1), (full_shade, c2) = frames
    assert np.array_equal(lean, full_trace)
    for key in ("closest_rays", "shadow_rays", "shaded_hits", "camera_rays"):
        assert c0[key] == c1[key], key
    difference = float(np.sqrt(np.mean((lean[..., :3] - full_shade[..., :3]) ** 2)))
    print(f"IMAGE-METRIC lean vs full shade kernel: rmse {difference:.3e} of a mean of {float(lean[..., :3].mean()):.3f}")
    assert np.isfinite(lean).all() and difference <= 1e-4 * float(lean[..., :3].mean())      # measured 2.5e-8 of a mean of 0.93
    for key in ("closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(c0[key] - c2[key]) <= max(4, c0[key] // 1000), key


def test_the_order_the_trace_kernel_takes_its_rays_in_does_not_change_the_frame():
    """Which instantiation of the trace kernel runs, and in which order it takes its rays, does not change the frame: the lean kernel (no coverage code; the default for an
    all-opaque scene) against the full one (HIPR_LEAN_TRACE=0), bit for bit, on a frame large enough that a launch holds many chunks. With HIPR_COHERENCE_SORT=1 a build that
    links the coherence-sort experiment (tools/experiments/ray_sort.hip, `tools/build_variant.sh sort "-DHIPR_RAY_SORT=1"`, loaded through HIPR_LIBRARY) also lists the rays of
    every fused launch by (kind, origin cell, direction octant) and k_trace_wide8<..., SORTED = true> takes them in that order -- every result is stored where it is in
    queue order and a path slot has at most one shadow ray per bounce, so the frame is the same again; the product library (round 6: built without the experiment) ignores
    the variable, and the first two frames below are then the same run twice."""
    import os
    from bifrost3d_amd.renderer import Context
    scene = Scene("atrium", param0=20000, param1=3)
    w, h, batch, passes = 320, 192, 8, 2
    frames = []
    for settings in (dict(), dict(HIPR_COHERENCE_SORT="1"), dict(HIPR_COHERENCE_SORT="1", HIPR_LEAN_TRACE="0")):
        saved = {k: os.environ.get(k) for k in ("HIPR_COHERENCE_SORT", "HIPR_LEAN_TRACE")}
        os.environ.update(settings)
        try:
            c = Context(0)
        finally:
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        try:
            c.upload_scene(scene)
            c.set_frame(w, h, 0, 1, batch)
            for p in range(passes):
                c.render_pass(scene.camera(w, h, accumulations=p * batch, max_bounce_count=6))
            c.synchronize()
            frames.append(c.read_accumulation())
        finally:
            c.close()
    assert np.isfinite(frames[0]).all() and float(frames[0][..., :3].mean()) > 0.05
    assert np.array_equal(frames[0], frames[1])
    assert np.array_equal(frames[0], frames[2])


def test_million_triangle_scene(ctx, oracle_q, verify_ctx):
    """BASELINE config 5's shape at test size: the 1 M-triangle atrium (seed 2), wide BVH of 250 k nodes. Stage parity bit-exact with
    counters, a small image against the oracle, and at 3840 x 2160 the size-independent properties (finite, every camera path
    traced, batching and two-phase tiling bit-identical)."""
    scene = Scene("atrium", param0=1000000, param1=2)
    assert scene.desc.triangle_count > 900000
    ctx.upload_scene(scene)
    assert ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT
    rng = np.random.default_rng(21)
    n = 40000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform(-14, 14, (n, 3))
    rays[:, 1] = np.abs(rays[:, 1]) * 0.7
    d = rng.normal(size=(n, 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 7] = np.inf
    skip = np.full(n, 0xFFFFFFFF, np.uint32)
    ctx.set_instrumentation(True)
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
    rays[:, 7] = rng.uniform(0.05, 30.0, n)
    gpu_s = ctx.debug_trace_shadow(rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu_s, (nodes, tris) = oracle_q.trace_shadow(scene.desc, rays, use_bvh=ctx.oracle_search())
    assert np.array_equal(gpu_s, cpu_s) and counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris

    w, h, spp = 64, 36, 4
    image, gc = render_gpu(ctx, scene, w, h, spp, 4)
    ref, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    close, rmse = image_metrics(image, ref)
    print(f"1M atrium: pixels within 1e-3: {close:.4f}, RMSE {rmse:.3e}, without the 2 worst pixels {rmse_without_worst(image, ref, 2):.3e}")
    # The jittered instances of this scene (three orders of scale, long thin triangles) put 3 % of a 4 spp frame's pixels on a path that parts from its exact copy
    # under the shade kernel's approximate arithmetic (0.972 of the pixels within 1e-3 relative, RMSE 2.2e-2 of a mean of 0.9: profiles/r03_image_metrics.txt).
    # VERDICT round 4 asked whether that is arithmetic or a defect (sliver triangles' geometric normals under contraction): round 5's answer is leg A below -- the
    # verification build of the same source renders this frame bit-identical to the oracle, so nothing but the rounding of the product's operations is in the 3 %
    # -- and leg B: the product's distance to the oracle IS its distance to the verification build on the device. The share of close pixels is the bar; the RMSE
    # (a handful of fireflies, re-rolled by any change of the arithmetic) a sanity bound of 5 % of the frame's mean.
    from conftest import verification_build_equals_oracle
    exact_image = verification_build_equals_oracle(verify_ctx, oracle_q, scene, w, h, spp, 4, "1M atrium")
    on_device = float(np.sqrt(np.mean((image[..., :3] - exact_image) ** 2)))
    assert abs(on_device - rmse) <= 0.1 * on_device + 1e-6, (on_device, rmse)
    assert close >= 0.96 and rmse <= 0.05 * float(ref[..., :3].mean()) and np.isfinite(image).all()
    assert abs(float(image[..., :3].mean()) - float(ref[..., :3].mean())) <= 0.01 * float(ref[..., :3].mean())
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])

    w, h = 3840, 2160
    full, counters = render_gpu(ctx, scene, w, h, 2, 4, samples_per_pass=2)
    assert np.isfinite(full).all() and counters["camera_rays"] == 2 * w * h and counters["shadow_rays"] > 0
    one_by_one, _ = render_gpu(ctx, scene, w, h, 2, 4, samples_per_pass=1)
    assert np.array_equal(one_by_one, full)
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    assembled = np.zeros_like(full)
    for phase in range(2):
        ctx.set_frame(w, h, phase, 2, 2)
        ctx.render_pass(scene.camera(w, h, accumulations=0, max_bounce_count=4), synchronize=True)
        part = ctx.read_accumulation()
        k = np.arange(part.shape[0])
        tile = (k // 64) * 2 + phase
        x, y = (tile % tiles_x) * 8 + (k % 64) % 8, (tile // tiles_x) * 8 + (k % 64) // 8
        valid = (x < w) & (y < h) & (tile < tiles_x * tiles_y)
        assembled[y[valid], x[valid]] = part[valid]
    assert np.array_equal(assembled, full)


def test_ten_million_triangle_scene_at_4k(ctx, oracle_q, verify_ctx):
    """BASELINE config 5 at its size on one GPU: the 10 M-triangle atrium (seed 2), 3840 x 2160. Stage parity on 20 k rays bit for bit with counters
    against the oracle's search over the same 8-wide tree, a small image against the oracle, and at full size the size-independent properties: finite,
    every camera path traced, batching bit-invariant, tiling over two phases assembling the full frame."""
    scene = Scene("atrium", param0=10000000, param1=2)
    assert scene.desc.triangle_count > 9000000 and scene.desc.wide8_slot_count > 0
    ctx.upload_scene(scene)
    assert ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT
    rng = np.random.default_rng(22)
    n = 20000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform(-14, 14, (n, 3))
    rays[:, 1] = np.abs(rays[:, 1]) * 0.7
    d = rng.normal(size=(n, 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 7] = np.inf
    skip = np.full(n, 0xFFFFFFFF, np.uint32)
    ctx.set_instrumentation(True)
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
    rays[:, 7] = rng.uniform(0.05, 30.0, n)
    gpu_s = ctx.debug_trace_shadow(rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu_s, (nodes, tris) = oracle_q.trace_shadow(scene.desc, rays, use_bvh=ctx.oracle_search())
    assert np.array_equal(gpu_s, cpu_s) and counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris
    assert (gpu[:, 3].view(np.uint32) != 0xFFFFFFFF).mean() > 0.25 and 0.02 < (gpu_s == 0).mean() < 0.99      # closest hits from behind one-sided surfaces are stepped over

    w, h, spp = 96, 54, 4
    image, gc = render_gpu(ctx, scene, w, h, spp, 4)
    ref, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    close, rmse = image_metrics(image, ref)
    print(f"10M atrium: pixels within 1e-3: {close:.4f}, RMSE {rmse:.3e}, without the 4 worst pixels {rmse_without_worst(image, ref, 4):.3e}")
    # as in the 1 M-triangle test: the verification build equals the oracle bit for bit on this frame too (leg A), the product's distance to the oracle is its
    # distance to the verification build (leg B); measured: 0.973 of the pixels within 1e-3, RMSE 3.9e-2 of a mean of 0.9
    from conftest import verification_build_equals_oracle
    exact_image = verification_build_equals_oracle(verify_ctx, oracle_q, scene, w, h, spp, 4, "10M atrium")
    on_device = float(np.sqrt(np.mean((image[..., :3] - exact_image) ** 2)))
    assert abs(on_device - rmse) <= 0.1 * on_device + 1e-6, (on_device, rmse)
    assert close >= 0.96 and rmse <= 0.08 * float(ref[..., :3].mean()) and np.isfinite(image).all()
    ctx.upload_scene(scene)      # (the verification context held its own copy; this context renders on)
    assert abs(float(image[..., :3].mean()) - float(ref[..., :3].mean())) <= 0.01 * float(ref[..., :3].mean())
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])

    w, h = 3840, 2160
    full, counters = render_gpu(ctx, scene, w, h, 2, 4, samples_per_pass=2)
    assert np.isfinite(full).all() and counters["camera_rays"] == 2 * w * h and counters["shadow_rays"] > 0
    one_by_one, _ = render_gpu(ctx, scene, w, h, 2, 4, samples_per_pass=1)
    assert np.array_equal(one_by_one, full)
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    assembled = np.zeros_like(full)
    for phase in range(2):
        ctx.set_frame(w, h, phase, 2, 2)
        ctx.render_pass(scene.camera(w, h, accumulations=0, max_bounce_count=4), synchronize=True)
        part = ctx.read_accumulation()
        k = np.arange(part.shape[0])
        tile = (k // 64) * 2 + phase
        x, y = (tile % tiles_x) * 8 + (k % 64) % 8, (tile // tiles_x) * 8 + (k % 64) // 8
        valid = (x < w) & (y < h) & (tile < tiles_x * tiles_y)
        assembled[y[valid], x[valid]] = part[valid]
    assert np.array_equal(assembled, full)
    ctx.set_frame(8, 8)       # give the 4K queues back


def test_device_group_renders_the_single_context_image(ctx, oracle_q):
    """hipr_group_*: a frame split over the members of a device group (tile % size == member, scene replicated, accumulation kept per
    member) and assembled on member 0 equals the single-context frame bit for bit -- f64 accumulation and half4 pixels -- also when the
    pass is batched. On this one-GPU box the three members share device 0, so the gather takes the copy path."""
    import ctypes as C
    import torch
    lib = ctx.lib
    scene = Scene("atrium", param0=20000, param1=3)
    w, h, spp, batch = 100, 60, 4, 2
    single, _ = render_gpu(ctx, scene, w, h, spp, 4, samples_per_pass=batch)
    reference = torch.zeros((h, w, 4), dtype=torch.float16, device="cuda")
    ctx.set_frame(w, h, 0, 1, batch)
    for a in range(0, spp, batch):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=4), reference.data_ptr(), w, synchronize=True)

    devices = (C.c_int * 3)(0, 0, 0)
    group = C.c_void_p()
    assert lib.hipr_group_create(devices, 3, C.byref(group)) == 0
    try:
        assert lib.hipr_group_size(group) == 3 and b"copy" in lib.hipr_group_gather_description(group).lower() or b"memcpy" in lib.hipr_group_gather_description(group).lower()
        tables = capi.load_tables()
        t = capi.HiprTables(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in tables])
        assert lib.hipr_group_upload_tables(group, C.byref(t)) == 0
        assert lib.hipr_group_upload_scene(group, C.byref(scene.desc)) == 0
        state = scene.state
        assert lib.hipr_group_set_scene_state(group, C.byref(state)) == 0
        assert lib.hipr_group_set_frame(group, w, h, batch) == 0
        pitch = w + 12
        frame = torch.zeros((h, pitch, 4), dtype=torch.float16, device="cuda")
        torch.cuda.synchronize()
        for a in range(0, spp, batch):
            cam = scene.camera(w, h, accumulations=a, max_bounce_count=4)
            assert lib.hipr_group_trace_pass(group, C.byref(cam)) == 0
            assert lib.hipr_group_accumulate_samples(group, 0, batch, a, C.c_void_p(frame.data_ptr()), pitch, 1) == 0
        accumulation = np.zeros((h, w, 4), np.float64)
        assert lib.hipr_group_read_accumulation(group, accumulation.ctypes.data_as(C.POINTER(C.c_double)), w * h) == 0
        counters = capi.HiprCounters()
        assert lib.hipr_group_get_counters(group, C.byref(counters)) == 0
    finally:
        lib.hipr_group_destroy(group)
    assert np.array_equal(accumulation, single)
    assert torch.equal(frame[:, :w].cpu(), reference.cpu())
    assert counters.camera_rays == w * h * spp


def test_device_group_over_distinct_devices_gathers_through_rccl(ctx):
    """The same contract over as many DISTINCT devices as the box has (2 ... 8): the members' tiles then travel by RCCL point-to-point inside the process
    (ncclCommInitAll; member 0 posts one ncclRecv per peer, every peer one ncclSend). Skipped on a one-GPU box, where the copy path above is all there is."""
    import ctypes as C
    import torch
    lib = ctx.lib
    n = min(int(lib.hipr_device_count()), 8)
    if n < 2:
        pytest.skip("one device: the RCCL transport needs distinct devices")
    scene = Scene("atrium", param0=20000, param1=3)
    w, h, spp, batch = 200, 120, 4, 2
    single, _ = render_gpu(ctx, scene, w, h, spp, 4, samples_per_pass=batch)
    reference = torch.zeros((h, w, 4), dtype=torch.float16, device="cuda:0")
    ctx.set_frame(w, h, 0, 1, batch)
    for a in range(0, spp, batch):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=4), reference.data_ptr(), w, synchronize=True)
    devices = (C.c_int * n)(*range(n))
    group = C.c_void_p()
    assert lib.hipr_group_create(devices, n, C.byref(group)) == 0
    try:
        assert b"rccl" in lib.hipr_group_gather_description(group).lower(), lib.hipr_group_gather_description(group)
        tables = capi.load_tables()
        t = capi.HiprTables(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in tables])
        assert lib.hipr_group_upload_tables(group, C.byref(t)) == 0
        assert lib.hipr_group_upload_scene(group, C.byref(scene.desc)) == 0
        state = scene.state
        assert lib.hipr_group_set_scene_state(group, C.byref(state)) == 0
        assert lib.hipr_group_set_frame(group, w, h, batch) == 0
        frame = torch.zeros((h, w, 4), dtype=torch.float16, device="cuda:0")
        torch.cuda.synchronize()
        for a in range(0, spp, batch):
            cam = scene.camera(w, h, accumulations=a, max_bounce_count=4)
            assert lib.hipr_group_trace_pass(group, C.byref(cam)) == 0
            assert lib.hipr_group_accumulate_samples(group, 0, batch, a, C.c_void_p(frame.data_ptr()), w, 1) == 0
        accumulation = np.zeros((h, w, 4), np.float64)
        assert lib.hipr_group_read_accumulation(group, accumulation.ctypes.data_as(C.POINTER(C.c_double)), w * h) == 0
    finally:
        lib.hipr_group_destroy(group)
    assert np.array_equal(accumulation, single)
    assert torch.equal(frame.cpu(), reference.cpu())


def test_device_group_reports_the_failing_member(ctx):
    """A member that fails on a worker thread: the status comes back, nothing hangs, and hipr_last_error() on the CALLING thread names the member and carries
    its message (a pass before any scene is uploaded fails on every member; so does a fold of samples that were never traced)."""
    import ctypes as C
    lib = ctx.lib
    devices = (C.c_int * 3)(0, 0, 0)
    group = C.c_void_p()
    assert lib.hipr_group_create(devices, 3, C.byref(group)) == 0
    try:
        assert lib.hipr_group_set_frame(group, 64, 64, 1) == 0
        cam = Scene("cornell").camera(64, 64, accumulations=0, max_bounce_count=1)
        assert lib.hipr_group_trace_pass(group, C.byref(cam)) != 0
        message = lib.hipr_last_error().decode()
        assert "device group member" in message and "must be set before rendering" in message, message
        assert lib.hipr_group_accumulate_samples(group, 0, 1, 0, None, 0, 1) != 0
        assert "device group member" in lib.hipr_last_error().decode()
    finally:
        lib.hipr_group_destroy(group)


def test_device_group_watchdog_ends_a_stalled_exchange(ctx, monkeypatch):
    """the tile exchange of a device group cannot hang. A member whose half of the exchange is held back past the deadline (test hook: a host
    function that sleeps on its gather stream) ends THAT call with HIPR_ERROR_TIMEOUT and a message naming the member; the group falls back to fresh streams and
    peer-to-peer copies, and the next call delivers the frame -- the single-context frame, bit for bit."""
    import ctypes as C
    import time
    import torch
    lib = ctx.lib
    scene = Scene("atrium", param0=20000, param1=3)
    w, h = 100, 60
    reference = torch.zeros((h, w, 4), dtype=torch.float16, device="cuda")
    ctx.upload_scene(scene)
    ctx.set_frame(w, h, 0, 1, 1)
    ctx.render_pass(scene.camera(w, h, accumulations=0, max_bounce_count=4), reference.data_ptr(), w, synchronize=True)
    monkeypatch.setenv("HIPR_GROUP_GATHER_TIMEOUT_MS", "300")
    monkeypatch.setenv("HIPR_GROUP_TEST_STALL_MEMBER", "1")
    devices = (C.c_int * 3)(0, 0, 0)
    group = C.c_void_p()
    assert lib.hipr_group_create(devices, 3, C.byref(group)) == 0
    try:
        tables = capi.load_tables()
        t = capi.HiprTables(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in tables])
        assert lib.hipr_group_upload_tables(group, C.byref(t)) == 0 and lib.hipr_group_upload_scene(group, C.byref(scene.desc)) == 0
        state = scene.state
        assert lib.hipr_group_set_scene_state(group, C.byref(state)) == 0 and lib.hipr_group_set_frame(group, w, h, 1) == 0
        frame = torch.zeros((h, w, 4), dtype=torch.float16, device="cuda")
        torch.cuda.synchronize()
        cam = scene.camera(w, h, accumulations=0, max_bounce_count=4)
        assert lib.hipr_group_trace_pass(group, C.byref(cam)) == 0
        t0 = time.time()
        status = lib.hipr_group_accumulate_samples(group, 0, 1, 0, C.c_void_p(frame.data_ptr()), w, 1)
        seconds = time.time() - t0
        message = lib.hipr_last_error().decode()
        assert status == -7 and "device group member 1" in message and "did not finish within 300 ms" in message, (status, message)      # HIPR_ERROR_TIMEOUT
        assert seconds < 5.0
        assert b"fallen back" in lib.hipr_group_gather_description(group)
        # the same accumulation again, AT ONCE: the held-back copy is still queued on its old stream and runs in the middle of what follows -- on the
        # buffers it was given, which the group has retired; the calls below use fresh streams and fresh buffers, so no tile can tear
        assert lib.hipr_group_trace_pass(group, C.byref(cam)) == 0
        assert lib.hipr_group_accumulate_samples(group, 0, 1, 0, C.c_void_p(frame.data_ptr()), w, 1) == 0, lib.hipr_last_error()
        # (the stalled call had folded accumulation 0 already: fold it into a cleared frame again for the comparison)
        assert lib.hipr_group_set_frame(group, w, h, 1) == 0
        assert lib.hipr_group_trace_pass(group, C.byref(cam)) == 0
        assert lib.hipr_group_accumulate_samples(group, 0, 1, 0, C.c_void_p(frame.data_ptr()), w, 1) == 0
    finally:
        lib.hipr_group_destroy(group)
    assert torch.equal(frame.cpu(), reference.cpu())


@pytest.mark.parametrize("quads", [1, 3, 12])
def test_refitted_scene_on_the_device_bit_exact(ctx, oracle_q, quads):
    """hipr_update_scene_geometry after SceneBuilder::update_model_transforms moved a model (BVH refit, topology kept): closest hits, shadow
    transmittance and counters equal the oracle's on the refitted description for all three searches, and the image equals the image
    of the pose reached by moving back and forth (the refit is exact: the same boxes either way)."""
    from test_coverage_cpu import cornell_box_rays
    scene = Scene("cornell", param0=quads)
    ctx.upload_scene(scene)
    variant = ctx.trace_variant()
    pose = dict(translation=(0.05, -0.30, 0.10), rotation=(0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))), scale=0.3)
    assert scene.move_model(6, **pose) is True
    ctx.update_scene_geometry(scene)
    assert ctx.trace_variant() == variant
    ctx.set_instrumentation(True)
    rays = cornell_box_rays(40000, 8)
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
    rays[:, 7] = np.random.default_rng(3).uniform(0.05, 2.0, len(rays))
    gpu_s = ctx.debug_trace_shadow(rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu_s, (nodes, tris) = oracle_q.trace_shadow(scene.desc, rays, use_bvh=ctx.oracle_search())
    assert np.array_equal(gpu_s, cpu_s) and counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris

    w, h, spp = 64, 36, 4
    ctx.set_frame(w, h)
    for a in range(spp):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=4), synchronize=True)
    moved_image = ctx.read_accumulation()
    reference, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    close, rmse = image_metrics(moved_image, reference)
    assert close >= 0.97 and rmse <= 0.01, (close, rmse)
    untouched = Scene("cornell", param0=quads)
    still, _ = render_gpu(ctx, untouched, w, h, spp, 4)
    assert not np.array_equal(still, moved_image)      # the box did move in the picture
    # a wrong-sized description is refused, the uploaded scene stays usable
    other = Scene("cornell", param0=quads + 1)
    assert ctx.lib.hipr_update_scene_geometry(ctx.handle, __import__("ctypes").byref(other.desc)) == -1


class SceneWithOtherLights:
    """A scene's description with the light array replaced (same geometry, materials, BVH): what a host with many lights uploads."""

    def __init__(self, scene, lights):
        self.scene = scene                      # keeps the arrays the description points at alive
        self.lights = (capi.HiprLight * len(lights))(*lights)
        self.desc = capi.HiprSceneDesc()
        C.memmove(C.byref(self.desc), C.byref(scene.desc), C.sizeof(capi.HiprSceneDesc))
        self.desc.lights = C.cast(self.lights, C.POINTER(capi.HiprLight))
        self.desc.light_count = len(lights)
        self.state = scene.state
        self.camera = scene.camera


def test_a_refit_may_move_a_light_but_not_change_its_type(ctx):
    """hipr_update_scene_geometry uploads the lights again (they move with their nodes) but not the environment's data, and the kernels were instantiated for the
    kinds of light the upload brought: a description whose light changes type is refused, one whose light moved is taken."""
    scene = Scene("cornell")
    ctx.upload_scene(scene)
    light = capi.HiprLight()
    C.memmove(C.byref(light), scene.desc.lights, C.sizeof(capi.HiprLight))
    kind = light.flags & 7                                          # HIPR_LIGHT_TYPE_MASK; 1 sphere, 4 presampled environment (include/hiprenderer_c.h:78-79)
    moved = capi.HiprLight()
    C.memmove(C.byref(moved), C.byref(light), C.sizeof(capi.HiprLight))
    moved.data[4] = light.data[4] - 0.05
    assert ctx.lib.hipr_update_scene_geometry(ctx.handle, C.byref(SceneWithOtherLights(scene, [moved]).desc)) == 0, ctx.lib.hipr_last_error()
    other = capi.HiprLight()
    C.memmove(C.byref(other), C.byref(light), C.sizeof(capi.HiprLight))
    other.flags = (light.flags & ~7) | (4 if kind != 4 else 1)
    assert ctx.lib.hipr_update_scene_geometry(ctx.handle, C.byref(SceneWithOtherLights(scene, [other]).desc)) == -1
    assert b"changes its type" in ctx.lib.hipr_last_error()
    ctx.set_frame(32, 18)
    ctx.render_pass(scene.camera(32, 18, max_bounce_count=2), synchronize=True)      # the context is still usable
    assert np.isfinite(ctx.read_accumulation()).all()


@pytest.mark.parametrize("light_count", [32, 33, 48])
def test_many_lights(ctx, oracle_q, light_count):
    """The shade kernel keeps light arrays of up to 32 entries in LDS and reads longer ones from global memory: the Cornell box lit by a ring of
    `light_count` small sphere lights that share the power of its one lamp, on both sides of that limit, against the oracle."""
    scene = Scene("cornell")
    original = scene.desc.lights[0]
    lights = []
    for k in range(light_count):
        light = capi.HiprLight()
        C.memmove(C.byref(light), C.byref(original), C.sizeof(capi.HiprLight))
        angle = 2.0 * np.pi * k / light_count
        for c in range(3):
            light.data[c] = original.data[c] / light_count                  # power
        light.data[3] = original.data[3] + 0.25 * np.cos(angle)               # position: a ring under the ceiling
        light.data[5] = original.data[5] + 0.25 * np.sin(angle)
        light.data[6] = 0.02                                                   # radius
        lights.append(light)
    many = SceneWithOtherLights(scene, lights)
    w, h, spp = 64, 36, 8
    gpu, gc = render_gpu(ctx, many, w, h, spp, 4)
    cpu, cc, _ = oracle_q.render(many.desc, many.state, many.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    close, rmse = image_metrics(gpu, cpu)
    print(f"cornell with {light_count} lights: pixels within 1e-3: {close:.4f}, RMSE {rmse:.3e}")
    assert np.isfinite(gpu).all() and close >= 0.995, close
    assert rmse <= 1e-3
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])
    assert float(gpu[..., :3].mean()) > 0.05
