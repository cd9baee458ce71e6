"""GPU parity tests: the HIP kernels, called through the C-ABI, against the CPU oracle on the same inputs.

Bit-exact for integer work (RNG) and for the correctly rounded f32 stages (camera rays, BVH traversal,
triangle intersection, shadow transmittance, traversal counters); statistical for shading, where libm
and ocml transcendentals differ in the last ulp (tolerances stated per test).
"""
import ctypes as C

import numpy as np
import pytest

from bifrost3d_amd import capi
from bifrost3d_amd.host import Scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from bifrost3d_amd.renderer import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle_q():
    from oracle_bindings import get_oracle
    return get_oracle(True)   # unorm16 tables, as uploaded to the device


@pytest.fixture(scope="module")
def cornell():
    return Scene("cornell")


@pytest.fixture(scope="module")
def atrium():
    return Scene("atrium", param0=20000, param1=3)


@pytest.fixture(scope="module")
def cornell_tessellated():
    """The Cornell box with every wall cut into 3 x 3 quads: 114 triangles, 35 BVH2 nodes -> the BVH2 kernels."""
    return Scene("cornell", param0=3)


EXPECTED_VARIANT = {"cornell": capi.TRACE_EXHAUSTIVE, "cornell_tessellated": capi.TRACE_BVH2, "atrium": capi.TRACE_WIDE8_PERSISTENT, "atrium_4wide": capi.TRACE_WIDE_PERSISTENT}


@pytest.fixture
def search(ctx):
    """Scene names ending in _4wide run on the 4-wide tree's kernels (the 8-wide tree with leaf records is the default for scenes of that size)."""
    def choose(scene_name):
        ctx.set_trace_variant(capi.TRACE_WIDE_PERSISTENT if scene_name.endswith("_4wide") else -1)
    yield choose
    ctx.set_trace_variant(-1)


def random_rays(rng, n, lo, hi, tmax=np.inf):
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 0.0
    rays[:, 4:7] = d
    rays[:, 7] = tmax
    return rays


def test_sobol_bit_exact(ctx, oracle_q):
    rng = np.random.default_rng(7)
    triples = np.stack([rng.integers(0, 4096, 200000), rng.integers(0, 2 ** 32, 200000), rng.integers(0, 64, 200000)], axis=1).astype(np.uint32)
    triples[:16, 0] = np.arange(16)
    assert np.array_equal(ctx.debug_sobol(triples), oracle_q.sobol4ui(triples))


@pytest.mark.parametrize("size", [(64, 40), (37, 23), (128, 72)])
def test_camera_rays_bit_exact(ctx, oracle_q, cornell, size):
    w, h = size
    ctx.upload_scene(cornell)
    ctx.set_frame(w, h)
    for accumulation in (0, 1, 5, 255):
        cam = cornell.camera(w, h)
        o, d, px = ctx.debug_generate(cam, accumulation)
        valid = px != 0xFFFFFFFF
        assert valid.sum() == w * h
        xy = np.stack([px[valid] & 0xFFFF, px[valid] >> 16], axis=1).astype(np.uint32)
        eo, ed = oracle_q.generate_rays(cam, w, h, accumulation, xy)
        assert np.array_equal(o[valid, :3].view(np.uint32), eo[:, :3].view(np.uint32))
        assert np.array_equal(d[valid, :3].view(np.uint32), ed[:, :3].view(np.uint32))


@pytest.mark.parametrize("scene_name", ["cornell", "cornell_tessellated", "atrium", "atrium_4wide"])
def test_closest_hit_bit_exact(ctx, oracle_q, cornell, cornell_tessellated, atrium, scene_name, search):
    scene = {"cornell": cornell, "cornell_tessellated": cornell_tessellated, "atrium": atrium, "atrium_4wide": atrium}[scene_name]
    search(scene_name)
    ctx.upload_scene(scene)
    ctx.set_instrumentation(True)
    rng = np.random.default_rng(11)
    lo, hi = (-14.0, 14.0) if scene_name.startswith("atrium") else (-0.6, 0.6)
    rays = random_rays(rng, 50000, lo, hi)
    if scene_name.startswith("atrium"):
        rays[:, 1] = np.abs(rays[:, 1]) * 0.7
    w, h = 96, 54
    cam = scene.camera(w, h)
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), axis=-1).reshape(-1, 2).astype(np.uint32)
    co, cd = oracle_q.generate_rays(cam, w, h, 3, xy)
    cam_rays = np.zeros((len(xy), 8), np.float32)
    cam_rays[:, 0:4] = co
    cam_rays[:, 4:7] = cd[:, :3]
    cam_rays[:, 7] = np.inf
    rays = np.concatenate([rays, cam_rays])
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    skip[::7] = rng.integers(0, scene.desc.triangle_count, len(skip[::7]))

    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    # more than 64 BVH2 nodes: persistent kernels over the compressed 8-wide BVH with leaf records (oracle mode 3; mode 2 = the 4-wide tree); at most
    # 64 triangles: exhaustive search (oracle mode 0). The oracle states the same search, so counters agree too.
    cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
    assert ctx.trace_variant() == EXPECTED_VARIANT[scene_name]
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32)), "t, u, v and primitive id must match bit for bit"
    assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
    # The BVH answer equals exhaustive search (oracle side), so traversal loses no hits. The only admissible
    # difference: two coincident surfaces (the Cornell boxes stand on the floor) whose hit distances differ in the
    # last ulp may be resolved to the other surface, because box culling compares against the best distance so far.
    brute, _ = oracle_q.trace_closest(scene.desc, rays[:8000], skip[:8000], use_bvh=False, with_lights=True)
    if ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT:
        # The 8-wide search as a GEOMETRIC search, on both sides: without the stepping over hits on the back of one-sided surfaces that it does by default
        # (hipr_set_backface_culling; above, device and oracle agreed bit for bit WITH it).
        ctx.set_backface_culling(False)
        oracle_q.set_backface_culling(False)
        try:
            ctx.set_instrumentation(True)
            ctx.reset_counters()
            gpu_geometric = ctx.debug_trace_closest(rays, skip)
            counters = ctx.counters()
            ctx.set_instrumentation(False)
            cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
        finally:
            ctx.set_backface_culling(True)
            oracle_q.set_backface_culling(True)
        assert np.array_equal(gpu_geometric.view(np.uint32), cpu.view(np.uint32)) and counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
        stepped_over = (gpu[:, 3].view(np.uint32) != gpu_geometric[:, 3].view(np.uint32)).mean()
        assert (0.05 < stepped_over < 0.5) if scene_name == "atrium" else stepped_over < 0.05
        gpu = gpu_geometric
        # a leaf record solves its triangles from the corner they share, not from their first vertex: the same hit, with t / u / v rounded differently
        differs = brute[:, 3].view(np.uint32) != cpu[:8000, 3].view(np.uint32)
        same = ~differs & np.isfinite(brute[:, 0])
        assert np.all(np.abs(brute[same, 0] - cpu[:8000][same, 0]) <= 2e-5 * (1.0 + np.abs(brute[same, 0])))
        assert np.all(np.abs(brute[same, 1:3] - cpu[:8000][same, 1:3]) <= 2e-3)
    else:
        differs = (brute.view(np.uint32) != cpu[:8000].view(np.uint32)).any(axis=1)
    assert differs.mean() <= 2e-3
    assert np.all(np.abs(brute[differs, 0] - cpu[:8000][differs, 0]) <= 2e-5 * (1.0 + np.abs(brute[differs, 0])))
    assert (gpu[:, 3].view(np.uint32) != 0xFFFFFFFF).mean() > 0.5


@pytest.mark.parametrize("scene_name", ["cornell", "cornell_tessellated", "atrium", "atrium_4wide"])
def test_shadow_rays_bit_exact(ctx, oracle_q, cornell, cornell_tessellated, atrium, scene_name, search):
    scene = {"cornell": cornell, "cornell_tessellated": cornell_tessellated, "atrium": atrium, "atrium_4wide": atrium}[scene_name]
    search(scene_name)
    ctx.upload_scene(scene)
    ctx.set_instrumentation(True)
    rng = np.random.default_rng(13)
    lo, hi = (-12.0, 12.0) if scene_name.startswith("atrium") else (-0.45, 0.45)
    rays = random_rays(rng, 40000, lo, hi)
    rays[:, 7] = rng.uniform(0.05, 30.0 if scene_name.startswith("atrium") else 3.0, len(rays)).astype(np.float32)
    gpu = ctx.debug_trace_shadow(rays)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu, (nodes, tris) = oracle_q.trace_shadow(scene.desc, rays, use_bvh=ctx.oracle_search())
    assert np.array_equal(gpu, cpu)
    assert counters["shadow_nodes"] == nodes and counters["shadow_triangles"] == tris
    assert 0.05 < (gpu == 0).mean() < 0.99


@pytest.mark.parametrize("count", [50000, 777, 64, 33, 1])
def test_the_kernels_the_renderer_runs_return_the_instrumented_kernels_hits(ctx, oracle_q, atrium, count):
    """The bit-exact tests above run the COUNTING instantiations of the trace kernels (set_instrumentation: node / triangle counters); the renderer runs the ones
    without counters -- other template arguments, other register allocation, and the place where an experiment on the traversal lands first (round 6: the paired record
    iterations of tools/experiments/wide8_paired_leaves.diff.txt were held to this test). Same hits, bit for bit, for full waves, ragged ones and a single ray: closest
    hits with and without a triangle to skip, and shadow transmittances."""
    ctx.upload_scene(atrium)
    ctx.set_instrumentation(False)
    rng = np.random.default_rng(count)
    rays = random_rays(rng, count, -14.0, 14.0)
    rays[:, 1] = np.abs(rays[:, 1]) * 0.7
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    skip[::5] = rng.integers(0, atrium.desc.triangle_count, len(skip[::5]))
    gpu = ctx.debug_trace_closest(rays, skip)
    cpu, _ = oracle_q.trace_closest(atrium.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    shadow_rays = rays.copy()
    shadow_rays[:, 7] = rng.uniform(0.05, 30.0, len(rays)).astype(np.float32)
    assert np.array_equal(ctx.debug_trace_shadow(shadow_rays), oracle_q.trace_shadow(atrium.desc, shadow_rays, use_bvh=ctx.oracle_search())[0])
    if count >= 50000:
        assert 0.2 < (gpu[:, 3].view(np.uint32) != 0xFFFFFFFF).mean() < 0.999


def render_gpu(ctx, scene, w, h, spp, max_bounce, first=0, samples_per_pass=1, tile_phase=0, tile_stride=1):
    ctx.upload_scene(scene)
    ctx.set_frame(w, h, tile_phase, tile_stride, samples_per_pass)
    ctx.reset_counters()
    for a in range(first, first + spp, samples_per_pass):
        cam = scene.camera(w, h, accumulations=a, max_bounce_count=max_bounce)
        ctx.render_pass(cam, synchronize=True)
    return ctx.read_accumulation(), ctx.counters()


def rmse(a, b):
    return float(np.sqrt(np.mean((a[..., :3] - b[..., :3]) ** 2)))


# The fast shade unit against the exact one ON THE DEVICE, equal seed, as measured on the MI355X (round 5, GPUTEST log; unchanged in round 6: the specified transcendentals of
# the exact mode round like the f64 ones they replace): the figure each image test holds the product to, x 1.5 (a defect in a branch only the fast build
# compiles -- reciprocal division, v_exp / v_log pow, v_sin / v_cos, contraction -- moves these, and the wide sanity bounds below would not notice).
RECORDED_FAST_VS_EXACT = {"atrium_20k": 1.948e-5, "atrium_20k_textured": 1.218e-3, "environment_cornell": 5.429e-7, "environment_atrium": 5.667e-3, "material": 1.462e-4,
                          "material_coat": 2.734e-5, "glass": 7.033e-3}


def image_bar(name, gpu, cpu, close_at_least, rmse_at_most, band=1e-3, outliers=0, exact=None):
    """The image bar of the PRODUCT build, whose shade unit uses hardware-approximate arithmetic like the reference's --use_fast_math PTX: the share of pixels
    within `band` relative of the oracle's and the RMSE.
    `exact` = (verify_ctx, oracle, scene, w, h, spp, bounces) adds the two legs of round 5. (A) The VERIFICATION build of the same source renders the same frame and
    must equal the oracle (f64 transcendentals) bit for bit: the CODE is then exact, whatever the product's figures are. (B) The product's RMSE against the oracle
    must be its RMSE against the verification build on the device (within 10 %): the figure IS the fast arithmetic's path divergence (profiles/
    r05_divergence_sites.txt: perturbations of the last bits growing along the path until the copies land on another triangle / sample another direction), zero
    mean, and its size in one small frame re-rolls with any change of the arithmetic (the glass scene: 5e-6 in round 2, 7e-3 in round 5 after two reciprocals were
    restated) -- so `rmse_at_most` is a sanity bound there and the share of close pixels, which moves little, is the bar.
    `outliers`: that many pixels with the largest error are left out of the RMSE (tests without leg A only)."""
    rel = np.abs(gpu[..., :3] - cpu[..., :3]) / (np.abs(cpu[..., :3]) + 1e-3)
    close = float((rel.max(axis=-1) <= band).mean())
    value = rmse(gpu, cpu)
    if outliers:
        squared = ((gpu[..., :3] - cpu[..., :3]) ** 2).sum(axis=-1).ravel()
        kept = np.sort(squared)[:len(squared) - outliers]
        print(f"IMAGE-METRIC {name}: rmse of all pixels {value:.3e}, without the {outliers} worst {float(np.sqrt(kept.sum() / (3 * len(kept)))):.3e}")
        value = float(np.sqrt(kept.sum() / (3 * len(kept))))
    print(f"IMAGE-METRIC {name}: close({band:g}) {close:.4f} rmse {value:.3e} mean {float(cpu[..., :3].mean()):.3f}")
    assert np.isfinite(gpu).all()
    assert close >= close_at_least, (name, close)
    assert value <= rmse_at_most, (name, value)
    if exact is not None:
        from conftest import verification_build_equals_oracle
        verify_ctx, oracle, scene, w, h, spp, bounces = exact
        exact_image = verification_build_equals_oracle(verify_ctx, oracle, scene, w, h, spp, bounces, name)      # leg A
        on_device = float(np.sqrt(np.mean((gpu[..., :3] - exact_image) ** 2)))
        print(f"IMAGE-METRIC {name}: product vs verification build on the device {on_device:.3e}, product vs oracle {rmse(gpu, cpu):.3e}")
        assert abs(on_device - rmse(gpu, cpu)) <= 0.1 * on_device + 1e-6, (name, on_device, rmse(gpu, cpu))       # leg B
        if name in RECORDED_FAST_VS_EXACT:      # leg C: the divergence itself, held to what was recorded
            assert on_device <= 1.5 * RECORDED_FAST_VS_EXACT[name] + 5e-6, (name, on_device, RECORDED_FAST_VS_EXACT[name])


def test_background_colour_G10(ctx):
    """RendererFixture.render_background_color, tests/OptiXRendererTests/RendererTest.h:142-153: 16x12, +-1e-4 (half output)."""
    import torch
    scene = Scene("empty_ortho", param0=16, param1=12)
    ctx.upload_scene(scene)
    ctx.set_frame(16, 12)
    out = torch.zeros((12, 16, 4), dtype=torch.float16, device="cuda")
    ctx.render_pass(scene.camera(16, 12), out.data_ptr(), 16, synchronize=True)
    torch.cuda.synchronize()
    px = out.float().cpu().numpy()
    assert np.all(np.abs(px[..., 0] - 0.1) <= 1e-4) and np.all(np.abs(px[..., 1] - 0.5) <= 1e-4) and np.all(np.abs(px[..., 2] - 2.0) <= 1e-4)
    assert np.all(px[..., 3] == 1.0)


def test_cornell_image_matches_oracle(ctx, oracle_q, cornell):
    """Full path: 8 accumulations of the Cornell box (reference materials), bounces 4. Shading uses sin/cos/pow whose
    last ulp differs between glibc and ocml, so individual paths may take different discrete decisions: the bar is
    statistical, but tight: measured on the MI355X (round 2) every pixel is within 1e-3 relative and the RMSE is 1e-6; the bar is >= 99.9 % of the pixels and RMSE <= 1e-4."""
    w, h, spp = 64, 36, 8
    gpu, gc = render_gpu(ctx, cornell, w, h, spp, 4)
    cpu, cc, _ = oracle_q.render(cornell.desc, cornell.state, cornell.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("cornell", gpu, cpu, 0.999, 1e-4)
    assert gc["camera_rays"] == cc["camera_rays"] == w * h * spp
    for key in ("closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(gc[key] - cc[key]) <= 0.002 * cc[key], (key, gc[key], cc[key])


def test_diffuse_cornell_image_matches_oracle(ctx, oracle_q):
    """BASELINE.json config 2 material set (all Diffuse), same bar as above."""
    scene = Scene("cornell", diffuse_only=True)
    w, h, spp = 64, 36, 8
    gpu, _ = render_gpu(ctx, scene, w, h, spp, 4)
    cpu, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("cornell_diffuse", gpu, cpu, 0.999, 1e-4)


def test_atrium_image_matches_oracle(ctx, oracle_q, atrium, verify_ctx):
    """DefaultShading with coat + metals + directional and sphere light, 20 k triangles."""
    w, h, spp = 48, 27, 4
    gpu, _ = render_gpu(ctx, atrium, w, h, spp, 4)
    cpu, _, _ = oracle_q.render(atrium.desc, atrium.state, atrium.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("atrium_20k", gpu, cpu, 0.995, 1e-2, band=2e-3, exact=(verify_ctx, oracle_q, atrium, w, h, spp, 4))      # measured: 0.9992 of the pixels close


def test_textured_atrium_image_matches_oracle(ctx, oracle_q, verify_ctx):
    """Round 4: the stand-in with what the real Sponza brings and the plain one does not -- a tint / roughness texture on every material, cut-out cloth banners
    (30 % of the triangles are not statically opaque) -- so the FULL kernels render it: texture samplers in k_shade, coverage lookups for shadow rays and
    k_trace_wide8<..., COVERAGE = true>. Image against the oracle under the usual statistical bar; the ray counters agree; and the scene is not the plain one
    (its image differs, and rays pass through the banners' holes: more shadow rays reach their light)."""
    textured = Scene("atrium", param0=20000, param1=3, textured=True)
    plain = Scene("atrium", param0=20000, param1=3)
    assert textured.desc.texture_count == 10 and textured.desc.triangle_count == plain.desc.triangle_count
    w, h, spp = 64, 36, 8
    ctx.upload_scene(textured)
    gpu, counters = render_gpu(ctx, textured, w, h, spp, 4)
    cpu, cpu_counters, _ = oracle_q.render(textured.desc, textured.state, textured.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("atrium_20k_textured", gpu, cpu, 0.99, 1e-2, band=2e-3, exact=(verify_ctx, oracle_q, textured, w, h, spp, 4))
    for key in ("camera_rays", "closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(counters[key] - cpu_counters[key]) <= max(4, 0.002 * cpu_counters[key]), (key, counters[key], cpu_counters[key])
    reference, _ = render_gpu(ctx, plain, w, h, spp, 4)
    assert rmse(gpu, reference) > 1e-2


def test_tiling_and_batching_are_bit_invariant(ctx, cornell):
    """The image must not depend on how pixels are split over GPUs (tile_stride) or on samples_per_pass:
    the RNG is a pure function of (pixel, accumulation, bounce) and every path owns its radiance slot."""
    w, h, spp = 40, 24, 4
    full, _ = render_gpu(ctx, cornell, w, h, spp, 4)
    batched, _ = render_gpu(ctx, cornell, w, h, spp, 4, samples_per_pass=4)
    assert np.array_equal(full, batched)
    tiles_x = (w + 7) // 8
    for stride in (2, 3):
        assembled = np.zeros_like(full)
        for phase in range(stride):
            part, _ = render_gpu(ctx, cornell, w, h, spp, 4, tile_phase=phase, tile_stride=stride)
            for k in range(part.shape[0]):
                tile = (k // 64) * stride + phase
                x, y = (tile % tiles_x) * 8 + (k % 64) % 8, (tile // tiles_x) * 8 + (k % 64) // 8
                if x < w and y < h and tile < tiles_x * ((h + 7) // 8):
                    assembled[y, x] = part[k]
        assert np.array_equal(assembled, full), stride


def test_scatter_tiles_roundtrip(ctx):
    import torch
    w, h, ranks = 40, 24, 3
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    per_rank = ((tiles_x * tiles_y + ranks - 1) // ranks) * 64
    compact = torch.arange(ranks * per_rank * 4, dtype=torch.int16, device="cuda").reshape(ranks, per_rank, 4)
    out = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda")
    ctx.scatter_tiles(compact.data_ptr(), per_rank, ranks, w, h, out.data_ptr(), w)
    ctx.synchronize()
    torch.cuda.synchronize()
    o, c = out.cpu().numpy(), compact.cpu().numpy()
    for y in range(h):
        for x in range(w):
            tile = (y // 8) * tiles_x + x // 8
            assert np.array_equal(o[y, x], c[tile % ranks, (tile // ranks) * 64 + (x % 8) + (y % 8) * 8])


def test_errors_are_reported_not_swallowed(ctx):
    lib = ctx.lib
    assert lib.hipr_upload_scene(ctx.handle, None) == -1
    assert b"null scene" in lib.hipr_last_error()
    bad = capi.HiprFrameDesc(0, 0, 0, 1, 1)
    assert lib.hipr_set_frame(ctx.handle, C.byref(bad)) == -1


AOV_ENTRIES = [("depth", capi.ENTRY_DEPTH), ("albedo", capi.ENTRY_ALBEDO), ("tint", capi.ENTRY_TINT), ("roughness", capi.ENTRY_ROUGHNESS),
               ("shading_normal", capi.ENTRY_SHADING_NORMAL), ("primitive_id", capi.ENTRY_PRIMITIVE_ID), ("denoiser_albedo", capi.ENTRY_DENOISER_ALBEDO)]


@pytest.mark.parametrize("name,entry", AOV_ENTRIES)
@pytest.mark.parametrize("scene_name", ["cornell", "atrium"])
def test_aov_entry_points_match_oracle(ctx, oracle_q, cornell, atrium, scene_name, name, entry):
    """The visualisation backends (SimpleRGPs.cu:227-340): depth, tint, roughness, normals and primitive ids are one
    closest hit + attribute fetch (tolerance 1e-5 absolute: the camera ray and hit are bit-exact, the colour is a few
    f32 operations); albedo goes through the rho tables and pow (tolerance 2e-3). denoiser_albedo is the feature image of the reference's
    denoising backend (SimpleRGPs.cu:149-201): every material as DefaultShading, light hits as radiance / (1 + radiance)."""
    scene = cornell if scene_name == "cornell" else atrium
    w, h, spp = 48, 27, 2
    ctx.set_entry_point(entry)
    try:
        gpu, _ = render_gpu(ctx, scene, w, h, spp, 4)
    finally:
        ctx.set_entry_point(capi.ENTRY_PATH_TRACING)
    cpu, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, entry=entry)
    tol = 2e-3 if "albedo" in name else 1e-5
    if name == "depth":   # world units, scaled by the scene size on both sides
        tol = 1e-5 * max(1.0, float(np.nanmax(np.where(np.isfinite(cpu[..., 0]), cpu[..., 0], 0.0))))
    g, c = gpu[..., :3], cpu[..., :3]
    if name == "depth":
        # A miss adds |origin - 1e30 * direction|, which overflows f32 to +inf (as in the reference), and the running mean
        # of inf is NaN from the second accumulation on: the non-finite pixels must be the same set on both sides.
        assert np.array_equal(np.isfinite(g), np.isfinite(c))
        assert 0.05 < np.isfinite(c).mean()
        g, c = np.where(np.isfinite(g), g, 0.0), np.where(np.isfinite(c), c, 0.0)
    err = np.abs(g - c).max(axis=-1)
    assert (err <= tol).mean() >= 0.999, (name, float(err.max()), float((err <= tol).mean()))
    assert np.isfinite(g).all()


def test_tint_quad_G10(ctx):
    """RendererFixture.render_tint, RendererTest.h:155-172: 4x3 ortho quad, red = (x+.5)/4, green = (y+.5)/3, +-0.003."""
    w, h = 4, 3
    scene = Scene("quad", param0=w, param1=h)
    ctx.set_entry_point(capi.ENTRY_TINT)
    try:
        gpu, _ = render_gpu(ctx, scene, w, h, 1, 4)
    finally:
        ctx.set_entry_point(capi.ENTRY_PATH_TRACING)
    xs = (np.arange(w) + 0.5) / w
    ys = (np.arange(h) + 0.5) / h
    assert np.abs(gpu[..., 0] - xs[None, :]).max() <= 0.003
    assert np.abs(gpu[..., 1] - ys[:, None]).max() <= 0.003


@pytest.mark.parametrize("scene_name", ["cornell", "atrium"])
def test_wavefront_count_does_not_change_the_image(ctx, cornell, atrium, scene_name):
    """hipr_set_wavefront_count(n) splits a pass into n shares of the path slots on n streams (one shades while another traces;
    the default is 2). Every path still owns its radiance slot and sees the same kernel order, so the accumulation must not change."""
    scene = cornell if scene_name == "cornell" else atrium
    w, h, spp = 640, 416, 2          # 266 240 pixels: enough for four wavefronts (one per 65 536 path slots at most)
    images, counters = {}, {}
    try:
        for count in (1, 2, 3, 4):
            ctx.set_wavefront_count(count)
            images[count], counters[count] = render_gpu(ctx, scene, w, h, spp, 4)
    finally:
        ctx.set_wavefront_count(0)
        ctx.set_frame(w, h)
    for count in (2, 3, 4):
        assert np.array_equal(images[1], images[count]), count
        for key in ("camera_rays", "closest_rays", "shadow_rays"):
            assert counters[1][key] == counters[count][key], (count, key)


def test_slot_order_and_the_deal_to_the_wavefronts_do_not_change_the_image(ctx, atrium):
    """Round 4: path slots are pixel-major (slot = pixel * samples_per_pass + sample) and dealt to the wavefronts in groups of 64, round robin; k_accumulate
    brings the samples of 256 pixels through LDS eight at a time. None of it may show: a frame whose size is no multiple of the 8 x 8 tiles, samples per pass
    that neither divide 64 nor fit one LDS chunk (3, 5, 30), one to three wavefronts -- the running mean equals the one of single-sample passes on one
    wavefront, bit for bit, and so do the ray counters; folding a traced pass in two parts (hipr_accumulate_samples) gives the same frame as folding it at once."""
    import ctypes as C
    w, h, accumulations = 652, 412, 30           # 82 x 52 tiles, the last column and row partly outside the frame
    try:
        ctx.set_wavefront_count(1)
        reference, reference_counters = render_gpu(ctx, atrium, w, h, accumulations, 4, samples_per_pass=1)
        for samples, count in ((3, 2), (5, 3), (30, 2), (15, 1)):
            ctx.set_wavefront_count(count)
            image, counters = render_gpu(ctx, atrium, w, h, accumulations, 4, samples_per_pass=samples)
            assert np.array_equal(image, reference), (samples, count)
            for key in ("camera_rays", "closest_rays", "shadow_rays", "shaded_hits"):
                assert counters[key] == reference_counters[key], (samples, count, key)
        # a pass of 10 traced once and folded as 4 + 6 samples
        ctx.set_wavefront_count(2)
        ctx.upload_scene(atrium)
        ctx.set_frame(w, h, 0, 1, 10)
        for first in range(0, accumulations, 10):
            cam = atrium.camera(w, h, accumulations=first, max_bounce_count=4)
            ctx._check(ctx.lib.hipr_trace_pass(ctx.handle, C.byref(cam)), "hipr_trace_pass")
            ctx._check(ctx.lib.hipr_accumulate_samples(ctx.handle, 0, 4, first, None, 0, 0), "hipr_accumulate_samples")
            ctx._check(ctx.lib.hipr_accumulate_samples(ctx.handle, 4, 6, first + 4, None, 0, 1), "hipr_accumulate_samples")
        assert np.array_equal(ctx.read_accumulation(), reference)
    finally:
        ctx.set_wavefront_count(0)
        ctx.set_frame(w, h)


def test_tessellated_cornell_renders_the_same_image(ctx, oracle_q, cornell_tessellated):
    """The BVH2 kernels end to end: the tessellated box is the same surface set, so the image matches the oracle's (BVH2) render
    of it under the usual statistical bar."""
    w, h, spp = 64, 36, 8
    gpu, _ = render_gpu(ctx, cornell_tessellated, w, h, spp, 4)
    cpu, _, _ = oracle_q.render(cornell_tessellated.desc, cornell_tessellated.state, cornell_tessellated.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("cornell_tessellated", gpu, cpu, 0.999, 1e-4)


@pytest.mark.parametrize("scene_name", ["cornell", "atrium"])
def test_environment_map_image_matches_oracle(ctx, oracle_q, scene_name, verify_ctx):
    """A latitude-longitude environment map (procedural sky with a sun): importance sampled through the presampled light
    samples in next event estimation, evaluated with MIS where paths escape. Pixels that see the sky directly are a texture
    lookup (1e-4 relative); the lit image holds to the usual statistical bar."""
    scene = Scene("cornell", environment=True) if scene_name == "cornell" else Scene("atrium", param0=20000, param1=3, environment=True)
    assert scene.desc.environment and scene.desc.light_count >= 2
    w, h, spp = 64, 36, 8
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 4)
    cpu, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("environment_" + scene_name, gpu, cpu, 0.99, (1e-4 if scene_name == "cornell" else 0.03) * max(1.0, float(cpu[..., :3].mean())), band=2e-3,
              exact=(verify_ctx, oracle_q, scene, w, h, spp, 4))
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])
    # the environment changes the picture: brighter than the constant-tint render of the same scene
    plain = Scene("cornell") if scene_name == "cornell" else Scene("atrium", param0=20000, param1=3)
    dark, _ = render_gpu(ctx, plain, w, h, 2, 4)
    assert abs(float(gpu[..., :3].mean()) - float(dark[..., :3].mean())) > 0.01


@pytest.mark.parametrize("binary_container", [False, True])
def test_loaded_gltf_image_matches_oracle(ctx, oracle_q, tmp_path, binary_container):
    """file -> glTFLoader -> SimpleViewer defaults -> flattened scene -> the C-ABI: a glTF scene (instanced meshes under TRS and
    matrix nodes, sheared copies, unindexed geometry) renders the same image on the device as in the oracle."""
    pytest.importorskip("scipy")
    from test_loaders_cpu import _write_synthetic_gltf
    scene = Scene("file:" + _write_synthetic_gltf(str(tmp_path), binary_container))
    assert scene.desc.triangle_count == 260 and scene.desc.light_count == 1
    w, h, spp = 64, 36, 8
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 4)
    cpu, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("gltf_binary" if binary_container else "gltf_text", gpu, cpu, 0.999, 1e-4, band=2e-3)
    assert gc["closest_rays"] == cc["closest_rays"] or abs(gc["closest_rays"] - cc["closest_rays"]) <= 0.003 * cc["closest_rays"]
    # the camera placed from the scene bounds (a scene size away from its centre) sees the model: part of the frame is not the sky-blue environment
    sky = np.array([0.68, 0.92, 1.0])
    assert (np.abs(gpu[..., :3] - sky).max(axis=-1) > 0.05).mean() > 0.01


@pytest.mark.parametrize("coat", [False, True])
def test_material_scene_image_matches_oracle(ctx, oracle_q, coat, verify_ctx):
    """BASELINE config 3: the viewer's material scene (seven shader balls from rough dielectric to polished gold, optionally
    coated, on the checkered floor whose texture carries roughness in alpha, repeat wrapping, nearest magnification), 179 k
    triangles through the wide-BVH kernels, 32 bounces."""
    scene = Scene("material", coat=coat)
    assert scene.camera(64, 36).max_bounce_count == 32
    w, h, spp = 96, 54, 4
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 32)
    assert ctx.trace_variant() == capi.TRACE_WIDE8_PERSISTENT
    cpu, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=32), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("material_coat" if coat else "material", gpu, cpu, 0.995, 1e-2, band=2e-3, exact=(verify_ctx, oracle_q, scene, w, h, spp, 32))
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])
    # the floor's checker is there: neighbouring texels of very different brightness below the horizon
    lower = gpu[: h // 3, :, :3].mean(axis=-1)
    assert lower.max() > 4 * lower.min()


def test_glass_scene_image_matches_oracle(ctx, oracle_q, verify_ctx):
    """SURVEY 8(f)2, the viewer's glass scene (apps/SimpleViewer/Scenes/Glass.cpp): TransmissiveShading at image level -- a frosted
    glass shader ball, a smooth lens and a diamond (total internal reflection, 32 bounces) on the textured floor under a
    directional and a large sphere light. Refraction chains amplify last-ulp differences, so a few more pixels than elsewhere
    may leave the 2e-3 band; the ray counts still agree to 0.3 %."""
    scene = Scene("glass")
    assert scene.desc.light_count == 2 and scene.camera(64, 36).max_bounce_count == 32
    w, h, spp = 96, 54, 4
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 32)
    cpu, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=32), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("glass", gpu, cpu, 0.99, 2e-2, band=2e-3, exact=(verify_ctx, oracle_q, scene, w, h, spp, 32))      # measured: 0.9992 of the pixels close; RMSE 7e-3 (round 2: 5e-6)
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])


def test_spot_light_image_matches_oracle(ctx, oracle_q):
    """a spot light (disc emitter with a cone, SpotLightImpl.h) next to the Cornell box's sphere light: sampled in next
    event estimation, hit by paths, part of closest-hit selection like the sphere light."""
    scene, plain = Scene("cornell", spot=True), Scene("cornell")
    assert scene.desc.light_count == 2 and scene.desc.lights[1].flags != scene.desc.lights[0].flags      # a sphere and a spot light
    w, h, spp = 64, 36, 8
    gpu, gc = render_gpu(ctx, scene, w, h, spp, 4)
    cpu, cc, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    image_bar("spot", gpu, cpu, 0.999, 1e-4, band=2e-3)
    for key in ("closest_rays", "shadow_rays"):
        assert abs(gc[key] - cc[key]) <= 0.003 * cc[key], (key, gc[key], cc[key])
    dark, _ = render_gpu(ctx, plain, w, h, spp, 4)
    assert float(gpu[..., :3].mean()) > 1.03 * float(dark[..., :3].mean())      # the spot adds light (half the frame is sky)


@pytest.mark.parametrize("scene_name", ["cornell", "atrium_full"])
def test_full_size_properties(ctx, scene_name):
    """BASELINE.json's sizes (1920x1080; the 251 k-triangle atrium of configs[3]) through size-independent properties, where the
    oracle would take minutes: the frame is finite, every camera path is traced, a frame split over two tile phases (what two
    GPUs would render) assembles bit for bit into the single-GPU frame, batching four accumulations is bit-identical to four
    passes, and the running mean converges (4 and 8 accumulations agree within Monte Carlo noise)."""
    scene = Scene("cornell") if scene_name == "cornell" else Scene("atrium", param0=260000, param1=1)
    w, h = 1920, 1080
    full, counters = render_gpu(ctx, scene, w, h, 4, 4, samples_per_pass=4)
    assert np.isfinite(full).all() and counters["camera_rays"] == 4 * w * h
    assert counters["closest_rays"] >= counters["camera_rays"] and counters["shadow_rays"] > 0

    one_by_one, _ = render_gpu(ctx, scene, w, h, 4, 4, samples_per_pass=1)
    assert np.array_equal(one_by_one, full)

    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    assembled = np.zeros_like(full)
    for phase in range(2):
        part, _ = render_gpu(ctx, scene, w, h, 4, 4, samples_per_pass=4, tile_phase=phase, tile_stride=2)
        k = np.arange(part.shape[0])
        tile = (k // 64) * 2 + phase
        x, y = (tile % tiles_x) * 8 + (k % 64) % 8, (tile // tiles_x) * 8 + (k % 64) // 8
        valid = (x < w) & (y < h) & (tile < tiles_x * tiles_y)
        assembled[y[valid], x[valid]] = part[valid]
    assert np.array_equal(assembled, full)

    eight, _ = render_gpu(ctx, scene, w, h, 8, 4, samples_per_pass=4)      # twice the accumulations: the same image within Monte Carlo noise
    assert abs(float(full[..., :3].mean()) - float(eight[..., :3].mean())) < 0.01 * float(full[..., :3].mean())
    assert rmse(full, eight) < 0.75 * rmse(full, np.zeros_like(full))


def test_camera_rays_match_the_reference_ground_truth(ctx, cornell):
    """K1 against the 'Unity QED' rays of the reference's camera tests (BifrostTests/Scene/CameraTest.h:305-383) and its
    orthographic expectations (:78-111), for translated and rotated cameras."""
    from test_host_cpu import check_camera_rays_against_the_reference
    ctx.upload_scene(cornell)

    def generate(cam, w, h, pixels):
        ctx.set_frame(w, h)
        o, d, px = ctx.debug_generate(cam, 0)
        index = {int(p): i for i, p in enumerate(px)}
        rows = [index[int(x) | (int(y) << 16)] for x, y in pixels]
        return o[rows], d[rows]

    check_camera_rays_against_the_reference(generate)


@pytest.mark.gpu
def test_parallelogram_items_of_any_vertex_order_bit_exact(ctx, oracle_q, tmp_path):
    """Nine parallelograms whose two triangles are stored in every combination of vertex rotations (all selector values of the
    exhaustive-search items): t, barycentrics, triangle id and counters of k_trace_closest_small equal the oracle's bit for bit,
    and so does the shadow kernel's transmittance."""
    from test_host_cpu import parallelogram_rays, write_parallelogram_rotations_obj
    scene = Scene("file:" + write_parallelogram_rotations_obj(tmp_path / "rotations.obj"))
    ctx.upload_scene(scene)
    assert ctx.trace_variant() == capi.TRACE_EXHAUSTIVE
    rays = parallelogram_rays(40000, 9)
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    skip[::5] = np.random.default_rng(1).integers(0, 18, len(skip[::5]))
    ctx.set_instrumentation(True)
    ctx.reset_counters()
    gpu = ctx.debug_trace_closest(rays, skip)
    counters = ctx.counters()
    ctx.set_instrumentation(False)
    cpu, (_, tested) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=0, with_lights=True)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert counters["closest_triangles"] == tested == 9 * len(rays)
    assert set(gpu[:, 3].view(np.uint32).tolist()) >= set(range(18))
    rays[:, 7] = np.random.default_rng(2).uniform(0.5, 6.0, len(rays))
    assert np.array_equal(ctx.debug_trace_shadow(rays), oracle_q.trace_shadow(scene.desc, rays, use_bvh=0)[0])
