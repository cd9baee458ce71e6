"""GPU parity at the edges of the input space (through the C-ABI, against the pinned CPU oracle):

  * frames of one pixel, of less than one 8 x 8 tile, and ragged in both directions (the reference's launch covers exactly frame_size,
    OR/Renderer.cpp:1215-1219; HIPRenderer pads tiles with dead lanes);
  * max_bounce_count 0 and 1 (MonteCarlo.cu:230: a path continues while bounces <= max_bounce_count);
  * rays along the axes, with zero and negative-zero direction components, starting on surfaces, of zero length: the compressed trees' slab tests
    see 1 / 0 and 0 * inf there;
  * a mesh with degenerate triangles (repeated vertices, collinear corners, coincident duplicates) on every search.
"""
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
    return get_oracle(True)


def render(ctx, scene, w, h, spp, bounces, batch=1):
    ctx.set_frame(w, h, 0, 1, batch)
    ctx.reset_counters()
    for a in range(0, spp, batch):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=bounces))
    ctx.synchronize()
    return ctx.read_accumulation(), ctx.counters()


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (8, 8), (9, 7), (17, 8), (250, 3)])
def test_tiny_and_ragged_frames(ctx, oracle_q, size):
    w, h = size
    scene = Scene("cornell")
    ctx.upload_scene(scene)
    spp = 8
    gpu, counters = render(ctx, scene, w, h, spp, 4, batch=4)
    cpu, oracle_counters, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, accumulations=0, max_bounce_count=4), w, h, spp, use_bvh=ctx.oracle_search())
    assert gpu.shape[:2] == (h, w) and np.isfinite(gpu).all()
    assert counters["camera_rays"] == w * h * spp == oracle_counters["camera_rays"]      # dead lanes of partial tiles are neither traced nor counted
    relative = np.abs(gpu[..., :3] - cpu[..., :3]) / (np.abs(cpu[..., :3]) + 1e-3)
    assert float((relative.max(axis=-1) <= 1e-3).mean()) >= 0.97, relative.max()
    for key in ("closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(counters[key] - oracle_counters[key]) <= max(2, oracle_counters[key] // 5000), (key, counters[key], oracle_counters[key])


@pytest.mark.parametrize("name,kwargs", [("cornell", {}), ("atrium", dict(param0=20000, param1=3))])
@pytest.mark.parametrize("bounces", [0, 1])
def test_shortest_paths(ctx, oracle_q, name, kwargs, bounces):
    scene = Scene(name, **kwargs)
    ctx.upload_scene(scene)
    w, h, spp = 64, 36, 4
    gpu, counters = render(ctx, scene, w, h, spp, bounces)
    cpu, oracle_counters, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, accumulations=0, max_bounce_count=bounces), w, h, spp, use_bvh=ctx.oracle_search())
    relative = np.abs(gpu[..., :3] - cpu[..., :3]) / (np.abs(cpu[..., :3]) + 1e-3)
    assert np.isfinite(gpu).all() and float((relative.max(axis=-1) <= 1e-3).mean()) >= 0.99
    # a path of max_bounce_count b traces at most b + 1 segments (+ retraces of rejected hits, none in these scenes) and as many shadow rays
    assert counters["camera_rays"] == w * h * spp
    assert counters["closest_rays"] <= (bounces + 2) * w * h * spp and counters["shadow_rays"] <= (bounces + 1) * w * h * spp
    for key in ("closest_rays", "shadow_rays", "shaded_hits"):
        assert abs(counters[key] - oracle_counters[key]) <= max(2, oracle_counters[key] // 5000), (key, counters[key], oracle_counters[key])


def awkward_rays(lo, hi, seed):
    """Rays the slab tests like least: along +-axes (two direction components +0 or -0), in the axis planes (one zero component), from points on the
    scene's own grid planes, with tmin = tmax, and of nearly zero extent."""
    rng = np.random.default_rng(seed)
    rows = []
    for axis in range(3):
        for sign in (1.0, -1.0):
            for zero in (0.0, -0.0):
                d = np.full(3, zero, np.float32)
                d[axis] = sign
                for _ in range(150):
                    o = rng.uniform(lo, hi, 3).astype(np.float32)
                    rows.append(np.concatenate([o, [0.0], d, [np.inf]]))
    for axis in range(3):      # one zero component
        for _ in range(600):
            d = rng.normal(size=3)
            d[axis] = 0.0 if rng.random() < 0.5 else -0.0
            d = d / np.linalg.norm(d)
            rows.append(np.concatenate([rng.uniform(lo, hi, 3), [0.0], d, [np.inf]]))
    for _ in range(600):       # origins snapped to quarter units: on walls, box faces and BVH split planes
        o = np.round(rng.uniform(lo, hi, 3) * 4.0) / 4.0
        d = rng.normal(size=3)
        rows.append(np.concatenate([o, [0.0], d / np.linalg.norm(d), [np.inf]]))
    for _ in range(300):       # tmin well inside the scene, some with tmin beyond every hit
        d = rng.normal(size=3)
        rows.append(np.concatenate([rng.uniform(lo, hi, 3), [rng.uniform(0.0, hi - lo)], d / np.linalg.norm(d), [np.inf]]))
    return np.asarray(rows, np.float32)


@pytest.mark.parametrize("name,kwargs,variant", [("cornell", {}, capi.TRACE_EXHAUSTIVE), ("cornell", dict(param0=12), capi.TRACE_BVH2),
                                                 ("atrium", dict(param0=20000, param1=3), capi.TRACE_WIDE_PERSISTENT), ("atrium", dict(param0=20000, param1=3), capi.TRACE_WIDE8_PERSISTENT)])
def test_awkward_rays_bit_exact(ctx, oracle_q, name, kwargs, variant):
    scene = Scene(name, **kwargs)
    ctx.set_trace_variant(variant)
    try:
        ctx.upload_scene(scene)
        assert ctx.trace_variant() == variant
        lo, hi = (-12.0, 12.0) if name == "atrium" else (-0.5, 0.5)
        rays = awkward_rays(lo, hi, 5)
        skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
        ctx.set_instrumentation(True)
        ctx.reset_counters()
        gpu = ctx.debug_trace_closest(rays, skip)
        counters = ctx.counters()
        ctx.set_instrumentation(False)
        cpu, (nodes, tris) = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
        assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
        assert counters["closest_nodes"] == nodes and counters["closest_triangles"] == tris
        hit = gpu[:, 3].view(np.uint32) != 0xFFFFFFFF
        assert 0.1 < hit.mean() <= 1.0 and np.isfinite(gpu[hit, 0]).all()      # the 8-wide search steps over hits on the back of one-sided surfaces
        # the same rays as shadow rays of finite and of zero extent
        shadow = rays.copy()
        shadow[:, 3] = 0.0
        shadow[:, 7] = np.random.default_rng(6).uniform(0.0, hi - lo, len(rays)).astype(np.float32)
        shadow[::11, 7] = 0.0
        assert np.array_equal(ctx.debug_trace_shadow(shadow), oracle_q.trace_shadow(scene.desc, shadow, use_bvh=ctx.oracle_search())[0])
    finally:
        ctx.set_trace_variant(-1)


def write_degenerate_obj(path):
    """A unit quad and a pyramid over it, salted with triangles of zero area: a corner used twice, three times, collinear corners, and exact duplicates
    of good triangles (coincident surfaces: the lower triangle id wins, MonteCarlo's closest hit has no such tie but the searches must agree)."""
    rng = np.random.default_rng(3)
    vertices = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (2, 0, 0), (0.5, 0.5, 1), (0.5, 0.5, 1)]
    faces = [(1, 2, 3), (2, 4, 3), (1, 2, 5), (1, 1, 2), (6, 7, 6), (3, 4, 6), (1, 3, 6), (2, 4, 6), (1, 2, 6), (1, 2, 3), (3, 4, 6)]
    for _ in range(120):      # a cloud of small triangles, every fourth degenerate
        base = len(vertices)
        c = rng.uniform(-1.0, 2.0, 3)
        a, b, d = c + rng.normal(scale=0.1, size=3), c + rng.normal(scale=0.1, size=3), c + rng.normal(scale=0.1, size=3)
        kind = rng.integers(0, 4)
        if kind == 0:
            d = a + 0.5 * (b - a)      # collinear
        vertices += [tuple(a), tuple(b), tuple(d)]
        faces.append((base + 1, base + 2, base + 3) if kind != 1 else (base + 1, base + 2, base + 1))
    with open(path, "w") as f:
        for v in vertices:
            f.write("v %.9g %.9g %.9g\n" % v)
        for face in faces:
            f.write("f %d %d %d\n" % face)
    return str(path)


@pytest.mark.parametrize("variant", [capi.TRACE_EXHAUSTIVE, capi.TRACE_BVH2, capi.TRACE_WIDE_PERSISTENT, capi.TRACE_WIDE8_PERSISTENT])
def test_degenerate_triangles(ctx, oracle_q, tmp_path, variant):
    scene = Scene("file:" + write_degenerate_obj(tmp_path / "degenerate.obj"))
    ctx.set_trace_variant(variant)
    try:
        ctx.upload_scene(scene)
        assert ctx.trace_variant() == variant
        rng = np.random.default_rng(8)
        rays = np.zeros((30000, 8), np.float32)
        rays[:, 0:3] = rng.uniform(-1.5, 2.5, (len(rays), 3))
        d = rng.normal(size=(len(rays), 3))
        rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
        rays[:, 7] = np.inf
        skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
        gpu = ctx.debug_trace_closest(rays, skip)
        cpu, _ = oracle_q.trace_closest(scene.desc, rays, skip, use_bvh=ctx.oracle_search(), with_lights=True)
        assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
        hit = gpu[:, 3].view(np.uint32) != 0xFFFFFFFF
        assert hit.mean() > 0.03 and np.isfinite(gpu[hit, :3]).all()
        # the image: finite, and the oracle's
        w, h = 64, 36
        image, _ = render(ctx, scene, w, h, 4, 4)
        cpu_image, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, accumulations=0, max_bounce_count=4), w, h, 4, use_bvh=ctx.oracle_search())
        relative = np.abs(image[..., :3] - cpu_image[..., :3]) / (np.abs(cpu_image[..., :3]) + 1e-3)
        assert np.isfinite(image).all() and float((relative.max(axis=-1) <= 1e-3).mean()) >= 0.97
    finally:
        ctx.set_trace_variant(-1)
