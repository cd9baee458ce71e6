"""The oracle's own searches at the edges of the input space (CPU): rays along the axes, with zero / negative-zero direction components, from points on
the scene's grid planes, and a mesh salted with degenerate triangles -- every tree search of the oracle finds what its exhaustive search finds. This is
the root of trust of tests/test_gpu_edges.py, which holds the HIP kernels to these searches bit for bit."""
import numpy as np
import pytest

from bifrost3d_amd.host import Scene
from oracle_bindings import get_oracle
from test_coverage_cpu import hit_records


@pytest.fixture(scope="module")
def oracle():
    return get_oracle(True)


@pytest.fixture(autouse=True)
def geometric_search(oracle):
    """These are comparisons of geometric searches: the 8-wide search's stepping over the back of one-sided triangles is off (tests/test_wide8_cpu.py has its tests)."""
    oracle.set_backface_culling(False)
    yield
    oracle.set_backface_culling(True)


def awkward_rays(lo, hi, seed):
    from test_gpu_edges import awkward_rays as rays      # one definition for both sides
    return rays(lo, hi, seed)


@pytest.mark.parametrize("name,kwargs,lo,hi", [("cornell", dict(param0=12), -0.5, 0.5), ("atrium", dict(param0=20000, param1=3), -12.0, 12.0)])
def test_awkward_rays_find_what_exhaustive_search_finds(oracle, name, kwargs, lo, hi):
    scene = Scene(name, **kwargs)
    rays = awkward_rays(lo, hi, 5)
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    brute, _ = oracle.trace_closest(scene.desc, rays, skip, use_bvh=0, with_lights=False)
    assert np.isfinite(brute[brute[:, 3].view(np.uint32) != 0xFFFFFFFF, 0]).all()
    _, b_instance, b_primitive, _ = hit_records(scene, brute)
    for search in (1, 2, 3):
        found, _ = oracle.trace_closest(scene.desc, rays, skip, use_bvh=search, with_lights=False)
        _, instance, primitive, _ = hit_records(scene, found)
        differs = (instance != b_instance) | (primitive != b_primitive)
        # coincident surfaces (boxes standing on the floor, a ray IN a wall's plane) may resolve to the other triangle at the same distance; nothing else may differ
        both = differs & (found[:, 3].view(np.uint32) != 0xFFFFFFFF) & (brute[:, 3].view(np.uint32) != 0xFFFFFFFF)
        # a ray that STARTS on a surface (origins snapped to the walls' planes) hits it at t = 1e-13 or not at all, by the last bit of the triangle test: the
        # records of the 8-wide tree solve a triangle from the corner its pair shares, the other searches from its first vertex. Paths never ask: a path's
        # next ray starts off the surface (offset_ray_origin) and skips the triangle it left.
        on_surface = np.minimum(np.abs(found[:, 0]), np.abs(brute[:, 0])) <= 1e-9
        both &= ~on_surface
        assert np.all(np.abs(found[both, 0] - brute[both, 0]) <= 1e-4 * (1.0 + np.abs(brute[both, 0]))), search
        lost = differs & ~both & ~on_surface
        assert lost.mean() <= 2e-3, (search, int(lost.sum()), len(rays))      # rays that graze a silhouette edge exactly in a box plane


def test_degenerate_triangles_are_harmless(oracle, tmp_path):
    from test_gpu_edges import write_degenerate_obj
    scene = Scene("file:" + write_degenerate_obj(tmp_path / "degenerate.obj"))
    assert scene.desc.triangle_count == 131 and scene.desc.wide8_slot_count > 0
    rng = np.random.default_rng(8)
    rays = np.zeros((20000, 8), np.float32)
    rays[:, 0:3] = rng.uniform(-1.5, 2.5, (len(rays), 3))
    d = rng.normal(size=(len(rays), 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 7] = np.inf
    skip = np.full(len(rays), 0xFFFFFFFF, np.uint32)
    brute, _ = oracle.trace_closest(scene.desc, rays, skip, use_bvh=0, with_lights=False)
    hit = brute[:, 3].view(np.uint32) != 0xFFFFFFFF
    assert hit.mean() > 0.03 and np.isfinite(brute[hit, :3]).all()
    # no hit on a zero-area triangle: its barycentrics would be the quotient of two zeros
    tris = scene.triangles()
    ids = brute[hit, 3].view(np.uint32)
    corners = tris[ids][:, :9].view(np.float32).reshape(-1, 3, 3)
    area = np.linalg.norm(np.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0]), axis=1)
    assert (area > 0).all()
    _, b_instance, b_primitive, _ = hit_records(scene, brute)
    for search in (1, 2, 3):
        found, _ = oracle.trace_closest(scene.desc, rays, skip, use_bvh=search, with_lights=False)
        _, instance, primitive, _ = hit_records(scene, found)
        differs = (instance != b_instance) | (primitive != b_primitive)
        both = differs & hit & (found[:, 3].view(np.uint32) != 0xFFFFFFFF)
        assert np.all(np.abs(found[both, 0] - brute[both, 0]) <= 1e-4 * (1.0 + np.abs(brute[both, 0])))      # the duplicated triangles: same place, other id
        assert (differs & ~both).mean() <= 1e-3
    image, _, _ = oracle.render(scene.desc, scene.state, scene.camera(48, 27, accumulations=0, max_bounce_count=4), 48, 27, 4, use_bvh=3)
    assert np.isfinite(image).all()
