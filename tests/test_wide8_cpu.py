"""The 8-wide tree with leaf records (include/hiprenderer_c.h "wide8", host/Wide8Builder.cpp) on the CPU: structure (every triangle in exactly one record,
pairs share an edge, quantised child boxes contain what is below them), the oracle's search over it against its other searches, refit, and the range
checks of the public header. The HIP kernel that walks the tree is held to the oracle's search bit for bit by the -m gpu tests."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from bifrost3d_amd import capi
from bifrost3d_amd.host import Scene
from oracle_bindings import get_oracle
from test_coverage_cpu import cornell_box_rays, deep_chain_rays, opacity_rays, write_deep_chain_obj

ROOT = Path(__file__).resolve().parent.parent
NONE = 0xFFFFFFFF


@pytest.fixture(scope="module")
def oracle():
    return get_oracle(True)


@pytest.fixture(autouse=True)
def geometric_search(request, oracle):
    """The tests of this file compare the 8-wide search with the other searches as GEOMETRIC searches: its stepping over hits on the back of one-sided
    triangles (hipr_set_backface_culling, on by default) is off for them, and tested by the cases marked `culling` at the end."""
    oracle.set_backface_culling("culling" in request.keywords)
    yield
    oracle.set_backface_culling(True)


def slots_of(desc):
    return np.ctypeslib.as_array(C.cast(desc.wide8_slots, C.POINTER(C.c_uint32)), shape=(desc.wide8_slot_count, 16)).copy()


def walk(desc):
    """Yields (slot, words, is_node, depth) for every slot reachable from the root, each exactly once."""
    slots = slots_of(desc)
    seen = np.zeros(len(slots), bool)
    stack = [(0, True, 1)]
    seen[0] = True
    while stack:
        slot, is_node, depth = stack.pop()
        yield slot, slots[slot], is_node, depth
        if not is_node:
            continue
        w = slots[slot]
        base, valid, inner = int(w[3]) & 0xFFFFFF, int(w[3]) >> 24, int(w[2]) >> 24
        rank = 0
        for position in range(8):
            if valid >> position & 1:
                child = base + rank
                rank += 1
                assert not seen[child], "a slot is reached twice"
                seen[child] = True
                stack.append((child, bool(inner >> position & 1), depth + 1))
    assert seen.all(), "unreachable slots"


def node_child_boxes(desc, w):
    """Decoded child boxes [position] -> (lo, hi) of a node, the way the traversal decodes them (float32 origin, power-of-two scale)."""
    packed = int(w[0]) | int(w[1]) << 32
    gmin, cell = np.array(list(desc.wide8_grid_min), np.float32), np.array(list(desc.wide8_grid_cell), np.float32)
    boxes = {}
    q = w[4:16].view(np.uint8).reshape(2, 3, 8)      # [lo / hi][axis][position]
    valid = int(w[3]) >> 24
    origin = np.zeros(3, np.float64)
    scale = np.zeros(3, np.float64)
    for a in range(3):
        m = (packed >> (21 * a)) & 0x1FFFFF
        origin[a] = float(np.float32(np.float64(np.float32(m)) * np.float64(cell[a]) + np.float64(gmin[a])))      # fma(float(m), cell, min): one rounding
        scale[a] = 2.0 ** (int((int(w[2]) >> (8 * a)) & 0xFF) - 127)
    for position in range(8):
        if valid >> position & 1:
            boxes[position] = (origin + q[0, :, position] * scale, origin + q[1, :, position] * scale)
        else:
            assert (q[0, :, position] == 255).all() and (q[1, :, position] == 0).all()
    return boxes


@pytest.mark.parametrize("name,kwargs", [("cornell", dict(param0=3)), ("atrium", dict(param0=20000, param1=3)), ("opacity", dict(param0=8)), ("material", dict())])
def test_structure(name, kwargs):
    scene = Scene(name, **kwargs)
    d = scene.desc
    assert d.wide8_slot_count > 0 and capi.load_library().hipr_validate_scene(C.byref(d)) == 0
    tris = scene.triangles()                       # (n, 12): v0, v1, v2, instance, primitive, flags (leaf order)
    positions = tris[:, :9].view(np.float32).reshape(-1, 3, 3)
    count = np.zeros(d.triangle_count, int)
    exact = {}                                     # slot -> exact box of what is below it
    order = list(walk(d))
    height = max(depth for _, _, is_node, depth in order if is_node)
    assert height == d.wide8_height
    pairs = 0
    for slot, w, is_node, _ in order:
        if is_node:
            continue
        f = w[:12].view(np.float32)
        a, e1, e2, e3 = f[0:3], f[3:6], f[6:9], f[9:12]
        ia, ib, flags = int(w[12]), int(w[13]), int(w[14])
        count[ia] += 1
        lo, hi = positions[ia].min(axis=0), positions[ia].max(axis=0)
        # record corner k of A: weights (w, u, v) -> the selectors say which of them belong to A's vertices 1 and 2
        corners_a = [a, a + e1, a + e2]
        su, sv = (flags >> 8) & 3, (flags >> 10) & 3
        assert {su, sv} <= {0, 1, 2} and su != sv
        assert np.allclose(corners_a[su], positions[ia][1], rtol=0, atol=1e-5 * (1 + np.abs(positions[ia]).max()))
        assert np.allclose(corners_a[sv], positions[ia][2], rtol=0, atol=1e-5 * (1 + np.abs(positions[ia]).max()))
        assert bool(flags & 1) == bool(int(tris[ia, 11]) & 1)
        if ib != NONE:
            pairs += 1
            count[ib] += 1
            assert tris[ia, 9] == tris[ib, 9], "pairs are made inside one instance"
            corners_b = [a, a + e2, a + e3]
            su, sv = (flags >> 12) & 3, (flags >> 14) & 3
            assert su != sv
            assert np.allclose(corners_b[su], positions[ib][1], rtol=0, atol=1e-5 * (1 + np.abs(positions[ib]).max()))
            assert np.allclose(corners_b[sv], positions[ib][2], rtol=0, atol=1e-5 * (1 + np.abs(positions[ib]).max()))
            shared = sum(any((p == q).all() for q in positions[ib]) for p in positions[ia])
            assert shared == 2
            assert bool(flags & 2) == bool(int(tris[ib, 11]) & 1)
            lo, hi = np.minimum(lo, positions[ib].min(axis=0)), np.maximum(hi, positions[ib].max(axis=0))
        exact[slot] = (lo.astype(np.float64), hi.astype(np.float64))
    assert (count == 1).all(), "every triangle in exactly one record"
    if name != "material":
        assert pairs > 0
    # bottom up: children have larger slot numbers than their parents
    for slot, w, is_node, _ in sorted(order, key=lambda item: -item[0]):
        if not is_node:
            continue
        boxes = node_child_boxes(d, w)
        base, valid = int(w[3]) & 0xFFFFFF, int(w[3]) >> 24
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        rank = 0
        for position in range(8):
            if not (valid >> position & 1):
                continue
            child = base + rank
            rank += 1
            assert child > slot
            clo, chi = exact[child]
            qlo, qhi = boxes[position]
            assert (qlo <= clo).all() and (qhi >= chi).all(), "a quantised child box must contain what is below it"
            # and not by much: at most two cells of the node's grid per side
            scale = 2.0 ** (np.array([(int(w[2]) >> (8 * a)) & 0xFF for a in range(3)]) - 127.0)
            assert ((clo - qlo) <= 2.0 * scale + 1e-30).all() and ((qhi - chi) <= 2.0 * scale + 1e-30).all()
            lo, hi = np.minimum(lo, clo), np.maximum(hi, chi)
        exact[slot] = (lo, hi)


def scene_rays(name, scene, n, seed):
    if name == "cornell":
        return cornell_box_rays(n, seed)
    if name == "opacity":
        return opacity_rays(n, seed, tmax=False)
    rng = np.random.default_rng(seed)
    lo = np.array(list(scene.desc.wide8_grid_min))
    extent = np.array(list(scene.desc.wide8_grid_cell)) * 2097151.0
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = lo + rng.random((n, 3)) * extent
    direction = rng.normal(size=(n, 3))
    rays[:, 4:7] = direction / np.linalg.norm(direction, axis=1, keepdims=True)
    rays[:, 7] = np.inf
    return rays


@pytest.mark.parametrize("name,kwargs", [("cornell", dict(param0=3)), ("atrium", dict(param0=20000, param1=3)), ("opacity", dict(param0=8)), ("material", dict())])
def test_the_search_over_the_tree_finds_what_the_other_searches_find(oracle, name, kwargs):
    """Closest hits: the same triangle as the BVH2 search (which equals exhaustive search, test_host_cpu.py) at the same place -- the record solves its
    triangles from the corner they share, so t, u, v may differ in rounding. Shadow rays: the same transmittance."""
    scene = Scene(name, **kwargs)
    d = scene.desc
    rays = scene_rays(name, scene, 30000, 5)
    skip = np.full(len(rays), NONE, np.uint32)
    skip[::5] = np.random.default_rng(1).integers(0, d.triangle_count, len(skip[::5]))
    two, _ = oracle.trace_closest(d, rays, skip, use_bvh=1, with_lights=True)
    eight, (nodes, tris) = oracle.trace_closest(d, rays, skip, use_bvh=3, with_lights=True)
    ids2, ids8 = two[:, 3].view(np.uint32), eight[:, 3].view(np.uint32)
    assert (ids2 != ids8).mean() <= 2e-3           # coincident surfaces may resolve to the other one
    same = (ids2 == ids8) & (ids2 != NONE)
    assert same.mean() > 0.3
    assert np.all(np.abs(two[same, 0] - eight[same, 0]) <= 2e-5 * (1.0 + np.abs(two[same, 0])))
    assert np.abs(two[same, 1:3] - eight[same, 1:3]).max() <= 5e-3 and np.median(np.abs(two[same, 1:3] - eight[same, 1:3])) <= 1e-6
    assert nodes > 0 and tris > 0
    shadow = rays.copy()
    shadow[:, 7] = np.random.default_rng(2).uniform(0.05, 3.0 if name in ("cornell", "opacity") else 30.0, len(rays))
    t2, _ = oracle.trace_shadow(d, shadow, use_bvh=1)
    t8, _ = oracle.trace_shadow(d, shadow, use_bvh=3)
    assert (t2 != t8).mean() <= 1e-3               # a hit exactly at the end of the segment under one rounding and not the other
    if name == "opacity":
        assert ((t8 > 0) & (t8 < 1)).mean() > 0.01   # partial coverage is exercised


def test_images_of_the_two_wide_trees_agree(oracle):
    scene = Scene("atrium", param0=20000, param1=3)
    w, h, spp = 48, 27, 8
    four, c4, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=2)
    eight, c8, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=3)
    assert abs(c4["closest_rays"] - c8["closest_rays"]) <= 0.002 * c4["closest_rays"]
    assert abs(four[..., :3].mean() - eight[..., :3].mean()) <= 0.02 * four[..., :3].mean()
    assert c8["closest_nodes"] < 0.85 * c4["closest_nodes"]          # what the tree is for: fewer node visits per ray


def test_refit_keeps_the_topology_and_the_hits(oracle):
    scene = Scene("cornell", param0=3)
    before = slots_of(scene.desc)
    rays = cornell_box_rays(20000, 31)
    pose = dict(translation=(0.05, -0.30, 0.10), rotation=(0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))), scale=0.3)
    assert scene.move_model(6, **pose) is True
    d = scene.desc
    assert capi.load_library().hipr_validate_scene(C.byref(d)) == 0
    after = slots_of(d)
    nodes = np.array([slot for slot, _, is_node, _ in walk(d) if is_node])
    leaves = np.array([slot for slot, _, is_node, _ in walk(d) if not is_node])
    assert np.array_equal(before[nodes, 3], after[nodes, 3]) and np.array_equal(before[nodes, 2] >> 24, after[nodes, 2] >> 24)      # children and kinds unchanged
    assert np.array_equal(before[leaves, 12:15], after[leaves, 12:15])                                                             # records keep their triangles and selectors
    assert (before[leaves, :12] != after[leaves, :12]).any(axis=1).sum() >= 6                                                       # the moved box's records changed
    two, _ = oracle.trace_closest(d, rays, use_bvh=1, with_lights=False)
    eight, _ = oracle.trace_closest(d, rays, use_bvh=3, with_lights=False)
    assert (two[:, 3].view(np.uint32) != eight[:, 3].view(np.uint32)).mean() <= 2e-3
    # moving back restores the built tree bit for bit
    assert scene.move_model(6, translation=(0.2, -0.35, -0.2), rotation=(0.0, float(np.sin(np.pi / 12)), 0.0, float(np.cos(np.pi / 12))), scale=0.3) is True
    assert np.array_equal(slots_of(scene.desc), before)


def test_deep_trees(oracle, tmp_path):
    """The degenerate chain scene: the tree's height follows the chain, and the search's stack high-water mark stays below it."""
    scene = Scene("file:" + write_deep_chain_obj(tmp_path / "chain.obj", count=200))
    assert 18 <= scene.desc.wide8_height <= 33
    rays = deep_chain_rays(20000, 9)
    oracle.lib.oracle_wide8_stack_high_water(1)
    eight, _ = oracle.trace_closest(scene.desc, rays, use_bvh=3, with_lights=True)
    high = oracle.lib.oracle_wide8_stack_high_water(1)
    assert 12 < high < scene.desc.wide8_height
    two, _ = oracle.trace_closest(scene.desc, rays, use_bvh=1, with_lights=True)
    assert (two[:, 3].view(np.uint32) != eight[:, 3].view(np.uint32)).mean() <= 2e-3


def test_validate_scene_rejects_a_broken_tree():
    lib = capi.load_library()
    scene = Scene("opacity", param0=8)
    desc = scene.desc
    assert lib.hipr_validate_scene(C.byref(desc)) == 0
    Slot = C.c_uint32 * 16

    def mutated(mutate):
        d = capi.HiprSceneDesc()
        C.memmove(C.byref(d), C.byref(desc), C.sizeof(capi.HiprSceneDesc))
        array = (Slot * desc.wide8_slot_count)()
        C.memmove(array, desc.wide8_slots, C.sizeof(Slot) * desc.wide8_slot_count)
        mutate(array)
        d.wide8_slots = C.cast(array, C.POINTER(Slot))
        return d, array

    def rejected(d, what):
        status = lib.hipr_validate_scene(C.byref(d))
        message = lib.hipr_last_error().decode()
        assert status == -1 and what in message, (status, message)

    leaf = next(slot for slot, _, is_node, _ in walk(desc) if not is_node)

    def children_outside(array):
        array[0][3] = (array[0][3] & 0xFF000000) | (desc.wide8_slot_count - 1)
    d, keep = mutated(children_outside)
    rejected(d, "outside")

    def cycle(array):
        array[0][3] = (array[0][3] & 0xFF000000) | 0       # the root's children start at the root
    d, keep = mutated(cycle)
    rejected(d, "reached twice")

    def bad_triangle(array):
        array[leaf][12] = desc.triangle_count + 5
    d, keep = mutated(bad_triangle)
    rejected(d, "references triangles")

    def bad_selector(array):
        array[leaf][14] |= 3 << 8
    d, keep = mutated(bad_selector)
    rejected(d, "corner selector")

    def inner_of_nothing(array):
        valid = array[0][3] >> 24
        empty = next(p for p in range(8) if not (valid >> p & 1)) if valid != 0xFF else None
        if empty is None:
            array[0][3] &= 0x7FFFFFFF
            empty = 7
        array[0][2] |= 1 << (24 + empty)
    d, keep = mutated(inner_of_nothing)
    rejected(d, "8-wide")
    d = capi.HiprSceneDesc()
    C.memmove(C.byref(d), C.byref(desc), C.sizeof(capi.HiprSceneDesc))
    d.wide8_grid_cell[1] = 0.0
    rejected(d, "grid")


# ---- stepping over the back of one-sided surfaces (hipr_set_backface_culling) -------------------------------------------------------------------------------

def refused(scene, rays, hits):
    """Which of the hits the hit program refuses for being on the back of a one-sided surface (MonteCarlo.cu:147-164), decided the way IT decides:
    normalised geometric normal . direction >= 0 on a material that is neither thin-walled, a cut-out nor transmissive."""
    tris = scene.triangles()
    ids = hits[:, 3].view(np.uint32)
    on_triangle = (ids != NONE) & ((ids & 0x80000000) == 0)
    safe = np.where(on_triangle, ids, 0)
    corners = tris[safe][:, :9].view(np.float32).reshape(-1, 3, 3)
    normal = np.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0])
    normal /= np.maximum(np.linalg.norm(normal, axis=1, keepdims=True), 1e-30)
    behind = (normal * rays[:, 4:7]).sum(axis=1) >= 0.0
    one_sided = (tris[safe][:, 11] & capi.TRIANGLE_ONE_SIDED) != 0
    return on_triangle & behind & one_sided


def write_mixed_winding_obj(path):
    """Sheets of quads in three orientations whose second triangle is wound against the first in every other quad (bit 4 of a record's flags), plus a few
    quads standing exactly on a sheet (coincident surfaces). The OBJ default material is one-sided."""
    rng = np.random.default_rng(5)
    vertices, faces = [], []
    for sheet in range(6):
        axis, level = sheet % 3, 0.3 * (sheet // 3) - 0.15 + 0.07 * sheet
        for i in range(6):
            for j in range(6):
                corners = []
                for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                    p = [0.0, 0.0, 0.0]
                    p[axis] = level
                    p[(axis + 1) % 3] = (i + du) / 6.0 - 0.5
                    p[(axis + 2) % 3] = (j + dv) / 6.0 - 0.5
                    corners.append(tuple(p))
                base = len(vertices)
                vertices += corners
                first = (base + 1, base + 2, base + 3)
                second = (base + 1, base + 3, base + 4) if (i + j + sheet) % 2 == 0 else (base + 1, base + 4, base + 3)      # same edge, other way round
                rotate = int(rng.integers(0, 3))
                faces += [first, second[rotate:] + second[:rotate]]
    with open(path, "w") as f:
        for v in vertices:
            f.write("v %.9g %.9g %.9g\n" % v)
        for face in faces:
            f.write("f %d %d %d\n" % face)
    return str(path)


@pytest.mark.culling
@pytest.mark.parametrize("name,kwargs,lo,hi", [("atrium", dict(param0=20000, param1=3), -12.0, 12.0), ("cornell", dict(param0=3), -0.5, 0.5), ("mixed_winding", {}, -0.7, 0.7)])
def test_stepping_over_refused_hits_finds_what_the_retrace_finds(oracle, tmp_path, name, kwargs, lo, hi):
    """With the culling on, the 8-wide search returns for every ray what tracing again from just past every refused hit ends on -- except where a refused
    triangle COINCIDES with another surface (the Cornell boxes stand on the floor: a ray inside a box leaves through its bottom AND the floor at one distance):
    the retrace starts past both, the stepping search finds the floor (include/hiprenderer_c.h, hipr_set_backface_culling)."""
    scene = Scene("file:" + write_mixed_winding_obj(tmp_path / "mixed.obj")) if name == "mixed_winding" else Scene(name, **kwargs)
    if name == "mixed_winding":
        flags = slots_of(scene.desc)[[slot for slot, _, is_node, _ in walk(scene.desc) if not is_node], 14]
        assert ((flags >> 4) & 1).any() and not ((flags >> 4) & 1).all()
    rng = np.random.default_rng(17)
    rays = np.zeros((40000, 8), np.float32)
    rays[:, 0:3] = rng.uniform(lo, hi, (len(rays), 3))
    if name == "atrium":
        rays[:, 1] = np.abs(rays[:, 1]) * 0.7
    d = rng.normal(size=(len(rays), 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 7] = np.inf
    skip = np.full(len(rays), NONE, np.uint32)
    stepped, (nodes_on, _) = oracle.trace_closest(scene.desc, rays, skip, use_bvh=3, with_lights=False)
    oracle.set_backface_culling(False)
    expected, (nodes_off, _) = oracle.trace_closest(scene.desc, rays, skip, use_bvh=3, with_lights=False)
    retraces, chain = 0, []
    pending = refused(scene, rays, expected)
    assert pending.mean() > (0.05 if name != "cornell" else 0.0)
    current = rays.copy()
    for _ in range(64):
        if not pending.any():
            break
        retraces += int(pending.sum())
        chain.append(np.where(pending, expected[:, 0], np.float32(np.nan)))
        current[pending, 3] = np.nextafter(expected[pending, 0], np.float32(np.inf))      # tmin just past the refused hit, as the hit program sets it
        again, (nodes_again, _) = oracle.trace_closest(scene.desc, current[pending], skip[pending], use_bvh=3, with_lights=False)
        nodes_off += nodes_again
        expected[pending] = again
        pending_rays = np.flatnonzero(pending)
        still = refused(scene, current[pending], again)
        pending = np.zeros(len(rays), bool)
        pending[pending_rays[still]] = True
    assert not pending.any()
    different = ~(stepped.view(np.uint32) == expected.view(np.uint32)).all(axis=1)
    # (a) hits inside the facing margin (grazing by 1e-4) are left to the hit program: the stepping search returns the refused hit itself
    left_to_the_hit_program = refused(scene, rays, stepped)
    # (b) a surface at the distance of a refused hit: found by the stepping search, skipped by the retrace
    refused_distances = np.stack(chain, axis=1) if chain else np.zeros((len(rays), 0), np.float32)
    coincident = (np.abs(refused_distances - stepped[:, :1]) <= 4e-7 * np.abs(stepped[:, :1])).any(axis=1)
    assert (different & ~left_to_the_hit_program & ~coincident).sum() == 0
    assert (different & left_to_the_hit_program).mean() <= 2e-4
    assert (different & coincident).mean() <= (1e-2 if name == "cornell" else 1e-4)
    if name != "cornell":
        assert retraces > 0.1 * len(rays) and nodes_on < nodes_off      # one traversal with fewer node visits than the retraces together


@pytest.mark.culling
@pytest.mark.parametrize("name,kwargs,bounces", [("atrium", dict(param0=20000, param1=3), 4), ("opacity", dict(param0=8), 16), ("material", {}, 8), ("cornell", dict(param0=3), 6)])
def test_frames_do_not_depend_on_the_culling(oracle, name, kwargs, bounces):
    """Same paths, same frames, bit for bit; only the refused hits and their retraces are gone (atrium: a sixth of the closest-hit queries)."""
    scene = Scene(name, **kwargs)
    w, h, spp = 64, 36, 4
    camera = scene.camera(w, h, max_bounce_count=bounces)
    stepping, c_on, _ = oracle.render(scene.desc, scene.state, camera, w, h, spp, use_bvh=3)
    oracle.set_backface_culling(False)
    retracing, c_off, _ = oracle.render(scene.desc, scene.state, camera, w, h, spp, use_bvh=3)
    four_wide, c_four, _ = oracle.render(scene.desc, scene.state, camera, w, h, spp, use_bvh=2)
    assert np.array_equal(stepping, retracing)
    assert c_on["shaded_hits"] == c_off["shaded_hits"] and c_on["shadow_rays"] == c_off["shadow_rays"]
    assert c_on["closest_rays"] + c_off["rejected_hits"] - c_on["rejected_hits"] == c_off["closest_rays"]      # one query less per refused hit stepped over
    assert c_off["rejected_hits"] == c_four["rejected_hits"]
    if name == "atrium":
        assert c_off["rejected_hits"] > 0.1 * c_off["closest_rays"] and c_on["rejected_hits"] <= 0.001 * c_off["rejected_hits"]
    if name == "opacity":
        assert c_on["rejected_hits"] == c_off["rejected_hits"] > 0      # coverage below the drawn number: nothing the traversal could decide


@pytest.mark.culling
def test_one_sided_flags_follow_the_materials(oracle):
    scene = Scene("atrium", param0=20000, param1=3)
    d = scene.desc
    tris = scene.triangles()
    one_sided = (tris[:, 11] & capi.TRIANGLE_ONE_SIDED) != 0
    assert 0.3 < one_sided.mean() <= 1.0
    for t in np.random.default_rng(2).integers(0, len(tris), 500):
        m = d.materials[d.instances[int(tris[t, 9])].material_index]
        expected = not (m.flags & (capi.MATERIAL_CUTOUT | capi.MATERIAL_THIN_WALLED)) and m.shading_model != capi.SHADING_TRANSMISSIVE
        assert bool(one_sided[t]) == expected
    # the records carry the flags of their triangles, the winding bit only where there is a second triangle, and a margin
    slots = slots_of(d)
    leaves = np.array([slot for slot, _, is_node, _ in walk(d) if not is_node])
    flags, margin = slots[leaves, 14], slots[leaves, 15].view(np.float32)
    assert np.array_equal((flags >> 2) & 1, one_sided[slots[leaves, 12]].astype(np.uint32))
    paired = slots[leaves, 13] != NONE
    assert np.array_equal(((flags >> 3) & 1)[paired], one_sided[slots[leaves, 13][paired]].astype(np.uint32))
    assert not ((flags >> 4) & 1).any()      # a consistently wound mesh: the second triangle of a pair follows the shared edge the other way (test above: mixed windings)
    assert (margin > 0).all() and np.isfinite(margin).all()
    # a flag on a triangle whose material is two-sided is refused by the validation
    thin = [i for i in range(d.material_count) if d.materials[i].flags & capi.MATERIAL_THIN_WALLED]
    assert thin
    victim = next(t for t in range(len(tris)) if d.instances[int(tris[t, 9])].material_index in thin)
    assert not (tris[victim, 11] & capi.TRIANGLE_ONE_SIDED)
    d.triangles[victim].flags |= capi.TRIANGLE_ONE_SIDED
    try:
        assert capi.load_library().hipr_validate_scene(C.byref(d)) != 0
    finally:
        d.triangles[victim].flags &= ~capi.TRIANGLE_ONE_SIDED
    assert capi.load_library().hipr_validate_scene(C.byref(d)) == 0


def test_the_reinsertion_optimised_bvh2_passes_this_suite():
    """host/BvhOptimizer.cpp (opt-in, HIPR_BVH_REINSERTION = passes): subtrees of the finished BVH2 are taken out and hung where the SAH cost falls most before the
    tree is collapsed. Whatever it does to the topology, the tree must stay a valid one: this file's checks (records pair triangles that share an edge, quantised
    boxes contain what is below them, the oracle's search over the 8-wide tree equals its other searches and its brute force, refit keeps the topology) run again
    in a process that builds every scene with three passes of it. The switch is read once per process, hence the child."""
    import os, subprocess, sys
    env = dict(os.environ, HIPR_BVH_REINSERTION="3")
    env.pop("PYTEST_CURRENT_TEST", None)
    done = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "not reinsertion_optimised", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-1000:]
    assert "passed" in done.stdout
