"""Model and image ingestion on the CPU: the host library's PNG decoder against Pillow, and the glTF loader's world-space
triangles against an independent evaluation of the same file in numpy (glTF 2.0 node transforms, then the reference's
right- to left-handed mirroring, extensions/glTFLoader/glTFLoader/glTFLoader.cpp:279-285,665-681).

The files are written by the tests. When the reference checkout is present (this container only) its own model and image
resources (apps/SimpleViewer/Resources) are run through the same checks; they are not copied into the repository.
"""
import base64
import json
import os
import struct

import numpy as np
import pytest

from bifrost3d_amd import capi, host

REFERENCE_RESOURCES = "/root/reference/apps/SimpleViewer/Resources"


# ---------------------------------------------------------------------------------------------------------------------
# PNG
# ---------------------------------------------------------------------------------------------------------------------

def _expected_pixels(image):
    """What StbImageLoader's conventions make of a Pillow image: 8 bit, grey / grey+alpha / RGB / RGBA, 2 channels widened to RGBA."""
    from PIL import Image
    if image.mode in ("I;16", "I;16B", "I"):
        return (np.asarray(image).astype(np.uint32) >> 8).astype(np.uint8)[..., None]
    if image.mode == "1":
        return np.asarray(image.convert("L"))[..., None]
    if image.mode == "P":
        return np.asarray(image.convert("RGBA" if "transparency" in image.info else "RGB"))
    if image.mode == "LA":
        la = np.asarray(image)
        return np.stack([la[..., 0], la[..., 0], la[..., 0], la[..., 1]], axis=-1)
    array = np.asarray(image)
    return array[..., None] if array.ndim == 2 else array


@pytest.mark.parametrize("mode", ["L", "LA", "RGB", "RGBA", "P", "P+tRNS", "1", "I;16"])
def test_png_decoder_matches_pillow(tmp_path, mode):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    width, height = 37, 23      # odd sizes: packed rows of the 1 bit image end mid-byte
    y, x = np.mgrid[0:height, 0:width]
    smooth = ((x * 5 + y * 3) % 256).astype(np.uint8)       # gradients make the encoder pick Sub / Up / Average / Paeth rows
    noise = rng.integers(0, 256, (height, width), dtype=np.uint8)
    if mode == "L":
        image = Image.fromarray(np.where(y % 2 == 0, smooth, noise), "L")
    elif mode == "LA":
        image = Image.fromarray(np.stack([smooth, noise], axis=-1), "LA")
    elif mode == "RGB":
        image = Image.fromarray(np.stack([smooth, smooth[::-1], noise], axis=-1), "RGB")
    elif mode == "RGBA":
        image = Image.fromarray(np.stack([smooth, noise, smooth[:, ::-1], noise[::-1]], axis=-1), "RGBA")
    elif mode in ("P", "P+tRNS"):
        image = Image.fromarray((noise % 13).astype(np.uint8), "P")
        image.putpalette([int(v) for v in rng.integers(0, 256, 13 * 3)])
        if mode == "P+tRNS":
            image.info["transparency"] = bytes([0, 128, 255, 17])
    elif mode == "1":
        image = Image.fromarray(noise > 127).convert("1")
    else:
        image = Image.fromarray((noise.astype(np.uint16) << 8) | smooth)    # uint16 -> mode I;16
    path = str(tmp_path / "image.png")
    save_arguments = {"transparency": image.info["transparency"]} if mode == "P+tRNS" else {}
    image.save(path, **save_arguments)

    expected = _expected_pixels(Image.open(path))
    decoded = host.load_png(path)
    assert decoded.shape == expected.shape
    assert np.array_equal(decoded, expected)
    assert np.array_equal(host.load_png(path, flip=True), expected[::-1])      # load(): bottom row first


def test_png_decoder_rejects_what_it_cannot_read(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    path = str(tmp_path / "interlaced.png")
    png = bytearray()
    Image.fromarray(np.zeros((4, 4), np.uint8), "L").save(str(tmp_path / "plain.png"))
    png[:] = open(str(tmp_path / "plain.png"), "rb").read()
    # flip the IHDR interlace byte (and do not bother with the CRC: the decoder must refuse on the flag alone)
    png[8 + 8 + 12] = 1
    open(path, "wb").write(png)
    with pytest.raises(capi.HiprError):
        host.load_png(path)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + b"\0" * 5)
    with pytest.raises(capi.HiprError):
        host.load_png(path)
    with pytest.raises(capi.HiprError):
        host.load_png(str(tmp_path / "absent.png"))


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE_RESOURCES, "WorldMask.png")), reason="reference resources are only in the build container")
def test_png_decoder_on_the_reference_resource():
    Image = pytest.importorskip("PIL.Image")
    path = os.path.join(REFERENCE_RESOURCES, "WorldMask.png")
    expected = _expected_pixels(Image.open(path))
    assert np.array_equal(host.load_png(path), expected)


# ---------------------------------------------------------------------------------------------------------------------
# glTF
# ---------------------------------------------------------------------------------------------------------------------

_COMPONENT_DTYPE = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_TYPE_COUNT = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _read_gltf(path):
    data = open(path, "rb").read()
    binary = None
    if path.endswith(".glb"):
        magic, version, length = struct.unpack_from("<III", data, 0)
        assert magic == 0x46546C67 and version == 2
        at, document = 12, None
        while at + 8 <= length:
            chunk_length, chunk_type = struct.unpack_from("<II", data, at)
            body = data[at + 8:at + 8 + chunk_length]
            if chunk_type == 0x4E4F534A:
                document = json.loads(body)
            elif chunk_type == 0x004E4942:
                binary = body
            at += 8 + ((chunk_length + 3) & ~3)
    else:
        document = json.loads(data)
    buffers = []
    for buffer in document.get("buffers", []):
        uri = buffer.get("uri")
        if uri is None:
            buffers.append(binary)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(";base64,", 1)[1]))
        else:
            buffers.append(open(os.path.join(os.path.dirname(path), uri), "rb").read())
    return document, buffers


def _accessor(document, buffers, index):
    accessor = document["accessors"][index]
    view = document["bufferViews"][accessor["bufferView"]]
    dtype = np.dtype(_COMPONENT_DTYPE[accessor["componentType"]])
    components = _TYPE_COUNT[accessor["type"]]
    stride = view.get("byteStride") or dtype.itemsize * components
    offset = view.get("byteOffset", 0) + accessor.get("byteOffset", 0)
    raw = np.frombuffer(buffers[view["buffer"]], dtype=np.uint8)
    rows = np.lib.stride_tricks.as_strided(raw[offset:], shape=(accessor["count"], dtype.itemsize * components), strides=(stride, 1))
    return np.ascontiguousarray(rows).view(dtype).reshape(accessor["count"], components)


def _node_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T      # column major in the file
    t = np.array(node.get("translation", [0, 0, 0]), dtype=np.float64)
    x, y, z, w = node.get("rotation", [0, 0, 0, 1])
    s = np.array(node.get("scale", [1, 1, 1]), dtype=np.float64)
    # The reference takes node scales as uniform (glTFLoader.cpp:269-271): the volume preserving mean stands in for the three
    # factors. Non-uniform scaling reaches it through node matrices, whose remainder is baked into the meshes.
    s = np.full(3, np.cbrt(s.prod()))
    rotation = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    matrix = np.eye(4)
    matrix[:3, :3] = rotation * s[None, :]
    matrix[:3, 3] = t
    return matrix


def _expected_world_triangles(path):
    """(n, 3, 3) world-space corners of every TRIANGLES primitive of the default scene, in the renderer's left-handed
    frame: glTF world space with X negated and the first two corners of every triangle swapped."""
    document, buffers = _read_gltf(path)
    triangles = []

    def visit(node_index, parent):
        node = document["nodes"][node_index]
        world = parent @ _node_matrix(node)
        if "mesh" in node:
            for primitive in document["meshes"][node["mesh"]]["primitives"]:
                if primitive.get("mode", 4) != 4:
                    continue
                positions = _accessor(document, buffers, primitive["attributes"]["POSITION"]).astype(np.float64)
                indices = _accessor(document, buffers, primitive["indices"]).reshape(-1).astype(np.int64) if "indices" in primitive else np.arange(len(positions))
                indices = indices[:len(indices) // 3 * 3].reshape(-1, 3)
                transformed = positions @ world[:3, :3].T + world[:3, 3]
                triangles.append(transformed[indices])
        for child in node.get("children", []):
            visit(child, world)

    for root in document["scenes"][document["scene"]]["nodes"]:
        visit(root, np.eye(4))
    corners = np.concatenate(triangles)
    corners[..., 0] *= -1.0
    return corners[:, [1, 0, 2], :]


def _loaded_world_triangles(path):
    scene = host.Scene("file:" + path)
    raw = scene.triangles()     # (n, 12) uint32 words: v0, v1, v2 as floats, then instance / primitive / flags
    return raw[:, :9].copy().view(np.float32).reshape(-1, 3, 3).astype(np.float64), scene


def _assert_same_triangles(actual, expected):
    from scipy.spatial import cKDTree
    assert actual.shape == expected.shape
    scale = np.abs(expected).max()
    tolerance = 2e-5 * scale
    assert np.allclose(actual.min(axis=(0, 1)), expected.min(axis=(0, 1)), atol=tolerance)
    assert np.allclose(actual.max(axis=(0, 1)), expected.max(axis=(0, 1)), atol=tolerance)
    # The BVH build reorders triangles: match every loaded triangle to an expected one by its corners (order inside the
    # triangle matters: it carries the winding), and require the match to be one to one up to exact duplicates.
    tree = cKDTree(expected.reshape(-1, 9))
    distance, match = tree.query(actual.reshape(-1, 9))
    assert distance.max() <= 4 * tolerance, f"worst corner mismatch {distance.max()} of scene scale {scale}"
    normals_actual = np.cross(actual[:, 1] - actual[:, 0], actual[:, 2] - actual[:, 0])
    normals_expected = np.cross(expected[match, 1] - expected[match, 0], expected[match, 2] - expected[match, 0])
    assert np.allclose(normals_actual, normals_expected, atol=8 * tolerance * scale)
    assert len(np.unique(match)) >= len(np.unique(np.round(expected.reshape(-1, 9) / tolerance).astype(np.int64), axis=0)) * 0.999


def _write_synthetic_gltf(directory, binary_container):
    """A small scene that exercises what the loader has to get right: an interleaved vertex buffer, 8 / 16 / 32 bit and absent
    indices, a node hierarchy mixing TRS and matrices, non-uniform scale and shear (the residual baked into mesh copies),
    a mesh instanced under several nodes, and a LINES primitive that must be skipped."""
    rng = np.random.default_rng(11)
    blob = bytearray()
    views, accessors = [], []

    def add(array, stride=None):
        while len(blob) % 4:
            blob.append(0)
        offset = len(blob)
        blob.extend(array.tobytes())
        view = {"buffer": 0, "byteOffset": offset, "byteLength": array.nbytes}
        if stride:
            view["byteStride"] = stride
        views.append(view)
        return len(views) - 1

    component = {np.dtype(np.uint8): 5121, np.dtype(np.uint16): 5123, np.dtype(np.uint32): 5125, np.dtype(np.float32): 5126}

    def accessor(view, dtype, count, kind, byte_offset=0, bounds=None):
        entry = {"bufferView": view, "componentType": component[np.dtype(dtype)], "count": count, "type": kind}
        if byte_offset:
            entry["byteOffset"] = byte_offset
        if bounds is not None:
            entry["min"], entry["max"] = [float(v) for v in bounds.min(axis=0)], [float(v) for v in bounds.max(axis=0)]
        accessors.append(entry)
        return len(accessors) - 1

    # a 6 x 6 vertex height field, positions and normals interleaved
    grid = 6
    gy, gx = np.mgrid[0:grid, 0:grid]
    positions = np.stack([gx / (grid - 1) - 0.5, 0.2 * rng.random((grid, grid)), gy / (grid - 1) - 0.5], axis=-1).reshape(-1, 3).astype(np.float32)
    normals = np.tile(np.array([0, 1, 0], np.float32), (len(positions), 1))
    vertex_view = add(np.concatenate([positions, normals], axis=1), stride=24)
    position_accessor = accessor(vertex_view, np.float32, len(positions), "VEC3", bounds=positions)
    normal_accessor = accessor(vertex_view, np.float32, len(positions), "VEC3", byte_offset=12)
    quads = [(y * grid + x, y * grid + x + 1, (y + 1) * grid + x + 1, (y + 1) * grid + x) for y in range(grid - 1) for x in range(grid - 1)]
    indices = np.array([[a, d, c, a, c, b] for a, b, c, d in quads]).reshape(-1)
    index_accessors = [accessor(add(indices.astype(t)), t, len(indices), "SCALAR") for t in (np.uint8, np.uint16, np.uint32)]
    # an unindexed soup of 5 triangles
    soup = rng.random((15, 3)).astype(np.float32)
    soup_accessor = accessor(add(soup), np.float32, len(soup), "VEC3", bounds=soup)
    line_accessor = accessor(add(np.array([0, 1], np.uint16)), np.uint16, 2, "SCALAR")

    def grid_primitive(index_accessor):
        return {"attributes": {"POSITION": position_accessor, "NORMAL": normal_accessor}, "indices": index_accessor, "material": 0}

    meshes = [{"name": "Grids", "primitives": [grid_primitive(index_accessors[0]), grid_primitive(index_accessors[1]),
                                               {"attributes": {"POSITION": position_accessor}, "indices": line_accessor, "mode": 1}]},
              {"name": "Grid32", "primitives": [grid_primitive(index_accessors[2])]},
              {"name": "Soup", "primitives": [{"attributes": {"POSITION": soup_accessor}}]}]
    shear = np.eye(4)
    shear[0, 1], shear[2, 1], shear[:3, 3] = 0.4, -0.3, (0.5, 2.0, -1.0)
    half_turn = float(np.sqrt(0.5))
    squash = _node_matrix({"rotation": [0.2705980500730985, 0.0, 0.0, 0.9626833975842174], "translation": [0.0, 0.5, 0.0]}) @ np.diag([1.0, 0.25, 3.0, 1.0])
    nodes = [{"name": "Root", "children": [1, 2, 5], "translation": [0.5, 0.0, -2.0], "rotation": [0.0, half_turn, 0.0, half_turn], "scale": [1.5, 1.5, 1.5]},
             {"name": "Plain", "mesh": 0, "translation": [1.0, 0.25, 0.0]},
             {"name": "Squashed", "mesh": 1, "matrix": [float(v) for v in squash.T.reshape(-1)], "children": [3, 4]},
             {"name": "SharedA", "mesh": 1, "translation": [0.0, 1.0, 0.0]},
             {"name": "SharedB", "mesh": 2, "matrix": [float(v) for v in shear.T.reshape(-1)]},
             {"name": "SoupNode", "mesh": 2, "translation": [-2.0, 0.0, 0.0], "scale": [1.0, 2.0, 4.0]},     # taken as 2, 2, 2
             {"name": "SecondRoot", "mesh": 1, "translation": [0.0, -1.0, 3.0]}]
    document = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 6]}], "nodes": nodes, "meshes": meshes,
                "materials": [{"name": "Grey", "pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.5, 0.5, 1.0], "metallicFactor": 0.0}}],
                "accessors": accessors, "bufferViews": views}
    if binary_container:
        document["buffers"] = [{"byteLength": len(blob)}]
        text = json.dumps(document).encode()
        text += b" " * (-len(text) % 4)
        blob.extend(b"\0" * (-len(blob) % 4))
        path = os.path.join(directory, "synthetic.glb")
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(text) + 8 + len(blob)))
            f.write(struct.pack("<II", len(text), 0x4E4F534A) + text)
            f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))
    else:
        document["buffers"] = [{"byteLength": len(blob), "uri": "synthetic.bin"}]
        open(os.path.join(directory, "synthetic.bin"), "wb").write(bytes(blob))
        path = os.path.join(directory, "synthetic.gltf")
        json.dump(document, open(path, "w"))
    return path


@pytest.mark.parametrize("binary_container", [False, True])
def test_gltf_world_triangles_match_an_independent_evaluation(tmp_path, binary_container):
    pytest.importorskip("scipy")
    path = _write_synthetic_gltf(str(tmp_path), binary_container)
    expected = _expected_world_triangles(path)
    assert len(expected) == 4 * 50 + 2 * 5 + 50      # grids under Plain (2 primitives), Squashed, SharedA, SecondRoot; soups under SharedB, SoupNode
    actual, scene = _loaded_world_triangles(path)
    _assert_same_triangles(actual, expected)
    desc = scene.desc
    assert desc.instance_count == 7                 # one MeshModel per (node, TRIANGLES primitive)
    assert desc.light_count == 1                    # the viewer's default directional light: the file has no lights
    assert tuple(round(v, 2) for v in scene.state.environment_tint) == (0.68, 0.92, 1.0)


@pytest.mark.skipif(not os.path.isdir(REFERENCE_RESOURCES), reason="reference resources are only in the build container")
@pytest.mark.parametrize("name", ["Diamond.glb", "Shaderball.gltf"])
def test_gltf_loader_on_the_reference_resources(name):
    pytest.importorskip("scipy")
    path = os.path.join(REFERENCE_RESOURCES, name)
    expected = _expected_world_triangles(path)
    actual, scene = _loaded_world_triangles(path)
    _assert_same_triangles(actual, expected)
    assert scene.desc.node_count > 0 and scene.desc.wide_node_count > 0


def test_scene_load_reports_files_it_cannot_read(tmp_path):
    with pytest.raises(capi.HiprError):
        host.Scene("file:" + str(tmp_path / "absent.glb"))
    open(str(tmp_path / "empty.gltf"), "w").write("{}")
    with pytest.raises(capi.HiprError):
        host.Scene("file:" + str(tmp_path / "empty.gltf"))
    with pytest.raises(capi.HiprError):
        host.Scene("file:" + str(tmp_path / "model.fbx"))


# ---- BASELINE config 3 on the reference's own asset (build container only: the asset is not copied) ---------------------------------------

@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE_RESOURCES, "Shaderball.gltf")), reason="reference resources are only in the build container")
def test_material_scene_on_the_reference_shader_ball_against_the_stand_in():
    """apps/SimpleViewer/Scenes/Material.cpp:143-188 builds config 3 from Resources/Shaderball.gltf; the GPU box has no reference tree, so bench.py and the
    -m gpu tests render a procedural stand-in of the ball. Here, where the asset is, the scene is built from it through the same code path: the 8-wide tree's
    search finds what the BVH2 search finds on it, and the stand-in is held to the real asset's figures: same triangle count (+-5 %), same share of rays that
    hit, same rays per path, same mean radiance -- and, since round 4, the same COST per ray: node visits, triangle tests and shadow-ray node visits within
    10 % of the asset's (4.3 / 5.0 / 3.5; round 3's regular-grid sphere cost 2.8 / 1.6 / 2.5 and flattered config 3's Mrays/s). The stand-in's layered shells,
    openings and slanted, uneven tessellation (host/MaterialScene.cpp) were tuned against these counters; profiles/r04_config3_real_asset_statistics.txt."""
    from oracle_bindings import get_oracle
    Scene = host.Scene
    oracle = get_oracle(True)
    real = Scene("material:" + os.path.join(REFERENCE_RESOURCES, "Shaderball.gltf"))
    stand_in = Scene("material")
    assert 170000 <= real.desc.triangle_count <= 185000                                   # 7 x (13 332 + 11 952) + the floor's 8
    assert abs(stand_in.desc.triangle_count - real.desc.triangle_count) <= 0.05 * real.desc.triangle_count
    w, h, spp = 128, 72, 2
    figures = {}
    for name, scene in (("real", real), ("stand_in", stand_in)):
        cam = scene.camera(w, h, max_bounce_count=32)
        image, c, _ = oracle.render(scene.desc, scene.state, cam, w, h, spp, use_bvh=3)
        figures[name] = dict(nodes=c["closest_nodes"] / c["closest_rays"], triangles=c["closest_triangles"] / c["closest_rays"], shadow_nodes=c["shadow_nodes"] / max(1, c["shadow_rays"]),
                             hits=c["shaded_hits"] / c["closest_rays"], rays_per_path=(c["closest_rays"] + c["shadow_rays"]) / c["camera_rays"], mean=float(image[..., :3].mean()))
    print("CONFIG-3-STATISTICS", figures)
    for key, tolerance in (("hits", 0.15), ("rays_per_path", 0.2), ("mean", 0.2)):
        assert abs(figures["stand_in"][key] - figures["real"][key]) <= tolerance * figures["real"][key], (key, figures)
    for key in ("nodes", "triangles", "shadow_nodes"):      # what a ray costs: within 10 % of the asset
        assert abs(figures["stand_in"][key] - figures["real"][key]) <= 0.10 * figures["real"][key], (key, figures)
    # the real asset under both searches: same hits for the camera rays
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), axis=-1).reshape(-1, 2).astype(np.uint32)
    o, d = oracle.generate_rays(real.camera(w, h), w, h, 1, xy)
    rays = np.zeros((len(xy), 8), np.float32)
    rays[:, 0:4] = o
    rays[:, 4:7] = d[:, :3]
    rays[:, 7] = np.inf
    two, _ = oracle.trace_closest(real.desc, rays, use_bvh=1, with_lights=True)
    eight, _ = oracle.trace_closest(real.desc, rays, use_bvh=3, with_lights=True)
    assert (two[:, 3].view(np.uint32) != eight[:, 3].view(np.uint32)).mean() <= 2e-3
    assert (eight[:, 3].view(np.uint32) != 0xFFFFFFFF).mean() > 0.5
