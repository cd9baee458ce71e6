"""K3 -- the whole BSDF / next-event-estimation / sampling stack -- held to the oracle BIT for bit, in the CPU suite ("Parity first").

The device code of the shade stage is plain IEEE f32 arithmetic in a fixed order. tests/native/DeviceShadeHost.hip compiles THOSE headers (csrc/device_shading.h,
csrc/shade_kernel.h shade_path, csrc/kernels.h' samplers) for the host through hipcc's host pass, with the arithmetic of the verification build
(libhiprenderer_verify.so: correctly rounded division and square root, no contraction, every transcendental evaluated in f64 and rounded once), and the oracle
evaluates its transcendentals the same way (oracle_set_f64_transcendentals). Two independent implementations then agree on every word:

  * the three shading models' sample() and evaluate_with_PDF(), in the plain form and in the per-hit "terms" form k_shade calls, on random directions and numbers;
  * the light sources' sample_radiance();
  * shade_path itself against the oracle's hit programs (miss, light hit, path_tracing_closest_hit<>: ORS/MonteCarlo.cu:61-302), entry by entry over whole
    wavefronts of real paths, bounce after bounce: radiance added, next ray, BSDF PDF, throughput, bounce counter, shadow ray and the radiance it carries.

Round 4 could only say "within 2e-3 for 99 % of the samples" here, because the product's shade unit is built with hardware-approximate arithmetic like the reference's
--use_fast_math PTX. What the product's fast arithmetic then costs in image terms is measured on the GPU against the verification build (tests/test_gpu_verify_build.py).
The same comparison found, in round 5, two reciprocal-multiplies in the "terms" form that differed from the plain functions in the last bit for 3-60 % of the
evaluations, and the one place where device and oracle interpolated normals in a different order."""
import json
from pathlib import Path

import numpy as np
import pytest

from bifrost3d_amd.host import Scene
from device_host_bindings import LIB_PATH, RECORD_WORDS, DeviceShadeOnHost, camera_paths, oracle_shade
from test_oracle_goldens import normalize, shading_params
from test_oracle_lights import samples02, sphere_light, spot_light

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.skipif(not LIB_PATH.exists(), reason="tests/native/libdevice_shade_host.so is not built (python -c 'import __graft_entry__ as g; g.build()')")


@pytest.fixture(scope="module")
def exact_oracle():
    """The oracle with unorm16 tables (as uploaded to the device) and f64 transcendentals; restored afterwards."""
    from oracle_bindings import get_oracle
    o = get_oracle(True)
    before = o.lib.oracle_set_f64_transcendentals(1)
    yield o
    o.lib.oracle_set_f64_transcendentals(before)


@pytest.fixture(scope="module")
def cornell_on_host():
    d = DeviceShadeOnHost(Scene("cornell"))
    yield d
    d.close()


def same_bits(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("model, oracle_model, material", [(0, 4, "gold"), (0, 4, "plastic"), (0, 4, "coated_plastic"), (1, 6, "gold"), (1, 6, "plastic"), (2, 5, "frosted_glass")])
def test_shading_models_of_the_device_equal_the_oracle_bit_for_bit(exact_oracle, cornell_on_host, goldens, model, oracle_model, material):
    """HIPR_SHADING_* as k_shade evaluates them (hipr_debug_shading's kernel, on the host) against the oracle's DefaultShading / DiffuseShading / TransmissiveShading:
    4 outgoing directions x 6000 samples and evaluations each, authored and random roughness, with and without a path-regularisation hint, both forms."""
    rng = np.random.default_rng(11 + 7 * model + len(material))
    n = 6000
    for trial in range(4):
        wo = normalize([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.02, 1.0)])
        params = np.array(shading_params(goldens["materials"][material]), np.float32)
        if trial >= 2:
            params[3], params[9] = rng.uniform(0.02, 1.0), rng.uniform(0.1, 20.0)      # roughness; max_PDF_hint (path regularisation raises the roughness)
        u = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        wi = rng.normal(size=(n, 3)).astype(np.float32)
        wi /= np.linalg.norm(wi, axis=1, keepdims=True)
        if model != 2:
            wi[:, 2] = np.abs(wi[:, 2])
        sampled, evaluated = exact_oracle.bsdf_sample(oracle_model, params, wo, u), exact_oracle.bsdf_eval(oracle_model, params, wo, wi)
        assert np.isfinite(sampled[:, 0:3]).all() and (np.abs(sampled[:, 3]) > 1e-6).mean() > 0.5      # the comparison is not over empty samples
        for terms in (False, True):
            ours = cornell_on_host.shading(model, params, wo, u, mode=0, terms=terms)
            assert same_bits(ours, sampled).all(), (material, trial, terms, "sample", int((~same_bits(ours, sampled).all(axis=1)).sum()))
            ours = cornell_on_host.shading(model, params, wo, wi, mode=1, terms=terms)[:, :4]
            assert same_bits(np.ascontiguousarray(ours), evaluated).all(), (material, trial, terms, "evaluate")


def test_light_sources_of_the_device_equal_the_oracle_bit_for_bit(exact_oracle, cornell_on_host):
    u = samples02(exact_oracle, 1024)
    rng = np.random.default_rng(3)
    for light in (sphere_light((0.3, 2.0, -0.4), 0.5, 7.0), sphere_light((0.0, 1.0, 0.0), 0.0, 3.0), spot_light((0.1, 3.0, 0.2), (0.0, -1.0, 0.0), 0.7, 5.0, 0.6),
                  spot_light((0.0, 2.0, 0.0), (0.6, -0.8, 0.0), 0.0, 5.0, 0.8), spot_light((0.0, 4.0, 0.0), (0.0, -1.0, 0.0), 2.5, 5.0, 0.3)):
        for _ in range(3):
            position = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
            assert same_bits(cornell_on_host.light(light, position, u), exact_oracle.light_sample(light, position, u)).all()


def wavefronts(oracle, scene, width, height, accumulation, bounces, device, models=7, textures=2, check=None):
    """The paths of one accumulation followed bounce by bounce on the host: the oracle traces (its searches are the device's, bit for bit: tests/test_gpu_parity.py) and
    BOTH shade every queue entry; the oracle's records feed the next bounce. Returns (entries shaded, entries with any differing word, {word name: count})."""
    cam = scene.camera(width, height, accumulations=accumulation, max_bounce_count=bounces)
    search = 0 if scene.desc.triangle_count <= 64 else (3 if scene.desc.wide8_slot_count else 1)
    rays, throughput, last, hashes, accumulations = camera_paths(oracle, cam, width, height, accumulation)
    entries, differing, by_word, kinds = 0, 0, {}, {"continues": 0, "shadow": 0, "shaded": 0, "miss_or_light": 0}
    for _ in range(bounces + 2):
        if len(rays) == 0:
            break
        trace = rays.copy()
        trace[:, 7] = np.inf
        hits, _ = oracle.trace_closest(scene.desc, trace, skip=last, use_bvh=search, with_lights=True)
        ours = device.shade(cam, rays, throughput, hits, last, hashes, accumulations, models=models, textures=textures)
        theirs = oracle_shade(oracle, scene, cam, rays, throughput, hits, last, hashes, accumulations)
        same = same_bits(ours, theirs)
        entries += len(rays)
        differing += int((~same.all(axis=1)).sum())
        for k in np.where(~same.all(axis=0))[0]:
            by_word[RECORD_WORDS.get(int(k), int(k))] = by_word.get(RECORD_WORDS.get(int(k), int(k)), 0) + int((~same[:, k]).sum())
        flags = theirs[:, 0].view(np.uint32)
        for bit, name in ((1, "continues"), (2, "shadow"), (4, "shaded")):
            kinds[name] += int(((flags & bit) != 0).sum())
        kinds["miss_or_light"] += int(((flags & 4) == 0).sum())
        if check is not None:
            check(ours, theirs)
        on = (flags & 1) != 0
        rays, throughput = np.ascontiguousarray(theirs[on, 4:12]), np.ascontiguousarray(theirs[on, 12:16])
        last, hashes, accumulations = np.ascontiguousarray(theirs[on, 16]).view(np.uint32), hashes[on], accumulations[on]
    return entries, differing, by_word, kinds


SCENES = {
    "cornell": (lambda: Scene("cornell"), 4, 7, 2),
    "cornell_all_diffuse_kernel": (lambda: Scene("cornell", diffuse_only=True), 4, 2, 0),      # k_shade<2, ..., TEXTURES = 0>, the instantiation BASELINE config 2 runs
    "cornell_default_kernel": (lambda: Scene("cornell"), 4, 1, 0),                              # k_shade<1, ..., 0>: the headline's instantiation
    "cornell_spot_light": (lambda: Scene("cornell", spot=True), 4, 7, 2),
    "cornell_environment_map": (lambda: Scene("cornell", environment=True), 4, 7, 2),           # presampled environment light, map lookups weighted by MIS on escape
    "opacity_cutouts_and_partial_coverage": (lambda: Scene("opacity"), 32, 7, 2),               # coverage textures: rejected hits retraced with tmin bumped
    "material_coat_textured_floor": (lambda: Scene("material", coat=True), 32, 7, 2),
    "glass_transmissive": (lambda: Scene("glass"), 32, 7, 2),
    "atrium_17k_vertex_normals": (lambda: Scene("atrium", param0=20000, param1=1), 4, 1, 0),
    "atrium_17k_textured_cutouts": (lambda: Scene("atrium", param0=20000, param1=1, textured=True), 4, 1, 1),      # k_shade<1, ..., TEXTURES = 1>: 8-bit material textures
}


@pytest.mark.parametrize("name", list(SCENES))
def test_the_shade_stage_of_the_device_equals_the_hit_programs_of_the_oracle_bit_for_bit(exact_oracle, name):
    make, bounces, models, textures = SCENES[name]
    scene = make()
    device = DeviceShadeOnHost(scene)
    try:
        total, kinds_total = 0, {}
        for accumulation in (0, 5):      # accumulation 0 shoots through the pixel centres (ORS/SimpleRGPs.cu:68), every other one a jittered sample
            entries, differing, by_word, kinds = wavefronts(exact_oracle, scene, 64, 36, accumulation, bounces, device, models, textures)
            assert differing == 0, (name, accumulation, differing, entries, by_word)
            total += entries
            for k, v in kinds.items():
                kinds_total[k] = kinds_total.get(k, 0) + v
        # the comparison went through every part of the record: paths that continued, shadow rays, shaded hits and rays that escaped or hit a light
        assert total >= 2 * 64 * 36 and kinds_total["shaded"] > 1000 and kinds_total["continues"] > 1000 and kinds_total["shadow"] > 500, (name, total, kinds_total)
        if "environment" in name or "material" in name or "atrium" in name:
            assert kinds_total["miss_or_light"] > 100
    finally:
        device.close()


def test_the_comparison_sees_a_last_bit(exact_oracle):
    """Power of the test above: with the oracle back on glibc's f32 transcendentals (its default, the checker of the product build) the same comparison reports
    differences -- last-bit ones in directions and throughput --, so 'zero entries differ' is a statement about arithmetic, not about a comparison that cannot fail."""
    scene = Scene("cornell")
    device = DeviceShadeOnHost(scene)
    worst = [0.0]

    def check(ours, theirs):
        finite = np.isfinite(ours) & np.isfinite(theirs)
        worst[0] = max(worst[0], float(np.max(np.abs(ours[finite][:] - theirs[finite][:]) / (np.abs(theirs[finite]) + 1e-3))))
    try:
        exact_oracle.lib.oracle_set_f64_transcendentals(0)
        entries, differing, by_word, _ = wavefronts(exact_oracle, scene, 64, 36, 5, 4, device, check=check)
    finally:
        exact_oracle.lib.oracle_set_f64_transcendentals(1)
        device.close()
    assert 0 < differing < 0.5 * entries and ("direction.x" in by_word or "throughput.x" in by_word), (differing, entries, by_word)
    assert worst[0] < 1e-3      # and what differs does so at rounding level: no path took another branch in this small frame, or if one did, the test above would have caught it


def test_texel_index_arithmetic_of_the_samplers():
    """wrap_coord / wrap_next of csrc/kernels.h (round 5: a mask for power-of-two sizes, the neighbour texel from the wrapped one) against the definition -- Python's floor
    modulo for repeat, a clamp otherwise -- for every size 1 ... 130 and coordinates from far below zero to far beyond the size, both texels of a bilinear tap."""
    import ctypes as C
    from device_host_bindings import library
    lib = library()
    lib.dsh_wrap.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    coordinates = np.concatenate([np.arange(-700, 700), np.array([-2**20, -65537, -65536, 65535, 65536, 2**20, 2**30, -2**30])]).astype(np.int32)
    out = np.zeros(2 * len(coordinates), np.int32)
    for n in list(range(1, 131)) + [256, 1000, 1024, 4096, 5000]:
        for repeat in (1, 0):
            lib.dsh_wrap(coordinates.ctypes.data_as(C.POINTER(C.c_int)), len(coordinates), n, repeat, out.ctypes.data_as(C.POINTER(C.c_int)))
            i = coordinates.astype(np.int64)
            first = i % n if repeat else np.clip(i, 0, n - 1)
            second = (i + 1) % n if repeat else np.clip(i + 1, 0, n - 1)
            assert np.array_equal(out[0::2], first) and np.array_equal(out[1::2], second), (n, repeat)
