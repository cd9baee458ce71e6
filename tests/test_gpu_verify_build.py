"""K3 parity made exact: the renderer's EXACT arithmetic mode against the oracle, bit for bit, and the fast mode against the exact mode. ("verify" / "verification build"
below is the exact mode: until round 6 it was a library of its own, libhiprenderer_verify.so; now hipr_set_arithmetic(ctx, HIPR_ARITHMETIC_EXACT) selects the second
build of the shade unit inside libhiprenderer.so.)

The exact shade unit is the product's source compiled like the traversal unit -- correctly rounded division and square root, no contraction, denormals kept -- with
sin, cos and pow as the specified binary64 sequences of csrc/spec_math.h (atan2 / asin of the environment lookup: the f64 libm rounded once); the oracle restates
the same sequences on request (oracle_set_f64_transcendentals, oracle/vecmath.h spec::). Then

  (i)   whole lit images of the verification build EQUAL the oracle's: every pixel's f64 running mean bit-identical, on every scene type the renderer has -- the
        statistical image bars of rounds 1-4 (RMSE within n x what was measured) become an equality for the code, and what remains statistical is one number:
  (ii)  the product against the verification build ON THE DEVICE at equal seed -- what the shade unit's hardware-approximate arithmetic (the reference's
        --use_fast_math, extensions/OptiXRenderer/CMakeLists.txt:82) does to an image: zero-mean path divergence, measured here and held to 1.5 x the recorded figure;
  (iii) the product against the ORACLE (the figure of bench.py's rmse_vs_oracle and of tests/test_gpu_statistics.py) is held to (ii) measured in the same run;
  (iv)  the shade stage itself (hipr_debug_shade, shade_path entry by entry) equals the oracle's hit programs word for word on the verification build, and on the
        product the same records are counted decision by decision: how many entries take another discrete decision (hit accepted or refused, shadow ray emitted,
        another light candidate kept, another lobe or branch sampled) under the fast arithmetic.
The same comparison runs WITHOUT a GPU on the host build of the device code (tests/test_device_code_on_host_cpu.py)."""
import json
from pathlib import Path

import numpy as np
import pytest

from bifrost3d_amd import capi
from bifrost3d_amd.host import Scene
from device_host_bindings import RECORD_WORDS, camera_paths, oracle_shade

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def product():
    from bifrost3d_amd.renderer import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def verify():
    from bifrost3d_amd.renderer import Context
    c = Context(0, arithmetic="exact")
    assert c.arithmetic == "exact"
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle_q():
    from oracle_bindings import get_oracle
    return get_oracle(True)


class exact_transcendentals:
    def __init__(self, oracle):
        self.oracle = oracle

    def __enter__(self):
        self.before = self.oracle.lib.oracle_set_f64_transcendentals(1)
        return self.oracle

    def __exit__(self, *args):
        self.oracle.lib.oracle_set_f64_transcendentals(self.before)


def render(ctx, scene, w, h, spp, bounces):
    ctx.upload_scene(scene)
    batch = min(spp, 32)
    ctx.set_frame(w, h, 0, 1, batch)
    for a in range(0, spp, batch):
        ctx.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=bounces))
    ctx.synchronize()
    return ctx.read_accumulation()[..., :3].astype(np.float64)


def rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


def compare_rms(a, b):      # extensions/ImageOperations/ImageOperations/Compare.h:23-43
    d = np.abs(a - b)
    return float(np.sqrt(np.mean((0.2126 * d[..., 0] + 0.7152 * d[..., 1] + 0.0722 * d[..., 2]) ** 2)))


IMAGES = {      # scene, bounces, frame, accumulations
    "cornell": (lambda: Scene("cornell"), 4, (160, 90), 64),
    "cornell_all_diffuse": (lambda: Scene("cornell", diffuse_only=True), 4, (160, 90), 64),
    "cornell_spot_light": (lambda: Scene("cornell", spot=True), 4, (160, 90), 32),
    "cornell_environment_map": (lambda: Scene("cornell", environment=True), 4, (160, 90), 32),
    "opacity": (lambda: Scene("opacity"), 32, (160, 90), 64),
    "material_coat_32_bounces": (lambda: Scene("material", coat=True), 32, (160, 90), 64),
    "glass_32_bounces": (lambda: Scene("glass"), 32, (160, 90), 64),
    "atrium_251k_headline": (lambda: Scene("atrium", param0=260000, param1=1), 4, (160, 90), 64),
    "atrium_251k_textured_cutouts": (lambda: Scene("atrium", param0=260000, param1=1, textured=True), 4, (160, 90), 32),
    "atrium_1M_sliver_triangles": (lambda: Scene("atrium", param0=1000000, param1=2), 4, (96, 54), 8),
}


@pytest.mark.parametrize("name", list(IMAGES))
def test_images_of_the_verification_build_equal_the_oracle_bit_for_bit(verify, oracle_q, name):
    make, bounces, (w, h), spp = IMAGES[name]
    scene = make()
    ours = render(verify, scene, w, h, spp, bounces)
    with exact_transcendentals(oracle_q) as oracle:
        theirs, counters, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=bounces), w, h, spp, use_bvh=verify.oracle_search())
    theirs = theirs[..., :3].astype(np.float64)
    identical = (ours == theirs).all(axis=-1)
    print(f"VERIFY {name}: {w}x{h}x{spp} spp, {int(scene.desc.triangle_count)} triangles, mean radiance {theirs.mean():.4f}: {identical.mean():.6f} of the pixels bit-identical, "
          f"RMSE {rmse(ours, theirs):.3e}")
    assert np.isfinite(ours).all() and theirs.mean() > 0.01 and counters["shaded_hits"] > w * h
    # f64-evaluated transcendentals of two libraries round to the same f32 except with probability ~2^-26 per call: at most a pixel or two of a frame may differ
    assert identical.mean() >= 0.9999 and rmse(ours, theirs) <= 1e-5, (name, float(identical.mean()), rmse(ours, theirs))


def followed_wavefronts(oracle, scene, w, h, accumulation, bounces, shade_stages):
    """The paths of one accumulation bounce by bounce: the oracle traces and shades (its records feed the next bounce); every stage of `shade_stages` shades the same
    entries. Yields (oracle records, [records of each stage])."""
    cam = scene.camera(w, h, accumulations=accumulation, max_bounce_count=bounces)
    search = 0 if scene.desc.triangle_count <= 64 else (3 if scene.desc.wide8_slot_count else 1)
    rays, throughput, last, hashes, accumulations = camera_paths(oracle, cam, w, h, accumulation)
    for _ in range(bounces + 2):
        if len(rays) == 0:
            return
        trace = rays.copy()
        trace[:, 7] = np.inf
        hits, _ = oracle.trace_closest(scene.desc, trace, skip=last, use_bvh=search, with_lights=True)
        theirs = oracle_shade(oracle, scene, cam, rays, throughput, hits, last, hashes, accumulations)
        yield theirs, [stage.debug_shade(cam, rays, throughput, hits, last, hashes, accumulations) for stage in shade_stages]
        on = (theirs[:, 0].view(np.uint32) & 1) != 0
        rays, throughput = np.ascontiguousarray(theirs[on, 4:12]), np.ascontiguousarray(theirs[on, 12:16])
        last, hashes, accumulations = np.ascontiguousarray(theirs[on, 16]).view(np.uint32), hashes[on], accumulations[on]


@pytest.mark.parametrize("name", ["cornell", "glass", "atrium17k", "atrium17k_textured"])
def test_the_shade_stage_on_the_device_equals_the_hit_programs_of_the_oracle(verify, oracle_q, name):
    scene, bounces = {"cornell": (lambda: Scene("cornell"), 4), "glass": (lambda: Scene("glass"), 12), "atrium17k": (lambda: Scene("atrium", param0=20000, param1=1), 4),
                      "atrium17k_textured": (lambda: Scene("atrium", param0=20000, param1=1, textured=True), 4)}[name]
    scene = scene()
    verify.upload_scene(scene)
    entries = 0
    with exact_transcendentals(oracle_q) as oracle:
        for accumulation in (0, 3):
            for theirs, (ours,) in followed_wavefronts(oracle, scene, 96, 54, accumulation, bounces, [verify]):
                same = (ours.view(np.uint32) == theirs.view(np.uint32)) | (np.isnan(ours) & np.isnan(theirs))
                assert same.all(), (name, accumulation, int((~same.all(axis=1)).sum()), {RECORD_WORDS.get(int(k), int(k)): int((~same[:, k]).sum()) for k in np.where(~same.all(axis=0))[0]})
                entries += len(theirs)
    assert entries > 3 * 96 * 54      # two accumulations of camera rays and what continued


# What the product's fast arithmetic does, measured on the MI355X in round 5 (profiles/r05_fast_math_attribution.txt): equal-seed RMSE of the product against the
# verification build, 160 x 90 x 64 spp. Every single approximation of the shade unit (hardware sin / cos; approximate division and square root; contraction) alone
# already gives 60-85 % of the product's figure: an ulp-level perturbation applied to every path saturates into path divergence, whatever its size.
FAST_MATH = {"atrium_251k_headline": (2.40e-3, 1.72e-3), "material_coat_32_bounces": (1.48e-3, 9.9e-4), "cornell": (3.9e-6, 1.8e-6)}


@pytest.mark.parametrize("name", list(FAST_MATH))
def test_the_fast_arithmetic_of_the_product_is_zero_mean_path_divergence_of_the_recorded_size(product, verify, oracle_q, name):
    make, bounces, (w, h), spp = IMAGES[name]
    scene = make()
    fast, exact = render(product, scene, w, h, spp, bounces), render(verify, scene, w, h, spp, bounces)
    libm, _, _ = oracle_q.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=bounces), w, h, spp, use_bvh=product.oracle_search())
    libm = libm[..., :3].astype(np.float64)
    d = (fast - exact).reshape(-1, 3)
    mean, standard_error = d.mean(axis=0), d.std(axis=0) / np.sqrt(len(d))
    on_device, against_oracle = rmse(fast, exact), rmse(fast, libm)
    recorded_rmse, recorded_compare = FAST_MATH[name]
    print(f"FASTMATH {name}: product vs verification build RMSE {on_device:.3e} (Compare::rms {compare_rms(fast, exact):.3e}); product vs oracle {against_oracle:.3e}; "
          f"mean signed difference {mean} +- {standard_error} ({np.abs(mean) / np.maximum(standard_error, 1e-30)} sigma); pixels differing {((fast != exact).any(axis=-1)).mean():.4f}")
    # no bias beyond the rounding of the arithmetic itself: where no path diverges (the Cornell box) the approximate operations show as a SYSTEMATIC relative
    # difference of a few 1e-7 -- measured 4e-7, 4 to 25 sigma of a noise of 1e-8 --, six orders of magnitude below the image; allowed: 2e-6 of the mean radiance
    assert np.all(np.abs(mean) <= 3.0 * standard_error + 2e-6 * exact.mean()), (mean, standard_error)
    assert on_device <= 1.5 * recorded_rmse and compare_rms(fast, exact) <= 1.5 * recorded_compare      # (ii)
    assert abs(against_oracle - on_device) <= 0.1 * on_device + 1e-7                                   # (iii): the oracle's figure IS the fast arithmetic's


def test_decisions_of_the_product_under_its_fast_arithmetic(product, verify, oracle_q):
    """Entry by entry: the product's shade stage against the verification build's on the SAME queue entries (the 251 k-triangle atrium, paths followed by the oracle).
    Counts, per shaded hit, the discrete decisions that fall differently -- this is where the equal-seed image difference comes from, not the rounding of the values
    (which is ~1e-6 relative in entries that decide alike)."""
    scene = Scene("atrium", param0=260000, param1=1)
    product.upload_scene(scene)
    verify.upload_scene(scene)
    totals = {"entries": 0, "shaded": 0, "accepted_or_refused": 0, "shadow_ray_emitted": 0, "other_light_candidate_kept": 0, "other_direction_sampled": 0, "path_ended_or_not": 0}
    relative = []
    with exact_transcendentals(oracle_q) as oracle:
        for accumulation in (1, 2):
            for theirs, (fast, exact) in followed_wavefronts(oracle, scene, 160, 90, accumulation, 4, [product, verify]):
                assert np.array_equal(exact.view(np.uint32), theirs.view(np.uint32))
                ff, fe = fast[:, 0].view(np.uint32), exact[:, 0].view(np.uint32)
                shaded = (fe & 4) != 0
                totals["entries"] += len(fe)
                totals["shaded"] += int(shaded.sum())
                totals["accepted_or_refused"] += int(((ff ^ fe) & 4 != 0).sum())
                both = shaded & ((ff & 4) != 0)
                totals["shadow_ray_emitted"] += int((both & ((ff ^ fe) & 2 != 0)).sum())
                shadows = both & ((ff & 2) != 0) & ((fe & 2) != 0)
                totals["other_light_candidate_kept"] += int((shadows & (np.abs(fast[:, 21:24] - exact[:, 21:24]).max(axis=1) > 1e-3)).sum())
                totals["path_ended_or_not"] += int((both & ((ff ^ fe) & 1 != 0)).sum())
                on = both & ((ff & 1) != 0) & ((fe & 1) != 0)
                other_direction = on & (np.abs(fast[:, 8:11] - exact[:, 8:11]).max(axis=1) > 1e-2)
                totals["other_direction_sampled"] += int(other_direction.sum())
                alike = on & ~other_direction
                relative.append(np.abs(fast[alike, 12:15] - exact[alike, 12:15]).max(axis=1) / (np.abs(exact[alike, 12:15]).max(axis=1) + 1e-6))
    relative = np.concatenate(relative)
    rates = {k: v / max(1, totals["shaded"]) for k, v in totals.items() if k not in ("entries", "shaded")}
    print(f"DECISIONS atrium 251k, 2 accumulations of 160x90: {totals}; per shaded hit {rates}; throughput of entries that decide alike: median relative difference "
          f"{np.median(relative):.2e}, 99th percentile {np.quantile(relative, 0.99):.2e}, max {relative.max():.2e}")
    assert totals["shaded"] > 50000
    assert sum(rates.values()) <= 1e-4      # measured: none in 61 293 shaded hits; where the paths do part, and why: profiles/r05_divergence_sites.txt (tools/divergence_sites.py)
    assert np.quantile(relative, 0.99) < 1e-3
    # the branches only the fast build compiles (reciprocal division, v_exp / v_log pow, v_sin / v_cos, contraction), checked directly on single evaluations that
    # decide alike: the typical entry within a few ulp of the exact unit's (measured: median 2.1e-7), and no entry further off than a cancellation explains
    # (measured: the worst of 116 k entries 2.3e-2, the 99th percentile above) -- a wrong constant or a swapped operand in a fast-only branch moves every entry
    assert np.median(relative) < 2e-6 and relative.max() < 0.1, (float(np.median(relative)), float(relative.max()))


AOV_ENTRIES = [("depth", capi.ENTRY_DEPTH), ("albedo", capi.ENTRY_ALBEDO), ("tint", capi.ENTRY_TINT), ("roughness", capi.ENTRY_ROUGHNESS),
               ("shading_normal", capi.ENTRY_SHADING_NORMAL), ("primitive_id", capi.ENTRY_PRIMITIVE_ID), ("denoiser_albedo", capi.ENTRY_DENOISER_ALBEDO)]


@pytest.mark.parametrize("name, entry", AOV_ENTRIES)
@pytest.mark.parametrize("scene_name", ["cornell", "atrium17k", "atrium17k_textured"])
def test_aov_entry_points_of_the_verification_build_equal_the_oracle(verify, oracle_q, scene_name, name, entry):
    """The visualisation backends (ORS/SimpleRGPs.cu:227-340) and the denoiser's feature image: first-hit attributes, the rho tables, the primitive-id hash. The
    product's test holds them to 1e-5 (2e-3 for the albedos); with exact arithmetic on both sides they are equal bit for bit, NaN for NaN (a miss of the depth entry adds
    |origin - 1e30 direction| = inf, whose running mean is NaN from the second accumulation on -- on both sides in the same pixels)."""
    scene = {"cornell": lambda: Scene("cornell"), "atrium17k": lambda: Scene("atrium", param0=20000, param1=1),
             "atrium17k_textured": lambda: Scene("atrium", param0=20000, param1=1, textured=True)}[scene_name]()
    w, h, spp = 96, 54, 3
    verify.upload_scene(scene)
    verify.set_frame(w, h, 0, 1, 1)
    verify.set_entry_point(entry)
    try:
        for a in range(spp):
            verify.render_pass(scene.camera(w, h, accumulations=a, max_bounce_count=4), synchronize=True)
        ours = verify.read_accumulation()[..., :3]
    finally:
        verify.set_entry_point(capi.ENTRY_PATH_TRACING)
    with exact_transcendentals(oracle_q) as oracle:
        theirs, _, _ = oracle.render(scene.desc, scene.state, scene.camera(w, h, max_bounce_count=4), w, h, spp, use_bvh=verify.oracle_search(), entry=entry)
    theirs = theirs[..., :3]
    same = (ours == theirs) | (np.isnan(ours) & np.isnan(theirs))
    assert same.all(), (scene_name, name, int((~same.all(axis=-1)).sum()))
    assert np.isfinite(theirs).mean() > 0.05 and np.nanmax(np.abs(theirs)) > 0


def test_specified_transcendentals_on_the_device(verify, product, oracle_q):
    """csrc/spec_math.h as gfx950 evaluates it (hipr_debug_math on the exact shade unit) against the oracle's restatement, bit for bit: the binary64 operations the
    sequences are made of are correctly rounded on both machines, so not one of 3 M results may differ. (tests/test_spec_math_cpu.py holds both to glibc.) The fast
    unit's hardware approximations are within their documented few ulp of the same values."""
    import ctypes as C
    fp = C.POINTER(C.c_float)
    lib = oracle_q.lib
    lib.oracle_spec_math.argtypes = [C.c_int, C.c_int, fp, fp, fp]

    def oracle_math(function, x, y):
        out = np.zeros_like(x)
        lib.oracle_spec_math(function, x.size, x.ctypes.data_as(fp), y.ctypes.data_as(fp), out.ctypes.data_as(fp))
        return out

    rng = np.random.default_rng(5)
    azimuths = (np.float32(2.0) * np.float32(np.pi)) * rng.random(1_000_000, dtype=np.float32)
    wide = rng.uniform(-1.0e5, 1.0e5, 250_000).astype(np.float32)
    zeros = np.zeros(1_250_000, np.float32)
    x = np.concatenate([azimuths, wide])
    for function in (0, 1):
        assert np.array_equal(verify.debug_math(function, x).view(np.uint32), oracle_math(function, x, zeros).view(np.uint32))
        fast = product.debug_math(function, azimuths)
        assert np.abs(fast - oracle_math(function, azimuths, zeros[:azimuths.size])).max() < 2e-6      # v_sin_f32 / v_cos_f32 on [0, 2 pi]
    base = np.concatenate([rng.random(500_000, dtype=np.float32), np.exp(rng.uniform(-87.0, 88.0, 250_000)).astype(np.float32),
                           np.array([0.0, 1.0, 1e-45, np.inf, -1.0, np.nan], np.float32)])
    for exponents in (np.full_like(base, 0.25), np.full_like(base, 0.1), np.full_like(base, 2.4), rng.uniform(-4.0, 4.0, base.size).astype(np.float32)):
        ours, theirs = verify.debug_math(2, base, exponents), oracle_math(2, base, exponents)
        same = (ours.view(np.uint32) == theirs.view(np.uint32)) | (np.isnan(ours) & np.isnan(theirs))
        assert same.all(), (base[~same][:4], exponents[~same][:4], ours[~same][:4], theirs[~same][:4])
    unit = rng.random(100_000, dtype=np.float32) * np.float32(0.999) + np.float32(0.001)
    quarter = np.full_like(unit, 0.25)
    assert np.abs(product.debug_math(2, unit, quarter) / oracle_math(2, unit, quarter) - 1.0).max() < 1e-5      # v_exp_f32(y * v_log_f32(x))
