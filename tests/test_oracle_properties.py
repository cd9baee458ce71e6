"""The reference's remaining BSDF and shading-model property tests replayed against the CPU oracle: power
conservation, Helmholtz reciprocity, PDF positivity, grazing-angle and Fresnel behaviour, Snell's law, hemisphere symmetry,
GGX against its reflection / transmission halves, rho tables against Monte Carlo estimates, thin sheets, MIS heuristics.
Paths relative to /root/reference/tests/OptiXRendererTests/. Together with test_oracle_goldens.py (regression vectors and
variance constants) and test_oracle_lights.py this is every test the reference holds for the code on the path."""
import math

import numpy as np
import pytest

from bifrost3d_amd import capi
from oracle_bindings import MODEL_DEFAULT, MODEL_GGX, MODEL_GGX_R, MODEL_GGX_T, MODEL_OREN_NAYAR, MODEL_TRANSMISSIVE
from test_oracle_goldens import normalize, pmj3, rho_estimate, shading_params, w_from_cos_theta

NAN = float("nan")
FLT_MAX = 3.4028234663852886e38


@pytest.fixture(scope="module")
def pmj(oracle):
    return oracle.pmjbn(16384)


def valid(pdf):
    return np.abs(pdf) > 1e-6       # PDF::is_valid, OR/Types.h: MIN_VALID_PDF


def oren_nayar(roughness):
    return [1, 1, 1, roughness, 1]


def ggx_r(alpha, specularity=1.0):
    return [alpha, specularity, specularity, specularity]


def ggx(alpha, ior, tint=1.0, specularity=None, oracle=None):
    spec = oracle.lib.oracle_dielectric_specularity(1.0, ior) if specularity is None else specularity
    return [alpha, spec, ior, tint, tint, tint]


GOLD = dict(tint=[1.0, 0.766, 0.336], roughness=0.02, specularity=1.0, metallic=1.0, coat=0.0, coat_roughness=0.0)
PLASTIC = dict(tint=[0.02, 0.27, 0.33], roughness=0.7, specularity=0.02, metallic=0.0, coat=0.0, coat_roughness=0.0)
COATED_PLASTIC = dict(PLASTIC, coat=1.0, coat_roughness=0.7)
SMOOTH_GLASS = dict(tint=[0.95, 0.97, 0.95], roughness=0.0, specularity=0.04, metallic=0.0, coat=0.0, coat_roughness=0.0)


def helmholtz_reciprocity(oracle, model, params, wo, u):       # BSDFTestUtils.h:110-120
    s = oracle.bsdf_sample(model, params, wo, u)
    ok = valid(s[:, 3])
    if ok.any():
        wi = s[ok, 4:7]
        swapped = np.array([oracle.bsdf_eval(model, params, w, wo[None, :], which=1)[0, 0:3] for w in wi])
        np.testing.assert_allclose(swapped, s[ok, 0:3], atol=1e-4)


def pdf_positivity(oracle, pmj, model, params, wo, count):       # BSDFTestUtils.h:143-155
    wi = np.zeros((count, 3), np.float32)
    fp = capi.C.POINTER(capi.c_f)
    for i in range(count):
        u = np.ascontiguousarray(pmj[i], np.float32)
        oracle.lib.oracle_uniform_sphere(u.ctypes.data_as(fp), wi[i].ctypes.data_as(fp))
    r = oracle.bsdf_eval(model, params, wo, wi)
    assert (r[:, 0:3] >= 0.0).all()
    lit = (r[:, 0:3] > 0.0).any(axis=1)
    assert (np.abs(r[lit, 3]) > 0.0).all()


# ---- BSDFs/OrenNayarTest.h -------------------------------------------------------------------------------------------------------

def test_oren_nayar_power_conservation(oracle, pmj):       # :51-59
    wo = normalize([1, 1, 1])
    for roughness in (0.0, 0.2, 0.4, 0.6, 0.8, 1.0):
        mean, _, _ = rho_estimate(oracle, MODEL_OREN_NAYAR, oren_nayar(roughness), wo, pmj3(pmj, 2048))
        assert np.all(np.abs(mean - 1.0) <= 0.00045), (roughness, mean)


def test_oren_nayar_helmholtz_reciprocity(oracle, pmj):       # :61-67
    for roughness in (0.0, 0.5, 1.0):
        helmholtz_reciprocity(oracle, MODEL_OREN_NAYAR, oren_nayar(roughness), normalize([1, 1, 1]), pmj3(pmj, 16))


# ---- BSDFs/GGXTest.h: GGX_R ----------------------------------------------------------------------------------------------------------

def test_ggx_r_power_conservation(oracle, pmj):       # :74-83
    u = pmj3(pmj, 1024)
    for c in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0):
        for alpha in (0.0, 0.0675, 0.125, 0.25, 0.5, 1.0):
            mean, _, _ = rho_estimate(oracle, MODEL_GGX_R, ggx_r(alpha), w_from_cos_theta(c), u)
            assert np.all(mean <= 1.0), (c, alpha, mean)


def test_ggx_r_helmholtz_reciprocity(oracle, pmj):       # :85-91
    for alpha in (0.0675, 0.125, 0.25, 0.5, 1.0):
        helmholtz_reciprocity(oracle, MODEL_GGX_R, ggx_r(alpha), normalize([1, 1, 1]), pmj3(pmj, 16))


def test_ggx_r_pdf_positivity(oracle, pmj):       # :101-109
    for c in (-0.8, -0.4, 0.1, 0.5, 0.9):
        for alpha in (0.2, 0.6, 1.0):
            pdf_positivity(oracle, pmj, MODEL_GGX_R, ggx_r(alpha), w_from_cos_theta(c), 128)


def test_ggx_r_minimal_alpha_and_grazing_angles(oracle):       # :118-161
    min_alpha = 0.0      # alpha_from_roughness(0) clamps to the minimal alpha inside the BSDF
    incident, grazing = np.array([0, 0, 1], np.float32), normalize([0.0, 1.0, 0.001])
    mirrored = np.array([grazing[0], -grazing[1], grazing[2]], np.float32)
    for wo, wi in ((incident, incident), (grazing, incident), (grazing, grazing), (grazing, mirrored)):
        assert not np.isnan(oracle.bsdf_eval(MODEL_GGX_R, ggx_r(min_alpha), wo, wi[None, :], which=1)).any()
    flat = np.array([0, 1, 0], np.float32)
    for alpha in (0.0, 0.5, 1.0):       # fully_grazing_evaluates_to_black
        for wo, wi in ((flat, incident), (incident, flat), (flat, flat)):
            assert abs(float(oracle.bsdf_eval(MODEL_GGX_R, ggx_r(alpha), wo, wi[None, :], which=1)[0, 0])) < 1e-30


# ---- BSDFs/GGXTest.h: GGX_T ----------------------------------------------------------------------------------------------------------

def test_ggx_t_power_conservation(oracle, pmj):       # :283-293
    u = pmj3(pmj, 1024)
    for ior in (0.5, 0.9, 1.1, 1.5):
        for c in (-1.0, -0.7, -0.4, -0.1, 0.1, 0.4, 0.7, 1.0):
            for alpha in (0.0, 0.0675, 0.125, 0.25, 0.5, 1.0):
                mean, _, _ = rho_estimate(oracle, MODEL_GGX_T, [alpha, ior], w_from_cos_theta(c), u)
                assert mean[0] <= 1.0, (ior, c, alpha, mean)


def test_ggx_t_pdf_positivity(oracle, pmj):       # :317-325
    for c in (-0.8, -0.4, 0.1, 0.5, 0.9):
        for alpha in (0.2, 0.6, 1.0):
            pdf_positivity(oracle, pmj, MODEL_GGX_T, [alpha, 0.5], w_from_cos_theta(c), 128)


def test_ggx_t_consistent_sampling_across_hemispheres(oracle):       # :337-356
    u = np.array([[0.5, 0.5, 0.5]], np.float32)
    for c in (-1.0, -0.4, -0.1, 0.1, 0.4, 1.0):
        positive = w_from_cos_theta(c)
        negative = positive * np.array([1, 1, -1], np.float32)
        for alpha in (0.0675, 0.125, 0.25, 0.5, 1.0):
            for ior in (0.5, 0.9, 1.1, 1.5):
                p, n = oracle.bsdf_sample(MODEL_GGX_T, [alpha, ior], positive, u)[0], oracle.bsdf_sample(MODEL_GGX_T, [alpha, ior], negative, u)[0]
                assert p[3] == n[3]
                np.testing.assert_allclose(p[0:3], n[0:3], rtol=4e-7)
                np.testing.assert_allclose(p[4:7], n[4:7] * np.array([1, 1, -1]), rtol=4e-7, atol=1e-12)


def test_ggx_t_fully_grazing_evaluates_to_black(oracle):       # :358-378
    incident, flat = np.array([0, 0, -1], np.float32), np.array([0, 1, 0], np.float32)
    for alpha in (0.0, 0.5, 1.0):
        for ior in (0.5, 0.9, 1.1, 1.5):
            for wo, wi in ((flat, incident), (incident, flat), (flat, flat)):
                assert abs(float(oracle.bsdf_eval(MODEL_GGX_T, [alpha, ior], wo, wi[None, :], which=1)[0, 0])) < 1e-30


def test_ggx_t_snells_law(oracle):       # :380-395
    for c in (0.2, 0.5, 0.9):
        wo = w_from_cos_theta(c)
        wi = oracle.bsdf_sample(MODEL_GGX_T, [0.0, 2.0], wo, np.array([[0.5, 0.5, 0.5]], np.float32))[0, 4:7]
        sin_o, sin_i = math.sqrt(max(0.0, 1 - float(wo[2]) ** 2)), math.sqrt(max(0.0, 1 - float(wi[2]) ** 2))
        assert 1.0 * sin_o == pytest.approx(2.0 * sin_i, abs=1e-6)


# ---- BSDFs/GGXTest.h: GGX ------------------------------------------------------------------------------------------------------------

def test_ggx_power_conservation(oracle, pmj):       # :455-465
    u = pmj3(pmj, 1024)
    for ior in (0.5, 0.9, 1.1, 1.5):
        for c in (-1.0, -0.7, -0.4, -0.1, 0.1, 0.4, 0.7, 1.0):
            for alpha in (0.0, 0.0675, 0.125, 0.25, 0.5, 1.0):
                mean, _, _ = rho_estimate(oracle, MODEL_GGX, ggx(alpha, ior, oracle=oracle), w_from_cos_theta(c), u)
                assert np.all(mean <= 1.0 + 1e-5), (ior, c, alpha, mean)


def test_ggx_pdf_positivity(oracle, pmj):       # :480-490
    for c in (-0.8, -0.4, 0.1, 0.5, 0.9):
        for alpha in (0.2, 0.6, 1.0):
            pdf_positivity(oracle, pmj, MODEL_GGX, ggx(alpha, 1.5, oracle=oracle), w_from_cos_theta(c), 128)


def test_ggx_reflection_reflectance_equals_ggx_r(oracle, pmj):       # :492-511
    for c in (0.2, 1.0):
        wo = w_from_cos_theta(c)
        for alpha in (0.0675, 0.25, 1.0):
            both, _, both_direction = rho_estimate(oracle, MODEL_GGX, ggx(alpha, 1.5, specularity=1.0), wo, pmj3(pmj, 4096))      # transmission disabled
            only, _, only_direction = rho_estimate(oracle, MODEL_GGX_R, ggx_r(alpha), wo, pmj3(pmj, 2048))
            assert np.all(np.abs(both - only) <= 0.001), (c, alpha, both, only)
            assert float(both_direction @ only_direction) == pytest.approx(1.0, abs=0.002)


def test_ggx_transmission_reflectance_equals_ggx_t(oracle, pmj):       # :513-536
    u = pmj3(pmj, 4096)
    for ior in (0.5, 1.5):
        for c in (0.4, 1.0):
            wo = w_from_cos_theta(c)
            for alpha in (0.0675, 0.25, 1.0):
                # GGXWrapper with disable_reflection: samples that stay in wo's hemisphere are rejected
                s = oracle.bsdf_sample(MODEL_GGX, ggx(alpha, ior, specularity=0.0), wo, u)
                keep = valid(s[:, 3]) & (np.sign(s[:, 6]) != np.sign(wo[2]))
                weights = np.zeros((len(s), 3))
                weights[keep] = s[keep, 0:3].astype(np.float64) * np.abs(s[keep, 6:7]) / np.abs(s[keep, 3:4])
                direction = (weights.sum(axis=1)[:, None] * s[:, 4:7]).sum(axis=0)
                # GGXTransmissionWrapper(alpha, ior, specularity = 0): GGX_T ignores Fresnel, the wrapper multiplies 1 - schlick(0, wo . h) in
                t = oracle.bsdf_sample(MODEL_GGX_T, [alpha, ior], wo, u)
                ok = valid(t[:, 3])
                halfway = wo[None, :].astype(np.float64) + ior * t[:, 4:7].astype(np.float64)
                halfway /= np.linalg.norm(halfway, axis=1, keepdims=True)
                halfway[halfway[:, 2] < 0.0] *= -1.0
                transmitted = 1.0 - (1.0 - halfway @ wo.astype(np.float64)) ** 5
                only = np.zeros((len(t), 3))
                only[ok] = t[ok, 0:3].astype(np.float64) * transmitted[ok, None] * np.abs(t[ok, 6:7]) / np.abs(t[ok, 3:4])
                only_direction = (only.sum(axis=1)[:, None] * t[:, 4:7]).sum(axis=0)
                assert np.all(np.abs(weights.mean(axis=0) - only.mean(axis=0)) <= 0.0015), (ior, c, alpha, weights.mean(axis=0), only.mean(axis=0))
                assert float(direction / np.linalg.norm(direction) @ (only_direction / np.linalg.norm(only_direction))) == pytest.approx(1.0, abs=0.002)


def test_ggx_black_transmission_never_sampled(oracle, pmj):       # :571-591
    u = pmj3(pmj, 128)
    for alpha in (0.2, 0.6, 1.0):
        for c in (0.1, 0.5, 0.9):
            s = oracle.bsdf_sample(MODEL_GGX, ggx(alpha, 1.5, tint=0.0, oracle=oracle), w_from_cos_theta(c), u)
            assert (s[:, 6] >= 0.0).all()


def test_ggx_fully_grazing_evaluates_to_black(oracle):       # :593-617
    grazing_wo, grazing_wi = np.array([0, 1, 0], np.float32), np.array([0, -1, 0], np.float32)
    for alpha in (0.0, 0.5, 1.0):
        for ior in (0.5, 0.9, 1.1, 1.5):
            params = ggx(alpha, ior, specularity=1.0)
            for z in (-0.1, 0.0, 0.1):
                offset = np.array([0, 0, z], np.float32)
                assert abs(float(oracle.bsdf_eval(MODEL_GGX, params, grazing_wo, normalize(grazing_wi + offset)[None, :], which=1)[0, 0])) < 1e-30
                assert abs(float(oracle.bsdf_eval(MODEL_GGX, params, normalize(grazing_wo + offset), grazing_wi[None, :], which=1)[0, 0])) < 1e-30


# ---- ShadingModels/DefaultShadingTest.h --------------------------------------------------------------------------------------------

def sample02_3(oracle, count):       # ShadingModelTestUtils.h consistency_test: (sample02(i), (i + 0.5) / count)
    return np.array([list(oracle.sample02(i)) + [(i + 0.5) / count] for i in range(count)], np.float32)


def shading_consistency(oracle, model, material, wo, count):       # ShadingModelTestUtils.h:49-68
    params = shading_params(material, float(wo[2]))
    s = oracle.bsdf_sample(model, params, wo, sample02_3(oracle, count))
    assert (np.abs(s[:, 3]) >= 0.0).all()
    ok = valid(s[:, 3])
    if ok.any():
        assert (s[ok, 0] >= 0.0).all()
        r = oracle.bsdf_eval(model, params, wo, s[ok, 4:7])
        np.testing.assert_allclose(r[:, 0:3], s[ok, 0:3], rtol=2e-5, atol=1e-12)
        np.testing.assert_allclose(np.abs(r[:, 3]), np.abs(s[ok, 3]), rtol=2e-5)


def test_default_shading_function_consistency(oracle):       # :110-124
    wo = normalize([1, 1, 1])
    for material in (GOLD, PLASTIC, COATED_PLASTIC):
        for roughness in (0.2, 0.4, 0.6, 0.8, 1.0):
            shading_consistency(oracle, MODEL_DEFAULT, dict(material, roughness=roughness), wo, 32)


def test_default_shading_pdf_positivity(oracle, pmj):       # :126-142
    for material in (GOLD, PLASTIC, COATED_PLASTIC):
        for c in (-0.8, -0.4, 0.1, 0.5, 0.9):
            wo = w_from_cos_theta(c)
            for roughness in (0.2, 0.6, 1.0):
                pdf_positivity(oracle, pmj, MODEL_DEFAULT, shading_params(dict(material, roughness=roughness), float(wo[2])), wo, 128)


def test_default_shading_fresnel(oracle):       # :144-196
    up = np.array([0, 0, 1], np.float32)
    grazing_wo, grazing_wi = normalize([0.0, 1.0, 0.001]), normalize([0.0, -1.0, 0.001])
    red = dict(tint=[1.0, 0.0, 0.0], roughness=0.02, specularity=0.0, metallic=0.0, coat=0.0, coat_roughness=0.0)
    incident = oracle.bsdf_eval(MODEL_DEFAULT, shading_params(red, 1.0), up, up[None, :])[0]
    assert incident[0] > 0.0 and abs(incident[1]) < 1e-6 and abs(incident[2]) < 1e-6      # incident reflectivity of a non-metal is its (red) diffuse tint
    grazing = oracle.bsdf_eval(MODEL_DEFAULT, shading_params(red, float(grazing_wo[2])), grazing_wo, grazing_wi[None, :])[0]
    assert grazing[0] > 0.99 and grazing[0] == pytest.approx(grazing[1], rel=1e-6) and grazing[0] == pytest.approx(grazing[2], rel=1e-6)      # grazing: white
    metal = oracle.bsdf_eval(MODEL_DEFAULT, shading_params(GOLD, 1.0), up, up[None, :])[0]
    np.testing.assert_allclose(metal[0:3] * (GOLD["tint"][0] / metal[0]), GOLD["tint"], atol=1e-6)      # metals reflect their tint
    metal_grazing = oracle.bsdf_eval(MODEL_DEFAULT, shading_params(GOLD, float(grazing_wo[2])), grazing_wo, grazing_wi[None, :])[0]
    assert metal_grazing[1] > 0.99 and metal_grazing[0] == pytest.approx(metal_grazing[1], rel=0.01) and metal_grazing[2] == pytest.approx(metal_grazing[1], rel=0.01)


def test_default_shading_rho_matches_monte_carlo_estimate(oracle, pmj):       # :198-232
    u = pmj3(pmj, 8192)
    for wo in (np.array([0, 0, 1], np.float32), normalize([1, 0, 1]), w_from_cos_theta(1.0 / 31.0)):
        for roughness in (0.25, 0.75):
            for metallic in (0.0, 0.5, 1.0):
                for coat in (0.0, 0.5, 1.0):
                    for coat_roughness in (0.25, 0.75):
                        material = dict(tint=[1.0, 0.5, 0.25], roughness=roughness, specularity=0.04, metallic=metallic, coat=coat, coat_roughness=coat_roughness)
                        params = shading_params(material, float(wo[2]))
                        expected, _, _ = rho_estimate(oracle, MODEL_DEFAULT, params, wo, u)
                        actual = oracle.default_shading_info(params, float(wo[2]))["rho"]
                        tolerance = 0.015 * (2 - roughness) * (2 - coat_roughness)
                        assert np.all(np.abs(np.asarray(actual) - expected) <= tolerance * expected), (wo, material, expected, actual)


# ---- ShadingModels/TransmissiveShadingTest.h ----------------------------------------------------------------------------------------

def transmissive_rho(oracle, params, cos_theta):
    out = np.zeros(3, np.float32)
    fp = capi.C.POINTER(capi.c_f)
    oracle.lib.oracle_transmissive_rho(np.ascontiguousarray(params, np.float32).ctypes.data_as(fp), cos_theta, out.ctypes.data_as(fp))
    return out


def test_transmissive_shading_rho_matches_monte_carlo_estimate(oracle, pmj):       # :64-92
    u = pmj3(pmj, 16384)
    for wo in (np.array([0, 0, 1], np.float32), normalize([1, 0, 1]), w_from_cos_theta(1.0 / 15.0)):
        for roughness in (0.25, 0.75):
            params = shading_params(dict(tint=[1.0, 0.5, 0.25], roughness=roughness, specularity=0.04, metallic=0.0, coat=0.0, coat_roughness=0.0), float(wo[2]))
            expected, _, _ = rho_estimate(oracle, MODEL_TRANSMISSIVE, params, wo, u)
            actual = transmissive_rho(oracle, params, float(wo[2]))
            assert np.all(np.abs(actual - expected) <= 0.015 * (2 - roughness) * expected), (wo, roughness, expected, actual)


def test_transmissive_shading_white_hot_furnace(oracle, pmj):       # :94-114
    u = pmj3(pmj, 8192)
    for medium_ior in (1.26667, 1.5, 3.01667):
        specularity = oracle.lib.oracle_dielectric_specularity(1.0, medium_ior)
        for roughness in (0.2, 0.7):
            for c in (0.1, 0.4, 0.7, 1.0):
                wo = w_from_cos_theta(c)
                params = shading_params(dict(tint=[1.0, 1.0, 1.0], roughness=roughness, specularity=specularity, metallic=0.0, coat=0.0, coat_roughness=0.0), float(wo[2]))
                mean, _, _ = rho_estimate(oracle, MODEL_TRANSMISSIVE, params, wo, u)
                assert mean[0] == pytest.approx(1.0, abs=0.026), (medium_ior, roughness, c, mean)


def test_transmissive_shading_function_consistency(oracle):       # :116-129
    for c in (-1.0, -0.7, -0.4, -0.1, 0.1, 0.4, 0.7, 1.0):
        for roughness in (0.2, 0.4, 0.6, 0.8, 1.0):
            shading_consistency(oracle, MODEL_TRANSMISSIVE, dict(SMOOTH_GLASS, roughness=roughness), w_from_cos_theta(c), 32)


def test_transmissive_shading_pdf_positivity(oracle, pmj):       # :131-142
    for c in (-0.8, -0.4, 0.1, 0.5, 0.9):
        wo = w_from_cos_theta(c)
        for roughness in (0.2, 0.6, 1.0):
            pdf_positivity(oracle, pmj, MODEL_TRANSMISSIVE, shading_params(dict(SMOOTH_GLASS, roughness=roughness), float(wo[2])), wo, 1024)


def test_transmissive_shading_fresnel(oracle):       # :144-174
    # roughness just above the minimal alpha: GGX::roughness_from_alpha(MIN_ALPHA + 1e-6)
    material = dict(SMOOTH_GLASS, specularity=0.0, roughness=math.sqrt(capi_min_alpha() + 1e-6))
    up = np.array([0, 0, 1], np.float32)
    incident = oracle.bsdf_eval(MODEL_TRANSMISSIVE, shading_params(material, 1.0), up, up[None, :])[0]
    assert np.all(np.abs(incident[0:3]) < 1e-30)       # no reflection at normal incidence with zero specularity
    grazing_wo, grazing_wi = normalize([0.0, 1.0, 0.001]), normalize([0.0, -1.0, 0.001])
    grazing = oracle.bsdf_eval(MODEL_TRANSMISSIVE, shading_params(material, float(grazing_wo[2])), grazing_wo, grazing_wi[None, :])[0]
    assert grazing[0] > 0.99 and grazing[0] == pytest.approx(grazing[1], rel=1e-6) and grazing[0] == pytest.approx(grazing[2], rel=1e-6)


def capi_min_alpha():
    return 1e-4             # GGX::MIN_ALPHA, ORS/BSDFs/GGX.h:33


def test_transmissive_shading_snells_law(oracle):       # :176-201
    specularity = oracle.lib.oracle_dielectric_specularity(1.0, 2.0)
    for c in (0.2, 0.5, 0.9):
        wo = w_from_cos_theta(c)
        params = shading_params(dict(SMOOTH_GLASS, specularity=specularity), float(wo[2]))
        wi = oracle.bsdf_sample(MODEL_TRANSMISSIVE, params, wo, np.array([[0.5, 0.5, 1.0]], np.float32))[0, 4:7]      # third sample 1: always the BTDF
        assert wi[2] < 0.0
        sin_o, sin_i = math.sqrt(max(0.0, 1 - float(wo[2]) ** 2)), math.sqrt(max(0.0, 1 - float(wi[2]) ** 2))
        assert 1.0 * sin_o == pytest.approx(2.0 * sin_i, abs=1e-3)


# ---- ShadingModels/UtilsTest.h: thin sheets ------------------------------------------------------------------------------------------

def smooth_thin_sheet_reflectance(oracle, cos_theta_o, medium_ior, tint):       # BSDFTestUtils.h:228-265, the closed form of the bounce series
    specularity = oracle.lib.oracle_dielectric_specularity(1.0, medium_ior)
    tint_per_side = np.sqrt(np.asarray(tint, np.float64))
    refracted = capi.c_f()
    if not oracle.lib.oracle_refract_cos(-abs(cos_theta_o), medium_ior, capi.C.byref(refracted)):
        return np.ones(3), np.zeros(3)

    def schlick(f0, c):
        return f0 + (1 - f0) * (1 - c) ** 5

    # dielectric_schlick_fresnel(specularity, cos_theta, ior_i_over_o) = plain Schlick when entering the denser medium
    r0, ri = schlick(specularity, cos_theta_o), schlick(specularity, abs(refracted.value))
    t0, ti = (1 - r0) * tint_per_side, (1 - ri) * tint_per_side
    return r0 + (ri * t0 * ti) / (1 - ri * ri), (t0 * ti) / (1 - ri * ri)


def test_smooth_thin_sheets_follow_the_closed_form(oracle):       # :115-165
    tint = np.array([1.0, 0.5, 0.25], np.float32)
    tint_per_side = np.sqrt(tint).astype(np.float32)
    fp = capi.C.POINTER(capi.c_f)
    for medium_ior in (1.26667, 1.5, 3.01667):
        specularity = oracle.lib.oracle_dielectric_specularity(1.0, medium_ior)
        for c in (0.3, 0.5, 1.0):
            expected_reflected, expected_transmitted = smooth_thin_sheet_reflectance(oracle, c, medium_ior, tint)
            integrated = np.zeros(6, np.float32)
            wo = w_from_cos_theta(c)
            oracle.lib.oracle_integrate_thin_sheet(tint_per_side.ctypes.data_as(fp), 0.0, specularity, medium_ior, wo.ctypes.data_as(fp), 4096, 32, integrated.ctypes.data_as(fp))
            np.testing.assert_allclose(integrated[0:3], expected_reflected, atol=0.01)       # smooth_ggx_thin_sheet_reflects_according_to_expectation
            np.testing.assert_allclose(integrated[3:6], expected_transmitted, atol=0.01)
            approximated = np.zeros(6, np.float32)
            oracle.lib.oracle_thin_sheet(c, 0.0, medium_ior, tint.ctypes.data_as(fp), approximated.ctypes.data_as(fp))
            np.testing.assert_allclose(approximated[0:3], expected_reflected, atol=0.025)    # approx_smooth_ggx_thin_sheet_is_nearly_exact_for_smooth_surfaces
            np.testing.assert_allclose(approximated[3:6], expected_transmitted, atol=0.025)


# ---- MiscTest.h: MIS heuristics ------------------------------------------------------------------------------------------------------

def test_balance_and_power_heuristic_invariants(oracle):       # :54-79, :141-164
    b, p = oracle.lib.oracle_balance_heuristic, oracle.lib.oracle_power_heuristic
    inf = float("inf")
    assert b(1.0, 1.0) == pytest.approx(0.5) and b(1.0, 3.0) == pytest.approx(0.25) and b(1.0, NAN) == 1.0
    assert b(1.0, FLT_MAX) == pytest.approx(1.0 / FLT_MAX) and b(FLT_MAX, 1.0) == pytest.approx(1.0)
    assert b(0.5 * FLT_MAX, FLT_MAX) == pytest.approx(0.0, abs=1e-30) and b(FLT_MAX, 0.5 * FLT_MAX) == pytest.approx(1.0)
    assert b(1.0, inf) == 0.0 and b(inf, 1.0) == pytest.approx(1.0)
    assert b(0.0, 0.0) == 0.0 and b(0.0, 1.0) == 0.0 and b(0.0, FLT_MAX) == 0.0
    assert p(1.0, 1.0) == pytest.approx(0.5) and p(1.0, 3.0) == pytest.approx(0.1) and p(1.0, NAN) == 1.0
    assert p(1.0, FLT_MAX) == pytest.approx(0.0, abs=1e-30) and p(FLT_MAX, 1.0) == pytest.approx(1.0)
    assert p(0.0, 0.0) == 0.0 and p(0.0, 1.0) == 0.0 and p(0.0, FLT_MAX) == 0.0
    root = math.sqrt(FLT_MAX)
    assert p(0.9 * root, root) == pytest.approx(0.0, abs=1e-30) and p(root, 0.9 * root) == pytest.approx(1.0)


def test_the_specified_normal_interpolation_order_is_the_references_up_to_rounding():
    """device and oracle take every vertex normal to world space first (M n_i, not normalised), interpolate there and normalise once; the reference
    interpolates in object space, normalises, transforms with rtTransformNormal and normalises again (ORS/TriangleAttributes.cu:58-61, ORS/MonteCarlo.cu:176). For the
    transforms an instance can carry -- rotation, uniform scale, translation (BF/Math/Transform.h:28-34) -- M is linear with M^-T proportional to M, so the two orders give
    the same direction up to f32 rounding. Held here in f32 arithmetic, statement for statement, on 20 000 random triangles / transforms / barycentrics (ADVICE round 5:
    the oracle states the device's order, so the equivalence with the reference's order is checked on its own)."""
    rng = np.random.default_rng(17)
    n = 20000
    f = np.float32

    def unit(v):
        return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(f)

    normals = unit(rng.normal(size=(n, 3, 3)))                       # three vertex normals per triangle, a few degrees to wildly apart
    normals[: n // 2] = unit(normals[: n // 2, :1] + 0.2 * rng.normal(size=(n // 2, 3, 3)))      # half of them a smooth surface: neighbours within ~10 degrees
    quaternion = unit(rng.normal(size=(n, 4)))
    w, x, y, z = (quaternion[:, k] for k in (3, 0, 1, 2))
    rotation = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                         np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                         np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1).astype(f)
    scale = np.exp(rng.uniform(-3, 3, n)).astype(f)
    M = (rotation * scale[:, None, None]).astype(f)
    u, v = rng.random(n, dtype=f), rng.random(n, dtype=f)
    flip = u + v > 1
    u, v = np.where(flip, 1 - u, u).astype(f), np.where(flip, 1 - v, v).astype(f)
    weights = np.stack([f(1) - u - v, u, v], -1).astype(f)

    def transform(matrix, vectors):      # f32 products and sums, one row at a time like the device's M[0] * o.x + M[1] * o.y + M[2] * o.z
        return ((matrix[..., 0] * vectors[..., None, 0]).astype(f) + (matrix[..., 1] * vectors[..., None, 1]).astype(f) + (matrix[..., 2] * vectors[..., None, 2]).astype(f)).astype(f)

    def interpolate(three):
        return ((three[:, 0] * weights[:, 0, None]).astype(f) + (three[:, 1] * weights[:, 1, None]).astype(f) + (three[:, 2] * weights[:, 2, None]).astype(f)).astype(f)

    specified = unit(interpolate(np.stack([transform(M, normals[:, k]) for k in range(3)], 1)))
    inverse_transpose = (rotation / scale[:, None, None]).astype(f)      # rtTransformNormal: the inverse transpose of M
    reference = unit(transform(inverse_transpose, unit(interpolate(normals))))
    length = np.linalg.norm(interpolate(normals).astype(np.float64), axis=-1)      # opposing normals cancel: the shorter the sum, the more its rounding shows in the direction
    apart = np.linalg.norm(specified.astype(np.float64) - reference.astype(np.float64), axis=-1)
    smooth = slice(0, n // 2)
    assert length[smooth].min() > 0.5 and apart[smooth].max() < 1e-6, (float(length[smooth].min()), float(apart[smooth].max()))      # a few ulp of a unit vector
    assert np.median(apart) < 2.5e-7 and (apart * length).max() < 2e-6, (float(np.median(apart)), float((apart * length).max()))
