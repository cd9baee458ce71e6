"""Pins the CPU oracle against the golden vectors of the reference's own unit tests.

Each test names the reference test it replays (paths relative to /root/reference/tests/OptiXRendererTests/).
All CPU, no GPU. The oracle runs its float-table ("host") mode here, as the reference tests do.
"""
import math

import numpy as np
import pytest

from oracle_bindings import (MODEL_DEFAULT, MODEL_GGX, MODEL_GGX_R, MODEL_GGX_T, MODEL_OREN_NAYAR, MODEL_TRANSMISSIVE,
                             get_oracle)
from bifrost3d_amd import capi

NAN = float("nan")


def unorm16(v):
    return float(np.float32(int(min(max(v, 0.0), 1.0) * 65535.0 + 0.5)) / np.float32(65535.0))


def shading_params(m, cos_theta=NAN, max_pdf_hint=NAN):
    """Parameter block of MODEL_DEFAULT / MODEL_TRANSMISSIVE. coat values pass through UNorm16 like Material does."""
    return np.array(list(m["tint"]) + [m["roughness"], m["specularity"], m["metallic"], unorm16(m["coat"]), unorm16(m["coat_roughness"]),
                                       cos_theta, max_pdf_hint], np.float32)


def normalize(v):
    v = np.asarray(v, np.float32)
    return (v * (np.float32(1.0) / np.sqrt(np.dot(v, v), dtype=np.float32))).astype(np.float32)


def w_from_cos_theta(c):
    c = np.float32(c)
    return np.array([np.sqrt(np.float32(1) - c * c), 0.0, c], np.float32)


def rng3_sample02(o, count):
    return np.array([list(o.sample02(s)) + [(s + 0.5) / count] for s in range(count)], np.float32)


@pytest.fixture(scope="module")
def pmj(oracle):
    return oracle.pmjbn(16384)


def pmj3(pmj, n):
    third = ((np.arange(n, dtype=np.float32) + np.float32(0.5)) / np.float32(n)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([pmj[:n], third[:, None]], axis=1), np.float32)


def rho_estimate(oracle, model, params, wo, u):
    """directional_hemispherical_reflectance_function of BSDFTestUtils.h:48-93: mean and std-dev of f |cos| / pdf."""
    s = oracle.bsdf_sample(model, params, wo, u)
    pdf = np.abs(s[:, 3])
    valid = pdf > 1e-6
    w = np.zeros((len(s), 3), np.float32)
    w[valid] = (s[valid, 0:3] * np.abs(s[valid, 6:7])) / pdf[valid, None]
    w64 = w.astype(np.float64)
    mean = w64.mean(axis=0)
    std = np.sqrt(np.maximum((w64 ** 2).mean(axis=0) - mean ** 2, 0.0))
    direction = (w64.sum(axis=1)[:, None] * s[:, 4:7].astype(np.float64)).sum(axis=0)
    n = np.linalg.norm(direction)
    return mean, std, (direction / n if n > 0 else direction)


# ------------------------------------------------------------------------------------------------
# G1 / G2: exact regression vectors
# ------------------------------------------------------------------------------------------------
def test_G1_default_shading_regression(oracle, goldens):
    g = goldens["G1_default_shading_regression"]
    u = rng3_sample02(oracle, 2)
    k = 0
    for name in g["materials"]:
        params = shading_params(goldens["materials"][name])
        for wo in g["wos"]:
            wo = normalize(wo)
            out = oracle.bsdf_sample(MODEL_DEFAULT, params, wo, u)
            for s in range(2):
                expected = np.array(g["responses"][k]); k += 1
                np.testing.assert_allclose(out[s, 0:3], expected[0:3], rtol=g["relative_tolerance"], err_msg=f"{name} wo={wo} s={s}")
                assert abs(abs(out[s, 3]) - expected[3]) <= expected[3] * g["relative_tolerance"], (name, wo, s, out[s, 3], expected[3])
    assert k == 18


def test_G2_transmissive_shading_regression(oracle, goldens):
    g = goldens["G2_transmissive_shading_regression"]
    m = goldens["materials"][g["material"]]
    u = rng3_sample02(oracle, 2)
    k = 0
    for c in g["cos_theta_os"]:
        wo = w_from_cos_theta(c)
        params = shading_params(m, cos_theta=float(wo[2]))
        wo[2] = abs(wo[2])
        out = oracle.bsdf_sample(MODEL_TRANSMISSIVE, params, wo, u)
        for s in range(2):
            expected = np.array(g["responses"][k]); k += 1
            np.testing.assert_allclose(out[s, 0:3], expected[0:3], rtol=g["relative_tolerance"], err_msg=f"cos={c} s={s}")
            assert abs(abs(out[s, 3]) - expected[3]) <= expected[3] * g["relative_tolerance"]
    assert k == 8


# ------------------------------------------------------------------------------------------------
# G3: sampling standard deviations (BSDF_sampling_variance_test, BSDFTestUtils.h:95-106)
# ------------------------------------------------------------------------------------------------
def average_std_dev(oracle, model, params, u, cos_thetas):
    total = np.zeros(3)
    for c in cos_thetas:
        mean, std, _ = rho_estimate(oracle, model, params, w_from_cos_theta(c), u)
        total += std / mean
    return total / len(cos_thetas)


def test_G3_ggx_r_std_dev(oracle, goldens, pmj):
    g = goldens["G3_sampling_std_dev"]
    u = pmj3(pmj, g["sample_count"])
    sd = average_std_dev(oracle, MODEL_GGX_R, [g["ggx_r"]["alpha"], 1, 1, 1], u, g["cos_thetas"])
    assert np.all(np.abs(sd - g["ggx_r"]["expected"]) <= g["tolerance"]), sd


def test_G3_ggx_t_std_dev(oracle, goldens, pmj):
    g = goldens["G3_sampling_std_dev"]
    u = pmj3(pmj, g["sample_count"])
    for ior, expected in zip(g["ggx_t"]["iors"], g["ggx_t"]["expected"]):
        sd = average_std_dev(oracle, MODEL_GGX_T, [g["ggx_t"]["alpha"], ior], u, g["cos_thetas"])
        assert np.all(np.abs(sd - expected) <= g["tolerance"]), (ior, sd)


def test_G3_ggx_std_dev(oracle, goldens, pmj):
    g = goldens["G3_sampling_std_dev"]
    u = pmj3(pmj, g["sample_count"])
    for ior, expected in zip(g["ggx"]["iors"], g["ggx"]["expected"]):
        sd = average_std_dev(oracle, MODEL_GGX, [g["ggx"]["alpha"], g["ggx"]["specularity"], ior, 1, 1, 1], u, g["cos_thetas"])
        assert np.all(np.abs(sd - expected) <= g["tolerance"]), (ior, sd)


def test_G3_oren_nayar_std_dev(oracle, goldens, pmj):
    g = goldens["G3_sampling_std_dev"]
    u = pmj3(pmj, g["sample_count"])
    for roughness, expected in zip(g["oren_nayar"]["roughness"], g["oren_nayar"]["expected"]):
        sd = average_std_dev(oracle, MODEL_OREN_NAYAR, [1, 1, 1, roughness, 1], u, g["cos_thetas"])
        assert np.all(np.abs(sd - expected) <= g["tolerance"]), (roughness, sd)


# ------------------------------------------------------------------------------------------------
# G4: precomputed table spot checks
# ------------------------------------------------------------------------------------------------
def test_G4_ggx_rho_tables(oracle, goldens, pmj):
    g = goldens["G4_tables"]["ggx_rho"]
    u = pmj3(pmj, g["sample_count"])
    middle_cos = (32 // 2) / 31.0
    middle_rough = (32 // 2) / 31.0
    middle_alpha = max(1e-4, np.float32(middle_rough) ** 2)
    out = np.zeros(2, np.float32)
    for c in (0.000001, middle_cos, 1.0):
        wo = w_from_cos_theta(c)
        for alpha in (1e-9, middle_alpha, 1.0):
            roughness = math.sqrt(alpha)
            oracle.lib.oracle_specular_rho(c, roughness, out.ctypes.data_as(capi.C.POINTER(capi.c_f)))
            base = rho_estimate(oracle, MODEL_GGX_R, [alpha, 0, 0, 0], wo, u)[0][0]
            full = rho_estimate(oracle, MODEL_GGX_R, [alpha, 1, 1, 1], wo, u)[0][0]
            assert abs(base - out[0]) <= g["tolerance"], (c, alpha, base, out[0])
            assert abs(full - out[1]) <= g["tolerance"], (c, alpha, full, out[1])


def test_G4_dielectric_rho_tables(oracle, goldens, pmj):
    g = goldens["G4_tables"]["dielectric_rho"]
    u = pmj3(pmj, g["sample_count"])
    middle = (16 // 2) / 15.0
    middle_alpha = max(1e-4, np.float32(middle) ** 2)
    iors = [0.331492, 1.0 / 1.5, 0.789474, 1.26667, 1.5, 3.01667]
    out = np.zeros(2, np.float32)
    for c in (1 / 15.0, middle, 1.0):
        wo = w_from_cos_theta(c)
        for alpha in (1e-4, middle_alpha, 1.0):
            roughness = float(np.sqrt(np.float32(alpha)))
            for ior in iors:
                specularity = oracle.lib.oracle_dielectric_specularity(1.0, ior)
                rho = rho_estimate(oracle, MODEL_GGX, [alpha, specularity, ior, 1, 0, 0], wo, u)[0]
                oracle.lib.oracle_dielectric_rho(c, roughness, ior, out.ctypes.data_as(capi.C.POINTER(capi.c_f)))
                assert abs(rho[0] - out[0]) <= g["total_tolerance"], (c, alpha, ior, rho[0], out[0])
                assert abs(rho[1] - out[1]) <= g["reflected_tolerance"], (c, alpha, ior, rho[1], out[1])


def test_G4_alpha_from_max_PDF_brackets(oracle, goldens):
    n = goldens["G4_tables"]["alpha_from_pdf"]["sample_count"]
    max_alpha_error = 1.0 / 32
    for i in range(n):
        s = oracle.sample02(i)
        cos_theta, max_pdf = float(s[0]), float(s[1] / (1 - s[1]))
        alpha = oracle.lib.oracle_estimate_alpha(cos_theta, max_pdf)
        wo = w_from_cos_theta(cos_theta)
        wi = np.array([-wo[0], -wo[1], wo[2]], np.float32)
        pdf = abs(oracle.bsdf_eval(MODEL_GGX_R, [alpha, 1, 1, 1], wo, wi, which=2)[0, 3])
        shifted = min(max(alpha + max_alpha_error * (-1 if pdf < max_pdf else 1), 0.0), 1.0)
        shifted_pdf = abs(oracle.bsdf_eval(MODEL_GGX_R, [shifted, 1, 1, 1], wo, wi, which=2)[0, 3])
        passed = (pdf <= max_pdf <= shifted_pdf) or (shifted_pdf <= max_pdf <= pdf)
        invalid = (shifted == 0.0 and shifted_pdf < max_pdf) or (shifted == 1.0 and max_pdf < shifted_pdf)
        assert passed or invalid, (i, cos_theta, max_pdf, alpha, pdf, shifted_pdf)


def test_minimum_roughness_edge_cases(oracle):
    """ShadingModelUtils.GGX_minimum_roughness_edge_case_handling, ORT/ShadingModels/UtilsTest.h:65-99"""
    f = oracle.lib.oracle_min_roughness_from_PDF
    assert f(0.5, -1.0) == 0.0                       # delta dirac
    assert f(0.5, float("inf")) == pytest.approx(0.0, abs=1e-7)
    assert f(0.5, NAN) == pytest.approx(0.0, abs=1e-7)
    assert f(0.5, 0.0) == pytest.approx(1.0)


# ------------------------------------------------------------------------------------------------
# G5: analytic identities
# ------------------------------------------------------------------------------------------------
def test_G5_white_hot_room(oracle):
    """DefaultShadingModel.white_hot_room, ORT/ShadingModels/DefaultShadingTest.h:234-252: rho == 1 (EXPECT_FLOAT_EQ)."""
    for metallic in (0.0, 0.5, 1.0):
        for roughness in (0.0, 0.5, 1.0):
            for a in range(5):
                c = 1.0 - a * 0.2
                m = dict(tint=[1, 1, 1], roughness=roughness, specularity=0.04, metallic=metallic, coat=0, coat_roughness=0)
                rho = oracle.default_shading_info(shading_params(m), c)["rho"]
                assert abs(rho[0] - 1.0) <= 4 * np.finfo(np.float32).eps, (metallic, roughness, c, rho)


def test_G5_power_conservation(oracle, goldens, pmj):
    """DefaultShadingModel.power_conservation, ORT/ShadingModels/DefaultShadingTest.h:60-78 (8192 samples)."""
    u = pmj3(pmj, 8192)
    for roughness in (0.0, 0.5, 0.9):
        for c in (0.1, 0.4, 0.7, 1.0):
            m = dict(tint=[1, 1, 1], roughness=roughness, specularity=0.02, metallic=0.0, coat=0, coat_roughness=0)
            mean, _, _ = rho_estimate(oracle, MODEL_DEFAULT, shading_params(m), w_from_cos_theta(c), u)
            assert np.all(np.abs(mean - 1.0) <= goldens["G5_identities"]["power_conservation"]["tolerance"]), (roughness, c, mean)


def test_G5_sampling_probabilities(oracle, goldens):
    tol = goldens["G5_identities"]["sampling_probabilities"]["tolerance"]
    for c in (0.5, 1.0):
        for roughness in (0.25, 0.75):
            for metallic in (0.0, 1.0):
                for coat in (0.0, 0.5, 1.0):
                    for coat_roughness in (0.25, 0.75):
                        m = dict(tint=[1, 1, 1], roughness=roughness, specularity=0.04, metallic=metallic, coat=coat, coat_roughness=coat_roughness)
                        info = oracle.default_shading_info(shading_params(m), c)
                        total = info["diffuse_probability"] + info["specular_probability"] + info["coat_probability"]
                        assert abs(total - 1.0) <= tol


def test_G5_metallic_interpolation(oracle, goldens):
    tol = goldens["G5_identities"]["metallic_lerp"]["tolerance"]
    for roughness in (0.0, 0.5, 1.0):
        for metallic in (0.25, 0.5, 0.75):
            for c in (0.2, 0.4, 0.6, 0.8, 1.0):
                def rho(mt):
                    m = dict(tint=[1.0, 0.5, 0.25], roughness=roughness, specularity=0.04, metallic=mt, coat=0, coat_roughness=0)
                    return oracle.default_shading_info(shading_params(m), c)["rho"]
                lerp = rho(0.0) + np.float32(metallic) * (rho(1.0) - rho(0.0))
                assert np.all(np.abs(lerp - rho(metallic)) <= tol)


def test_G5_coat_interpolation(oracle, goldens):
    """DefaultShadingModel.coat_interpolation (DefaultShadingTest.h:326-408): a partial coat is the interpolation of the uncoated and
    the fully coated material once all three are brought to the same effective roughness (a rough coat roughens the base; the
    partially coated material's input roughness is found by the reference's bisection). rho within 1 %, specularity bounded."""
    plastic = goldens["materials"]["plastic"]

    def model(roughness, coat, coat_roughness, cos_theta):
        m = dict(plastic, roughness=roughness, coat=coat, coat_roughness=coat_roughness)
        return oracle.default_shading_info(shading_params(m), cos_theta)

    def with_target_roughness(coat, coat_roughness, cos_theta, target):
        roughness, adjustment = 0.5, 0.25                      # in f64 like the reference; the material takes the f32 value
        previous = np.float32(roughness)
        while True:
            info = model(float(np.float32(roughness)), coat, coat_roughness, cos_theta)
            if abs(float(info["roughness"]) - target) < 1e-8:
                return info
            roughness += -adjustment if info["roughness"] > target else adjustment
            if np.float32(roughness) == previous:
                return info
            previous = np.float32(roughness)
            adjustment *= 0.5

    for coat_roughness in (0.0, 0.5, 1.0):
        for cos_theta in (0.2, 0.4, 0.6, 0.8, 1.0):
            coated = model(plastic["roughness"], 1.0, coat_roughness, cos_theta)
            if coat_roughness > 0.0:
                assert np.float32(plastic["roughness"]) < coated["roughness"]
            plain = model(float(coated["roughness"]), 0.0, coat_roughness, cos_theta)
            assert abs(float(plain["roughness"]) - float(coated["roughness"])) <= 4 * np.spacing(coated["roughness"])      # EXPECT_FLOAT_EQ: 4 ULP
            for coat in (0.25, 0.5, 0.75):
                partial = with_target_roughness(coat, coat_roughness, cos_theta, float(coated["roughness"]))
                assert abs(float(partial["roughness"]) - float(coated["roughness"])) <= 1e-6
                assert coated["specularity"][0] <= partial["specularity"][0] <= plain["specularity"][0]
                expected = plain["rho"] + np.float32(coat) * (coated["rho"] - plain["rho"])
                assert np.all(np.abs(expected - partial["rho"]) <= 0.01 * np.abs(expected)), (coat_roughness, cos_theta, coat, expected, partial["rho"])


def test_G5_default_shading_consistency(oracle, goldens):
    """ShadingModelTestUtils::consistency_test (ShadingModelTestUtils.h:51-66): sample == evaluate_with_PDF within 2e-5."""
    wo = normalize([1, 1, 1])
    u = rng3_sample02(oracle, 32)
    for name in ("gold", "plastic", "coated_plastic"):
        for roughness in (0.2, 0.4, 0.6, 0.8, 1.0):
            m = dict(goldens["materials"][name]); m["roughness"] = roughness
            p = shading_params(m)
            s = oracle.bsdf_sample(MODEL_DEFAULT, p, wo, u)
            valid = np.abs(s[:, 3]) > 1e-6
            r = oracle.bsdf_eval(MODEL_DEFAULT, p, wo, s[:, 4:7])
            np.testing.assert_allclose(r[valid, 0:3], s[valid, 0:3], rtol=2e-5, atol=0)
            np.testing.assert_allclose(np.abs(r[valid, 3]), np.abs(s[valid, 3]), rtol=2e-5)


def test_G5_oren_nayar(oracle, goldens, pmj):
    g = goldens["G5_identities"]
    wo = normalize([1, 1, 1])
    u = pmj3(pmj, 2048)
    for roughness in (0.0, 0.2, 0.4, 0.6, 0.8, 1.0):
        mean, _, _ = rho_estimate(oracle, MODEL_OREN_NAYAR, [1, 1, 1, roughness, 1], wo, u)
        assert np.all(np.abs(mean - 1.0) <= g["oren_nayar_power"]["tolerance"]), (roughness, mean)
    albedo = np.array([0.25, 0.5, 0.75])
    for roughness in (0.25, 0.5, 0.75):
        for c in (0.1, 0.5, 0.9):
            mean, _, _ = rho_estimate(oracle, MODEL_OREN_NAYAR, list(albedo) + [roughness, 1], w_from_cos_theta(c), u)
            assert np.all(np.abs(mean - albedo) <= g["oren_nayar_albedo"]["tolerance"]), (roughness, c, mean)
    for c in (0.1, 0.5, 0.9):
        for roughness in (0.1, 0.5, 0.9):
            assert abs(oracle.lib.oracle_E_FON(c, roughness, 1) - oracle.lib.oracle_E_FON(c, roughness, 0)) <= g["E_FON"]["tolerance"]


def test_bsdf_consistency(oracle, pmj):
    """BSDF_consistency_test (BSDFTestUtils.h:122-141) for GGX_R (GGXTest.h:93-99), GGX_T (:306-316), GGX (:468-480), OrenNayar (OrenNayarTest.h:68-75)."""
    u = pmj3(pmj, 16)
    wo = normalize([1, 1, 1])

    def check(model, params, wo):
        s = oracle.bsdf_sample(model, params, wo, u)
        valid = np.abs(s[:, 3]) > 1e-6
        if not valid.any():
            return
        r = oracle.bsdf_eval(model, params, wo, s[:, 4:7])
        np.testing.assert_allclose(r[valid, 0:3], s[valid, 0:3], rtol=2e-5, atol=1e-12)
        np.testing.assert_allclose(np.abs(r[valid, 3]), np.abs(s[valid, 3]), rtol=2e-5)

    for alpha in (0.0675, 0.125, 0.25, 0.5, 1.0):
        check(MODEL_GGX_R, [alpha, 1, 1, 1], wo)
    for roughness in (0.0, 0.5, 1.0):
        check(MODEL_OREN_NAYAR, [1, 1, 1, roughness, 1], wo)
    for ior in (0.5, 0.9, 1.1, 1.5):
        for c in (-1.0, -0.4, -0.1, 0.1, 0.4, 1.0):
            for alpha in (0.0675, 0.125, 0.25, 0.5, 1.0):
                check(MODEL_GGX_T, [alpha, ior], w_from_cos_theta(c))
            for alpha in (0.0675, 0.25, 1.0):
                for tint in (0.5, 1.0):
                    spec = oracle.lib.oracle_dielectric_specularity(1.0, ior)
                    check(MODEL_GGX, [alpha, spec, ior, tint, tint, tint], w_from_cos_theta(c))


def test_ggx_specular_limits(oracle, pmj):
    """GGX.sample_according_to_specularity-style check (GGXTest.h:540-558): smooth black-tinted glass reflects exactly `specularity`."""
    u = pmj3(pmj, 1024)
    for c in (-1.0, 1.0):
        for ior in (0.5, 1.5):
            for spec in (0.0, 0.5, 1.0):
                mean, _, _ = rho_estimate(oracle, MODEL_GGX, [1e-4, spec, ior, 0, 0, 0], np.array([0, 0, c], np.float32), u)
                assert abs(mean[0] - spec) <= 1e-5, (c, ior, spec, mean)


# ------------------------------------------------------------------------------------------------
# G6: thin sheet approximation + Sobol sampler statistics
# ------------------------------------------------------------------------------------------------
def test_G6_thin_sheet_rmse(oracle, goldens):
    g = goldens["G6_thin_sheet"]
    tint = np.array(g["transmission_tint"], np.float32)
    tint_side = np.sqrt(tint).astype(np.float32)
    fp = capi.C.POINTER(capi.c_f)
    se_r, se_t, n = np.zeros(3), np.zeros(3), 0
    out = np.zeros(6, np.float32)
    for roughness in g["roughness"]:
        alpha = max(1e-4, float(np.float32(roughness) * np.float32(roughness)))
        for ior in g["iors"]:
            spec = oracle.lib.oracle_dielectric_specularity(1.0, ior)
            for c in g["cos_thetas"]:
                wo = w_from_cos_theta(c)
                oracle.lib.oracle_integrate_thin_sheet(tint_side.ctypes.data_as(fp), alpha, spec, ior, wo.ctypes.data_as(fp),
                                                       g["path_count"], g["bounce_count"], out.ctypes.data_as(fp))
                expected = out.astype(np.float64).copy()
                expected /= (expected[0] + expected[3])
                approx = np.zeros(6, np.float32)
                oracle.lib.oracle_thin_sheet(c, roughness, ior, tint.ctypes.data_as(fp), approx.ctypes.data_as(fp))
                se_r += (expected[0:3] - approx[0:3]) ** 2
                se_t += (expected[3:6] - approx[3:6]) ** 2
                n += 1
    rmse_r, rmse_t = np.sqrt(se_r / n), np.sqrt(se_t / n)
    assert np.all(np.abs(rmse_r - g["expected_reflection_rmse"]) <= g["tolerance"]), rmse_r
    assert np.all(np.abs(rmse_t - g["expected_transmission_rmse"]) <= g["tolerance"]), rmse_t


# ------------------------------------------------------------------------------------------------
# G7: MIS / PDF semantics, specularity constants, normals
# ------------------------------------------------------------------------------------------------
def test_G7_PDF_wrapper_semantics(oracle):
    """MonteCarlo.PDF (MiscTest.h:81-139): plain, too small, delta-dirac and invalid PDFs, before and after disable_MIS; and
    GGX.zero_roughness_converted_to_effectively_smooth_alpha (GGXTest.h:449-453)."""
    import ctypes as C
    oracle.lib.oracle_pdf_semantics.argtypes = [C.c_int, C.c_float, C.c_int, C.POINTER(C.c_float)]

    def state(kind, value=0.0, disable=False):
        out = np.zeros(4, np.float32)
        oracle.lib.oracle_pdf_semantics(kind, value, int(disable), out.ctypes.data_as(C.POINTER(C.c_float)))
        return float(out[0]), bool(out[1]), bool(out[2]), bool(out[3])          # value, is_valid, use_for_MIS, is_delta_dirac

    assert state(0, 0.5) == (0.5, True, True, False)
    assert state(0, 0.5, disable=True) == (0.5, True, False, True)
    tiny = float(np.float32(1e-6) * np.float32(0.5))                             # MIN_VALID_PDF * 0.5
    assert state(0, tiny) == (tiny, False, False, False)
    assert state(0, tiny, disable=True) == (tiny, False, False, True)
    assert state(1, 1.0)[1:] == (True, False, True) and state(1, 1.0, disable=True)[1:] == (True, False, True)
    assert state(2)[1:] == (False, False, True) and state(2, disable=True)[1:] == (False, False, True)
    assert oracle.lib.oracle_ggx_effectively_smooth_roughness(C.c_float(0.0)) == 1
    assert oracle.lib.oracle_ggx_effectively_smooth_roughness(C.c_float(0.1)) == 0


def test_G7_balance_heuristic(oracle):
    b, p = oracle.lib.oracle_balance_heuristic, oracle.lib.oracle_power_heuristic
    assert b(1.0, 1.0) == pytest.approx(0.5)
    assert b(1.0, 3.0) == pytest.approx(0.25)
    assert b(1.0, NAN) == 1.0
    almost_inf = 3.0e38
    assert p(0.9 * math.sqrt(almost_inf), math.sqrt(almost_inf)) == pytest.approx(0.0)
    assert p(math.sqrt(almost_inf), 0.9 * math.sqrt(almost_inf)) == pytest.approx(1.0)


def test_G7_specularity_constants(oracle, goldens):
    g = goldens["G7_misc"]["specularity"]
    lib = oracle.lib
    assert lib.oracle_dielectric_specularity(1.0, g["water_ior"]) == pytest.approx(g["water"], rel=4e-7)
    assert lib.oracle_dielectric_specularity(1.0, g["glass_ior"]) == pytest.approx(g["glass"], rel=4e-7)
    assert lib.oracle_dielectric_ior_from_specularity(g["water"]) == pytest.approx(g["water_ior"], rel=4e-7)
    assert lib.oracle_dielectric_ior_from_specularity(g["glass"]) == pytest.approx(g["glass_ior"], rel=4e-7)
    c = goldens["G7_misc"]["conductors"]
    one = [1.0, 1.0, 1.0]
    for metal in ("gold", "titanium"):
        spec = oracle.vec3_call("oracle_conductor_specularity", one, c[f"{metal}_ior"], c[f"{metal}_extinction"])
        np.testing.assert_allclose(spec, c[f"{metal}_specularity"], rtol=c["relative_tolerance"])
        ior = oracle.vec3_call("oracle_conductor_ior_from_specularity", c[f"{metal}_specularity"], c[f"{metal}_extinction"])
        np.testing.assert_allclose(ior, c[f"{metal}_ior"], rtol=c["relative_tolerance"])
    for base_ior in (1.0, 1.2, 1.4):   # Specularity.scaling_dielectric_specularity_under_coat, MiscTest.h:216-230
        expected = lib.oracle_dielectric_specularity(1.5, base_ior)
        actual = lib.oracle_adjust_dielectric_specularity(1.5, lib.oracle_dielectric_specularity(1.0, base_ior))
        assert abs(expected - actual) <= 1e-7
    coat = [1.5, 1.5, 1.5]           # Specularity.scaling_conductor_specularity_under_coat, MiscTest.h:233-250
    for base_ior in (c["gold_ior"], c["titanium_ior"]):
        for extinction in (c["gold_extinction"], c["titanium_extinction"]):
            expected = oracle.vec3_call("oracle_conductor_specularity", coat, base_ior, extinction)
            through_air = oracle.vec3_call("oracle_conductor_specularity", one, base_ior, extinction)
            actual = oracle.vec3_call("oracle_adjust_conductor_specularity", coat, list(map(float, through_air)), extinction)
            np.testing.assert_allclose(actual, expected, atol=0.02)


def test_G7_fix_backfacing_shading_normal(oracle):
    n = [0, 0, 1]
    for w, target, eps in ((normalize([1, 0, -0.1]), 0.0, 1e-6), ([1, 0, 0], 0.002, 1e-5), (normalize([1, 0, -0.1]), 0.002, 1e-5)):
        fixed = oracle.vec3_call("oracle_fix_backfacing_shading_normal", list(map(float, w)), n, target)
        assert abs(float(np.dot(np.asarray(w, np.float32), fixed)) - target) <= eps
    same = oracle.vec3_call("oracle_fix_backfacing_shading_normal", list(map(float, normalize([1, 0, 1]))), n, 0.0)
    assert list(same) == [0, 0, 1]


def test_G7_refract_matches_optix_semantics(oracle):
    """Trigonometry.refract_overload_gives_same_result_as_optix_implementation, ORT/MiscTest.h:292-322."""
    fp = capi.C.POINTER(capi.c_f)
    up = np.array([0, 0, 1], np.float32)
    for s in range(16):
        u = oracle.sample02(s)
        wo = oracle.vec3_call("oracle_uniform_hemisphere", u)
        for ior in (0.33, 0.7, 1.5, 3.0):
            e, a = np.zeros(3, np.float32), np.zeros(3, np.float32)
            ok_e = oracle.lib.oracle_refract(wo.ctypes.data_as(fp), up.ctypes.data_as(fp), ior, e.ctypes.data_as(fp))
            ok_a = oracle.lib.oracle_refract_z(wo.ctypes.data_as(fp), ior, a.ctypes.data_as(fp))
            assert ok_e == ok_a
            if ok_a:
                np.testing.assert_allclose(a, e, atol=1e-6)
                c = np.zeros(1, np.float32)
                assert oracle.lib.oracle_refract_cos(float(wo[2]), ior, c.ctypes.data_as(fp))
                assert abs(c[0] - e[2]) <= 1e-6


def test_G7_octahedral_decode_is_unit_length(oracle):
    enc = np.array([[x * 3276, y * 3276] for x in range(-10, 11) for y in range(-10, 11)], np.int16)
    out = np.zeros((len(enc), 3), np.float32)
    oracle.lib.oracle_decode_octahedral(enc.ctypes.data_as(capi.C.POINTER(capi.C.c_int16)), len(enc), out.ctypes.data_as(capi.C.POINTER(capi.c_f)))
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=2e-7)
    assert np.allclose(out[(enc == 0).all(axis=1)], [0, 0, 1])


# ------------------------------------------------------------------------------------------------
# Sampler sanity: pure function, well distributed (the exact bits are unpinned by the reference).
# ------------------------------------------------------------------------------------------------
def test_scrambled_sobol_is_stratified(oracle):
    n = 256
    t = np.array([[i, 12345, 7] for i in range(n)], np.uint32)
    s = oracle.sobol4ui(t)
    for dim in range(4):   # Owen scrambling preserves the (0,1)-sequence property: one point per 1/256 stratum
        assert len(set((s[:, dim] >> 24).tolist())) == n
    assert np.array_equal(s, oracle.sobol4ui(t))
    assert not np.array_equal(s, oracle.sobol4ui(np.array([[i, 12346, 7] for i in range(n)], np.uint32)))


def test_G4_table_sizes_and_PDF_encoding(oracle):
    """ShadingModelUtils.{GGX_rho,dielectric_GGX_rho,GGX_minimum_roughness}_texture_size and ..._PDF_encoding_consistent
    (ORT/ShadingModels/UtilsTest.h:23-66): the table data and the code that indexes them agree on the sizes, and the
    minimum-roughness lookup encodes the PDF the way the table was fitted (Assets/Shading/EstimateGGXBoundedVNDFAlpha.cpp:88-100)."""
    from bifrost3d_amd import capi
    base, full, light, dense, alpha = capi.load_tables()
    assert base.size == 32 * 32 and full.size == 32 * 32                       # GGX_angle_sample_count x GGX_roughness_sample_count
    assert light.size == 16 * 16 * 16 * 2 and dense.size == 16 * 16 * 16 * 2      # dielectric angle x roughness x ior, (total, reflected)
    assert alpha.size == 32 * 32                                                # max_PDF_sample_count x wo_dot_normal_sample_count
    table = alpha.reshape(32, 32)       # rows: cos theta, columns: encoded PDF

    def bilinear(u, v):                 # BF/Math/ImageSampling.h:18-61, clamped texel-centre-free lookup over [0, 1]
        x, y = min(max(u, 0.0), 1.0) * 31, min(max(v, 0.0), 1.0) * 31
        x0, y0 = min(int(x), 30), min(int(y), 30)
        fx, fy = x - x0, y - y0
        return ((table[y0, x0] * (1 - fx) + table[y0, x0 + 1] * fx) * (1 - fy) + (table[y0 + 1, x0] * (1 - fx) + table[y0 + 1, x0 + 1] * fx) * fy)

    for pdf in (0.1, 1.0, 10.0, 1000.0, 100000.0):
        encoded = (pdf / (1.0 + pdf) - 0.13) / 0.87
        for cos_theta in (0.1, 0.5, 0.9):
            assert oracle.lib.oracle_estimate_alpha(cos_theta, pdf) == pytest.approx(float(bilinear(encoded, cos_theta)), rel=1e-4, abs=1e-6)
