"""The reference's own golden vectors for the shading models, evaluated by the DEVICE code of the shade kernel
(hipr_debug_shading runs DefaultShading / TransmissiveShading / DiffuseShading exactly as k_shade does, in the fast-math
translation unit): G1 DefaultShadingModel.regression_test (ORT/ShadingModels/DefaultShadingTest.h:410-447) and G2
TransmissiveShadingModel.regression_test (TransmissiveShadingTest.h:203-236), with the reference's tolerance (1e-4 relative),
plus device-vs-oracle agreement on random inputs."""
import json
from pathlib import Path

import numpy as np
import pytest

from bifrost3d_amd import capi
from test_oracle_goldens import normalize, rng3_sample02, shading_params, w_from_cos_theta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from bifrost3d_amd.renderer import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle_bindings import get_oracle
    return get_oracle(True)   # unorm16 tables, as on the device


@pytest.fixture(scope="module")
def goldens():
    return json.loads((Path(__file__).parent / "golden" / "reference_goldens.json").read_text())


def test_G1_default_shading_regression_on_device(ctx, oracle, goldens):
    g = goldens["G1_default_shading_regression"]
    u = rng3_sample02(oracle, 2)
    k = 0
    for name in g["materials"]:
        params = shading_params(goldens["materials"][name])
        for wo in g["wos"]:
            wo = normalize(wo)
            out = ctx.debug_shading(capi.SHADING_DEFAULT, params, wo, u)
            for s in range(2):
                expected = np.array(g["responses"][k]); k += 1
                np.testing.assert_allclose(out[s, 0:3], expected[0:3], rtol=g["relative_tolerance"], err_msg=f"{name} wo={wo} s={s}")
                assert abs(abs(out[s, 3]) - expected[3]) <= expected[3] * g["relative_tolerance"], (name, wo, s, out[s, 3], expected[3])
    assert k == 18


def test_G2_transmissive_shading_regression_on_device(ctx, oracle, goldens):
    g = goldens["G2_transmissive_shading_regression"]
    m = goldens["materials"][g["material"]]
    u = rng3_sample02(oracle, 2)
    k = 0
    for c in g["cos_theta_os"]:
        wo = w_from_cos_theta(c)
        params = shading_params(m, cos_theta=float(wo[2]))
        wo[2] = abs(wo[2])
        out = ctx.debug_shading(capi.SHADING_TRANSMISSIVE, params, wo, u)
        for s in range(2):
            expected = np.array(g["responses"][k]); k += 1
            np.testing.assert_allclose(out[s, 0:3], expected[0:3], rtol=g["relative_tolerance"], err_msg=f"cos={c} s={s}")
            assert abs(abs(out[s, 3]) - expected[3]) <= expected[3] * g["relative_tolerance"]
    assert k == 8


@pytest.mark.parametrize("model,oracle_model", [(0, 4), (1, 6), (2, 5)])   # HIPR_SHADING_* -> oracle MODEL_*_SHADING ids
def test_device_shading_models_follow_the_oracle(ctx, oracle, goldens, model, oracle_model):
    """Sampling and evaluation of the three shading models on random directions and numbers, device vs oracle. The device uses
    hardware-approximate divide / sqrt / sin / cos / pow (like the reference's --use_fast_math PTX): tolerance 2e-3 relative
    (+ 1e-5 absolute) on f and pdf for at least 99 % of the samples, the rest being discrete decisions (lobe choice, total internal
    reflection) that flip when a random number sits within rounding of a threshold."""
    rng = np.random.default_rng(17 + model)
    n = 4000
    for name in ("gold", "plastic", "coated_plastic") if model != 2 else ("frosted_glass",):
        if name not in goldens["materials"]:
            continue
        wo = normalize([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.05, 1.0)])
        params = shading_params(goldens["materials"][name])
        u = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        gpu = ctx.debug_shading(model, params, wo, u, mode=0)
        cpu = oracle.bsdf_sample(oracle_model, params, wo, u)
        same_lobe = np.abs(gpu[:, 4:7] - cpu[:, 4:7]).max(axis=1) <= 2e-3
        assert same_lobe.mean() >= 0.99, (name, float(same_lobe.mean()))
        err = np.abs(gpu[same_lobe, 0:4] - cpu[same_lobe, 0:4]) / (np.abs(cpu[same_lobe, 0:4]) + 1e-2)
        assert (err.max(axis=1) <= 2e-3).mean() >= 0.99, (name, float(np.quantile(err.max(axis=1), 0.99)))

        wi = rng.normal(size=(n, 3)).astype(np.float32)
        wi /= np.linalg.norm(wi, axis=1, keepdims=True)
        if model != 2:
            wi[:, 2] = np.abs(wi[:, 2])
        gpu = ctx.debug_shading(model, params, wo, wi, mode=1)
        cpu = oracle.bsdf_eval(oracle_model, params, wo, wi)
        # an impossible refraction configuration has f = 0 and an invalid (NaN) PDF on both sides
        finite = np.isfinite(cpu[:, 0:4]).all(axis=1)
        assert np.array_equal(finite, np.isfinite(gpu[:, 0:4]).all(axis=1))
        assert finite.mean() > 0.5
        err = np.abs(gpu[finite, 0:4] - cpu[finite, 0:4]) / (np.abs(cpu[finite, 0:4]) + 1e-2)
        assert (err.max(axis=1) <= 2e-3).mean() >= 0.99, (name, float(np.quantile(err.max(axis=1), 0.99)))


# ---- G8: the reference's light tests replayed on the device (ORT/LightSources/SphereLightTest.h, SpotLightTest.h) ---------------

def test_G8_lights_on_device(ctx, oracle):
    import math
    from test_oracle_lights import cosine_sample, pdf_is_valid, samples02, sphere_light, spot_light
    u1024, u16 = samples02(oracle, 1024), samples02(oracle, 16)

    # SphereLight.power_preservation_when_radius_changes
    normal = np.array([0.0, 1.0, 0.0])
    for radius, tolerance in {0.0: 1e-4, 1.0: 0.0001, 2.0: 0.001, 5.0: 0.001, 9.0: 0.004}.items():
        s = ctx.debug_light(sphere_light((0.0, 10.0, 0.0), radius, 10.0), np.zeros(3), u1024)
        luminances = s[:, 0].astype(np.float64) * (s[:, 4:7].astype(np.float64) @ normal) / np.abs(s[:, 3].astype(np.float64))
        assert abs(math.fsum(luminances) / 1024 * (4.0 * math.pi * 100.0) - 10.0) < tolerance, radius

    # device == oracle on the samples themselves (the shade TU's approximate divide / sqrt / sincos: 1e-5 relative)
    rng = np.random.default_rng(3)
    for light in (sphere_light((0.3, 2.0, -0.4), 0.5, 7.0), sphere_light((0.0, 1.0, 0.0), 0.0, 3.0), spot_light((0.1, 3.0, 0.2), (0.0, -1.0, 0.0), 0.7, 5.0, 0.6),
                  spot_light((0.0, 2.0, 0.0), (0.6, -0.8, 0.0), 0.0, 5.0, 0.8)):
        position = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
        gpu, cpu = ctx.debug_light(light, position, u1024[:256]), oracle.light_sample(light, position, u1024[:256])
        assert np.allclose(gpu, cpu, rtol=2e-5, atol=2e-6)

    # SpotLight.consistent_PDF_and_radiance
    light_position, light_direction = np.array([0.0, 10.0, 0.0]), np.array([0.0, -1.0, 0.0])
    for p in range(16):
        position = (light_position + 2.0 * light_direction + 2.0 * cosine_sample(u16[p])).astype(np.float32)
        for radius in (1.0, 4.0, 13.0):
            for cos_angle in (0.1, 0.5, 0.9):
                light = spot_light(light_position, light_direction, radius, 10.0, cos_angle)
                sampled = ctx.debug_light(light, position, u16)
                evaluated = ctx.debug_light(light, position, sampled[:, 4:7], mode=1)
                for s, e in zip(sampled, evaluated):
                    if pdf_is_valid(float(s[3])) or pdf_is_valid(float(e[3])):
                        assert float(e[3]) == pytest.approx(float(s[3]), rel=1e-4)
                    if s[0] > 0.0:
                        assert float(e[0]) == pytest.approx(float(s[0]), abs=1e-4)

    # SpotLight.pdf_rejects_rays_that_miss
    hit = np.array([1.0, 10.0, 0.0]) / math.sqrt(101.0)
    miss = np.array([3.0, 10.0, 0.0]) / math.sqrt(109.0)
    towards = ctx.debug_light(spot_light((0.0, 10.0, 0.0), (0.0, -1.0, 0.0), 2.0, 10.0, 0.5), np.zeros(3), np.stack([hit, miss]), mode=1)
    assert pdf_is_valid(float(towards[0, 3])) and not pdf_is_valid(float(towards[1, 3]))
    away = ctx.debug_light(spot_light((0.0, 10.0, 0.0), (0.0, 1.0, 0.0), 2.0, 10.0, 0.5), np.zeros(3), np.stack([hit, miss]), mode=1)
    assert not pdf_is_valid(float(away[0, 3])) and not pdf_is_valid(float(away[1, 3]))
    with pytest.raises(capi.HiprError):
        ctx.debug_light(sphere_light((0, 1, 0), 1.0, 1.0), np.zeros(3), np.stack([hit]), mode=1)
