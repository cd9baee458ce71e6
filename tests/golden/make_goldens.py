#!/usr/bin/env python3
"""Writes tests/golden/reference_goldens.json: the golden vectors of the reference's own tests.

Paths are relative to /root/reference/tests/OptiXRendererTests/ ("ORT/"). Only numbers and the
parameters they were produced with are recorded.
"""
import json
from pathlib import Path

G = {}

# Materials of ORT/ShadingModels/ShadingModelTestUtils.h:21-44 and ORT/ShadingModels/TransmissiveShadingTest.h:24-38
G["materials"] = {
    "gold": dict(tint=[1.0, 0.766, 0.336], roughness=0.02, metallic=1.0, specularity=1.0, coat=0.0, coat_roughness=0.0),
    "plastic": dict(tint=[0.02, 0.27, 0.33], roughness=0.7, metallic=0.0, specularity=0.02, coat=0.0, coat_roughness=0.0),
    "coated_plastic": dict(tint=[0.02, 0.27, 0.33], roughness=0.7, metallic=0.0, specularity=0.02, coat=1.0, coat_roughness=0.7),
    "frosted_glass": dict(tint=[0.95, 0.97, 0.95], roughness=0.2, metallic=0.0, specularity=0.04, coat=0.0, coat_roughness=0.0),
}

# G1: DefaultShadingModel.regression_test, ORT/ShadingModels/DefaultShadingTest.h:410-447.
# 3 materials x 3 wo x 2 samples, rng = (RNG::sample02(s), (s + 0.5) / 2); {f.r, f.g, f.b, pdf}, relative tolerance 1e-4.
G["G1_default_shading_regression"] = dict(
    source="ORT/ShadingModels/DefaultShadingTest.h:410-447",
    materials=["gold", "plastic", "coated_plastic"],
    wos=[[0.0, 0.0, 1.0], [1.0, 0.0, 1.0], [1.0, 0.0, 0.01]],  # normalised by the test
    relative_tolerance=1e-4,
    responses=[
        [497358.250000, 380976.437500, 167112.35938, 497357.968750], [124339.296875, 95243.906250, 41778.00000, 124339.195313],
        [994714.562500, 762453.062500, 335647.75000, 703369.687500], [249080.015625, 190921.531250, 84049.10156, 175985.171875],
        [4957685248.0, 4900781568.0, 4796215808.0, 49668972.0], [1455754624.0, 1439689728.0, 1410168448.0, 13442245.0],
        [0.011624, 0.076557, 0.09214, 0.010905], [0.012486, 0.092185, 0.11131, 0.230607],
        [0.012840, 0.122771, 0.14915, 0.034218], [0.011330, 0.121562, 0.14802, 0.254778],
        [0.051809, 0.085369, 0.09342, 0.286622], [0.013969, 0.145090, 0.17656, 0.218950],
        [0.019217, 0.081176, 0.09605, 0.0164565], [0.019548, 0.0975228, 0.116237, 0.228887],
        [0.017939, 0.128357, 0.15486, 0.0377722], [0.014534, 0.125507, 0.15214, 0.239682],
        [0.088401, 0.115091, 0.12150, 0.317704], [0.018240, 0.147322, 0.17830, 0.192018]])

# G2: TransmissiveShadingModel.regression_test, ORT/ShadingModels/TransmissiveShadingTest.h:203-236.
G["G2_transmissive_shading_regression"] = dict(
    source="ORT/ShadingModels/TransmissiveShadingTest.h:203-236",
    material="frosted_glass", cos_theta_os=[-0.7, -0.1, 0.4, 1.0], relative_tolerance=1e-4,
    responses=[
        [102.196815, 102.196815, 102.196815, 70.955925], [30.308733, 30.308733, 30.308733, 19.304911],
        [4075.826172, 4075.826172, 4075.826172, 445.397308], [4660.390625, 4660.390625, 4660.390625, 235.904617],
        [610.321655, 623.170593, 610.321655, 504.575867], [149.033539, 152.171082, 149.033539, 125.492897],
        [1633.225708, 1667.609497, 1633.225708, 1715.760010], [408.740875, 417.345978, 408.740875, 429.358185]])

# G3: sampling standard deviations, 1024 PMJ-BN samples, averaged over cos_theta {0.1,0.3,0.5,0.7,0.9,1.0}, +-0.01
G["G3_sampling_std_dev"] = dict(
    tolerance=0.01, cos_thetas=[0.1, 0.3, 0.5, 0.7, 0.9, 1.0], sample_count=1024,
    ggx_r=dict(source="ORT/BSDFs/GGXTest.h:111-116", alpha=0.75, expected=0.36),
    ggx_t=dict(source="ORT/BSDFs/GGXTest.h:327-335", alpha=0.75, iors=[0.5, 0.9, 1.1, 1.5], expected=[2.05, 0.53, 0.05, 0.08]),
    ggx=dict(source="ORT/BSDFs/GGXTest.h:560-569", alpha=0.75, specularity=0.5, iors=[0.5, 0.9, 1.1, 1.5], expected=[0.70, 0.57, 0.46, 0.46]),
    oren_nayar=dict(source="ORT/BSDFs/OrenNayarTest.h:77-84", roughness=[0.0, 0.25, 0.5, 0.75, 1.0], expected=[0.0, 0.074, 0.095, 0.114, 0.135]))

# G4: table spot checks
G["G4_tables"] = dict(
    ggx_rho=dict(source="ORT/BSDFs/GGXTest.h:163-189", tolerance=1e-4, sample_count=4096),
    dielectric_rho=dict(source="ORT/BSDFs/GGXTest.h:619-659", total_tolerance=0.0029, reflected_tolerance=0.0024, sample_count=8192),
    alpha_from_pdf=dict(source="ORT/BSDFs/GGXTest.h:191-233", sample_count=16))

# G5: analytic identities
G["G5_identities"] = dict(
    white_furnace=dict(source="ORT/ShadingModels/DefaultShadingTest.h:234-252"),
    power_conservation=dict(source="ORT/ShadingModels/DefaultShadingTest.h:60-78", tolerance=1e-3),
    sampling_probabilities=dict(source="ORT/ShadingModels/DefaultShadingTest.h:254-290", tolerance=2e-5),
    metallic_lerp=dict(source="ORT/ShadingModels/DefaultShadingTest.h:292-324", tolerance=1e-6),
    oren_nayar_power=dict(source="ORT/BSDFs/OrenNayarTest.h:50-58", tolerance=0.00045),
    oren_nayar_albedo=dict(source="ORT/BSDFs/OrenNayarTest.h:86-96", tolerance=0.0006),
    E_FON=dict(source="ORT/BSDFs/OrenNayarTest.h:98-107", tolerance=1e-3))

# G6: thin sheet RMSE regression, ORT/ShadingModels/UtilsTest.h:167-226
G["G6_thin_sheet"] = dict(
    source="ORT/ShadingModels/UtilsTest.h:167-226", tolerance=0.025,
    transmission_tint=[1.0, 0.5, 0.25], roughness=[0.0, 0.5, 1.0], cos_thetas=[0.3, 0.5, 1.0],
    iors=[0.331492, 1.0 / 1.5, 0.789474, 1.26667, 1.5, 3.01667], path_count=16384, bounce_count=32,
    expected_reflection_rmse=[0.2135, 0.2001, 0.1973], expected_transmission_rmse=[0.2001, 0.1005, 0.0503])

# G7: scalar constants, ORT/MiscTest.h
G["G7_misc"] = dict(
    specularity=dict(source="ORT/MiscTest.h:166-190", water_ior=1.333, glass_ior=1.5, water=0.02037318784, glass=0.04),
    conductors=dict(source="ORT/MiscTest.h:28-31,192-214", relative_tolerance=1e-5,
                    gold_ior=[0.1986, 0.54463, 1.2515], gold_extinction=[3.228, 2.1406, 1.7517],
                    titanium_ior=[2.6979, 2.4793, 2.3050], titanium_extinction=[3.7571, 3.3511, 3.0820],
                    gold_specularity=[0.932999, 0.687356, 0.384839],
                    titanium_specularity=[0.61167696422, 0.57501477894, 0.54852055032]),
    balance_heuristic=dict(source="ORT/MiscTest.h:54-65"))

# G10: renderer-level tests, ORT/RendererTest.h:142-194
G["G10_renderer"] = dict(
    background=dict(source="ORT/RendererTest.h:142-153", width=16, height=12, tolerance=1e-4),
    tint=dict(source="ORT/RendererTest.h:155-194", width=4, height=3, tolerance=0.003))

Path(__file__).with_name("reference_goldens.json").write_text(json.dumps(G, indent=1))
print("wrote reference_goldens.json")
