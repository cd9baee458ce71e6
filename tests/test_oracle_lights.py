"""Pins the oracle's light sources with the reference's own light tests:
tests/OptiXRendererTests/LightSources/SphereLightTest.h:22-136 and SpotLightTest.h:24-171 -- power preservation under changing
radius / cone angle, PDF consistency between sample_radiance and pdf, evaluate vs sampled radiance, rejection of rays that miss.
And G9: the flag and shading-model enums the ABI shares with the Bifrost side (Assets/FlagsTest.h:20-33)."""
import math

import numpy as np
import pytest

from bifrost3d_amd import capi

LIGHT_SPHERE, LIGHT_SPOT = 1, 5     # HIPR_LIGHT_SPHERE, HIPR_LIGHT_SPOT (include/hiprenderer_c.h)


def sphere_light(position, radius, power):
    light = capi.HiprLight()
    light.data[0:3] = [power] * 3
    light.data[3:6] = list(position)
    light.data[6] = radius
    light.flags = LIGHT_SPHERE
    return light


def spot_light(position, direction, radius, power, cos_angle):
    light = capi.HiprLight()
    light.data[0:3] = [power] * 3
    light.data[3:6] = list(position)
    light.data[6] = radius
    light.data[7:10] = list(direction)
    light.data[10] = cos_angle
    light.flags = LIGHT_SPOT
    return light


def samples02(oracle, count):
    return np.array([oracle.sample02(i) for i in range(count)], dtype=np.float32)


def pdf_is_valid(pdf):      # OR/Types.h PDF::is_valid: a usable, positive density
    return pdf > 0.0 and math.isfinite(pdf)


# ---- SphereLightTest.h ----------------------------------------------------------------------------------------------------------

def test_G8_sphere_light_power_preservation_when_radius_changes(oracle):
    u = samples02(oracle, 1024)
    shading_position, shading_normal = np.zeros(3, np.float32), np.array([0.0, 1.0, 0.0])
    tolerances = {0.0: 0.0, 1.0: 0.0001, 2.0: 0.001, 5.0: 0.001, 9.0: 0.004}
    for radius, tolerance in tolerances.items():
        light = sphere_light((0.0, 10.0, 0.0), radius, 10.0)
        s = oracle.light_sample(light, shading_position, u)         # radiance[3], PDF, direction[3], distance
        luminances = s[:, 0].astype(np.float64) * (s[:, 4:7].astype(np.float64) @ shading_normal) / np.abs(s[:, 3].astype(np.float64))
        power = math.fsum(luminances) / len(u) * (4.0 * math.pi * 10.0 * 10.0)
        if radius == 0.0:
            assert power == pytest.approx(10.0, rel=1e-6)           # the delta light: exact up to f32 rounding
        else:
            assert abs(power - 10.0) < tolerance, (radius, power)


def test_G8_sphere_light_consistent_PDF(oracle):
    u = samples02(oracle, 128)
    position = np.array([10.0, 0.0, 0.0], np.float32)
    for radius in (1.0, 2.0, 4.0, 7.0, 13.0):
        light = sphere_light((0.0, 10.0, 0.0), radius, 10.0)
        for s in oracle.light_sample(light, position, u):
            if pdf_is_valid(float(s[3])):
                assert oracle.light_pdf(light, position, s[4:7]) == pytest.approx(float(s[3]), abs=1e-4)


def test_G8_sphere_light_pdf_rejects_rays_that_miss(oracle):
    light = sphere_light((0.0, 10.0, 0.0), 2.0, 10.0)
    lit_position = np.zeros(3, np.float32)
    hit = np.array([1.0, 10.0, 0.0]); hit /= np.linalg.norm(hit)
    miss = np.array([3.0, 10.0, 0.0]); miss /= np.linalg.norm(miss)
    assert pdf_is_valid(oracle.light_pdf(light, lit_position, hit))
    assert not pdf_is_valid(oracle.light_pdf(light, lit_position, miss))


# ---- SpotLightTest.h ------------------------------------------------------------------------------------------------------------

def cosine_sample(u):       # OR/Distributions.h:166-179
    r, z, phi = math.sqrt(1.0 - u[0]), math.sqrt(u[0]), 2.0 * math.pi * u[1]
    return np.array([r * math.cos(phi), r * math.sin(phi), z])


def test_G8_spot_light_consistent_PDF_and_radiance(oracle):
    u = samples02(oracle, 16)
    light_position, light_direction = np.array([0.0, 10.0, 0.0]), np.array([0.0, -1.0, 0.0])
    center = light_position + 2.0 * light_direction
    for p in range(16):
        position = (center + 2.0 * cosine_sample(u[p])).astype(np.float32)
        for radius in (1.0, 4.0, 13.0):
            for cos_angle in (0.1, 0.5, 0.9):
                light = spot_light(light_position, light_direction, radius, 10.0, cos_angle)
                for s in oracle.light_sample(light, position, u):
                    sampled_pdf, pdf = float(s[3]), oracle.light_pdf(light, position, s[4:7])
                    if pdf_is_valid(sampled_pdf) or pdf_is_valid(pdf):
                        assert pdf == pytest.approx(sampled_pdf, rel=1e-4), (p, radius, cos_angle)      # EXPECT_PDF_EQ_PCT
                    if s[0] > 0.0:
                        assert float(oracle.light_evaluate(light, position, s[4:7])[0]) == pytest.approx(float(s[0]), abs=1e-4)


def test_G8_spot_light_pdf_rejects_rays_that_miss(oracle):
    lit_position = np.zeros(3, np.float32)
    hit = np.array([1.0, 10.0, 0.0]); hit /= np.linalg.norm(hit)
    miss = np.array([3.0, 10.0, 0.0]); miss /= np.linalg.norm(miss)
    light = spot_light((0.0, 10.0, 0.0), (0.0, -1.0, 0.0), 2.0, 10.0, 0.5)
    assert pdf_is_valid(oracle.light_pdf(light, lit_position, hit))
    assert not pdf_is_valid(oracle.light_pdf(light, lit_position, miss))
    away = spot_light((0.0, 10.0, 0.0), (0.0, 1.0, 0.0), 2.0, 10.0, 0.5)       # the light faces away
    assert not pdf_is_valid(oracle.light_pdf(away, lit_position, hit))
    assert not pdf_is_valid(oracle.light_pdf(away, lit_position, miss))


def estimate_spot_power(oracle, radius, cos_angle, light_uvs, disk_uvs, disk_depth=1.0):
    """SpotLightTest.h:86-126: the flux through a disk facing the light that covers its whole cone, by sampling the disk uniformly."""
    light = spot_light((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), radius, 1.0, cos_angle)
    disk_normal = np.array([0.0, 0.0, -1.0])
    depth_to_radius = math.sqrt((1.0 - cos_angle ** 2) / cos_angle ** 2)
    disk_radius = depth_to_radius * disk_depth + radius
    radiances = []
    for uv in disk_uvs:
        r, phi = math.sqrt(uv[0]) * disk_radius, 2.0 * math.pi * uv[1]       # Distributions::Disk::sample
        disk_position = np.array([r * math.cos(phi), r * math.sin(phi), disk_depth], np.float32)
        s = oracle.light_sample(light, disk_position, light_uvs)
        valid = (s[:, 3] > 0.0) & np.isfinite(s[:, 3])
        cos_theta = s[:, 4:7].astype(np.float64) @ disk_normal
        contribution = np.where(valid, s[:, 0].astype(np.float64) * cos_theta / np.where(valid, s[:, 3], 1.0), 0.0)
        radiances.append(math.fsum(contribution) / len(light_uvs))
    return float(np.mean(radiances)) * math.pi * disk_radius ** 2


@pytest.fixture(scope="module")
def power_samples(oracle):
    return oracle.pmjbn(256), samples02(oracle, 1024)      # the reference uses its blue-noise PMJ points for the light, sample02 for the disk


@pytest.mark.parametrize("radius", [0.0, 1.0, 2.0, 4.0])
def test_G8_spot_light_power_preservation_when_radius_changes(oracle, power_samples, radius):
    light_uvs, disk_uvs = power_samples
    assert abs(estimate_spot_power(oracle, radius, 0.5, light_uvs, disk_uvs) - 1.0) < 0.0025


@pytest.mark.parametrize("cos_angle", [0.3, 0.5, 0.7])
def test_G8_spot_light_power_preservation_when_angle_changes(oracle, power_samples, cos_angle):
    light_uvs, disk_uvs = power_samples
    assert abs(estimate_spot_power(oracle, 0.25, cos_angle, light_uvs, disk_uvs) - 1.0) < 0.0045


# ---- Assets/FlagsTest.h ---------------------------------------------------------------------------------------------------------

def test_G9_flags_and_shading_models_share_the_bifrost_layout():
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    header = (root / "include" / "hiprenderer_c.h").read_text()
    host = (root / "bifrost3d_amd" / "host" / "Bifrost.h").read_text()

    def abi(name):
        return int(re.search(rf"\b{name}\s*=\s*(\d+)", header).group(1))

    flags = dict(re.findall(r"(\w+) = (\d+)", re.search(r"enum class MaterialFlag[^{]*\{([^}]*)\}", host).group(1)))
    models = dict(re.findall(r"(\w+) = (\d+)", re.search(r"enum class ShadingModel[^{]*\{([^}]*)\}", host).group(1)))
    assert int(flags["ThinWalled"]) == abi("HIPR_MATERIAL_THIN_WALLED") and int(flags["Cutout"]) == abi("HIPR_MATERIAL_CUTOUT")
    assert int(models["Default"]) == abi("HIPR_SHADING_DEFAULT") == capi.SHADING_DEFAULT
    assert int(models["Diffuse"]) == abi("HIPR_SHADING_DIFFUSE") == capi.SHADING_DIFFUSE
    assert int(models["Transmissive"]) == abi("HIPR_SHADING_TRANSMISSIVE") == capi.SHADING_TRANSMISSIVE
    assert int(models["Count"]) == 3
