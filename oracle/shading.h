// oracle/shading.h -- CPU restatement of the three shading models and the light sources.
// TEST INFRASTRUCTURE ONLY (see oracle/vecmath.h).
//
// Reference files followed (relative to /root/reference/extensions/OptiXRenderer/OptiXRenderer/):
//   Shading/ShadingModels/DiffuseShading.h:21-50
//   Shading/ShadingModels/DefaultShading.h:41-298
//   Shading/ShadingModels/TransmissiveShading.h:22-97
//   Shading/LightSources/{LightImpl,SphereLightImpl,SpotLightImpl,DirectionalLightImpl}.h
//   Intersect.h:23-67, TBN.h:27-58, Utils.h:347-356, MonteCarlo.h:20-35
#pragma once

#include "bsdf.h"
#include "../include/hiprenderer_c.h"

namespace oracle {

// ---------------------------------------------------------------------------------------------
// Material parameter block after texture lookups: what the reference's shading model
// constructors read from `Material` + texcoord + vertex tint scale.
// ---------------------------------------------------------------------------------------------
struct MaterialInputs {
    float3 tint;
    float roughness;
    float specularity;
    float metallic;
    float coat;
    float coat_roughness;
};

inline float unorm16(uint16_t raw) { return raw / 65535.0f; }

inline MaterialInputs inputs_from_material(const HiprMaterial& m) {
    return {{m.tint[0], m.tint[1], m.tint[2]}, m.roughness, m.specularity, m.metallic, unorm16(m.coat), unorm16(m.coat_roughness)};
}

// ---------------------------------------------------------------------------------------------
// Diffuse
// ---------------------------------------------------------------------------------------------
struct DiffuseShading {
    float3 tint;
    float roughness;
    BSDFResponse evaluate_with_PDF(float3 wo, float3 wi) const {
        if (wo.z < 0.000001f || wi.z < 0.000001f)
            return BSDFResponse::none();
        return OrenNayar::evaluate_with_PDF(tint, roughness, wo, wi);
    }
    BSDFSample sample(float3 wo, float3 u) const {
        if (wo.z < 0.000001f)
            return BSDFSample::none();
        return OrenNayar::sample(tint, roughness, wo, make_float2(u));
    }
    float3 rho(float) const { return tint; }
};

// ---------------------------------------------------------------------------------------------
// Default: diffuse base + GGX specular + optional GGX coat.
// ---------------------------------------------------------------------------------------------
struct DefaultShading {
    float3 diffuse_tint;
    float roughness;
    float3 specularity;
    float specular_scale;
    float coat_scale;
    float coat_alpha;
    uint16_t specular_probability;
    uint16_t coat_probability;

    static float compute_specular_properties(float roughness, float specularity, float scale, float abs_cos_theta_o,
                                             float& alpha, float& reflection_scale, float& transmission_scale) {
        alpha = GGX::alpha_from_roughness(roughness);
        SpecularRho rho = SpecularRho::fetch(abs_cos_theta_o, roughness);
        reflection_scale = scale * rho.energy_loss_adjustment();
        float specular_rho = rho.rho(specularity) * reflection_scale;
        transmission_scale = 1.0f - specular_rho;
        return specular_rho;
    }

    void setup_shading(float3 tint, float in_roughness, float dielectric_specularity, float metallic, float in_coat_scale,
                       float coat_roughness, float cos_theta_o, float& coat_rho) {
        float abs_cos_theta_o = fabsf(cos_theta_o);
        roughness = in_roughness;
        float3 conductor_spec = tint;

        if (in_coat_scale > 0) {
            float coat_modulated_roughness = modulate_roughness_under_coat(in_roughness, coat_roughness);
            roughness = lerp(in_roughness, coat_modulated_roughness, in_coat_scale);
            if (dielectric_specularity < 1.0f) {
                float coated = adjust_dielectric_specularity_to_exterior_medium(COAT_IOR, dielectric_specularity);
                dielectric_specularity = lerp(dielectric_specularity, coated, in_coat_scale);
            }
            if (metallic > 0) {
                float3 coated = adjust_conductor_specularity_to_exterior_medium(make_float3(COAT_IOR), conductor_spec, make_float3(0.0f));
                conductor_spec = lerp(conductor_spec, coated, in_coat_scale);
                conductor_spec.x = std::isnan(conductor_spec.x) ? 1.0f : conductor_spec.x;
                conductor_spec.y = std::isnan(conductor_spec.y) ? 1.0f : conductor_spec.y;
                conductor_spec.z = std::isnan(conductor_spec.z) ? 1.0f : conductor_spec.z;
            }
        }

        float specular_alpha, dielectric_specular_transmission;
        compute_specular_properties(roughness, dielectric_specularity, 1.0f, abs_cos_theta_o, specular_alpha, specular_scale, dielectric_specular_transmission);
        float3 dielectric_tint = tint * dielectric_specular_transmission;

        specularity = lerp(make_float3(dielectric_specularity), conductor_spec, metallic);
        diffuse_tint = dielectric_tint * (1.0f - metallic);

        if (in_coat_scale > 0) {
            float coat_transmission;
            coat_rho = compute_specular_properties(coat_roughness, COAT_SPECULARITY, in_coat_scale, abs_cos_theta_o, coat_alpha, coat_scale, coat_transmission);
            specular_scale *= coat_transmission;
            diffuse_tint *= coat_transmission;
        } else {
            coat_rho = 0;
            coat_scale = 0;
            coat_alpha = 0;
        }
    }

    void setup_sampling_probabilities(float abs_cos_theta_o, float coat_rho) {
        float diffuse_rho_sum = sum(diffuse_rho(abs_cos_theta_o));
        float specular_rho_sum = sum(specular_rho(abs_cos_theta_o));
        float coat_rho_sum = 3 * coat_rho;
        float recip_total_rho = 1.0f / (diffuse_rho_sum + specular_rho_sum + coat_rho_sum);
        specular_probability = (unsigned short)(specular_rho_sum * recip_total_rho * 65535.0f + 0.5f);
        coat_probability = (unsigned short)(coat_rho_sum * recip_total_rho * 65535.0f + 0.5f);
    }

    // DefaultShading(const Material&, float abs_cos_theta_o), the host constructor the goldens use.
    DefaultShading(const MaterialInputs& m, float abs_cos_theta_o, float min_roughness = 0.0f) {
        float coat_rho;
        float cr = fmaxf(m.coat_roughness, min_roughness);
        float r = fmaxf(m.roughness, min_roughness);
        setup_shading(m.tint, r, m.specularity, m.metallic, m.coat, cr, abs_cos_theta_o, coat_rho);
        setup_sampling_probabilities(abs_cos_theta_o, coat_rho);
    }

    static DefaultShading with_max_PDF_hint(const MaterialInputs& m, float abs_cos_theta_o, PDF max_PDF_hint) {
        return DefaultShading(m, abs_cos_theta_o, GGXMinimumRoughness::from_PDF(abs_cos_theta_o, max_PDF_hint));
    }

    float specular_alpha() const { return GGX::alpha_from_roughness(roughness); }
    float get_diffuse_probability() const { return 1.0f - (specular_probability + coat_probability) / 65535.0f; }
    float get_specular_probability() const { return specular_probability / 65535.0f; }
    float get_coat_probability() const { return coat_probability / 65535.0f; }

    BSDFResponse evaluate_with_PDF(float3 wo, float3 wi) const {
        if (wo.z < 0.000001f || wi.z < 0.000001f)
            return BSDFResponse::none();
        BSDFResponse diffuse = OrenNayar::evaluate_with_PDF(diffuse_tint, roughness, wo, wi);
        BSDFResponse specular = GGX_R::evaluate_with_PDF(specular_alpha(), specularity, wo, wi);
        specular.reflectance *= specular_scale;
        BSDFResponse response;
        response.reflectance = diffuse.reflectance + specular.reflectance;
        response.pdf = diffuse.pdf * get_diffuse_probability() + specular.pdf * get_specular_probability();
        if (coat_scale > 0) {
            BSDFResponse coat = GGX_R::evaluate_with_PDF(coat_alpha, make_float3(COAT_SPECULARITY), wo, wi);
            response.reflectance += coat_scale * coat.reflectance;
            response.pdf += coat.pdf * get_coat_probability();
        }
        return response;
    }

    BSDFSample sample(float3 wo, float3 u) const {
        if (wo.z < 0.000001f)
            return BSDFSample::none();
        float sp = get_specular_probability();
        float cp = get_coat_probability();
        float dp = 1 - cp - sp;
        bool sample_coat = u.z < cp;
        bool sample_specular = !sample_coat && u.z < (cp + sp);
        bool sample_diffuse = !sample_coat && !sample_specular;

        BSDFSample s;
        if (sample_diffuse) {
            s = OrenNayar::sample(diffuse_tint, roughness, wo, make_float2(u));
            s.pdf *= dp;
        } else if (sample_specular) {
            s = GGX_R::sample(specular_alpha(), specularity, wo, make_float2(u));
            s.reflectance *= specular_scale;
            s.pdf *= sp;
        } else {
            s = GGX_R::sample(coat_alpha, make_float3(COAT_SPECULARITY), wo, make_float2(u));
            s.reflectance *= coat_scale;
            s.pdf *= cp;
        }
        if (s.pdf.invalid_or_delta_dirac())
            return s;

        if (!sample_diffuse) {
            BSDFResponse r = OrenNayar::evaluate_with_PDF(diffuse_tint, roughness, wo, s.direction);
            if (r.pdf.is_valid_and_not_delta_dirac()) {
                s.reflectance += r.reflectance;
                s.pdf += r.pdf * dp;
            }
        }
        if (!sample_specular) {
            BSDFResponse r = GGX_R::evaluate_with_PDF(specular_alpha(), specularity, wo, s.direction);
            if (r.pdf.is_valid_and_not_delta_dirac()) {
                s.reflectance += r.reflectance * specular_scale;
                s.pdf += r.pdf * sp;
            }
        }
        if (!sample_coat && coat_scale > 0) {
            BSDFResponse r = GGX_R::evaluate_with_PDF(coat_alpha, make_float3(COAT_SPECULARITY), wo, s.direction);
            if (r.pdf.is_valid_and_not_delta_dirac()) {
                s.reflectance += coat_scale * r.reflectance;
                s.pdf += r.pdf * cp;
            }
        }
        return s;
    }

    float3 diffuse_rho(float) const { return diffuse_tint; }
    float3 specular_rho(float abs_cos_theta) const { return SpecularRho::fetch(abs_cos_theta, roughness).rho(specularity) * specular_scale; }
    float coat_rho(float abs_cos_theta) const {
        return SpecularRho::fetch(abs_cos_theta, GGX::roughness_from_alpha(coat_alpha)).rho(COAT_SPECULARITY) * coat_scale;
    }
    float3 rho(float abs_cos_theta) const {
        float3 r = diffuse_rho(abs_cos_theta) + specular_rho(abs_cos_theta);
        if (coat_scale > 0.0f)
            r = r + coat_rho(abs_cos_theta);
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// Transmissive: rough dielectric R + T.
// ---------------------------------------------------------------------------------------------
struct TransmissiveShading {
    float3 transmission_tint;
    float specularity;
    float ggx_alpha;
    float ior_i_over_o;
    float energy_loss_adjustment;

    TransmissiveShading(const MaterialInputs& m, float cos_theta_o, float min_roughness = 0.0f) {
        float roughness = fmaxf(m.roughness, min_roughness);
        transmission_tint = m.tint;
        specularity = m.specularity;
        ggx_alpha = GGX::alpha_from_roughness(roughness);
        float medium_ior = dielectric_ior_from_specularity(specularity);
        bool entering = cos_theta_o >= 0.0f;
        float ior_o = entering ? AIR_IOR : medium_ior;
        float ior_i = entering ? medium_ior : AIR_IOR;
        ior_i_over_o = ior_i / ior_o;
        float rho = DielectricRho::fetch(fabsf(cos_theta_o), roughness, ior_i_over_o).total_rho;
        energy_loss_adjustment = 1.0f / rho;
    }
    static TransmissiveShading with_max_PDF_hint(const MaterialInputs& m, float cos_theta_o, PDF max_PDF_hint) {
        return TransmissiveShading(m, cos_theta_o, GGXMinimumRoughness::from_PDF(fabsf(cos_theta_o), max_PDF_hint));
    }
    BSDFResponse evaluate_with_PDF(float3 wo, float3 wi) const {
        if (wo.z < 0.000001f)
            return BSDFResponse::none();
        BSDFResponse r = GGX_RT::evaluate_with_PDF(transmission_tint, ggx_alpha, specularity, ior_i_over_o, wo, wi);
        r.reflectance *= energy_loss_adjustment;
        return r;
    }
    BSDFSample sample(float3 wo, float3 u) const {
        if (wo.z < 0.000001f)
            return BSDFSample::none();
        BSDFSample s = GGX_RT::sample(transmission_tint, ggx_alpha, specularity, ior_i_over_o, wo, u);
        s.reflectance *= energy_loss_adjustment;
        return s;
    }
    float3 rho(float abs_cos_theta_o) const {
        DielectricRho r = DielectricRho::fetch(abs_cos_theta_o, GGX::roughness_from_alpha(ggx_alpha), ior_i_over_o);
        float reflection = r.reflected_rho / r.total_rho;
        return reflection + (1 - reflection) * transmission_tint;
    }
};

// Thin sheet approximation (ORS/ShadingModels/Utils.h:139-166), pinned by golden G6.
struct ThinSheetThroughput { float3 reflected, transmitted; };
inline ThinSheetThroughput approx_thin_sheet_reflectance(float abs_cos_theta, float roughness, float ior_i_over_o, float3 transmission_tint) {
    float refracted_cos_theta;
    bool tir = !refract_cos(refracted_cos_theta, -abs_cos_theta, ior_i_over_o);
    if (tir)
        return {make_float3(1), make_float3(0)};
    DielectricRho rho0 = DielectricRho::fetch(abs_cos_theta, roughness, ior_i_over_o);
    float R0 = rho0.reflected_rho / rho0.total_rho;
    float T0 = 1 - R0;
    DielectricRho rhoi = DielectricRho::fetch(fabsf(refracted_cos_theta), roughness, ior_i_over_o);
    float Ri = rhoi.reflected_rho / rhoi.total_rho;
    float Ti = 1 - Ri;
    float3 T0Ti = T0 * Ti * transmission_tint;
    float3 transmitted = T0Ti / (1 - Ri * Ri);
    float3 reflected = R0 + Ri * transmitted;
    return {reflected, transmitted};
}

// ---------------------------------------------------------------------------------------------
// TBN, MIS, intersections
// ---------------------------------------------------------------------------------------------
inline void compute_tangents(float3 n, float3& tangent, float3& bitangent) {
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    tangent = {1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
    bitangent = {b, sign + n.y * n.y * a, -n.y};
}

struct TBN {
    float3 tangent, bitangent, normal;
    explicit TBN(float3 n) : normal(n) { compute_tangents(n, tangent, bitangent); }
    float3 to_local(float3 v) const { return {dot(tangent, v), dot(bitangent, v), dot(normal, v)}; }      // TBN * v
    float3 to_world(float3 v) const { return v.x * tangent + v.y * bitangent + v.z * normal; }           // v * TBN
};

inline float balance_heuristic(float pdf1, float pdf2) {
    float divisor = pdf1 + pdf2;
    float result = pdf1 / divisor;
    bool invalid = std::isinf(divisor) || std::isnan(result);
    return invalid ? (pdf1 <= pdf2 ? 0.0f : 1.0f) : result;
}
inline float power_heuristic(float pdf1, float pdf2) { return balance_heuristic(pdf1 * pdf1, pdf2 * pdf2); }
inline float MIS_weight(PDF a, PDF b) { return balance_heuristic(a.value(), b.value()); }

inline float ray_sphere(float3 o, float3 d, float3 center, float radius) {
    float3 to_sphere = o - center;
    float b = dot(to_sphere, d);
    float3 fbd = to_sphere - b * d;
    float disc = radius * radius - dot(fbd, fbd);
    if (disc > 0.0)
        return -b - sqrtf(disc);
    return nanf("");
}
inline float ray_plane(float3 o, float3 d, float3 plane_point, float3 plane_normal) {
    float dd = dot(plane_normal, plane_point);
    float n_dot_o = dot(plane_normal, o);
    float n_dot_d = dot(plane_normal, d);
    return (dd - n_dot_o) / n_dot_d;
}
inline float ray_disk(float3 o, float3 d, float3 center, float3 normal, float radius) {
    float t = ray_plane(o, d, center, normal);
    float3 v = (o + d * t) - center;
    if (dot(v, v) <= radius * radius && t >= 0.0f)
        return t;
    return nanf("");
}

// ---------------------------------------------------------------------------------------------
// Lights
// ---------------------------------------------------------------------------------------------
struct LightSample {
    float3 radiance;
    PDF pdf;
    float3 direction_to_light;
    float distance;
    static LightSample none() { return {{0, 0, 0}, PDF::delta_dirac(0), {0, 1, 0}, 0.0f}; }
};

struct SphereLight { float3 power, position; float radius; };
struct SpotLight { float3 power, position; float radius; float3 direction; float cos_angle; };
struct DirectionalLight { float3 radiance, direction; };

inline SphereLight as_sphere(const HiprLight& l) { return {{l.data[0], l.data[1], l.data[2]}, {l.data[3], l.data[4], l.data[5]}, l.data[6]}; }
inline SpotLight as_spot(const HiprLight& l) {
    return {{l.data[0], l.data[1], l.data[2]}, {l.data[3], l.data[4], l.data[5]}, l.data[6], {l.data[7], l.data[8], l.data[9]}, l.data[10]};
}
inline DirectionalLight as_directional(const HiprLight& l) { return {{l.data[0], l.data[1], l.data[2]}, {l.data[3], l.data[4], l.data[5]}}; }

namespace Lights {
static const float sphere_light_small_sin_theta_squared = 0.0f;
static const float spot_light_min_cone_angle_to_sample = 1e-5f;

inline float surface_area(const SphereLight& l) { return 4.0f * PIf * l.radius * l.radius; }
inline bool is_delta_light(const SphereLight& l, float3 position) {
    float3 v = l.position - position;
    return l.radius * l.radius / dot(v, v) <= sphere_light_small_sin_theta_squared;
}
inline LightSample sample_radiance(const SphereLight& l, float3 position, float2 u) {
    float3 to_light = l.position - position;
    float sin_theta_squared = l.radius * l.radius / dot(to_light, to_light);
    LightSample s;
    if (sin_theta_squared <= sphere_light_small_sin_theta_squared) {
        s.direction_to_light = to_light;
        s.distance = length(s.direction_to_light);
        s.direction_to_light /= s.distance;
        s.radiance = l.power / (4.0f * PIf * s.distance * s.distance);
        s.distance -= l.radius;
        s.pdf = PDF::delta_dirac(1);
    } else {
        float cos_theta = sqrtf(1.0f - sin_theta_squared);
        DirectionalSample cone = Dist::Cone::sample(cos_theta, u);
        const TBN tbn(normalize(to_light));
        s.direction_to_light = tbn.to_world(cone.direction);
        s.pdf = cone.pdf;
        s.distance = ray_sphere(position, s.direction_to_light, l.position, l.radius);
        if (s.distance <= 0.0f)   // false for NaN, as in the reference
            s.distance = dot(to_light, s.direction_to_light);
        float inv_divisor = 1.0f / (PIf * surface_area(l));
        s.radiance = l.power * inv_divisor;
    }
    s.distance = nextafterf(s.distance, 0.0f);
    return s;
}
inline PDF pdf(const SphereLight& l, float3 lit_position, float3 direction_to_light) {
    float3 to_center = l.position - lit_position;
    float sin_theta_squared = l.radius * l.radius / dot(to_center, to_center);
    if (sin_theta_squared < sphere_light_small_sin_theta_squared)
        return PDF::delta_dirac(0);
    float cos_theta_max = sqrtf(1.0f - sin_theta_squared);
    float cos_theta = dot(direction_to_light, normalize(to_center));
    float valid = cos_theta >= cos_theta_max ? 1.0f : 0.0f;
    return Dist::Cone::PDF(cos_theta_max) * valid;
}
inline float3 evaluate(const SphereLight& l, float3 position) {
    float inv_divisor = 1.0f / (is_delta_light(l, position) ? (4.0f * PIf) : (PIf * surface_area(l)));
    return l.power * inv_divisor;
}

inline float surface_area(const SpotLight& l) { return PIf * pow2(l.radius); }
inline bool is_delta_light(const SpotLight& l) { return l.radius == 0.0f; }
inline PDF pdf(const SpotLight& l, float3 lit_position, float3 direction_to_light) {
    float cos_theta = -dot(l.direction, direction_to_light);
    if (cos_theta > 0.0f && !is_delta_light(l)) {
        float t = ray_plane(lit_position, -l.direction, l.position, l.direction);
        float cone_radius = t * sqrtf(1.0f - pow2(l.cos_angle)) / l.cos_angle;
        if (l.radius > cone_radius && l.cos_angle > spot_light_min_cone_angle_to_sample)
            return Dist::Cone::PDF(l.cos_angle);
        float td = ray_disk(lit_position, direction_to_light, l.position, l.direction, l.radius);
        if (td >= 0.0f)
            return Dist::Disk::PDF(l.radius) * ((td * td) / cos_theta);
    }
    return PDF::delta_dirac(0);
}
inline float3 evaluate(const SpotLight& l, float3 lit_position, float3 direction_to_light) {
    float cos_theta = -dot(l.direction, direction_to_light);
    float normalization = TWO_PIf * (1 - l.cos_angle);
    if (is_delta_light(l)) {
        float3 d = l.position - lit_position;
        normalization *= dot(d, d);
    } else
        normalization *= surface_area(l) * cos_theta;
    float3 radiance = l.power / normalization;
    return (cos_theta > l.cos_angle) ? radiance : make_float3(0.0f);
}
inline LightSample sample_radiance(const SpotLight& l, float3 lit_position, float2 u) {
    LightSample s;
    if (is_delta_light(l)) {
        s.direction_to_light = l.position - lit_position;
        s.distance = length(s.direction_to_light);
        s.direction_to_light /= s.distance;
        s.pdf = 1.0f;
        s.radiance = evaluate(l, lit_position, s.direction_to_light);
        return s;
    }
    const TBN light_to_world(l.direction);
    float t = ray_plane(lit_position, -l.direction, l.position, l.direction);
    float cone_radius = t * sqrtf(1.0f - pow2(l.cos_angle)) / l.cos_angle;
    if (l.radius > cone_radius && l.cos_angle > spot_light_min_cone_angle_to_sample) {
        DirectionalSample cone = Dist::Cone::sample(l.cos_angle, u);
        s.direction_to_light = light_to_world.to_world(-cone.direction);
        s.distance = ray_plane(lit_position, s.direction_to_light, l.position, l.direction);
        s.pdf = cone.pdf;
        s.radiance = {0, 0, 0};
        float3 on_light = lit_position + s.direction_to_light * s.distance;
        float3 d = on_light - l.position;
        if (dot(d, d) < pow2(l.radius))
            s.radiance = evaluate(l, lit_position, s.direction_to_light);
    } else {
        float2 disk = Dist::Disk::sample(l.radius, u);
        float3 sampled_position = l.position + light_to_world.to_world(make_float3(disk, 0.0f));
        s.direction_to_light = sampled_position - lit_position;
        s.distance = length(s.direction_to_light);
        s.direction_to_light /= s.distance;
        float cos_theta = -dot(l.direction, s.direction_to_light);
        s.pdf = Dist::Disk::PDF(l.radius) * (pow2(s.distance) / cos_theta);
        s.radiance = evaluate(l, lit_position, s.direction_to_light);
    }
    s.distance = nextafterf(s.distance, 0.0f);
    return s;
}

inline LightSample sample_radiance(const DirectionalLight& l) {
    return {l.radiance, PDF::delta_dirac(1.0f), -l.direction, 1e30f};
}

inline LightSample sample_radiance(const HiprLight& light, float3 position, float2 u) {
    switch (light.flags & HIPR_LIGHT_TYPE_MASK) {
    case HIPR_LIGHT_SPHERE: return sample_radiance(as_sphere(light), position, u);
    case HIPR_LIGHT_DIRECTIONAL: return sample_radiance(as_directional(light));
    case HIPR_LIGHT_SPOT: return sample_radiance(as_spot(light), position, u);
    }
    return LightSample::none();
}

// evaluate_intersection<> of LightImpl.h:85-108 for the light a MonteCarlo ray hit.
inline float3 evaluate_intersection(const HiprLight& light, float3 ray_origin, float3 ray_direction, PDF bsdf_PDF) {
    float3 radiance;
    PDF light_PDF;
    switch (light.flags & HIPR_LIGHT_TYPE_MASK) {
    case HIPR_LIGHT_SPHERE:
        radiance = evaluate(as_sphere(light), ray_origin);
        light_PDF = pdf(as_sphere(light), ray_origin, ray_direction);
        break;
    case HIPR_LIGHT_SPOT:
        radiance = evaluate(as_spot(light), ray_origin, ray_direction);
        light_PDF = pdf(as_spot(light), ray_origin, ray_direction);
        break;
    default:
        return make_float3(1000.0f, 0, 1000);
    }
    if (bsdf_PDF.use_for_MIS())
        radiance *= MIS_weight(bsdf_PDF, light_PDF);
    return radiance;
}
} // namespace Lights

} // namespace oracle
