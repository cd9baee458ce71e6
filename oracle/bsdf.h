// oracle/bsdf.h -- CPU restatement of the reference's PDF wrapper, distributions, BSDFs and
// precomputed-table lookups. TEST INFRASTRUCTURE ONLY (see oracle/vecmath.h).
//
// Reference files followed (relative to /root/reference/extensions/OptiXRenderer/OptiXRenderer/):
//   Types.h:152-204 (PDF), :319-339 (BSDFResponse / BSDFSample)
//   Utils.h:29-31,51-57,80-130,146-204,210-272,363-367
//   Distributions.h:34-98 (Disk, Cone), :135-155 (UniformHemisphere), :186-260 (OrenNayerCLTC),
//                   :304-382 (GGX_VNDF), :392-461 (GGX_Bounded_VNDF), :104-130 (UniformSphere)
//   Shading/BSDFs/OrenNayar.h:41-127, Shading/BSDFs/GGX.h:27-441
//   Shading/ShadingModels/Utils.h:27-130 and core/Bifrost/Bifrost/Math/ImageSampling.h:18-61
// Pinned by the goldens replayed in tests/test_oracle_goldens.py (G1-G8 of SURVEY.md 8c).
#pragma once

#include "vecmath.h"

namespace oracle {

static const float PIf = 3.14159265358979323846f;
static const float TWO_PIf = 6.283185307f;
static const float RECIP_PIf = 0.31830988618379067153776752674503f;
static const float COAT_SPECULARITY = 0.04f;
static const float COAT_IOR = 1.5f;
static const float AIR_IOR = 1.0f;
static const float MIN_VALID_PDF = 0.000001f;

// ---------------------------------------------------------------------------------------------
// PDF wrapper: negative = delta dirac, NaN = invalid.
// ---------------------------------------------------------------------------------------------
struct PDF {
    float v;
    PDF() = default;
    PDF(float pdf) : v(pdf) {}
    static PDF invalid() { return PDF(nanf("")); }
    static PDF delta_dirac(float pdf = 1.0f) { return PDF(-pdf); }
    float value() const { return fabsf(v); }
    bool is_valid() const { return value() > MIN_VALID_PDF; }
    bool is_delta_dirac() const { return !(v >= 0.0f); }
    void disable_MIS() { if (v >= 0.0f) v = -v; }
    bool is_valid_and_not_delta_dirac() const { return v > MIN_VALID_PDF; }
    bool invalid_or_delta_dirac() const { return !(v > MIN_VALID_PDF); }
    bool use_for_MIS() const { return is_valid_and_not_delta_dirac(); }
    PDF& operator*=(float s) { v *= s; return *this; }
    PDF operator*(float s) const { return PDF(v * s); }
    PDF& operator+=(PDF r) { v += r.v; return *this; }
    PDF operator+(PDF r) const { return PDF(v + r.v); }
};

struct BSDFResponse { float3 reflectance; PDF pdf; static BSDFResponse none() { return {{0, 0, 0}, PDF(0.0f)}; } };
struct BSDFSample { float3 reflectance; PDF pdf; float3 direction; static BSDFSample none() { return {{0, 0, 0}, PDF(0.0f), {0, 0, 0}}; } };
struct DirectionalSample { float3 direction; float pdf; };

// ---------------------------------------------------------------------------------------------
// Scalar helpers (OR/Utils.h)
// ---------------------------------------------------------------------------------------------
inline float sum(float3 v) { return v.x + v.y + v.z; }
inline float signf(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
inline bool is_black(float3 c) { return c.x <= 0.0f && c.y <= 0.0f && c.z <= 0.0f; }
inline float saturate(float v) { return clampf(v, 0.0f, 1.0f); }
inline float pow2(float x) { return x * x; }
inline float3 pow2(float3 x) { return x * x; }
inline float pow4(float x) { float xx = x * x; return xx * xx; }
inline float pow5(float x) { float xx = x * x; return xx * xx * x; }
inline bool same_hemisphere(float3 wo, float3 wi) { return wo.z * wi.z >= 0.0f; }
inline void sincos(float theta, float& s, float& c) { s = exact_sinf(theta); c = exact_cosf(theta); }

inline float dielectric_specularity(float ior_o, float ior_i) { return pow2((ior_o - ior_i) / (ior_o + ior_i)); }
inline float3 conductor_specularity(float3 ior_o, float3 ior_i, float3 ext_i) {
    float3 e2 = pow2(ext_i);
    return (pow2(ior_o - ior_i) + e2) / (pow2(ior_o + ior_i) + e2);
}
inline float dielectric_ior_from_specularity(float specularity) { return 2.0f / (1.0f - sqrtf(specularity)) - 1.0f; }
inline float3 conductor_ior_from_specularity(float3 specularity, float3 ext_i) {
    float3 a = specularity - 1.0f;
    float3 b = 2.0f * specularity + 2.0f;
    float3 c = a + (specularity - 1.0f) * pow2(ext_i);
    float3 d = b * b - 4.0f * a * c;
    float3 sqrt_d = {sqrtf(d.x), sqrtf(d.y), sqrtf(d.z)};
    return (-b + sqrt_d) / (2.0f * a);
}
inline float adjust_dielectric_specularity_to_exterior_medium(float exterior_ior, float specularity_through_air) {
    return dielectric_specularity(exterior_ior, dielectric_ior_from_specularity(specularity_through_air));
}
inline float3 adjust_conductor_specularity_to_exterior_medium(float3 exterior_ior, float3 specularity_through_air, float3 ext) {
    return conductor_specularity(exterior_ior, conductor_ior_from_specularity(specularity_through_air, ext), ext);
}
inline float schlick_fresnel(float f0, float abs_cos_theta) { return f0 + (1.0f - f0) * pow5(1.0f - abs_cos_theta); }
inline float3 schlick_fresnel(float3 f0, float abs_cos_theta) {
    float t = pow5(1.0f - abs_cos_theta);
    return (1.0f - t) * f0 + t;
}
inline float dielectric_schlick_fresnel(float f0, float abs_cos_theta, float ior_i_over_o) {
    float sin2_theta = 1 - pow2(abs_cos_theta);
    if (sin2_theta >= pow2(ior_i_over_o))
        return 1.0f;
    float t = pow5(1.0f - abs_cos_theta);
    return (1.0f - t) * f0 + t;
}
inline float modulate_roughness_under_coat(float base_roughness, float coat_roughness) {
    float x_coat = 1 - AIR_IOR / COAT_IOR;
    float r4 = fminf(1, pow4(base_roughness) + 2.0f * x_coat * pow4(coat_roughness));
    return exact_powf(r4, 0.25f);
}
// refract against the normal (0,0,1), OR/Utils.h:242-272.
inline bool refract_z(float3& out, float3 wi, float ior_i_over_o) {
    float normal_z = 1;
    float cos_theta_i = wi.z;
    if (cos_theta_i > 0.0f) { normal_z = -1; cos_theta_i = -cos_theta_i; }
    else ior_i_over_o = 1.f / ior_i_over_o;
    float k = 1.0f - ior_i_over_o * ior_i_over_o * (1.0f - cos_theta_i * cos_theta_i);
    out = ior_i_over_o * wi - make_float3(0, 0, (ior_i_over_o * cos_theta_i + sqrtf(k)) * normal_z);
    return k >= 0.0f;
}
inline bool refract_cos(float& refraction_cos_theta, float cos_theta_i, float ior_i_over_o) {
    float normal_z = 1;
    float adjusted = cos_theta_i;
    if (cos_theta_i > 0.0f) { normal_z = -1; adjusted = -adjusted; }
    else ior_i_over_o = 1.f / ior_i_over_o;
    float k = 1.0f - pow2(ior_i_over_o) * (1.0f - pow2(adjusted));
    refraction_cos_theta = ior_i_over_o * cos_theta_i - (ior_i_over_o * adjusted + sqrtf(k)) * normal_z;
    return k >= 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Precomputed tables. `quantize` models the device upload to unorm16 (OR/Renderer.cpp:400-466);
// without it the lookups are the reference's CPU branch (BF/Assets/Shading/*.cpp sample_*).
// ---------------------------------------------------------------------------------------------
struct Tables {
    float ggx_base[32 * 32];   // GGX_with_fresnel, F0 = 0
    float ggx_full[32 * 32];   // GGX, F0 = 1
    float2 dielectric_light[16 * 16 * 16];
    float2 dielectric_dense[16 * 16 * 16];
    float alpha[32 * 32];
    bool loaded = false;

    static float quantize(float v) { return float((unsigned short)(v * 65535 + 0.5f)) / 65535.0f; }
    void set(const float* base, const float* full, const float* light, const float* dense, const float* alphas, bool quantize_unorm16) {
        auto q = [&](float v) { return quantize_unorm16 ? quantize(v) : v; };
        for (int i = 0; i < 1024; ++i) { ggx_base[i] = q(base[i]); ggx_full[i] = q(full[i]); alpha[i] = q(alphas[i]); }
        for (int i = 0; i < 4096; ++i) {
            dielectric_light[i] = {q(light[2 * i]), q(light[2 * i + 1])};
            dielectric_dense[i] = {q(dense[2 * i]), q(dense[2 * i + 1])};
        }
        loaded = true;
    }
};

inline Tables& tables() { static Tables t; return t; }

template <typename T> inline T lerp_t(T a, T b, float t);
template <> inline float lerp_t<float>(float a, float b, float t) { return a + (b - a) * t; }
template <> inline float2 lerp_t<float2>(float2 a, float2 b, float t) { return a + (b - a) * t; }

template <typename T>
inline T bilinear(const T* pixels, int width, int height, float u, float v) {
    u = clampf(u, 0.0f, 1.0f);
    float u_coord = u * (width - 1);
    int lower_u = int(u_coord);
    int upper_u = lower_u + 1 < width - 1 ? lower_u + 1 : width - 1;
    v = clampf(v, 0.0f, 1.0f);
    float v_coord = v * (height - 1);
    int lower_v = int(v_coord);
    int upper_v = lower_v + 1 < height - 1 ? lower_v + 1 : height - 1;
    float u_t = u_coord - lower_u;
    T lower = lerp_t(pixels[lower_v * width + lower_u], pixels[lower_v * width + upper_u], u_t);
    T upper = lerp_t(pixels[upper_v * width + lower_u], pixels[upper_v * width + upper_u], u_t);
    return lerp_t(lower, upper, v_coord - lower_v);
}

template <typename T>
inline T trilinear(const T* pixels, int width, int height, int depth, float u, float v, float w) {
    w = clampf(w, 0.0f, 1.0f);
    float w_coord = w * (depth - 1);
    int lower_w = int(w_coord);
    int upper_w = lower_w + 1 < depth - 1 ? lower_w + 1 : depth - 1;
    T lower = bilinear(pixels + lower_w * width * height, width, height, u, v);
    T upper = bilinear(pixels + upper_w * width * height, width, height, u, v);
    return lerp_t(lower, upper, w_coord - lower_w);
}

struct SpecularRho {
    float base, full;
    float rho(float specularity) const { return lerp(base, full, specularity); }
    float3 rho(float3 s) const { return {rho(s.x), rho(s.y), rho(s.z)}; }
    float energy_loss_adjustment() const { return 1.0f / full; }
    static SpecularRho fetch(float abs_cos_theta, float roughness) {
        const Tables& t = tables();
        return {bilinear(t.ggx_base, 32, 32, abs_cos_theta, roughness), bilinear(t.ggx_full, 32, 32, abs_cos_theta, roughness)};
    }
};

struct DielectricRho {
    float total_rho, reflected_rho;
    static DielectricRho fetch(float abs_cos_theta, float roughness, float ior_i_over_o) {
        const Tables& t = tables();
        float2 r;
        if (ior_i_over_o < 1.0f)
            r = trilinear(t.dielectric_light, 16, 16, 16, abs_cos_theta, roughness, (ior_i_over_o - 0.331492f) / 0.457982f);
        else
            r = trilinear(t.dielectric_dense, 16, 16, 16, abs_cos_theta, roughness, (ior_i_over_o - 1.26667f) / 1.75f);
        return {r.x, r.y};
    }
};

namespace GGXMinimumRoughness {
inline float encode_PDF(float pdf) {
    float non_linear_PDF = pdf / (1.0f + pdf);
    float encoded = (non_linear_PDF - 0.13f) / 0.87f;
    if (std::isnan(encoded)) encoded = 1.0f;   // device: fminf(1, x) clamps NaN to 1 (ORS/ShadingModels/Utils.h:123-128)
    return fminf(1.0f, encoded);
}
inline float estimate_alpha(float abs_cos_theta, float max_PDF) {
    return bilinear(tables().alpha, 32, 32, encode_PDF(max_PDF), abs_cos_theta);
}
inline float from_PDF(float abs_cos_theta, PDF max_PDF) {
    if (max_PDF.is_delta_dirac())
        return 0.0f;
    return sqrtf(estimate_alpha(abs_cos_theta, max_PDF.value()));
}
} // namespace GGXMinimumRoughness

// ---------------------------------------------------------------------------------------------
// Distributions
// ---------------------------------------------------------------------------------------------
namespace Dist {

namespace Disk {
inline float PDF(float radius) { return 1.0f / (PIf * pow2(radius)); }
inline float2 sample(float radius, float2 u) {
    float r = sqrtf(u.x) * radius;
    float phi = 2.0f * PIf * u.y;
    return {r * exact_cosf(phi), r * exact_sinf(phi)};
}
}

namespace Cone {
inline float PDF(float cos_theta_max) { return 1.0f / (2.0f * PIf * (1.0f - cos_theta_max)); }
inline DirectionalSample sample(float cos_theta_max, float2 u) {
    float cos_theta = (1.0f - u.x) + u.x * cos_theta_max;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * PIf * u.y;
    float sin_phi, cos_phi;
    sincos(phi, sin_phi, cos_phi);
    return {{cos_phi * sin_theta, sin_phi * sin_theta, cos_theta}, PDF(cos_theta_max)};
}
}

namespace UniformSphere {
inline float PDF() { return 0.25f * RECIP_PIf; }
inline DirectionalSample sample(float2 rs) {
    float2 u = 2.0f * rs - 1.0f;
    float d = 1 - (fabsf(u.x) + fabsf(u.y));
    float r = 1 - fabsf(d);
    float phi = (r == 0) ? 0 : (PIf / 4) * ((fabsf(u.x) - fabsf(u.y)) / r + 1);
    float sin_phi, cos_phi;
    sincos(phi, sin_phi, cos_phi);
    float f = r * sqrtf(2 - r * r);
    return {{f * signf(u.x) * cos_phi, f * signf(u.y) * sin_phi, signf(d) * (1 - r * r)}, PDF()};
}
}

namespace UniformHemisphere {
inline float PDF() { return 0.5f * RECIP_PIf; }
inline DirectionalSample sample(float2 u) {
    float z = u.x;
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float phi = TWO_PIf * u.y;
    float sin_phi, cos_phi;
    sincos(phi, sin_phi, cos_phi);
    return {{r * cos_phi, r * sin_phi, z}, PDF()};
}
}

namespace OrenNayarCLTC {
inline float3 apply_tangent_basis(const Matrix2x2& tangents, float3 w) {
    float2 xy = tangents * make_float2(w);
    return make_float3(xy, w.z);
}
inline Matrix2x2 orthonormal_tangents_ltc(float3 w) {
    float2 wh = make_float2(w);
    float len_sqr = dot(wh, wh);
    float2 X = len_sqr > 0.0f ? wh / sqrtf(len_sqr) : make_float2(1, 0);
    float2 Y = make_float2(-X.y, X.x);
    Matrix2x2 res;
    res.setCol(0, X);
    res.setCol(1, Y);
    return res;
}
inline void coefficients(float cos_theta, float roughness, float& a, float& b, float& c, float& d) {
    a = 1.0f + roughness * (0.303392f + (-0.518982f + 0.111709f * cos_theta) * cos_theta + (-0.276266f + 0.335918f * cos_theta) * roughness);
    b = roughness * (-1.16407f + 1.15859f * cos_theta + (0.150815f - 0.150105f * cos_theta) * roughness) / (cos_theta * cos_theta * cos_theta - 1.43545f);
    c = 1.0f + (0.20013f + (-0.506373f + 0.261777f * cos_theta) * cos_theta) * roughness;
    d = ((0.540852f + (-1.01625f + 0.475392f * cos_theta) * cos_theta) * roughness) / (-1.0743f + cos_theta * (0.0725628f + cos_theta));
}
inline DirectionalSample sample(float roughness, float3 wo, float2 u) {
    float a, b, c, d;
    coefficients(wo.z, roughness, a, b, c, d);
    float radius = sqrtf(u.x);
    float phi = 2.0f * PIf * u.y;
    float sin_phi, cos_phi;
    sincos(phi, sin_phi, cos_phi);
    float x = radius * cos_phi;
    float y = radius * sin_phi;
    float vz = 1.0f / sqrtf(d * d + 1.0f);
    float s = 0.5f * (1.0f + vz);
    x = -lerp(sqrtf(1.0f - y * y), x, s);
    float3 wh = make_float3(x, y, sqrtf(fmaxf(1.0f - (x * x + y * y), 0.0f)));
    float pdf_wh = wh.z / (PIf * s);
    float3 wi = make_float3(a * wh.x + b * wh.z, c * wh.y, d * wh.x + wh.z);
    float wi_magnitude = length(wi);
    float determinant_M = c * (a - b * d);
    float pdf_wi = pdf_wh * wi_magnitude * wi_magnitude * wi_magnitude / determinant_M;
    Matrix2x2 from_LTC = orthonormal_tangents_ltc(wo);
    wi = normalize(apply_tangent_basis(from_LTC, wi));
    return {wi, pdf_wi};
}
inline float PDF(float roughness, float3 wo, float3 wi_shading) {
    Matrix2x2 to_LTC = orthonormal_tangents_ltc(wo).transpose();
    float3 wi = apply_tangent_basis(to_LTC, wi_shading);
    float a, b, c, d;
    coefficients(wo.z, roughness, a, b, c, d);
    float determinant_M = c * (a - b * d);
    float3 wh = make_float3(c * (wi.x - b * wi.z), (a - b * d) * wi.y, -c * (d * wi.x - a * wi.z));
    float wh_magnitude_squared = dot(wh, wh);
    float vz = 1.0f / sqrtf(d * d + 1.0f);
    float s = 0.5f * (1.0f + vz);
    return determinant_M * determinant_M / pow2(wh_magnitude_squared) * fmaxf(wh.z, 0.0f) / (PIf * s);
}
}

namespace GGX_VNDF {
inline float D(float alpha, float3 h) {
    float m = pow2(h.x / alpha) + pow2(h.y / alpha) + pow2(h.z);
    return 1 / (PIf * alpha * alpha * pow2(m));
}
inline float lambda(float alpha, float3 w) {
    return 0.5f * (-1 + sqrtf(1 + (pow2(alpha * w.x) + pow2(alpha * w.y)) / pow2(w.z)));
}
inline float3 sample_halfway(float alpha, float3 wo, float2 u) {
    float3 wo_std = normalize(make_float3(alpha * wo.x, alpha * wo.y, wo.z));
    float phi = 2.0f * PIf * u.y;
    float z = fmaf(1.0f - u.x, 1.0f + wo_std.z, -wo_std.z);
    float sin_theta = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
    float sin_phi, cos_phi;
    sincos(phi, sin_phi, cos_phi);
    float3 c = make_float3(sin_theta * cos_phi, sin_theta * sin_phi, z);
    float3 wi_std = c + wo_std;
    return normalize(make_float3(alpha * wi_std.x, alpha * wi_std.y, fmaxf(0.0f, wi_std.z)));
}
inline float PDF(float alpha, float3 wo, float3 h) {
    float recip_G1 = 1.0f + lambda(alpha, wo);
    return dot(wo, h) * D(alpha, h) / (recip_G1 * fabsf(wo.z));
}
inline DirectionalSample sample(float alpha, float3 wo, float2 u) {
    float3 h = sample_halfway(alpha, wo, u);
    return {h, PDF(alpha, wo, h)};
}
}

namespace GGX_Bounded_VNDF {
inline float D(float alpha, float3 h) { return GGX_VNDF::D(alpha, h); }
inline float3 sample_reflection(float alpha, float3 wo, float2 u) {
    float3 wo_std = normalize(make_float3(wo.x * alpha, wo.y * alpha, wo.z));
    float phi = 2.0f * PIf * u.y;
    float a = alpha;
    float s = 1.0f + length(make_float2(wo));
    float a2 = a * a, s2 = s * s;
    float k = (1.0f - a2) * s2 / (s2 + a2 * wo.z * wo.z);
    float b = wo.z >= 0 ? k * wo_std.z : wo_std.z;
    float z = fmaf(1.0f - u.x, 1.0f + b, -b);
    float sin_theta = sqrtf(fmaxf(1.0f - z * z, 0.0f));
    float sin_phi, cos_phi;
    sincos(phi, sin_phi, cos_phi);
    float3 o_std = {sin_theta * cos_phi, sin_theta * sin_phi, z};
    float3 h_std = wo_std + o_std;
    float3 h = normalize(make_float3(h_std.x * alpha, h_std.y * alpha, h_std.z));
    return reflect(-wo, h);
}
inline float reflection_PDF(float alpha, float3 wo, float3 wi) {
    float3 h = normalize(wo + wi);
    float ndf = D(alpha, h);
    float2 ao = alpha * make_float2(wo);
    float len2 = dot(ao, ao);
    float t = sqrtf(len2 + wo.z * wo.z);
    if (wo.z >= 0.0f) {
        float s = 1.0f + length(make_float2(wo));
        float a2 = alpha * alpha, s2 = s * s;
        float k = (1.0f - a2) * s2 / (s2 + a2 * wo.z * wo.z);
        return ndf / (2.0f * (k * wo.z + t));
    }
    return ndf * (t - wo.z) / (2.0f * len2);
}
inline DirectionalSample sample(float alpha, float3 wo, float2 u) {
    float3 d = sample_reflection(alpha, wo, u);
    return {d, reflection_PDF(alpha, wo, d)};
}
}

} // namespace Dist

// ---------------------------------------------------------------------------------------------
// BSDFs
// ---------------------------------------------------------------------------------------------
namespace OrenNayar {
static const float constant1_FON = 0.5f - 2.0f / (3.0f * PIf);

inline float E_FON_exact(float cos_theta, float, float A, float B) {
    float Si = sqrtf(1.0f - (cos_theta * cos_theta));
    float G = Si * (exact_acosf(cos_theta) - Si * cos_theta) + (2.0f / 3.0f) * ((Si / cos_theta) * (1.0f - (Si * Si * Si)) - Si);
    return A + B * G * RECIP_PIf;
}
inline float E_FON_approx(float cos_theta, float, float A, float B) {
    float mucomp = 1.0f - cos_theta;
    float GoverPi = 0.0f;
    const float g[4] = {0.0714429953f, -0.332181442f, 0.491881867f, 0.0571085289f};
    for (int i = 0; i < 4; ++i)
        GoverPi = mucomp * (g[i] + GoverPi);
    return A + B * GoverPi;
}
inline float E_FON_exact(float cos_theta, float roughness) {
    float A = 1.0f / (1.0f + constant1_FON * roughness);
    return E_FON_exact(cos_theta, roughness, A, roughness * A);
}
inline float E_FON_approx(float cos_theta, float roughness) {
    float A = 1.0f / (1.0f + constant1_FON * roughness);
    return E_FON_approx(cos_theta, roughness, A, roughness * A);
}

inline float evaluate(float roughness, float3 wo, float3 wi, bool exact = false) {
    const float constant2_FON = 2.0f / 3.0f - 28.0f / (15.0f * PIf);
    float cos_theta_i = wi.z;
    float cos_theta_o = wo.z;
    float s = dot(wi, wo) - cos_theta_i * cos_theta_o;
    float s_over_t = s > 0.0f ? s / fmaxf(cos_theta_i, cos_theta_o) : s;
    float A = 1.0f / (1.0f + constant1_FON * roughness);
    float B = roughness * A;
    float f_single_scatter = RECIP_PIf * A * (1.0f + roughness * s_over_t);
    float EF_o = exact ? E_FON_exact(cos_theta_o, roughness, A, B) : E_FON_approx(cos_theta_o, roughness, A, B);
    float EF_i = exact ? E_FON_exact(cos_theta_i, roughness, A, B) : E_FON_approx(cos_theta_i, roughness, A, B);
    float average_EF = A * (1.0f + constant2_FON * roughness);
    float multi_scatter_rho = average_EF / (1.0f - (1.0f - average_EF));
    float f_multi_scatter = (multi_scatter_rho * RECIP_PIf) * fabsf(1.0f - EF_o) * fabsf(1.0f - EF_i) / fmaxf(1.0e-7f, 1.0f - average_EF);
    return f_single_scatter + f_multi_scatter;
}
inline float3 evaluate(float3 albedo, float roughness, float3 wo, float3 wi, bool exact = false) {
    return albedo * evaluate(roughness, wo, wi, exact);
}
inline float uniform_probability(float roughness, float cos_theta) {
    return exact_powf(roughness, 0.1f) * (0.162925f + cos_theta * (-0.372058f + (0.538233f - 0.290822f * cos_theta) * cos_theta));
}
inline PDF pdf(float roughness, float3 wo, float3 wi) {
    float up = uniform_probability(roughness, wo.z);
    float cltc_probability = 1.0f - up;
    float cltc_PDF = Dist::OrenNayarCLTC::PDF(roughness, wo, wi);
    return up * Dist::UniformHemisphere::PDF() + cltc_probability * cltc_PDF;
}
inline BSDFResponse evaluate_with_PDF(float3 albedo, float roughness, float3 wo, float3 wi, bool exact = false) {
    return {evaluate(albedo, roughness, wo, wi, exact), pdf(roughness, wo, wi)};
}
inline BSDFSample sample(float3 albedo, float roughness, float3 wo, float2 u, bool exact = false) {
    float up = uniform_probability(roughness, wo.z);
    float cltc_probability = 1.0f - up;
    DirectionalSample ds;
    float cltc_PDF;
    if (u.x <= up) {
        u.x = u.x / up;
        ds = Dist::UniformHemisphere::sample(u);
        cltc_PDF = Dist::OrenNayarCLTC::PDF(roughness, wo, ds.direction);
    } else {
        u.x = (u.x - up) / cltc_probability;
        ds = Dist::OrenNayarCLTC::sample(roughness, wo, u);
        cltc_PDF = ds.pdf;
    }
    ds.pdf = up * Dist::UniformHemisphere::PDF() + cltc_probability * cltc_PDF;
    BSDFSample r;
    r.direction = ds.direction;
    r.pdf = ds.pdf;
    r.reflectance = evaluate(albedo, roughness, wo, r.direction, exact);
    return r;
}
} // namespace OrenNayar

namespace GGX {
static const float MIN_ALPHA = 1e-4f;
inline float alpha_from_roughness(float roughness) { return fmaxf(MIN_ALPHA, roughness * roughness); }
inline float roughness_from_alpha(float alpha) { return sqrtf(alpha); }
inline bool effectively_smooth(float alpha) { return alpha <= MIN_ALPHA; }
inline float height_correlated_G(float alpha, float3 wo, float3 wi) {
    return 1.0f / (1.0f + Dist::GGX_VNDF::lambda(alpha, wo) + Dist::GGX_VNDF::lambda(alpha, wi));
}
}

namespace GGX_R {
inline float3 evaluate(float alpha, float3 specularity, float3 wo, float3 wi) {
    if (GGX::effectively_smooth(alpha))
        return make_float3(0.0f);
    if (wo.z * wi.z <= 0.0f)
        return make_float3(0.0f);
    float3 h = normalize(wo + wi);
    float G = GGX::height_correlated_G(alpha, wo, wi);
    float D = Dist::GGX_VNDF::D(alpha, h);
    float3 F = schlick_fresnel(specularity, dot(wo, h));
    return F * (D * G / (4.0f * wo.z * wi.z));
}
inline PDF pdf(float alpha, float3 wo, float3 wi) {
    if (GGX::effectively_smooth(alpha))
        return PDF::invalid();
    return Dist::GGX_Bounded_VNDF::reflection_PDF(alpha, wo, wi);
}
inline BSDFResponse evaluate_with_PDF(float alpha, float3 specularity, float3 wo, float3 wi) {
    return {evaluate(alpha, specularity, wo, wi), pdf(alpha, wo, wi)};
}
inline BSDFSample sample(float alpha, float3 specularity, float3 wo, float2 u) {
    BSDFSample r;
    if (GGX::effectively_smooth(alpha)) {
        r.direction = {-wo.x, -wo.y, wo.z};
        r.pdf = PDF::delta_dirac(1);
        r.reflectance = schlick_fresnel(specularity, fabsf(wo.z)) / fabsf(r.direction.z);
        return r;
    }
    DirectionalSample s = Dist::GGX_Bounded_VNDF::sample(alpha, wo, u);
    r.direction = s.direction;
    r.pdf = s.pdf;
    r.reflectance = evaluate(alpha, specularity, wo, r.direction);
    bool energyloss = r.direction.z < 0.0f;
    return energyloss ? BSDFSample::none() : r;
}
} // namespace GGX_R

namespace GGX_T {
inline float transmission_PDF_scale(float ior_i_over_o, float3 wo, float3 wi, float3 h) {
    float sqrt_denom = dot(wo, h) + ior_i_over_o * dot(wi, h);
    return pow2(ior_i_over_o / sqrt_denom) * fabsf(dot(wi, h));
}
inline float3 compute_halfway_vector(float ior_i_over_o, float3 wo, float3 wi) {
    float3 h = normalize(wo + ior_i_over_o * wi);
    if (h.z < 0.0f)
        h = -h;
    return h;
}
inline float evaluate(float alpha, float3 wo, float3 wi, float ior_i_over_o, float3 h) {
    if (GGX::effectively_smooth(alpha))
        return 0.0f;
    if (signf(wo.z) == signf(wi.z))
        return 0.0f;
    if (dot(wi, h) * wi.z <= 0 || dot(wo, h) * wo.z <= 0)
        return 0.0f;
    float G = GGX::height_correlated_G(alpha, wo, wi);
    float D = Dist::GGX_VNDF::D(alpha, h);
    float F = 1.0f;
    float f1 = fabsf(dot(wo, h) * dot(wi, h) / (wo.z * wi.z));
    float f2 = pow2(ior_i_over_o) * G * F * D / pow2(dot(wo, h) + ior_i_over_o * dot(wi, h));
    return f1 * f2;
}
inline float evaluate(float alpha, float ior_i_over_o, float3 wo, float3 wi) {
    return evaluate(alpha, wo, wi, ior_i_over_o, compute_halfway_vector(ior_i_over_o, wo, wi));
}
inline PDF pdf(float alpha, float ior_i_over_o, float3 wo, float3 wi) {
    if (GGX::effectively_smooth(alpha))
        return PDF::invalid();
    if (signf(wo.z) == signf(wi.z))
        return PDF::invalid();
    bool entering = wo.z >= 0.0f;
    if (!entering) { wo.z = -wo.z; wi.z = -wi.z; }
    float3 h = compute_halfway_vector(ior_i_over_o, wo, wi);
    if (dot(wo, h) < 0.0f || dot(wi, h) >= 0.0f)
        return PDF::invalid();
    return Dist::GGX_VNDF::PDF(alpha, wo, h) * transmission_PDF_scale(ior_i_over_o, wo, wi, h);
}
inline BSDFSample sample(float alpha, float ior_i_over_o, float3 wo, float2 u) {
    BSDFSample r;
    bool entering = wo.z >= 0.0f;
    if (!entering)
        wo.z = -wo.z;
    if (GGX::effectively_smooth(alpha)) {
        if (!refract_z(r.direction, -wo, ior_i_over_o))
            return BSDFSample::none();
        float reflectance = 1.0f / fabsf(r.direction.z);
        r.reflectance = make_float3(reflectance);
        r.pdf = PDF::delta_dirac(1);
    } else {
        float3 h = Dist::GGX_VNDF::sample_halfway(alpha, wo, u);
        r.pdf = Dist::GGX_VNDF::PDF(alpha, wo, h);
        if (!refract(r.direction, -wo, h, ior_i_over_o))
            return BSDFSample::none();
        r.pdf = r.pdf * transmission_PDF_scale(ior_i_over_o, wo, r.direction, h);
        bool energyloss = r.direction.z >= -0.0f;
        if (energyloss)
            return BSDFSample::none();
        r.reflectance = make_float3(evaluate(alpha, wo, r.direction, ior_i_over_o, h));
    }
    if (!entering)
        r.direction.z = -r.direction.z;
    return r;
}
} // namespace GGX_T

namespace GGX_RT {   // the combined reflection + transmission "GGX" namespace of GGX.h:267-441
inline float normalize_reflection_probability(float reflection_probability, float3 transmission_tint) {
    float transmission_probability = 1.0f - reflection_probability;
    float scaled_t = sum(transmission_tint) * transmission_probability;
    float scaled_r = 3 * reflection_probability;
    return scaled_r / (scaled_r + scaled_t);
}
inline float evaluate(float alpha, float specularity, float ior_i_over_o, float3 wo, float3 wi) {
    if (GGX::effectively_smooth(alpha) || wo.z == 0.0f || wi.z == 0.0f)
        return 0.0f;
    bool entering = wo.z >= 0.0f;
    if (!entering) { wo.z = -wo.z; wi.z = -wi.z; }
    bool is_reflection = same_hemisphere(wo, wi);
    float halfway_ior = is_reflection ? 1.0f : ior_i_over_o;
    float3 h = GGX_T::compute_halfway_vector(halfway_ior, wo, wi);
    float G = GGX::height_correlated_G(alpha, wo, wi);
    float D = Dist::GGX_VNDF::D(alpha, h);
    float F = dielectric_schlick_fresnel(specularity, dot(wo, h), ior_i_over_o);
    if (is_reflection)
        return F * D * G / (4.0f * wo.z * wi.z);
    if (dot(wi, h) * wi.z <= 0 || dot(wo, h) * wo.z <= 0)
        return 0.0f;
    float f1 = fabsf(dot(wo, h) * dot(wi, h) / (wo.z * wi.z));
    float f2 = (1 - F) * G * D * pow2(ior_i_over_o / (dot(wo, h) + ior_i_over_o * dot(wi, h)));
    return f1 * f2;
}
inline float3 evaluate(float3 transmission_tint, float alpha, float specularity, float ior_i_over_o, float3 wo, float3 wi) {
    float f = evaluate(alpha, specularity, ior_i_over_o, wo, wi);
    bool is_transmission = signf(wo.z) != signf(wi.z);
    return f * (is_transmission ? transmission_tint : make_float3(1));
}
inline PDF pdf(float3 transmission_tint, float alpha, float specularity, float ior_i_over_o, float3 wo, float3 wi) {
    if (GGX::effectively_smooth(alpha))
        return PDF::invalid();
    bool entering = wo.z >= 0.0f;
    if (!entering) { wo.z = -wo.z; wi.z = -wi.z; }
    bool is_reflection = same_hemisphere(wo, wi);
    float halfway_ior = is_reflection ? 1.0f : ior_i_over_o;
    float3 h = GGX_T::compute_halfway_vector(halfway_ior, wo, wi);
    bool backfacing_microfacet = !is_reflection && (dot(wo, h) < 0.0f || dot(wi, h) >= 0.0f);
    if (backfacing_microfacet)
        return PDF::invalid();
    PDF p = Dist::GGX_VNDF::PDF(alpha, wo, h);
    float reflection_probability = dielectric_schlick_fresnel(specularity, dot(wo, h), ior_i_over_o);
    float nrp = normalize_reflection_probability(reflection_probability, transmission_tint);
    p *= is_reflection ? nrp : (1 - nrp);
    if (is_reflection)
        p *= 1 / (4.0f * dot(wo, h));
    else
        p *= GGX_T::transmission_PDF_scale(ior_i_over_o, wo, wi, h);
    return p;
}
inline BSDFResponse evaluate_with_PDF(float3 transmission_tint, float alpha, float specularity, float ior_i_over_o, float3 wo, float3 wi) {
    return {evaluate(transmission_tint, alpha, specularity, ior_i_over_o, wo, wi), pdf(transmission_tint, alpha, specularity, ior_i_over_o, wo, wi)};
}
inline BSDFSample sample(float3 transmission_tint, float alpha, float specularity, float ior_i_over_o, float3 wo, float3 u) {
    BSDFSample r;
    bool entering = wo.z >= 0.0f;
    if (!entering)
        wo.z = -wo.z;
    if (GGX::effectively_smooth(alpha)) {
        float reflection_probability = dielectric_schlick_fresnel(specularity, fabsf(wo.z), ior_i_over_o);
        float nrp = normalize_reflection_probability(reflection_probability, transmission_tint);
        bool is_reflection = u.z < nrp;
        if (is_reflection) {
            r.pdf = PDF::delta_dirac(nrp);
            r.direction = {-wo.x, -wo.y, wo.z};
        } else {
            r.pdf = PDF::delta_dirac(1.0f - nrp);
            if (!refract_z(r.direction, -wo, ior_i_over_o))
                return BSDFSample::none();
        }
        float reflectance = (is_reflection ? reflection_probability : (1.0f - reflection_probability)) / fabsf(r.direction.z);
        r.reflectance = make_float3(reflectance);
    } else {
        DirectionalSample hs = Dist::GGX_VNDF::sample(alpha, wo, make_float2(u));
        float3 h = hs.direction;
        r.pdf = hs.pdf;
        float reflection_probability = dielectric_schlick_fresnel(specularity, dot(wo, h), ior_i_over_o);
        float nrp = normalize_reflection_probability(reflection_probability, transmission_tint);
        bool is_reflection = u.z < nrp;
        if (is_reflection) {
            r.direction = reflect(-wo, h);
            r.pdf *= nrp / (4.0f * dot(wo, h));
        } else {
            if (!refract(r.direction, -wo, h, ior_i_over_o))
                return BSDFSample::none();
            r.pdf *= 1 - nrp;
            r.pdf *= GGX_T::transmission_PDF_scale(ior_i_over_o, wo, r.direction, h);
        }
        bool energyloss = is_reflection ? r.direction.z < 0.0f : r.direction.z >= 0.0f;
        if (energyloss)
            return BSDFSample::none();
        r.reflectance = make_float3(evaluate(alpha, specularity, ior_i_over_o, wo, r.direction));
    }
    bool is_transmission = signf(wo.z) != signf(r.direction.z);
    if (is_transmission)
        r.reflectance *= transmission_tint;
    if (!entering)
        r.direction.z = -r.direction.z;
    return r;
}
} // namespace GGX_RT

} // namespace oracle
