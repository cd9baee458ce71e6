// device_shading.h -- BSDFs, shading models and light sources of the HIP path tracer (device code).
//
// What the reference computes in OR/Shading/BSDFs/{OrenNayar,GGX}.h, OR/Distributions.h,
// OR/Shading/ShadingModels/{Diffuse,Default,Transmissive}Shading.h, OR/Shading/ShadingModels/Utils.h and
// OR/Shading/LightSources/*Impl.h (OR = extensions/OptiXRenderer/OptiXRenderer), written for gfx950:
// plain f32 VALU code, no MFMA (branchy, no dense contraction), tables fetched with software
// bilinear filtering from unorm16 arrays that stay L1/L2 resident (4 KB + 32 KB + 2 KB).
#pragma once

#include "device_math.h"
#include "spec_math.h"
#include "../../include/hiprenderer_c.h"

namespace hipr {

#define HIPR_PI 3.14159265358979323846f
#define HIPR_TWO_PI 6.283185307f
#define HIPR_RECIP_PI 0.31830988618379067153776752674503f
#define HIPR_COAT_SPECULARITY 0.04f
#define HIPR_COAT_IOR 1.5f
#define HIPR_AIR_IOR 1.0f
#define HIPR_MIN_VALID_PDF 0.000001f
#define HIPR_GGX_MIN_ALPHA 1e-4f

// ---------------------------------------------------------------------------------------------
// PDF encoding of the reference (OR/Types.h:152-204): negative = delta dirac, NaN = invalid.
// ---------------------------------------------------------------------------------------------
HD float pdf_value(float p) { return fabsf(p); }
HD bool pdf_is_valid(float p) { return fabsf(p) > HIPR_MIN_VALID_PDF; }
HD bool pdf_is_delta(float p) { return !(p >= 0.0f); }
HD bool pdf_valid_not_delta(float p) { return p > HIPR_MIN_VALID_PDF; }
HD float pdf_disable_MIS(float p) { return p >= 0.0f ? -p : p; }
HD float pdf_invalid() { return __builtin_nanf(""); }

struct Response { f3 f; float pdf; };
struct Sample { f3 f; float pdf; f3 dir; };
HD Response response_none() { return {{0, 0, 0}, 0.0f}; }
HD Sample sample_none() { return {{0, 0, 0}, 0.0f, {0, 0, 0}}; }

// HIPR_FAST_MATH (set for the shade translation unit): hardware sin / cos / exp2 / log2 like the reference's
// --use_fast_math PTX (extensions/OptiXRenderer/CMakeLists.txt:82-83); otherwise the correctly rounded-ish ocml versions.
// HIPR_VERIFY_MATH (the EXACT arithmetic mode, hipr_set_arithmetic / shade.hip compiled a second time into the product library): sin, cos and pow are the
// specified f64 sequences of spec_math.h -- fixed chains of correctly rounded binary64 operations that oracle/vecmath.h restates, so device and oracle agree in
// every bit by construction -- and atan2 / asin (the environment map's lookup only, never on the headline path) are evaluated by the f64 libm of either side and
// rounded once (two f64 results within a few ulp of the true value round to the same f32 but for ~2^-26 of the arguments). With correctly rounded division and
// square root and no contraction (the traversal unit's flags) K3 is then checked exactly, not statistically (tests/test_gpu_verify_build.py).
#ifndef HIPR_VERIFY_MATH
#define HIPR_VERIFY_MATH 0
#endif
// How each transcendental is evaluated: 1 = the hardware's approximation (fast shade unit), 3 = the specified f64 sequence (exact mode), 2 = the f64 libm rounded once
// (round 5's verification build; still selectable for the A/B), 0 = ocml's f32 function. Separately settable for the attribution experiment of round 5
// (tools/fast_math_attribution.sh: which approximation moves how many paths).
#ifndef HIPR_SINCOS_KIND
#define HIPR_SINCOS_KIND (HIPR_VERIFY_MATH ? 3 : (HIPR_FAST_MATH ? 1 : 0))
#endif
#ifndef HIPR_POW_KIND
#define HIPR_POW_KIND (HIPR_VERIFY_MATH ? 3 : (HIPR_FAST_MATH ? 1 : 0))
#endif
#ifndef HIPR_ATAN_ASIN_KIND
#define HIPR_ATAN_ASIN_KIND (HIPR_VERIFY_MATH ? 2 : 0)
#endif
#ifndef HIPR_NATIVE_POW
#define HIPR_NATIVE_POW 1
#endif
HD void sincos_(float a, float& s, float& c) {
#if HIPR_SINCOS_KIND == 3
    spec_sincos(a, s, c);
#elif HIPR_SINCOS_KIND == 2
    s = float(sin(double(a))); c = float(cos(double(a)));
#elif HIPR_SINCOS_KIND == 1
    s = __sinf(a); c = __cosf(a);
#else
    s = sinf(a); c = cosf(a);
#endif
}
HD float pow_(float x, float y) {
#if HIPR_POW_KIND == 3
    return spec_pow(x, y);
#elif HIPR_POW_KIND == 2
    return float(pow(double(x), double(y)));
#elif HIPR_POW_KIND == 1 && HIPR_NATIVE_POW
    // x^y as exp2(y * log2(x)) on the hardware's v_log_f32 / v_exp_f32, what --use_fast_math makes of powf in the reference's PTX (HIP's __powf is the full ocml pow).
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
#elif HIPR_POW_KIND == 1
    return __powf(x, y);
#else
    return powf(x, y);
#endif
}
HD float atan2_(float y, float x) {
#if HIPR_ATAN_ASIN_KIND == 2
    return float(atan2(double(y), double(x)));
#else
    return atan2f(y, x);
#endif
}
HD float asin_(float x) {
#if HIPR_ATAN_ASIN_KIND == 2
    return float(asin(double(x)));
#else
    return asinf(x);
#endif
}

HD float sgn(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
HD float pow2(float x) { return x * x; }
HD f3 pow2(f3 x) { return x * x; }
HD float pow4(float x) { float xx = x * x; return xx * xx; }
HD float pow5(float x) { float xx = x * x; return xx * xx * x; }
HD float saturate(float v) { return clampf(v, 0.0f, 1.0f); }

HD float dielectric_specularity(float ior_o, float ior_i) { return pow2((ior_o - ior_i) / (ior_o + ior_i)); }
HD float dielectric_ior_from_specularity(float s) { return 2.0f / (1.0f - sqrtf(s)) - 1.0f; }
HD float adjust_dielectric_specularity(float exterior_ior, float s) { return dielectric_specularity(exterior_ior, dielectric_ior_from_specularity(s)); }
HD f3 conductor_specularity(f3 ior_o, f3 ior_i, f3 ext) {
    f3 e2 = pow2(ext);
    return (pow2(ior_o - ior_i) + e2) / (pow2(ior_o + ior_i) + e2);
}
HD f3 conductor_ior_from_specularity(f3 s, f3 ext) {
    f3 a = s - 1.0f;
    f3 b = 2.0f * s + 2.0f;
    f3 c = a + (s - 1.0f) * pow2(ext);
    f3 d = b * b - 4.0f * a * c;
    f3 sq = {sqrtf(d.x), sqrtf(d.y), sqrtf(d.z)};
    return (-b + sq) / (2.0f * a);
}
HD f3 schlick_fresnel3(f3 f0, float abs_cos) { float t = pow5(1.0f - abs_cos); return (1.0f - t) * f0 + t; }
HD float dielectric_schlick_fresnel(float f0, float abs_cos, float ior_i_over_o) {
    float sin2 = 1 - pow2(abs_cos);
    if (sin2 >= pow2(ior_i_over_o)) return 1.0f;
    float t = pow5(1.0f - abs_cos);
    return (1.0f - t) * f0 + t;
}
HD float modulate_roughness_under_coat(float base, float coat) {
    float x_coat = 1 - HIPR_AIR_IOR / HIPR_COAT_IOR;
    return pow_(fminf(1, pow4(base) + 2.0f * x_coat * pow4(coat)), 0.25f);
}
HD bool refract_z(f3& out, f3 wi, float ior) {
    float nz = 1, c = wi.z;
    if (c > 0.0f) { nz = -1; c = -c; } else ior = 1.f / ior;
    float k = 1.0f - ior * ior * (1.0f - c * c);
    out = ior * wi - mk3(0, 0, (ior * c + sqrtf(k)) * nz);
    return k >= 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Precomputed tables (uploaded by hipr_upload_tables as unorm16, OR/Renderer.cpp:400-466).
// ---------------------------------------------------------------------------------------------
struct DeviceTables {
    const ushort2* ggx_rho;        // 32 x 32   (x = F0 0 "base", y = F0 1 "full"), index [roughness][cos]
    const ushort2* dielectric_rho; // 32 x 16 x 16 (total, reflected), slices 0..15 light medium, 16..31 dense
    const unsigned short* alpha;   // 32 x 32, index [cos][encoded pdf]
};

HD float un16(unsigned short v) { return float(v) / 65535.0f; }
HD f2 un16(ushort2 v) { return {float(v.x) / 65535.0f, float(v.y) / 65535.0f}; }

struct BilinearTap { int i00, i01, i10, i11; float tu, tv; };
HD BilinearTap bilinear_tap(int width, int height, float u, float v) {
    u = clampf(u, 0.0f, 1.0f);
    float uc = u * (width - 1);
    int lu = int(uc);
    int uu = min(lu + 1, width - 1);
    v = clampf(v, 0.0f, 1.0f);
    float vc = v * (height - 1);
    int lv = int(vc);
    int uv = min(lv + 1, height - 1);
    return {lv * width + lu, lv * width + uu, uv * width + lu, uv * width + uu, uc - lu, vc - lv};
}
HD float lerp_ba(float a, float b, float t) { return a + (b - a) * t; }
HD f2 lerp_ba(f2 a, f2 b, float t) { return a + (b - a) * t; }

HD f2 fetch_specular_rho(const DeviceTables& t, float abs_cos, float roughness) {   // (base, full)
    BilinearTap b = bilinear_tap(32, 32, abs_cos, roughness);
    f2 lo = lerp_ba(un16(t.ggx_rho[b.i00]), un16(t.ggx_rho[b.i01]), b.tu);
    f2 hi = lerp_ba(un16(t.ggx_rho[b.i10]), un16(t.ggx_rho[b.i11]), b.tu);
    return lerp_ba(lo, hi, b.tv);
}
HD f2 fetch_dielectric_rho(const DeviceTables& t, float abs_cos, float roughness, float ior_i_over_o) {   // (total, reflected)
    bool light = ior_i_over_o < 1.0f;
    float w = light ? (ior_i_over_o - 0.331492f) / 0.457982f : (ior_i_over_o - 1.26667f) / 1.75f;
    w = clampf(w, 0.0f, 1.0f);
    float wc = w * 15.0f;
    int lw = int(wc);
    int uw = min(lw + 1, 15);
    const ushort2* base = t.dielectric_rho + (light ? 0 : 16 * 256);
    BilinearTap b = bilinear_tap(16, 16, abs_cos, roughness);
    auto slice = [&](int s) {
        const ushort2* p = base + s * 256;
        f2 lo = lerp_ba(un16(p[b.i00]), un16(p[b.i01]), b.tu);
        f2 hi = lerp_ba(un16(p[b.i10]), un16(p[b.i11]), b.tu);
        return lerp_ba(lo, hi, b.tv);
    };
    return lerp_ba(slice(lw), slice(uw), wc - lw);
}
HD float min_roughness_from_PDF(const DeviceTables& t, float abs_cos, float max_PDF) {
    if (pdf_is_delta(max_PDF)) return 0.0f;
    float p = pdf_value(max_PDF);
    float non_linear = p / (1.0f + p);
    float encoded = (non_linear - 0.13f) / 0.87f;
    encoded = (encoded != encoded) ? 1.0f : fminf(1.0f, encoded);
    BilinearTap b = bilinear_tap(32, 32, encoded, abs_cos);
    float lo = lerp_ba(un16(t.alpha[b.i00]), un16(t.alpha[b.i01]), b.tu);
    float hi = lerp_ba(un16(t.alpha[b.i10]), un16(t.alpha[b.i11]), b.tu);
    return sqrtf(lerp_ba(lo, hi, b.tv));
}

// ---------------------------------------------------------------------------------------------
// Distributions
// ---------------------------------------------------------------------------------------------


HD f3 cone_sample(float cos_theta_max, f2 u, float& pdf) {
    float cos_theta = (1.0f - u.x) + u.x * cos_theta_max;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float s, c;
    sincos_(2.0f * HIPR_PI * u.y, s, c);
    pdf = 1.0f / (2.0f * HIPR_PI * (1.0f - cos_theta_max));
    return {c * sin_theta, s * sin_theta, cos_theta};
}

namespace cltc {
HD void coefficients(float cos_theta, float r, float& a, float& b, float& c, float& d) {
    a = 1.0f + r * (0.303392f + (-0.518982f + 0.111709f * cos_theta) * cos_theta + (-0.276266f + 0.335918f * cos_theta) * r);
    b = r * (-1.16407f + 1.15859f * cos_theta + (0.150815f - 0.150105f * cos_theta) * r) / (cos_theta * cos_theta * cos_theta - 1.43545f);
    c = 1.0f + (0.20013f + (-0.506373f + 0.261777f * cos_theta) * cos_theta) * r;
    d = ((0.540852f + (-1.01625f + 0.475392f * cos_theta) * cos_theta) * r) / (-1.0743f + cos_theta * (0.0725628f + cos_theta));
}
// 2x2 tangent frame [X Y] with X along the projection of w; apply(M, v) = (M * v.xy, v.z).
HD void tangents(f3 w, f2& X, f2& Y) {
    f2 wh = {w.x, w.y};
    float len2 = dot(wh, wh);
    X = len2 > 0.0f ? wh / sqrtf(len2) : mk2(1, 0);
    Y = {-X.y, X.x};
}
HD float pdf(float roughness, f3 wo, f3 wi_shading) {
    f2 X, Y;
    tangents(wo, X, Y);
    // transpose of [X Y]: rows X, Y
    f3 wi = {X.x * wi_shading.x + X.y * wi_shading.y, Y.x * wi_shading.x + Y.y * wi_shading.y, wi_shading.z};
    float a, b, c, d;
    coefficients(wo.z, roughness, a, b, c, d);
    float det = c * (a - b * d);
    f3 wh = {c * (wi.x - b * wi.z), (a - b * d) * wi.y, -c * (d * wi.x - a * wi.z)};
    float wh2 = dot(wh, wh);
    float vz = 1.0f / sqrtf(d * d + 1.0f);
    float s = 0.5f * (1.0f + vz);
    return det * det / pow2(wh2) * fmaxf(wh.z, 0.0f) / (HIPR_PI * s);
}
HD f3 sample(float roughness, f3 wo, f2 u, float& out_pdf) {
    float a, b, c, d;
    coefficients(wo.z, roughness, a, b, c, d);
    float radius = sqrtf(u.x);
    float sp, cp;
    sincos_(2.0f * HIPR_PI * u.y, sp, cp);
    float x = radius * cp;
    float y = radius * sp;
    float vz = 1.0f / sqrtf(d * d + 1.0f);
    float s = 0.5f * (1.0f + vz);
    x = -lerp(sqrtf(1.0f - y * y), x, s);
    f3 wh = {x, y, sqrtf(fmaxf(1.0f - (x * x + y * y), 0.0f))};
    float pdf_wh = wh.z / (HIPR_PI * s);
    f3 wi = {a * wh.x + b * wh.z, c * wh.y, d * wh.x + wh.z};
    float m = length(wi);
    float det = c * (a - b * d);
    out_pdf = pdf_wh * m * m * m / det;
    f2 X, Y;
    tangents(wo, X, Y);
    // [X Y] * wi.xy: rows (X.x, Y.x), (X.y, Y.y)
    f3 local = {X.x * wi.x + Y.x * wi.y, X.y * wi.x + Y.y * wi.y, wi.z};
    return normalize(local);
}
} // namespace cltc

namespace vndf {
HD float D(float alpha, f3 h) {
    float m = pow2(h.x / alpha) + pow2(h.y / alpha) + pow2(h.z);
    return 1 / (HIPR_PI * alpha * alpha * pow2(m));
}
HD float lambda(float alpha, f3 w) { return 0.5f * (-1 + sqrtf(1 + (pow2(alpha * w.x) + pow2(alpha * w.y)) / pow2(w.z))); }
HD float G(float alpha, f3 wo, f3 wi) { return 1.0f / (1.0f + lambda(alpha, wo) + lambda(alpha, wi)); }
HD f3 sample_halfway(float alpha, f3 wo, f2 u) {
    f3 wo_std = normalize(mk3(alpha * wo.x, alpha * wo.y, wo.z));
    float phi = 2.0f * HIPR_PI * u.y;
    float z = fmaf(1.0f - u.x, 1.0f + wo_std.z, -wo_std.z);
    float sin_theta = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
    float sp, cp;
    sincos_(phi, sp, cp);
    f3 c = {sin_theta * cp, sin_theta * sp, z};
    f3 wi_std = c + wo_std;
    return normalize(mk3(alpha * wi_std.x, alpha * wi_std.y, fmaxf(0.0f, wi_std.z)));
}
HD float pdf(float alpha, f3 wo, f3 h) {
    float recip_G1 = 1.0f + lambda(alpha, wo);
    return dot(wo, h) * D(alpha, h) / (recip_G1 * fabsf(wo.z));
}
HD f3 bounded_sample_reflection(float alpha, f3 wo, f2 u) {
    f3 wo_std = normalize(mk3(wo.x * alpha, wo.y * alpha, wo.z));
    float phi = 2.0f * HIPR_PI * u.y;
    float s = 1.0f + length(mk2(wo.x, wo.y));
    float a2 = alpha * alpha, s2 = s * s;
    float k = (1.0f - a2) * s2 / (s2 + a2 * wo.z * wo.z);
    float b = wo.z >= 0 ? k * wo_std.z : wo_std.z;
    float z = fmaf(1.0f - u.x, 1.0f + b, -b);
    float sin_theta = sqrtf(fmaxf(1.0f - z * z, 0.0f));
    float sp, cp;
    sincos_(phi, sp, cp);
    f3 o_std = {sin_theta * cp, sin_theta * sp, z};
    f3 h_std = wo_std + o_std;
    f3 h = normalize(mk3(h_std.x * alpha, h_std.y * alpha, h_std.z));
    return reflect(-wo, h);
}
HD float bounded_reflection_pdf(float alpha, f3 wo, f3 wi) {
    f3 h = normalize(wo + wi);
    float ndf = D(alpha, h);
    f2 ao = alpha * mk2(wo.x, wo.y);
    float len2 = dot(ao, ao);
    float t = sqrtf(len2 + wo.z * wo.z);
    if (wo.z >= 0.0f) {
        float s = 1.0f + length(mk2(wo.x, wo.y));
        float a2 = alpha * alpha, s2 = s * s;
        float k = (1.0f - a2) * s2 / (s2 + a2 * wo.z * wo.z);
        return ndf / (2.0f * (k * wo.z + t));
    }
    return ndf * (t - wo.z) / (2.0f * len2);
}
} // namespace vndf

// ---------------------------------------------------------------------------------------------
// Oren-Nayar (EON) with CLTC + uniform mixture sampling
// ---------------------------------------------------------------------------------------------
namespace oren_nayar {
#define HIPR_FON_C1 (0.5f - 2.0f / (3.0f * HIPR_PI))
HD float E_FON_approx(float cos_theta, float A, float B) {
    float mucomp = 1.0f - cos_theta;
    float g = 0.0f;
    g = mucomp * (0.0714429953f + g);
    g = mucomp * (-0.332181442f + g);
    g = mucomp * (0.491881867f + g);
    g = mucomp * (0.0571085289f + g);
    return A + B * g;
}
HD float evaluate(float roughness, f3 wo, f3 wi) {
    const float c2 = 2.0f / 3.0f - 28.0f / (15.0f * HIPR_PI);
    float ci = wi.z, co = wo.z;
    float s = dot(wi, wo) - ci * co;
    float s_over_t = s > 0.0f ? s / fmaxf(ci, co) : s;
    float A = 1.0f / (1.0f + HIPR_FON_C1 * roughness);
    float B = roughness * A;
    float single = HIPR_RECIP_PI * A * (1.0f + roughness * s_over_t);
    float EF_o = E_FON_approx(co, A, B);
    float EF_i = E_FON_approx(ci, A, B);
    float avg_EF = A * (1.0f + c2 * roughness);
    float ms_rho = avg_EF / (1.0f - (1.0f - avg_EF));
    float multi = (ms_rho * HIPR_RECIP_PI) * fabsf(1.0f - EF_o) * fabsf(1.0f - EF_i) / fmaxf(1.0e-7f, 1.0f - avg_EF);
    return single + multi;
}
HD float uniform_probability(float roughness, float cos_theta) {
    return pow_(roughness, 0.1f) * (0.162925f + cos_theta * (-0.372058f + (0.538233f - 0.290822f * cos_theta) * cos_theta));
}
HD float pdf(float roughness, f3 wo, f3 wi) {
    float up = uniform_probability(roughness, wo.z);
    float cp = 1.0f - up;
    return up * (0.5f * HIPR_RECIP_PI) + cp * cltc::pdf(roughness, wo, wi);
}
HD Response evaluate_with_PDF(f3 albedo, float roughness, f3 wo, f3 wi) {
    return {albedo * evaluate(roughness, wo, wi), pdf(roughness, wo, wi)};
}
HD Sample sample(f3 albedo, float roughness, f3 wo, f2 u) {
    float up = uniform_probability(roughness, wo.z);
    float cp = 1.0f - up;
    f3 dir;
    float cltc_pdf;
    if (u.x <= up) {
        u.x = u.x / up;
        float z = u.x;
        float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        float s, c;
        sincos_(HIPR_TWO_PI * u.y, s, c);
        dir = {r * c, r * s, z};
        cltc_pdf = cltc::pdf(roughness, wo, dir);
    } else {
        u.x = (u.x - up) / cp;
        dir = cltc::sample(roughness, wo, u, cltc_pdf);
    }
    Sample r;
    r.dir = dir;
    r.pdf = up * (0.5f * HIPR_RECIP_PI) + cp * cltc_pdf;
    r.f = albedo * evaluate(roughness, wo, dir);
    return r;
}
} // namespace oren_nayar

// ---------------------------------------------------------------------------------------------
// GGX reflection (bounded VNDF sampling) and the combined rough dielectric
// ---------------------------------------------------------------------------------------------
HD float ggx_alpha_from_roughness(float r) { return fmaxf(HIPR_GGX_MIN_ALPHA, r * r); }
HD bool ggx_smooth(float alpha) { return alpha <= HIPR_GGX_MIN_ALPHA; }

namespace ggx_r {
HD f3 evaluate(float alpha, f3 specularity, f3 wo, f3 wi) {
    if (ggx_smooth(alpha)) return mk3(0.0f);
    if (wo.z * wi.z <= 0.0f) return mk3(0.0f);
    f3 h = normalize(wo + wi);
    float G = vndf::G(alpha, wo, wi);
    float D = vndf::D(alpha, h);
    f3 F = schlick_fresnel3(specularity, dot(wo, h));
    return F * (D * G / (4.0f * wo.z * wi.z));
}
HD float pdf(float alpha, f3 wo, f3 wi) {
    if (ggx_smooth(alpha)) return pdf_invalid();
    return vndf::bounded_reflection_pdf(alpha, wo, wi);
}
HD Response evaluate_with_PDF(float alpha, f3 specularity, f3 wo, f3 wi) { return {evaluate(alpha, specularity, wo, wi), pdf(alpha, wo, wi)}; }
HD Sample sample(float alpha, f3 specularity, f3 wo, f2 u) {
    Sample r;
    if (ggx_smooth(alpha)) {
        r.dir = {-wo.x, -wo.y, wo.z};
        r.pdf = -1.0f;
        r.f = schlick_fresnel3(specularity, fabsf(wo.z)) / fabsf(r.dir.z);
        return r;
    }
    r.dir = vndf::bounded_sample_reflection(alpha, wo, u);
    r.pdf = vndf::bounded_reflection_pdf(alpha, wo, r.dir);
    r.f = evaluate(alpha, specularity, wo, r.dir);
    return r.dir.z < 0.0f ? sample_none() : r;
}
} // namespace ggx_r

namespace ggx_rt {
HD float transmission_PDF_scale(float ior, f3 wo, f3 wi, f3 h) {
    float sd = dot(wo, h) + ior * dot(wi, h);
    return pow2(ior / sd) * fabsf(dot(wi, h));
}
HD f3 halfway(float ior, f3 wo, f3 wi) {
    f3 h = normalize(wo + ior * wi);
    return h.z < 0.0f ? -h : h;
}
HD float normalize_reflection_probability(float rp, f3 tint) {
    float st = sum(tint) * (1.0f - rp);
    float sr = 3 * rp;
    return sr / (sr + st);
}
HD float evaluate1(float alpha, float specularity, float ior, f3 wo, f3 wi) {
    if (ggx_smooth(alpha) || wo.z == 0.0f || wi.z == 0.0f) return 0.0f;
    if (!(wo.z >= 0.0f)) { wo.z = -wo.z; wi.z = -wi.z; }
    bool is_reflection = wo.z * wi.z >= 0.0f;
    f3 h = halfway(is_reflection ? 1.0f : ior, wo, wi);
    float G = vndf::G(alpha, wo, wi);
    float D = vndf::D(alpha, h);
    float F = dielectric_schlick_fresnel(specularity, dot(wo, h), ior);
    if (is_reflection) return F * D * G / (4.0f * wo.z * wi.z);
    if (dot(wi, h) * wi.z <= 0 || dot(wo, h) * wo.z <= 0) return 0.0f;
    float f1 = fabsf(dot(wo, h) * dot(wi, h) / (wo.z * wi.z));
    float f2v = (1 - F) * G * D * pow2(ior / (dot(wo, h) + ior * dot(wi, h)));
    return f1 * f2v;
}
HD f3 evaluate(f3 tint, float alpha, float specularity, float ior, f3 wo, f3 wi) {
    float f = evaluate1(alpha, specularity, ior, wo, wi);
    bool is_transmission = sgn(wo.z) != sgn(wi.z);
    return f * (is_transmission ? tint : mk3(1));
}
HD float pdf(f3 tint, float alpha, float specularity, float ior, f3 wo, f3 wi) {
    if (ggx_smooth(alpha)) return pdf_invalid();
    if (!(wo.z >= 0.0f)) { wo.z = -wo.z; wi.z = -wi.z; }
    bool is_reflection = wo.z * wi.z >= 0.0f;
    f3 h = halfway(is_reflection ? 1.0f : ior, wo, wi);
    if (!is_reflection && (dot(wo, h) < 0.0f || dot(wi, h) >= 0.0f)) return pdf_invalid();
    float p = vndf::pdf(alpha, wo, h);
    float rp = dielectric_schlick_fresnel(specularity, dot(wo, h), ior);
    float nrp = normalize_reflection_probability(rp, tint);
    p *= is_reflection ? nrp : (1 - nrp);
    if (is_reflection) p *= 1 / (4.0f * dot(wo, h));
    else p *= transmission_PDF_scale(ior, wo, wi, h);
    return p;
}
HD Sample sample(f3 tint, float alpha, float specularity, float ior, f3 wo, f3 u) {
    Sample r;
    bool entering = wo.z >= 0.0f;
    if (!entering) wo.z = -wo.z;
    if (ggx_smooth(alpha)) {
        float rp = dielectric_schlick_fresnel(specularity, fabsf(wo.z), ior);
        float nrp = normalize_reflection_probability(rp, tint);
        bool is_reflection = u.z < nrp;
        if (is_reflection) { r.pdf = -nrp; r.dir = {-wo.x, -wo.y, wo.z}; }
        else {
            r.pdf = -(1.0f - nrp);
            if (!refract_z(r.dir, -wo, ior)) return sample_none();
        }
        r.f = mk3((is_reflection ? rp : (1.0f - rp)) / fabsf(r.dir.z));
    } else {
        f3 h = vndf::sample_halfway(alpha, wo, mk2(u.x, u.y));
        r.pdf = vndf::pdf(alpha, wo, h);
        float rp = dielectric_schlick_fresnel(specularity, dot(wo, h), ior);
        float nrp = normalize_reflection_probability(rp, tint);
        bool is_reflection = u.z < nrp;
        if (is_reflection) {
            r.dir = reflect(-wo, h);
            r.pdf *= nrp / (4.0f * dot(wo, h));
        } else {
            if (!refract(r.dir, -wo, h, ior)) return sample_none();
            r.pdf *= 1 - nrp;
            r.pdf *= transmission_PDF_scale(ior, wo, r.dir, h);
        }
        bool energyloss = is_reflection ? r.dir.z < 0.0f : r.dir.z >= 0.0f;
        if (energyloss) return sample_none();
        r.f = mk3(evaluate1(alpha, specularity, ior, wo, r.dir));
    }
    if (sgn(wo.z) != sgn(r.dir.z)) r.f *= tint;
    if (!entering) r.dir.z = -r.dir.z;
    return r;
}
} // namespace ggx_rt

// ---------------------------------------------------------------------------------------------
// Shading models. One struct holds the state of whichever model the material selects so the
// RIS loop and the BSDF sampling can be written once; `model` is wave-divergent only in scenes
// that mix shading models.
// ---------------------------------------------------------------------------------------------
struct MaterialInputs { f3 tint; float roughness, specularity, metallic, coat, coat_roughness; };

struct Shading {
    int model;
    // Diffuse: a = tint, s0 = roughness.
    // Default: a = diffuse tint, b = specularity, s0 = roughness, s1 = specular scale, s2 = coat scale, s3 = coat alpha, p0/p1 = specular/coat probability (u16).
    // Transmissive: a = transmission tint, s0 = specularity, s1 = alpha, s2 = ior_i_over_o, s3 = energy loss adjustment.
    f3 a, b;
    float s0, s1, s2, s3;
    uint32_t p0, p1;
};

HD float specular_properties(const DeviceTables& t, float roughness, float specularity, float scale, float abs_cos,
                             float& alpha, float& reflection_scale, float& transmission_scale) {
    alpha = ggx_alpha_from_roughness(roughness);
    f2 rho = fetch_specular_rho(t, abs_cos, roughness);
    reflection_scale = scale * (1.0f / rho.y);
    float specular_rho = lerp(rho.x, rho.y, specularity) * reflection_scale;
    transmission_scale = 1.0f - specular_rho;
    return specular_rho;
}

HD Shading make_diffuse(f3 tint, float roughness) {
    Shading s;
    s.model = HIPR_SHADING_DIFFUSE;
    s.a = tint;
    s.s0 = roughness;
    return s;
}

HD Shading make_default(const DeviceTables& t, const MaterialInputs& m, float cos_theta_o, float max_PDF_hint) {
    float min_roughness = min_roughness_from_PDF(t, cos_theta_o, max_PDF_hint);
    float coat_roughness = fmaxf(m.coat_roughness, min_roughness);
    float in_roughness = fmaxf(m.roughness, min_roughness);
    float abs_cos = fabsf(cos_theta_o);

    Shading s;
    s.model = HIPR_SHADING_DEFAULT;
    float roughness = in_roughness;
    float dielectric_spec = m.specularity;
    f3 conductor_spec = m.tint;
    if (m.coat > 0) {
        float modulated = modulate_roughness_under_coat(in_roughness, coat_roughness);
        roughness = lerp(in_roughness, modulated, m.coat);
        if (dielectric_spec < 1.0f) {
            float coated = adjust_dielectric_specularity(HIPR_COAT_IOR, dielectric_spec);
            dielectric_spec = lerp(dielectric_spec, coated, m.coat);
        }
        if (m.metallic > 0) {
            f3 coated = conductor_specularity(mk3(HIPR_COAT_IOR), conductor_ior_from_specularity(conductor_spec, mk3(0.0f)), mk3(0.0f));
            conductor_spec = lerp(conductor_spec, coated, m.coat);
            conductor_spec.x = (conductor_spec.x != conductor_spec.x) ? 1.0f : conductor_spec.x;
            conductor_spec.y = (conductor_spec.y != conductor_spec.y) ? 1.0f : conductor_spec.y;
            conductor_spec.z = (conductor_spec.z != conductor_spec.z) ? 1.0f : conductor_spec.z;
        }
    }
    float specular_alpha, specular_scale, dielectric_transmission;
    specular_properties(t, roughness, dielectric_spec, 1.0f, abs_cos, specular_alpha, specular_scale, dielectric_transmission);
    f3 dielectric_tint = m.tint * dielectric_transmission;
    f3 specularity = lerp(mk3(dielectric_spec), conductor_spec, m.metallic);
    f3 diffuse_tint = dielectric_tint * (1.0f - m.metallic);

    float coat_rho = 0, coat_scale = 0, coat_alpha = 0;
    if (m.coat > 0) {
        float coat_transmission;
        coat_rho = specular_properties(t, coat_roughness, HIPR_COAT_SPECULARITY, m.coat, abs_cos, coat_alpha, coat_scale, coat_transmission);
        specular_scale *= coat_transmission;
        diffuse_tint *= coat_transmission;
    }
    s.a = diffuse_tint;
    s.b = specularity;
    s.s0 = roughness;
    s.s1 = specular_scale;
    s.s2 = coat_scale;
    s.s3 = coat_alpha;

    // setup_sampling_probabilities: u16 quantised, rho fetched at cos_theta_o as passed in.
    f2 rho = fetch_specular_rho(t, cos_theta_o, roughness);
    f3 specular_rho = mk3(lerp(rho.x, rho.y, specularity.x), lerp(rho.x, rho.y, specularity.y), lerp(rho.x, rho.y, specularity.z)) * specular_scale;
    float recip_total = 1.0f / (sum(diffuse_tint) + sum(specular_rho) + 3 * coat_rho);
    s.p0 = (unsigned short)(sum(specular_rho) * recip_total * 65535.0f + 0.5f);
    s.p1 = (unsigned short)(3 * coat_rho * recip_total * 65535.0f + 0.5f);
    return s;
}

HD Shading make_transmissive(const DeviceTables& t, const MaterialInputs& m, float cos_theta_o, float max_PDF_hint) {
    float min_roughness = min_roughness_from_PDF(t, fabsf(cos_theta_o), max_PDF_hint);
    float roughness = fmaxf(m.roughness, min_roughness);
    Shading s;
    s.model = HIPR_SHADING_TRANSMISSIVE;
    s.a = m.tint;
    s.s0 = m.specularity;
    s.s1 = ggx_alpha_from_roughness(roughness);
    float medium_ior = dielectric_ior_from_specularity(m.specularity);
    bool entering = cos_theta_o >= 0.0f;
    float ior_o = entering ? HIPR_AIR_IOR : medium_ior;
    float ior_i = entering ? medium_ior : HIPR_AIR_IOR;
    s.s2 = ior_i / ior_o;
    s.s3 = 1.0f / fetch_dielectric_rho(t, fabsf(cos_theta_o), roughness, s.s2).x;
    return s;
}

// ---------------------------------------------------------------------------------------------
// The terms of the BSDFs above that depend on the outgoing direction and the material only. A hit evaluates its shading four times -- three light
// candidates and the BSDF sample's other lobes -- with the same wo: the terms are computed once per hit (round 4; the compiler hoisted them out of the
// candidate loop by itself but computed them again, up to three times, in the sampling code behind it). Same expressions in the same order as the
// functions they are taken from -- to the operation: a quotient stays a quotient by the same divisor (round 4 had multiplied by a stored reciprocal in two places,
// which the verification build of round 5 showed as last-ulp differences from the plain functions in 3-60 % of the evaluations; in the product's fast arithmetic
// x / d is x * v_rcp_f32(d) either way and the reciprocal of a per-hit divisor is hoisted by the compiler). tests/test_device_code_on_host_cpu.py holds both
// forms to the oracle bit for bit.
// ---------------------------------------------------------------------------------------------
// A divisor that is the same for every evaluation of a hit. In the product's arithmetic x / d IS x * v_rcp_f32(d) (-freciprocal-math), so the reciprocal is taken once
// per hit and the evaluations multiply -- instruction for instruction what the division compiles to there, minus the v_rcp_f32; in the correctly rounded builds
// (verification build, host build of the tests) the quotient stays a quotient by d, as the plain functions and the oracle write it.
#ifndef HIPR_RECIPROCAL_DIVISION
#define HIPR_RECIPROCAL_DIVISION HIPR_FAST_MATH      // the unit is built with -freciprocal-math: x / d is x * v_rcp_f32(d)
#endif
struct HitDivisor {
#if HIPR_RECIPROCAL_DIVISION
    float reciprocal;
#else
    float d;
#endif
};
HD HitDivisor hit_divisor(float d) {
#if HIPR_RECIPROCAL_DIVISION
    return {1.0f / d};
#else
    return {d};
#endif
}
HD float operator/(float x, HitDivisor h) {
#if HIPR_RECIPROCAL_DIVISION
    return x * h.reciprocal;
#else
    return x / h.d;
#endif
}

struct OrenNayarTerms {
    float roughness, B, pi_A;        // evaluate: single = pi_A * (1 + roughness * s_over_t)
    float m_o; HitDivisor den;       //           multi = m_o * |1 - EF_i| / den, m_o = (ms_rho / pi) * |1 - EF_o|, den = max(1e-7, 1 - avg_EF)
    float A;
    float up, cp;                    // uniform / CLTC mixture
    float Xx, Xy;                    // cltc::tangents: X; Y = (-X.y, X.x)
    float a, b, c, d, amb, det, s;   // cltc::coefficients, amb = a - b d, det = c amb, s = (1 + 1 / sqrt(d^2 + 1)) / 2
};
HD OrenNayarTerms oren_nayar_terms(float roughness, f3 wo) {
    OrenNayarTerms t;
    const float c2 = 2.0f / 3.0f - 28.0f / (15.0f * HIPR_PI);
    t.roughness = roughness;
    t.A = 1.0f / (1.0f + HIPR_FON_C1 * roughness);
    t.B = roughness * t.A;
    t.pi_A = HIPR_RECIP_PI * t.A;
    const float EF_o = oren_nayar::E_FON_approx(wo.z, t.A, t.B);
    const float avg_EF = t.A * (1.0f + c2 * roughness);
    const float ms_rho = avg_EF / (1.0f - (1.0f - avg_EF));
    t.m_o = (ms_rho * HIPR_RECIP_PI) * fabsf(1.0f - EF_o);
    t.den = hit_divisor(fmaxf(1.0e-7f, 1.0f - avg_EF));
    t.up = oren_nayar::uniform_probability(roughness, wo.z);
    t.cp = 1.0f - t.up;
    f2 X, Y;
    cltc::tangents(wo, X, Y);
    t.Xx = X.x; t.Xy = X.y;
    cltc::coefficients(wo.z, roughness, t.a, t.b, t.c, t.d);
    t.amb = t.a - t.b * t.d;
    t.det = t.c * t.amb;
    const float vz = 1.0f / sqrtf(t.d * t.d + 1.0f);
    t.s = 0.5f * (1.0f + vz);
    return t;
}
namespace oren_nayar {
HD float evaluate(const OrenNayarTerms& t, f3 wo, f3 wi) {
    const float ci = wi.z, co = wo.z;
    const float s = dot(wi, wo) - ci * co;
    const float s_over_t = s > 0.0f ? s / fmaxf(ci, co) : s;
    const float single = t.pi_A * (1.0f + t.roughness * s_over_t);
    const float EF_i = E_FON_approx(ci, t.A, t.B);
    const float multi = t.m_o * fabsf(1.0f - EF_i) / t.den;
    return single + multi;
}
HD float cltc_pdf(const OrenNayarTerms& t, f3 wi_shading) {       // cltc::pdf
    const f3 wi = {t.Xx * wi_shading.x + t.Xy * wi_shading.y, -t.Xy * wi_shading.x + t.Xx * wi_shading.y, wi_shading.z};
    const f3 wh = {t.c * (wi.x - t.b * wi.z), t.amb * wi.y, -t.c * (t.d * wi.x - t.a * wi.z)};
    const float wh2 = dot(wh, wh);
    return t.det * t.det / pow2(wh2) * fmaxf(wh.z, 0.0f) / (HIPR_PI * t.s);
}
HD Response evaluate_with_PDF(f3 albedo, const OrenNayarTerms& t, f3 wo, f3 wi) {
    return {albedo * evaluate(t, wo, wi), t.up * (0.5f * HIPR_RECIP_PI) + t.cp * cltc_pdf(t, wi)};
}
// The direction oren_nayar::sample draws and the CLTC density sample() reports for it (the uniform branch evaluates cltc::pdf, the CLTC branch its own expression).
HD f3 sample_direction(const OrenNayarTerms& t, f3 wo, f2 u, float& cltc_density) {
    if (u.x <= t.up) {
        u.x = u.x / t.up;
        const float z = u.x;
        const float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        float s, c;
        sincos_(HIPR_TWO_PI * u.y, s, c);
        const f3 dir = {r * c, r * s, z};
        cltc_density = cltc_pdf(t, dir);
        return dir;
    }
    u.x = (u.x - t.up) / t.cp;
    const float radius = sqrtf(u.x);
    float sp, cp;
    sincos_(2.0f * HIPR_PI * u.y, sp, cp);
    float x = radius * cp;
    const float y = radius * sp;
    x = -lerp(sqrtf(1.0f - y * y), x, t.s);
    const f3 wh = {x, y, sqrtf(fmaxf(1.0f - (x * x + y * y), 0.0f))};
    const float pdf_wh = wh.z / (HIPR_PI * t.s);
    const f3 wi = {t.a * wh.x + t.b * wh.z, t.c * wh.y, t.d * wh.x + wh.z};
    const float m = length(wi);
    cltc_density = pdf_wh * m * m * m / t.det;
    const f3 local = {t.Xx * wi.x - t.Xy * wi.y, t.Xy * wi.x + t.Xx * wi.y, wi.z};
    return normalize(local);
}
} // namespace oren_nayar

struct GGXTerms { float alpha, recip_G1, pdf_numerator; HitDivisor pdf_divisor; };     // recip_G1 = 1 + lambda(alpha, wo); bounded_reflection_pdf = D(h) * pdf_numerator / pdf_divisor
HD GGXTerms ggx_terms(float alpha, f3 wo) {
    GGXTerms t;
    t.alpha = alpha;
    t.recip_G1 = 1.0f + vndf::lambda(alpha, wo);
    const f2 ao = alpha * mk2(wo.x, wo.y);
    const float len2 = dot(ao, ao);
    const float tt = sqrtf(len2 + wo.z * wo.z);
    if (wo.z >= 0.0f) {
        const float s = 1.0f + length(mk2(wo.x, wo.y));
        const float a2 = alpha * alpha, s2 = s * s;
        const float k = (1.0f - a2) * s2 / (s2 + a2 * wo.z * wo.z);
        t.pdf_numerator = 1.0f;      // ndf / (2 (k wo.z + t)); ndf * 1 is ndf
        t.pdf_divisor = hit_divisor(2.0f * (k * wo.z + tt));
    } else {
        t.pdf_numerator = tt - wo.z; // ndf * (t - wo.z) / (2 len2)
        t.pdf_divisor = hit_divisor(2.0f * len2);
    }
    return t;
}
namespace ggx_r {
HD Response evaluate_with_PDF(const GGXTerms& t, f3 specularity, f3 wo, f3 wi) {
    if (ggx_smooth(t.alpha)) return {mk3(0.0f), pdf_invalid()};
    const f3 h = normalize(wo + wi);
    const float D = vndf::D(t.alpha, h);
    Response r;
    r.pdf = D * t.pdf_numerator / t.pdf_divisor;
    if (wo.z * wi.z <= 0.0f) r.f = mk3(0.0f);
    else {
        const float G = 1.0f / (t.recip_G1 + vndf::lambda(t.alpha, wi));
        const f3 F = schlick_fresnel3(specularity, dot(wo, h));
        r.f = F * (D * G / (4.0f * wo.z * wi.z));
    }
    return r;
}
} // namespace ggx_r

// MODELS: bit i set = shading model i occurs in the scene. Kernels are instantiated per mask so a scene that only
// uses one model (the common case) carries neither the registers nor the branches of the others.
#define HIPR_HAS_DEFAULT(M) (((M) & 1) != 0)
#define HIPR_HAS_DIFFUSE(M) (((M) & 2) != 0)
#define HIPR_HAS_TRANSMISSIVE(M) (((M) & 4) != 0)

template <int MODELS>
HD Response shading_evaluate_with_PDF(const Shading& s, f3 wo, f3 wi) {
    if (HIPR_HAS_DIFFUSE(MODELS) && (MODELS == 2 || s.model == HIPR_SHADING_DIFFUSE)) {
        if (wo.z < 0.000001f || wi.z < 0.000001f) return response_none();
        return oren_nayar::evaluate_with_PDF(s.a, s.s0, wo, wi);
    }
    if (HIPR_HAS_TRANSMISSIVE(MODELS) && (MODELS == 4 || s.model == HIPR_SHADING_TRANSMISSIVE)) {
        if (wo.z < 0.000001f) return response_none();
        Response r = {ggx_rt::evaluate(s.a, s.s1, s.s0, s.s2, wo, wi), ggx_rt::pdf(s.a, s.s1, s.s0, s.s2, wo, wi)};
        r.f *= s.s3;
        return r;
    }
    if (!HIPR_HAS_DEFAULT(MODELS)) return response_none();
    if (wo.z < 0.000001f || wi.z < 0.000001f) return response_none();
    float sp = s.p0 / 65535.0f, cp = s.p1 / 65535.0f;
    float dp = 1.0f - (s.p0 + s.p1) / 65535.0f;
    float alpha = ggx_alpha_from_roughness(s.s0);
    Response d = oren_nayar::evaluate_with_PDF(s.a, s.s0, wo, wi);
    Response g = ggx_r::evaluate_with_PDF(alpha, s.b, wo, wi);
    g.f *= s.s1;
    Response r;
    r.f = d.f + g.f;
    r.pdf = d.pdf * dp + g.pdf * sp;
    if (s.s2 > 0) {
        Response c = ggx_r::evaluate_with_PDF(s.s3, mk3(HIPR_COAT_SPECULARITY), wo, wi);
        r.f += s.s2 * c.f;
        r.pdf += c.pdf * cp;
    }
    return r;
}

template <int MODELS>
HD Sample shading_sample(const Shading& s, f3 wo, f3 u) {
    if (wo.z < 0.000001f) return sample_none();
    if (HIPR_HAS_DIFFUSE(MODELS) && (MODELS == 2 || s.model == HIPR_SHADING_DIFFUSE))
        return oren_nayar::sample(s.a, s.s0, wo, mk2(u.x, u.y));
    if (HIPR_HAS_TRANSMISSIVE(MODELS) && (MODELS == 4 || s.model == HIPR_SHADING_TRANSMISSIVE)) {
        Sample r = ggx_rt::sample(s.a, s.s1, s.s0, s.s2, wo, u);
        r.f *= s.s3;
        return r;
    }
    if (!HIPR_HAS_DEFAULT(MODELS)) return sample_none();
    float sp = s.p0 / 65535.0f, cp = s.p1 / 65535.0f;
    float dp = 1 - cp - sp;
    float alpha = ggx_alpha_from_roughness(s.s0);
    bool sample_coat = u.z < cp;
    bool sample_specular = !sample_coat && u.z < (cp + sp);
    bool sample_diffuse = !sample_coat && !sample_specular;
    f2 u2 = {u.x, u.y};
    Sample r;
    if (sample_diffuse) {
        r = oren_nayar::sample(s.a, s.s0, wo, u2);
        r.pdf *= dp;
    } else if (sample_specular) {
        r = ggx_r::sample(alpha, s.b, wo, u2);
        r.f *= s.s1;
        r.pdf *= sp;
    } else {
        r = ggx_r::sample(s.s3, mk3(HIPR_COAT_SPECULARITY), wo, u2);
        r.f *= s.s2;
        r.pdf *= cp;
    }
    if (!pdf_valid_not_delta(r.pdf))
        return r;
    if (!sample_diffuse) {
        Response d = oren_nayar::evaluate_with_PDF(s.a, s.s0, wo, r.dir);
        if (pdf_valid_not_delta(d.pdf)) { r.f += d.f; r.pdf += d.pdf * dp; }
    }
    if (!sample_specular) {
        Response g = ggx_r::evaluate_with_PDF(alpha, s.b, wo, r.dir);
        if (pdf_valid_not_delta(g.pdf)) { r.f += g.f * s.s1; r.pdf += g.pdf * sp; }
    }
    if (!sample_coat && s.s2 > 0) {
        Response c = ggx_r::evaluate_with_PDF(s.s3, mk3(HIPR_COAT_SPECULARITY), wo, r.dir);
        if (pdf_valid_not_delta(c.pdf)) { r.f += s.s2 * c.f; r.pdf += c.pdf * cp; }
    }
    return r;
}

// The outgoing-direction terms of a hit's shading (Diffuse: the Oren-Nayar ones; Default: those and the two GGX lobes'; Transmissive: none).
struct ShadingTerms { OrenNayarTerms diffuse; GGXTerms specular, coat; };
template <int MODELS>
HD ShadingTerms shading_terms(const Shading& s, f3 wo) {
    ShadingTerms t = {};
    const bool transmissive = HIPR_HAS_TRANSMISSIVE(MODELS) && (MODELS == 4 || s.model == HIPR_SHADING_TRANSMISSIVE);
    if (MODELS == 4 || transmissive) return t;
    t.diffuse = oren_nayar_terms(s.s0, wo);
    if (HIPR_HAS_DEFAULT(MODELS) && (MODELS == 1 || s.model == HIPR_SHADING_DEFAULT)) {
        t.specular = ggx_terms(ggx_alpha_from_roughness(s.s0), wo);
        t.coat = ggx_terms(s.s3, wo);
    }
    return t;
}

template <int MODELS>
HD Response shading_evaluate_with_PDF(const Shading& s, const ShadingTerms& t, f3 wo, f3 wi) {
    if (HIPR_HAS_DIFFUSE(MODELS) && (MODELS == 2 || s.model == HIPR_SHADING_DIFFUSE)) {
        if (wo.z < 0.000001f || wi.z < 0.000001f) return response_none();
        return oren_nayar::evaluate_with_PDF(s.a, t.diffuse, wo, wi);
    }
    if (HIPR_HAS_TRANSMISSIVE(MODELS) && (MODELS == 4 || s.model == HIPR_SHADING_TRANSMISSIVE)) return shading_evaluate_with_PDF<MODELS>(s, wo, wi);
    if (!HIPR_HAS_DEFAULT(MODELS)) return response_none();
    if (wo.z < 0.000001f || wi.z < 0.000001f) return response_none();
    float sp = s.p0 / 65535.0f, cp = s.p1 / 65535.0f;
    float dp = 1.0f - (s.p0 + s.p1) / 65535.0f;
    Response d = oren_nayar::evaluate_with_PDF(s.a, t.diffuse, wo, wi);
    Response g = ggx_r::evaluate_with_PDF(t.specular, s.b, wo, wi);
    g.f *= s.s1;
    Response r;
    r.f = d.f + g.f;
    r.pdf = d.pdf * dp + g.pdf * sp;
    if (s.s2 > 0) {
        Response c = ggx_r::evaluate_with_PDF(t.coat, mk3(HIPR_COAT_SPECULARITY), wo, wi);
        r.f += s.s2 * c.f;
        r.pdf += c.pdf * cp;
    }
    return r;
}

// shading_sample with the lobes' evaluations written once: the chosen lobe only draws the DIRECTION (the two GGX lobes through one call), then every lobe is
// evaluated for it by the code the light candidates use, and the sum is formed in the order the reference forms it (DefaultShading.h sample: the sampled
// lobe's own response first, then diffuse, specular, coat as far as they were not the one sampled). A wave whose lanes picked different lobes used to run
// each lobe's evaluation twice.
template <int MODELS>
HD Sample shading_sample(const Shading& s, const ShadingTerms& t, f3 wo, f3 u) {
    if (wo.z < 0.000001f) return sample_none();
    if (HIPR_HAS_DIFFUSE(MODELS) && (MODELS == 2 || s.model == HIPR_SHADING_DIFFUSE)) {
        Sample r;
        float cltc_density;
        r.dir = oren_nayar::sample_direction(t.diffuse, wo, mk2(u.x, u.y), cltc_density);
        r.pdf = t.diffuse.up * (0.5f * HIPR_RECIP_PI) + t.diffuse.cp * cltc_density;
        r.f = s.a * oren_nayar::evaluate(t.diffuse, wo, r.dir);
        return r;
    }
    if (HIPR_HAS_TRANSMISSIVE(MODELS) && (MODELS == 4 || s.model == HIPR_SHADING_TRANSMISSIVE)) return shading_sample<MODELS>(s, wo, u);
    if (!HIPR_HAS_DEFAULT(MODELS)) return sample_none();
    float sp = s.p0 / 65535.0f, cp = s.p1 / 65535.0f;
    float dp = 1 - cp - sp;
    bool sample_coat = u.z < cp;
    bool sample_specular = !sample_coat && u.z < (cp + sp);
    bool sample_diffuse = !sample_coat && !sample_specular;
    f2 u2 = {u.x, u.y};
    Sample r;
    float cltc_density = 0.0f;
    if (sample_diffuse) r.dir = oren_nayar::sample_direction(t.diffuse, wo, u2, cltc_density);
    else {
        const float alpha = sample_specular ? t.specular.alpha : t.coat.alpha;
        const float scale = sample_specular ? s.s1 : s.s2, probability = sample_specular ? sp : cp;
        if (ggx_smooth(alpha)) {        // ggx_r::sample's mirror reflection: a delta PDF, returned as it is
            r.dir = {-wo.x, -wo.y, wo.z};
            r.pdf = -1.0f * probability;
            r.f = schlick_fresnel3(sample_specular ? s.b : mk3(HIPR_COAT_SPECULARITY), fabsf(wo.z)) / fabsf(r.dir.z) * scale;
            return r;
        }
        r.dir = vndf::bounded_sample_reflection(alpha, wo, u2);
        if (r.dir.z < 0.0f) return sample_none();
    }
    Response d = oren_nayar::evaluate_with_PDF(s.a, t.diffuse, wo, r.dir);
    if (sample_diffuse) d.pdf = t.diffuse.up * (0.5f * HIPR_RECIP_PI) + t.diffuse.cp * cltc_density;
    Response g = ggx_r::evaluate_with_PDF(t.specular, s.b, wo, r.dir);
    g.f *= s.s1;
    Response c = response_none();
    if (s.s2 > 0) {
        c = ggx_r::evaluate_with_PDF(t.coat, mk3(HIPR_COAT_SPECULARITY), wo, r.dir);
        c.f *= s.s2;
    }
    r.f = sample_diffuse ? d.f : (sample_specular ? g.f : c.f);
    r.pdf = sample_diffuse ? d.pdf * dp : (sample_specular ? g.pdf * sp : c.pdf * cp);
    if (!pdf_valid_not_delta(r.pdf))
        return r;
    if (!sample_diffuse && pdf_valid_not_delta(d.pdf)) { r.f += d.f; r.pdf += d.pdf * dp; }
    if (!sample_specular && pdf_valid_not_delta(g.pdf)) { r.f += g.f; r.pdf += g.pdf * sp; }
    if (!sample_coat && s.s2 > 0 && pdf_valid_not_delta(c.pdf)) { r.f += c.f; r.pdf += c.pdf * cp; }
    return r;
}

// ---------------------------------------------------------------------------------------------
// TBN, MIS, analytic intersections, lights
// ---------------------------------------------------------------------------------------------
struct Frame { f3 t, b, n; };
HD Frame make_frame(f3 n) {
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    return {{1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x}, {b, sign + n.y * n.y * a, -n.y}, n};
}
HD f3 to_local(const Frame& f, f3 v) { return {dot(f.t, v), dot(f.b, v), dot(f.n, v)}; }
HD f3 to_world(const Frame& f, f3 v) { return v.x * f.t + v.y * f.b + v.z * f.n; }

HD float balance_heuristic(float p1, float p2) {
    float divisor = p1 + p2;
    float result = p1 / divisor;
    bool invalid = __builtin_isinf(divisor) || (result != result);
    return invalid ? (p1 <= p2 ? 0.0f : 1.0f) : result;
}

HD float ray_sphere(f3 o, f3 d, f3 center, float radius) {
    f3 to = o - center;
    float b = dot(to, d);
    f3 fbd = to - b * d;
    float disc = radius * radius - dot(fbd, fbd);
    return disc > 0.0f ? -b - sqrtf(disc) : __builtin_nanf("");
}
HD float ray_plane(f3 o, f3 d, f3 p, f3 n) { return (dot(n, p) - dot(n, o)) / dot(n, d); }
HD float ray_disk(f3 o, f3 d, f3 center, f3 normal, float radius) {
    float t = ray_plane(o, d, center, normal);
    f3 v = (o + d * t) - center;
    return (dot(v, v) <= radius * radius && t >= 0.0f) ? t : __builtin_nanf("");
}

struct LightSample { f3 radiance; float pdf; f3 dir; float distance; };
HD LightSample light_sample_none() { return {{0, 0, 0}, -0.0f, {0, 1, 0}, 0.0f}; }

HD f3 L3(const HiprLight& l, int i) { return {l.data[i], l.data[i + 1], l.data[i + 2]}; }

HD float spot_pdf(const HiprLight& l, f3 lit, f3 dir) {
    f3 ldir = L3(l, 7), lpos = L3(l, 3);
    float radius = l.data[6], cos_angle = l.data[10];
    float cos_theta = -dot(ldir, dir);
    if (cos_theta > 0.0f && radius != 0.0f) {
        float t = ray_plane(lit, -ldir, lpos, ldir);
        float cone_radius = t * sqrtf(1.0f - pow2(cos_angle)) / cos_angle;
        if (radius > cone_radius && cos_angle > 1e-5f)
            return 1.0f / (2.0f * HIPR_PI * (1.0f - cos_angle));
        float td = ray_disk(lit, dir, lpos, ldir, radius);
        if (td >= 0.0f)
            return (1.0f / (HIPR_PI * pow2(radius))) * ((td * td) / cos_theta);
    }
    return -0.0f;
}
HD f3 spot_evaluate(const HiprLight& l, f3 lit, f3 dir) {
    f3 ldir = L3(l, 7), lpos = L3(l, 3);
    float radius = l.data[6], cos_angle = l.data[10];
    float cos_theta = -dot(ldir, dir);
    float normalization = HIPR_TWO_PI * (1 - cos_angle);
    if (radius == 0.0f) { f3 d = lpos - lit; normalization *= dot(d, d); }
    else normalization *= (HIPR_PI * pow2(radius)) * cos_theta;
    f3 radiance = L3(l, 0) / normalization;
    return cos_theta > cos_angle ? radiance : mk3(0.0f);
}

HD LightSample light_sample_radiance(const HiprLight& l, f3 p, f2 u) {
    uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
    LightSample s;
    if (type == HIPR_LIGHT_SPHERE) {
        f3 power = L3(l, 0), pos = L3(l, 3);
        float radius = l.data[6];
        f3 to_light = pos - p;
        float sin2 = radius * radius / dot(to_light, to_light);
        if (sin2 <= 0.0f) {
            s.dir = to_light;
            s.distance = length(s.dir);
            s.dir /= s.distance;
            s.radiance = power / (4.0f * HIPR_PI * s.distance * s.distance);
            s.distance -= radius;
            s.pdf = -1.0f;
        } else {
            float cos_theta = sqrtf(1.0f - sin2);
            float cone_pdf;
            f3 cone = cone_sample(cos_theta, u, cone_pdf);
            Frame f = make_frame(normalize(to_light));
            s.dir = to_world(f, cone);
            s.pdf = cone_pdf;
            s.distance = ray_sphere(p, s.dir, pos, radius);
            if (s.distance <= 0.0f)
                s.distance = dot(to_light, s.dir);
            s.radiance = power * (1.0f / (HIPR_PI * (4.0f * HIPR_PI * radius * radius)));
        }
        s.distance = nextafterf(s.distance, 0.0f);
        return s;
    }
    if (type == HIPR_LIGHT_DIRECTIONAL)
        return {L3(l, 0), -1.0f, -L3(l, 3), 1e30f};
    if (type == HIPR_LIGHT_SPOT) {
        f3 lpos = L3(l, 3), ldir = L3(l, 7);
        float radius = l.data[6], cos_angle = l.data[10];
        if (radius == 0.0f) {
            s.dir = lpos - p;
            s.distance = length(s.dir);
            s.dir /= s.distance;
            s.pdf = 1.0f;
            s.radiance = spot_evaluate(l, p, s.dir);
            return s;
        }
        Frame f = make_frame(ldir);
        float t = ray_plane(p, -ldir, lpos, ldir);
        float cone_radius = t * sqrtf(1.0f - pow2(cos_angle)) / cos_angle;
        if (radius > cone_radius && cos_angle > 1e-5f) {
            float cone_pdf;
            f3 cone = cone_sample(cos_angle, u, cone_pdf);
            s.dir = to_world(f, -cone);
            s.distance = ray_plane(p, s.dir, lpos, ldir);
            s.pdf = cone_pdf;
            s.radiance = mk3(0.0f);
            f3 d = (p + s.dir * s.distance) - lpos;
            if (dot(d, d) < pow2(radius))
                s.radiance = spot_evaluate(l, p, s.dir);
        } else {
            float r = sqrtf(u.x) * radius;
            float phi = 2.0f * HIPR_PI * u.y;
            float sphi, cphi;
            sincos_(phi, sphi, cphi);
            f3 sampled = lpos + to_world(f, mk3(r * cphi, r * sphi, 0.0f));
            s.dir = sampled - p;
            s.distance = length(s.dir);
            s.dir /= s.distance;
            float cos_theta = -dot(ldir, s.dir);
            s.pdf = (1.0f / (HIPR_PI * pow2(radius))) * (pow2(s.distance) / cos_theta);
            s.radiance = spot_evaluate(l, p, s.dir);
        }
        s.distance = nextafterf(s.distance, 0.0f);
        return s;
    }
    return light_sample_none();
}

// evaluate_intersection (OR/Shading/LightSources/LightImpl.h:85-108) for the light a path ray hit.
HD f3 light_evaluate_intersection(const HiprLight& l, f3 origin, f3 direction, float bsdf_PDF) {
    uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
    f3 radiance;
    float light_pdf;
    if (type == HIPR_LIGHT_SPHERE) {
        f3 power = L3(l, 0), pos = L3(l, 3);
        float radius = l.data[6];
        f3 to_center = pos - origin;
        float sin2 = radius * radius / dot(to_center, to_center);
        bool delta = sin2 <= 0.0f;
        radiance = power * (1.0f / (delta ? (4.0f * HIPR_PI) : (HIPR_PI * (4.0f * HIPR_PI * radius * radius))));
        if (sin2 < 0.0f) light_pdf = -0.0f;
        else {
            float cos_max = sqrtf(1.0f - sin2);
            float cos_theta = dot(direction, normalize(to_center));
            light_pdf = (1.0f / (2.0f * HIPR_PI * (1.0f - cos_max))) * (cos_theta >= cos_max ? 1.0f : 0.0f);
        }
    } else if (type == HIPR_LIGHT_SPOT) {
        radiance = spot_evaluate(l, origin, direction);
        light_pdf = spot_pdf(l, origin, direction);
    } else
        return mk3(1000.0f, 0, 1000);
    if (pdf_valid_not_delta(bsdf_PDF))
        radiance *= balance_heuristic(pdf_value(bsdf_PDF), pdf_value(light_pdf));
    return radiance;
}

} // namespace hipr
