// oracle/oracle_api.cpp -- extern "C" surface of the CPU oracle for ctypes (tests/, smoke(), bench cpu_baseline).
// TEST INFRASTRUCTURE ONLY (see oracle/vecmath.h). The product never links this.
#include "integrator.h"

#include <cstdio>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

static inline float3 f3(const float* p) { return {p[0], p[1], p[2]}; }
static inline void put3(float* o, float3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

enum BsdfModel {
    MODEL_OREN_NAYAR = 0,        // params: albedo[3], roughness, exact
    MODEL_GGX_R = 1,             // params: alpha, specularity[3]
    MODEL_GGX_T = 2,             // params: alpha, ior_i_over_o
    MODEL_GGX = 3,               // params: alpha, specularity, ior_i_over_o, tint[3]
    MODEL_DEFAULT_SHADING = 4,   // params: tint[3], roughness, specularity, metallic, coat, coat_roughness, cos_theta_o (NaN: wo.z), max_PDF_hint (NaN: none)
    MODEL_TRANSMISSIVE_SHADING = 5, // same parameter block
    MODEL_DIFFUSE_SHADING = 6,   // params: tint[3], roughness
};

static MaterialInputs inputs_of(const float* p) { return {{p[0], p[1], p[2]}, p[3], p[4], p[5], p[6], p[7]}; }

extern "C" {

int oracle_wide_stack_high_water(int reset) { return oracle::wide_stack_high_water(reset != 0); }
int oracle_wide8_stack_high_water(int reset) { return oracle::wide8_stack_high_water(reset != 0); }
void oracle_set_backface_culling(int enable) { oracle::set_backface_culling(enable != 0); }
int oracle_backface_culling() { return oracle::backface_culling() ? 1 : 0; }


void oracle_set_tables(const float* base, const float* full, const float* light, const float* dense, const float* alphas, int quantize_unorm16) {
    tables().set(base, full, light, dense, alphas, quantize_unorm16 != 0);
}

// ------------------------------------------------------------------------------------------- RNG
void oracle_pcg2d(uint32_t x, uint32_t y, uint32_t* out2) { uint2 r = rng::pcg2d(x, y); out2[0] = r.x; out2[1] = r.y; }
void oracle_sobol4ui(const uint32_t* acc_hash_dim, uint32_t n, uint32_t* out4) {
    for (uint32_t i = 0; i < n; ++i) {
        uint4 s = rng::sample4ui(acc_hash_dim[3 * i], acc_hash_dim[3 * i + 1], acc_hash_dim[3 * i + 2]);
        out4[4 * i] = s.x; out4[4 * i + 1] = s.y; out4[4 * i + 2] = s.z; out4[4 * i + 3] = s.w;
    }
}
void oracle_sobol4f(uint32_t accumulation, uint32_t pixel_hash, uint32_t dimension, float* out4) {
    float4 s = rng::sample4f(accumulation, pixel_hash, dimension);
    out4[0] = s.x; out4[1] = s.y; out4[2] = s.z; out4[3] = s.w;
}
void oracle_sample_offsets(float* out, int count) {
    for (int i = 0; i < count; ++i) { float4 s = rng::sample_offset(i); out[4 * i] = s.x; out[4 * i + 1] = s.y; out[4 * i + 2] = s.z; out[4 * i + 3] = s.w; }
}
void oracle_sample02(uint32_t n, float* out2) { float2 s = rng::sample02(n); out2[0] = s.x; out2[1] = s.y; }
uint32_t oracle_reverse_bits(uint32_t v) { return rng::reverse_bits(v); }
uint32_t oracle_jenkins_hash(uint32_t v) { return rng::jenkins_hash(v); }

// ------------------------------------------------------------------------------------ BSDF stack
// out7 per sample: reflectance[3], pdf (raw, sign carries delta-dirac), direction[3]
void oracle_bsdf_sample(int model, const float* p, const float* wo_n3, const float* u_n3, int n, float* out_n7) {
    for (int i = 0; i < n; ++i) {
        float3 wo = f3(wo_n3 + 3 * i), u = f3(u_n3 + 3 * i);
        BSDFSample s = BSDFSample::none();
        switch (model) {
        case MODEL_OREN_NAYAR: s = OrenNayar::sample(f3(p), p[3], wo, make_float2(u), p[4] != 0.0f); break;
        case MODEL_GGX_R: s = GGX_R::sample(p[0], f3(p + 1), wo, make_float2(u)); break;
        case MODEL_GGX_T: s = GGX_T::sample(p[0], p[1], wo, make_float2(u)); break;
        case MODEL_GGX: s = GGX_RT::sample(f3(p + 3), p[0], p[1], p[2], wo, u); break;
        case MODEL_DEFAULT_SHADING: {
            float cos_theta = std::isnan(p[8]) ? wo.z : p[8];
            DefaultShading m = std::isnan(p[9]) ? DefaultShading(inputs_of(p), cos_theta) : DefaultShading::with_max_PDF_hint(inputs_of(p), cos_theta, PDF(p[9]));
            s = m.sample(wo, u);
            break;
        }
        case MODEL_TRANSMISSIVE_SHADING: {
            float cos_theta = std::isnan(p[8]) ? wo.z : p[8];
            TransmissiveShading m = std::isnan(p[9]) ? TransmissiveShading(inputs_of(p), cos_theta) : TransmissiveShading::with_max_PDF_hint(inputs_of(p), cos_theta, PDF(p[9]));
            s = m.sample(wo, u);
            break;
        }
        case MODEL_DIFFUSE_SHADING: { DiffuseShading m = {f3(p), p[3]}; s = m.sample(wo, u); break; }
        }
        put3(out_n7 + 7 * i, s.reflectance);
        out_n7[7 * i + 3] = s.pdf.v;
        put3(out_n7 + 7 * i + 4, s.direction);
    }
}

// out4 per pair: reflectance[3], pdf (raw). which: 0 evaluate_with_PDF, 1 evaluate only, 2 pdf only (BSDF-level models)
void oracle_bsdf_eval(int model, const float* p, const float* wo_n3, const float* wi_n3, int n, int which, float* out_n4) {
    for (int i = 0; i < n; ++i) {
        float3 wo = f3(wo_n3 + 3 * i), wi = f3(wi_n3 + 3 * i);
        BSDFResponse r = BSDFResponse::none();
        switch (model) {
        case MODEL_OREN_NAYAR:
            if (which == 0) r = OrenNayar::evaluate_with_PDF(f3(p), p[3], wo, wi, p[4] != 0.0f);
            else if (which == 1) r.reflectance = OrenNayar::evaluate(f3(p), p[3], wo, wi, p[4] != 0.0f);
            else r.pdf = OrenNayar::pdf(p[3], wo, wi);
            break;
        case MODEL_GGX_R:
            if (which == 0) r = GGX_R::evaluate_with_PDF(p[0], f3(p + 1), wo, wi);
            else if (which == 1) r.reflectance = GGX_R::evaluate(p[0], f3(p + 1), wo, wi);
            else r.pdf = GGX_R::pdf(p[0], wo, wi);
            break;
        case MODEL_GGX_T:
            if (which != 2) r.reflectance = make_float3(GGX_T::evaluate(p[0], p[1], wo, wi));
            if (which != 1) r.pdf = GGX_T::pdf(p[0], p[1], wo, wi);
            break;
        case MODEL_GGX:
            if (which != 2) r.reflectance = GGX_RT::evaluate(f3(p + 3), p[0], p[1], p[2], wo, wi);
            if (which != 1) r.pdf = GGX_RT::pdf(f3(p + 3), p[0], p[1], p[2], wo, wi);
            break;
        case MODEL_DEFAULT_SHADING: {
            float cos_theta = std::isnan(p[8]) ? wo.z : p[8];
            DefaultShading m = std::isnan(p[9]) ? DefaultShading(inputs_of(p), cos_theta) : DefaultShading::with_max_PDF_hint(inputs_of(p), cos_theta, PDF(p[9]));
            r = m.evaluate_with_PDF(wo, wi);
            break;
        }
        case MODEL_TRANSMISSIVE_SHADING: {
            float cos_theta = std::isnan(p[8]) ? wo.z : p[8];
            TransmissiveShading m = std::isnan(p[9]) ? TransmissiveShading(inputs_of(p), cos_theta) : TransmissiveShading::with_max_PDF_hint(inputs_of(p), cos_theta, PDF(p[9]));
            r = m.evaluate_with_PDF(wo, wi);
            break;
        }
        case MODEL_DIFFUSE_SHADING: { DiffuseShading m = {f3(p), p[3]}; r = m.evaluate_with_PDF(wo, wi); break; }
        }
        put3(out_n4 + 4 * i, r.reflectance);
        out_n4[4 * i + 3] = r.pdf.v;
    }
}

// Shading model introspection: out = rho[3], diffuse/specular/coat probability, roughness, specularity[3]
void oracle_default_shading_info(const float* p, float cos_theta_o, float* out10) {
    DefaultShading m(inputs_of(p), cos_theta_o);
    put3(out10, m.rho(fabsf(cos_theta_o)));
    out10[3] = m.get_diffuse_probability();
    out10[4] = m.get_specular_probability();
    out10[5] = m.get_coat_probability();
    out10[6] = m.roughness;
    put3(out10 + 7, m.specularity);
}
void oracle_transmissive_rho(const float* p, float cos_theta_o, float* out3) {
    TransmissiveShading m(inputs_of(p), cos_theta_o);
    put3(out3, m.rho(fabsf(cos_theta_o)));
}
void oracle_thin_sheet(float abs_cos_theta, float roughness, float ior_i_over_o, const float* tint3, float* out6) {
    ThinSheetThroughput t = approx_thin_sheet_reflectance(abs_cos_theta, roughness, ior_i_over_o, f3(tint3));
    put3(out6, t.reflected);
    put3(out6 + 3, t.transmitted);
}

// integrate_over_thin_sheet of ORT/BSDFTestUtils.h:167-225 with the GGX sampler of ORT/ShadingModels/UtilsTest.h:186-191:
// paths bounce between the two faces of a sheet; rng = PracticalScrambledSobol::sample4f(path, 0, bounce).
void oracle_integrate_thin_sheet(const float* tint_per_side3, float alpha, float specularity, float medium_ior, const float* wo3,
                                 unsigned path_count, unsigned bounce_count, float* out6) {
    float3 tint = f3(tint_per_side3), wo = f3(wo3);
    double refl[3] = {0, 0, 0}, trans[3] = {0, 0, 0};
    for (unsigned i = 0; i < path_count; ++i) {
        float3 throughput = {1, 1, 1};
        float3 ray_wo = wo;
        bool terminate = false, escaped_is_reflection = false;
        for (unsigned bounce = 0; bounce < bounce_count && !terminate; ++bounce) {
            float hemisphere_sign = bounce == 0 ? 1.0f : -1.0f;
            ray_wo.z = hemisphere_sign * fabsf(ray_wo.z);
            float4 u = rng::sample4f(i, 0, bounce);
            bool entering = ray_wo.z >= 0.0f;
            float ior_i_over_o = entering ? (medium_ior / AIR_IOR) : (AIR_IOR / medium_ior);
            BSDFSample s = GGX_RT::sample(tint, alpha, specularity, ior_i_over_o, ray_wo, make_float3(u));
            if (s.pdf.is_valid())
                throughput *= (s.reflectance * fabsf(s.direction.z)) / s.pdf.value();
            else {
                throughput = make_float3(0.0f);
                terminate = true;
            }
            bool is_inside = bounce > 0;
            bool transmission_out = is_inside && signf(s.direction.z) != signf(ray_wo.z);
            bool initial_reflection = bounce == 0 && s.direction.z >= 0.0f;
            if (initial_reflection || transmission_out)
                terminate = true;
            ray_wo = s.direction;
            escaped_is_reflection = (bounce % 2) == 0;
        }
        double* dst = escaped_is_reflection ? refl : trans;
        dst[0] += throughput.x; dst[1] += throughput.y; dst[2] += throughput.z;
    }
    for (int c = 0; c < 3; ++c) { out6[c] = float(refl[c]) / float(path_count); out6[3 + c] = float(trans[c]) / float(path_count); }
}

// Table lookups and scalar helpers pinned by G4 / G7.
void oracle_specular_rho(float abs_cos_theta, float roughness, float* out2) { SpecularRho r = SpecularRho::fetch(abs_cos_theta, roughness); out2[0] = r.base; out2[1] = r.full; }
void oracle_dielectric_rho(float abs_cos_theta, float roughness, float ior_i_over_o, float* out2) {
    DielectricRho r = DielectricRho::fetch(abs_cos_theta, roughness, ior_i_over_o); out2[0] = r.total_rho; out2[1] = r.reflected_rho;
}
float oracle_estimate_alpha(float abs_cos_theta, float max_PDF) { return GGXMinimumRoughness::estimate_alpha(abs_cos_theta, max_PDF); }
float oracle_min_roughness_from_PDF(float abs_cos_theta, float max_PDF) { return GGXMinimumRoughness::from_PDF(abs_cos_theta, PDF(max_PDF)); }
float oracle_dielectric_specularity(float ior_o, float ior_i) { return dielectric_specularity(ior_o, ior_i); }
float oracle_dielectric_ior_from_specularity(float s) { return dielectric_ior_from_specularity(s); }
void oracle_conductor_specularity(const float* ior_o, const float* ior_i, const float* ext, float* out3) { put3(out3, conductor_specularity(f3(ior_o), f3(ior_i), f3(ext))); }
void oracle_conductor_ior_from_specularity(const float* spec, const float* ext, float* out3) { put3(out3, conductor_ior_from_specularity(f3(spec), f3(ext))); }
float oracle_adjust_dielectric_specularity(float exterior_ior, float spec) { return adjust_dielectric_specularity_to_exterior_medium(exterior_ior, spec); }
void oracle_adjust_conductor_specularity(const float* exterior_ior, const float* spec, const float* ext, float* out3) {
    put3(out3, adjust_conductor_specularity_to_exterior_medium(f3(exterior_ior), f3(spec), f3(ext)));
}
// PDF wrapper semantics (OR/Types.h:155-204). kind: 0 PDF(value), 1 PDF::delta_dirac(value), 2 PDF::invalid(). out4 = value(), is_valid, use_for_MIS, is_delta_dirac.
void oracle_pdf_semantics(int kind, float value, int disable_MIS, float* out4) {
    PDF pdf = kind == 0 ? PDF(value) : kind == 1 ? PDF::delta_dirac(value) : PDF::invalid();
    if (disable_MIS) pdf.disable_MIS();
    out4[0] = pdf.value(); out4[1] = pdf.is_valid(); out4[2] = pdf.use_for_MIS(); out4[3] = pdf.is_delta_dirac();
}
int oracle_ggx_effectively_smooth_roughness(float roughness) { return GGX::effectively_smooth(GGX::alpha_from_roughness(roughness)); }
float oracle_balance_heuristic(float a, float b) { return balance_heuristic(a, b); }
float oracle_power_heuristic(float a, float b) { return power_heuristic(a, b); }
float oracle_E_FON(float cos_theta, float roughness, int exact) { return exact ? OrenNayar::E_FON_exact(cos_theta, roughness) : OrenNayar::E_FON_approx(cos_theta, roughness); }
int oracle_refract(const float* wi3, const float* n3, float ior, float* out3) { float3 r; bool ok = refract(r, f3(wi3), f3(n3), ior); put3(out3, r); return ok; }
int oracle_refract_z(const float* wi3, float ior, float* out3) { float3 r; bool ok = refract_z(r, f3(wi3), ior); put3(out3, r); return ok; }
int oracle_refract_cos(float cos_theta_i, float ior, float* out) { return refract_cos(*out, cos_theta_i, ior); }
void oracle_uniform_hemisphere(const float* u2, float* out3) { put3(out3, Dist::UniformHemisphere::sample({u2[0], u2[1]}).direction); }
void oracle_uniform_sphere(const float* u2, float* out3) { put3(out3, Dist::UniformSphere::sample({u2[0], u2[1]}).direction); }

// --------------------------------------------------------------------------------------- lights
// out8: radiance[3], pdf raw, direction[3], distance
void oracle_light_sample(const HiprLight* light, const float* position3, const float* u_n2, int n, float* out_n8) {
    for (int i = 0; i < n; ++i) {
        LightSample s = Lights::sample_radiance(*light, f3(position3), {u_n2[2 * i], u_n2[2 * i + 1]});
        put3(out_n8 + 8 * i, s.radiance);
        out_n8[8 * i + 3] = s.pdf.v;
        put3(out_n8 + 8 * i + 4, s.direction_to_light);
        out_n8[8 * i + 7] = s.distance;
    }
}
float oracle_light_pdf(const HiprLight* light, const float* position3, const float* direction3) {
    switch (light->flags & HIPR_LIGHT_TYPE_MASK) {
    case HIPR_LIGHT_SPHERE: return Lights::pdf(as_sphere(*light), f3(position3), f3(direction3)).v;
    case HIPR_LIGHT_SPOT: return Lights::pdf(as_spot(*light), f3(position3), f3(direction3)).v;
    case HIPR_LIGHT_DIRECTIONAL: return PDF::delta_dirac(0.0f).v;
    }
    return nanf("");
}
void oracle_light_evaluate(const HiprLight* light, const float* position3, const float* direction3, float* out3) {
    switch (light->flags & HIPR_LIGHT_TYPE_MASK) {
    case HIPR_LIGHT_SPHERE: put3(out3, Lights::evaluate(as_sphere(*light), f3(position3))); return;
    case HIPR_LIGHT_SPOT: put3(out3, Lights::evaluate(as_spot(*light), f3(position3), f3(direction3))); return;
    }
    put3(out3, {0, 0, 0});
}
void oracle_light_evaluate_intersection(const HiprLight* light, const float* origin3, const float* direction3, float bsdf_pdf, float* out3) {
    put3(out3, Lights::evaluate_intersection(*light, f3(origin3), f3(direction3), PDF(bsdf_pdf)));
}

// ------------------------------------------------------------------------------------ geometry
void oracle_decode_octahedral(const int16_t* enc_n2, int n, float* out_n3) {
    for (int i = 0; i < n; ++i) {
        float2 f = {float(enc_n2[2 * i]), float(enc_n2[2 * i + 1])};
        float3 v = {f.x, f.y, 32767.0f - fabsf(f.x) - fabsf(f.y)};
        float t = fmaxf(-v.z, 0.0f);
        v.x += v.x >= 0 ? -t : t;
        v.y += v.y >= 0 ? -t : t;
        put3(out_n3 + 3 * i, normalize(v));
    }
}

void oracle_fix_backfacing_shading_normal(const float* w3, const float* n3, float target, float* out3) {
    float3 w = f3(w3), n = f3(n3);
    float c = dot(w, n);
    put3(out3, c < target ? normalize(n - (c - target) * w) : n);
}

// pixels: n pairs (x, y). Outputs float4 origin+tmin(0) and float4 direction+0 per pixel.
void oracle_generate_rays(const HiprCameraState* cam, int width, int height, uint32_t accumulation, const uint32_t* pixels_xy, uint32_t n,
                          float* out_origin_tmin, float* out_direction) {
    for (uint32_t i = 0; i < n; ++i) {
        float3 o, d;
        generate_camera_ray(*cam, int(pixels_xy[2 * i]), int(pixels_xy[2 * i + 1]), width, height, accumulation, o, d);
        put3(out_origin_tmin + 4 * i, o); out_origin_tmin[4 * i + 3] = 0.0f;
        put3(out_direction + 4 * i, d); out_direction[4 * i + 3] = 0.0f;
    }
}

static inline Ray ray_of(const float* r8) { return {f3(r8), r8[3], f3(r8 + 4), r8[7]}; }

// rays: float4 origin+tmin, float4 direction+tmax. out_hits: float4 {t, u, v, bits(id)}. counters2: nodes, triangles (may be NULL).
void oracle_trace_closest(const HiprSceneDesc* scene, const float* rays, const uint32_t* skip, uint32_t n, int use_bvh, int with_lights,
                          float* out_hits, uint64_t* counters2) {
    reset_search_items();
    TraversalCounters total;
#pragma omp parallel
    {
        TraversalCounters local;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < int64_t(n); ++i) {
            Ray ray = ray_of(rays + 8 * i);
            uint32_t sk = skip ? skip[i] : HIT_MISS;
            Hit h = use_bvh == 3 ? closest_hit_wide8(*scene, ray, sk, &local) : use_bvh == 2 ? closest_hit_wide(*scene, ray, sk, &local) : use_bvh ? closest_hit_bvh(*scene, ray, sk, &local) : closest_hit_bruteforce(*scene, ray, sk, &local);
            if (with_lights) intersect_lights(*scene, ray, h);
            out_hits[4 * i] = h.t; out_hits[4 * i + 1] = h.u; out_hits[4 * i + 2] = h.v; out_hits[4 * i + 3] = uint_as_float(h.id);
        }
#pragma omp critical
        { total.nodes += local.nodes; total.triangles += local.triangles; }
    }
    if (counters2) { counters2[0] = total.nodes; counters2[1] = total.triangles; }
}

// Shadow rays with unit radiance: one transmittance float per ray.
uint32_t oracle_search_item_count(const HiprSceneDesc* scene) { reset_search_items(); return search_item_count(*scene); }

void oracle_trace_shadow(const HiprSceneDesc* scene, const float* rays, uint32_t n, int use_bvh, float* out_transmittance, uint64_t* counters2) {
    reset_search_items();
    TraversalCounters total;
#pragma omp parallel
    {
        TraversalCounters local;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < int64_t(n); ++i) {
            Ray ray = ray_of(rays + 8 * i);
            float3 r = use_bvh == 3 ? shadow_wide8(*scene, ray, make_float3(1.0f), &local) : use_bvh == 2 ? shadow_wide(*scene, ray, make_float3(1.0f), &local)
                       : use_bvh ? shadow_bvh(*scene, ray, make_float3(1.0f), &local) : shadow_bruteforce(*scene, ray, make_float3(1.0f), &local);
            out_transmittance[i] = r.x;
        }
#pragma omp critical
        { total.nodes += local.nodes; total.triangles += local.triangles; }
    }
    if (counters2) { counters2[0] = total.nodes; counters2[1] = total.triangles; }
}

// Renders `accumulation_count` accumulations starting at cam->accumulations into accum_rgba (double4 per
// pixel, row-major, row 0 = bottom), running mean exactly as accumulate<> (ORS/SimpleRGPs.cu:74-107).
// counters9 follows HiprCounters (iterations unused). Returns elapsed seconds.
double oracle_render_entry(const HiprSceneDesc* scene, const HiprSceneState* state, const HiprCameraState* cam, int width, int height,
                           uint32_t accumulation_count, int use_bvh, int entry, double* accum_rgba, uint64_t* counters9) {
    reset_search_items();
    std::vector<float4> offsets(256);
    for (int i = 0; i < 256; ++i) offsets[i] = rng::sample_offset(i);
    RenderSettings settings;
    settings.use_bvh = use_bvh != 0;
    settings.use_wide = use_bvh == 2;   // 0 brute force, 1 BVH2, 2 compressed 4-wide BVH, 3 compressed 8-wide BVH with leaf records
    settings.use_wide8 = use_bvh == 3;
    RenderCounters total;
#ifdef _OPENMP
    double t0 = omp_get_wtime();
#endif
    // Pixels are independent (the running mean of a pixel only sees that pixel's samples in accumulation order), so the parallel loop runs over
    // blocks of 16 pixels with the accumulations inside: the same image bit for bit as a pass per accumulation, and every host core stays busy
    // on small frames (a 160 x 90 frame has 900 blocks; a row-parallel loop kept 23 threads busy).
    {
#pragma omp parallel
        {
            RenderCounters local;
            const long long pixels = (long long)width * height;
#pragma omp for schedule(dynamic, 1)
            for (long long block = 0; block < (pixels + 15) / 16; ++block)
                for (long long i = block * 16; i < std::min(pixels, block * 16 + 16); ++i) {
                    const int x = int(i % width), y = int(i / width);
                    double* px = accum_rgba + 4 * (size_t(y) * width + x);
                    for (uint32_t a = 0; a < accumulation_count; ++a) {
                        const uint32_t accumulation = cam->accumulations + a;
                        float3 r = entry == HIPR_ENTRY_PATH_TRACING ? path_trace_pixel(*scene, *state, *cam, offsets.data(), x, y, width, height, accumulation, settings, &local)
                                                                    : aov_pixel(*scene, *state, *cam, offsets.data(), x, y, width, height, accumulation, entry, settings);
                        if (accumulation != 0) {
                            double t = 1.0 / (accumulation + 1.0);
                            px[0] = px[0] + (double(r.x) - px[0]) * t;
                            px[1] = px[1] + (double(r.y) - px[1]) * t;
                            px[2] = px[2] + (double(r.z) - px[2]) * t;
                        } else {
                            px[0] = r.x; px[1] = r.y; px[2] = r.z;
                        }
                        px[3] = 1.0;
                    }
                }
#pragma omp critical
            {
                total.camera_rays += local.camera_rays; total.closest_rays += local.closest_rays; total.shadow_rays += local.shadow_rays;
                total.shaded_hits += local.shaded_hits; total.closest.nodes += local.closest.nodes; total.closest.triangles += local.closest.triangles;
                total.shadow.nodes += local.shadow.nodes; total.shadow.triangles += local.shadow.triangles;
                total.rejected_hits += local.rejected_hits;
            }
        }
    }
    if (counters9) {
        counters9[0] = total.camera_rays; counters9[1] = total.closest_rays; counters9[2] = total.shadow_rays; counters9[3] = total.shaded_hits;
        counters9[4] = total.closest.nodes; counters9[5] = total.closest.triangles; counters9[6] = total.shadow.nodes; counters9[7] = total.shadow.triangles;
        counters9[8] = total.rejected_hits;
    }
#ifdef _OPENMP
    return omp_get_wtime() - t0;
#else
    return 0.0;
#endif
}

double oracle_render(const HiprSceneDesc* scene, const HiprSceneState* state, const HiprCameraState* cam, int width, int height,
                     uint32_t accumulation_count, int use_bvh, double* accum_rgba, uint64_t* counters9) {
    return oracle_render_entry(scene, state, cam, width, height, accumulation_count, use_bvh, HIPR_ENTRY_PATH_TRACING, accum_rgba, counters9);
}

// K3 at stage level: the hit programs for n traced rays whose closest hits are given (integrator.cpp shade_hit_for_test); 32 words per entry.
void oracle_debug_shade(const HiprSceneDesc* scene, const HiprSceneState* state, const HiprCameraState* cam, uint32_t n, const float* rays_n8, const float* throughput_bounces_n4,
                        const float* hits_n4, const uint32_t* last_triangle, const uint32_t* pixel_hash, const uint32_t* accumulation, float* out_n32) {
    reset_search_items();
    std::vector<float4> offsets(256);
    for (int i = 0; i < 256; ++i) offsets[i] = rng::sample_offset(i);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < int64_t(n); ++i)
        shade_hit_for_test(*scene, *state, *cam, offsets.data(), rays_n8 + 8 * i, throughput_bounces_n4 + 4 * i, hits_n4 + 4 * i, last_triangle[i], pixel_hash[i], accumulation[i], out_n32 + 32 * i);
}

// 1: every transcendental of the path in f64, rounded once to f32 (vecmath.h exact_*): the checker of the verification build. Returns the previous setting.
int oracle_set_f64_transcendentals(int on) {
    const int before = oracle::g_f64_transcendentals ? 1 : 0;
    oracle::g_f64_transcendentals = on != 0;
    return before;
}

// The specified transcendentals of the exact arithmetic mode over arrays (vecmath.h spec::): function 0 sin, 1 cos, 2 pow(x, y).
void oracle_spec_math(int function, int n, const float* x, const float* y, float* out) {
    for (int i = 0; i < n; ++i)
        out[i] = function == 2 ? oracle::spec::pow(x[i], y[i]) : oracle::spec::sin_or_cos(x[i], function);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

} // extern "C"
