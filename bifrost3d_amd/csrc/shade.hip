// shade.hip -- translation unit of the shade kernel (see shade_kernel.h for why it is separate). Compiled TWICE into libhiprenderer.so (bifrost3d_amd/Makefile):
//   shade.o        HIPR_SHADE_EXACT unset: hardware-approximate division / sqrt / sin / cos / pow, contraction, denormals flushed -- the reference's --use_fast_math PTX
//                  (extensions/OptiXRenderer/CMakeLists.txt:82-83); the default, HIPR_ARITHMETIC_FAST.
//   shade_exact.o  HIPR_SHADE_EXACT=1: the traversal unit's flags (IEEE division / sqrt, no contraction, denormals kept) and the specified transcendentals of
//                  spec_math.h: every frame equals the CPU oracle's bit for bit (HIPR_ARITHMETIC_EXACT, hipr_set_arithmetic).
// Each build exports its host entry points as one ShadeUnit (launch.h); its kernels carry the build in their names (k_shade's ARITHMETIC argument, HIPR_UNIT).
#define HIPR_SHADE_TU 1
#ifndef HIPR_SHADE_EXACT
#define HIPR_SHADE_EXACT 0
#endif
#if HIPR_SHADE_EXACT
#define HIPR_FAST_MATH 0
#define HIPR_VERIFY_MATH 1
#define HIPR_UNIT(name) name##_exact
#else
#ifndef HIPR_FAST_MATH
#define HIPR_FAST_MATH 1
#endif
#define HIPR_UNIT(name) name##_fast
#endif
#include "shade_kernel.h"
#include "launch.h"

namespace hipr {

constexpr int UNIT_ARITHMETIC = HIPR_SHADE_EXACT ? HIPR_ARITHMETIC_EXACT : HIPR_ARITHMETIC_FAST;

template <int MODELS, bool AOV, int TEXTURES>
static void launch_models_with(const ShadeLaunch& a) {
    if (!AOV && a.nee_flags) {     // the two halves, one after the other (shade_kernel.h SHADE_PART_*)
        hipLaunchKernelGGL((k_shade<MODELS, AOV, SHADE_PART_NEE, TEXTURES, UNIT_ARITHMETIC>), dim3(a.grid), dim3(SHADE_BLOCK), 0, a.stream, a.scene, a.camera, a.frame, a.entry, a.in, a.hits, a.order, a.order_coat, a.listed, a.out,
                           a.shadows, a.radiance, a.in_count, a.out_counts, nullptr, nullptr, a.nee_flags, a.counters);
        hipLaunchKernelGGL((k_shade<MODELS, AOV, SHADE_PART_BSDF, TEXTURES, UNIT_ARITHMETIC>), dim3(a.grid), dim3(SHADE_BLOCK), 0, a.stream, a.scene, a.camera, a.frame, a.entry, a.in, a.hits, a.order, a.order_coat, a.listed, a.out,
                           a.shadows, a.radiance, a.in_count, a.out_counts, a.zero_a, a.zero_b, a.nee_flags, a.counters);
        return;
    }
    hipLaunchKernelGGL((k_shade<MODELS, AOV, SHADE_PART_ALL, TEXTURES, UNIT_ARITHMETIC>), dim3(a.grid), dim3(SHADE_BLOCK), 0, a.stream, a.scene, a.camera, a.frame, a.entry, a.in, a.hits, a.order, a.order_coat, a.listed, a.out, a.shadows,
                       a.radiance, a.in_count, a.out_counts, a.zero_a, a.zero_b, nullptr, a.counters);
}

// Scenes without a texture or an environment map run the instantiation without the samplers (shade_kernel.h TEXTURES); the AOV entries keep the one generic kernel.
template <int MODELS, bool AOV>
static void launch_models(const ShadeLaunch& a) {
    if (!AOV && !a.textures) launch_models_with<MODELS, AOV, 0>(a);
    else if (!AOV && !a.environment) launch_models_with<MODELS, AOV, 1>(a);      // material textures under a plain sky (the two-kernel form too: its frames are compared bit for bit)
    else launch_models_with<MODELS, AOV, 2>(a);
}

// The kernel is instantiated per set of shading models the uploaded scene uses (bit 0 Default, 1 Diffuse, 2 Transmissive).
static void launch_shade(int shading_models, const ShadeLaunch& a) {
    if (a.entry != HIPR_ENTRY_PATH_TRACING) { launch_models<7, true>(a); return; }   // AOV entries: one generic instantiation
    switch (shading_models) {
    case 1: launch_models<1, false>(a); break;
    case 2: launch_models<2, false>(a); break;
    case 4: launch_models<4, false>(a); break;
    case 3: launch_models<3, false>(a); break;
    case 5: launch_models<5, false>(a); break;
    case 6: launch_models<6, false>(a); break;
    default: launch_models<7, false>(a); break;
    }
}

// Stage-level parity entry point (hipr_debug_shading): the shading models exactly as the shade kernel evaluates them.
// model: 0 Default, 1 Diffuse, 2 Transmissive. params: tint[3], roughness, specularity, metallic, coat, coat_roughness,
// cos_theta_o (NaN: wo.z), max_PDF_hint (NaN: none). mode 0: sample(wo, u) -> f[3], pdf, direction[3]; mode 1:
// evaluate_with_PDF(wo, wi = third input) -> f[3], pdf, 0, 0, 0.
__global__ void HIPR_UNIT(k_debug_shading)(DeviceTables tables, int model, const float* params, const float* wo_n3, const float* in_n3, int n, int mode, float* out_n7) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 wo = {wo_n3[3 * i], wo_n3[3 * i + 1], wo_n3[3 * i + 2]}, in = {in_n3[3 * i], in_n3[3 * i + 1], in_n3[3 * i + 2]};
    MaterialInputs m;
    m.tint = {params[0], params[1], params[2]};
    m.roughness = params[3]; m.specularity = params[4]; m.metallic = params[5]; m.coat = params[6]; m.coat_roughness = params[7];
    const float cos_theta = params[8] != params[8] ? wo.z : params[8];
    const float hint = params[9] != params[9] ? -1.0f : params[9];
    Shading s;
    if (model == HIPR_SHADING_DIFFUSE) s = make_diffuse(m.tint, m.roughness);
    else if (model == HIPR_SHADING_TRANSMISSIVE) s = make_transmissive(tables, m, cos_theta, hint);
    else s = make_default(tables, m, cos_theta, hint);
    float* o = out_n7 + 7 * i;
    const ShadingTerms terms = shading_terms<7>(s, wo);     // the forms the shade kernel calls
    if (mode == 0) {
        const Sample r = shading_sample<7>(s, terms, wo, in);
        o[0] = r.f.x; o[1] = r.f.y; o[2] = r.f.z; o[3] = r.pdf; o[4] = r.dir.x; o[5] = r.dir.y; o[6] = r.dir.z;
    } else {
        const Response r = shading_evaluate_with_PDF<7>(s, terms, wo, in);
        o[0] = r.f.x; o[1] = r.f.y; o[2] = r.f.z; o[3] = r.pdf; o[4] = o[5] = o[6] = 0.0f;
    }
}

// The light sources as the shade kernel evaluates them. mode 0: sample_radiance(light, position, u = in.xy) -> radiance[3], PDF,
// direction_to_light[3], distance. mode 1 (spot lights): evaluate(light, position, direction = in) -> radiance[3], pdf(...), 0, 0, 0, 0.
__global__ void HIPR_UNIT(k_debug_light)(HiprLight light, const float* position3, const float* in_n3, int n, int mode, float* out_n8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 position = {position3[0], position3[1], position3[2]}, in = {in_n3[3 * i], in_n3[3 * i + 1], in_n3[3 * i + 2]};
    float* o = out_n8 + 8 * i;
    if (mode == 0) {
        const LightSample s = light_sample_radiance(light, position, mk2(in.x, in.y));
        o[0] = s.radiance.x; o[1] = s.radiance.y; o[2] = s.radiance.z; o[3] = s.pdf; o[4] = s.dir.x; o[5] = s.dir.y; o[6] = s.dir.z; o[7] = s.distance;
    } else {
        const f3 radiance = spot_evaluate(light, position, in);
        o[0] = radiance.x; o[1] = radiance.y; o[2] = radiance.z; o[3] = spot_pdf(light, position, in); o[4] = o[5] = o[6] = o[7] = 0.0f;
    }
}

// hipr_debug_shade: shade_path for n given queue entries, one record of 32 words each (tests/native/DeviceShadeHost.hip writes the same record from the host
// build of this code, oracle/integrator.cpp shade_hit_for_test from the oracle's hit programs): 0 flags (1 continues, 2 shadow ray, 4 shaded), 1-3 radiance added,
// 4-7 next origin + tmin, 8-11 next direction + BSDF PDF, 12-15 throughput + bits(bounces), 16 bits(last triangle), 17-20 shadow origin + tmax, 21-23 direction to
// the light, 24-26 radiance the shadow ray carries. The generic instantiation (all models, all samplers): the template arguments of k_shade only remove code.
__global__ __launch_bounds__(64) void HIPR_UNIT(k_debug_shade)(DeviceScene sc, HiprCameraState cam, uint32_t n, const float4* rays, const float4* throughput_bounces, const float4* hits,
                                                    const uint32_t* last_triangle, const uint32_t* pixel_hash, const uint32_t* accumulation, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ShadeInputs in;
    in.entry = i;
    in.meta = make_uint2(0u, last_triangle[i]);
    in.o = rays[2 * i]; in.d = rays[2 * i + 1]; in.t = throughput_bounces[i]; in.hit = hits[i];
    const ShadeGeometry geo = shade_fetch_geometry(sc, in);
    const HiprMaterial mat = shade_fetch_material(sc, in, geo);
    ShadeOutput so = {};
    shade_path<7, false, SHADE_PART_ALL, 2>(sc, cam, HIPR_ENTRY_PATH_TRACING, sc.sobol_tables, mk3(in.o.x, in.o.y, in.o.z), rays + 2 * i, mk3(in.d.x, in.d.y, in.d.z), in.d.w, mk3(in.t.x, in.t.y, in.t.z),
                                            __float_as_uint(in.t.w), in.meta.y, pixel_hash[i], accumulation[i], in.hit, geo, mat, false, so);
    float* o = out + 32 * size_t(i);
    for (int k = 0; k < 32; ++k) o[k] = 0.0f;
    o[0] = __uint_as_float((so.continues ? 1u : 0u) | (so.shadow ? 2u : 0u) | (so.shaded ? 4u : 0u));
    o[1] = so.add_radiance.x; o[2] = so.add_radiance.y; o[3] = so.add_radiance.z;
    if (so.continues) {
        o[4] = so.o.x; o[5] = so.o.y; o[6] = so.o.z; o[7] = so.tmin;
        o[8] = so.d.x; o[9] = so.d.y; o[10] = so.d.z; o[11] = so.bsdf_pdf;
        o[12] = so.throughput.x; o[13] = so.throughput.y; o[14] = so.throughput.z; o[15] = __uint_as_float(so.bounces);
        o[16] = __uint_as_float(so.last_triangle);
    }
    if (so.shadow) {
        o[17] = so.so.x; o[18] = so.so.y; o[19] = so.so.z; o[20] = so.stmax;
        o[21] = so.sd.x; o[22] = so.sd.y; o[23] = so.sd.z;
        o[24] = so.sradiance.x; o[25] = so.sradiance.y; o[26] = so.sradiance.z;
    }
}

static void launch_debug_shade(hipStream_t stream, const DeviceScene& scene, const HiprCameraState& camera, uint32_t n, const float4* rays, const float4* throughput_bounces, const float4* hits,
                        const uint32_t* last_triangle, const uint32_t* pixel_hash, const uint32_t* accumulation, float* out) {
    hipLaunchKernelGGL(HIPR_UNIT(k_debug_shade), dim3((n + 63) / 64), dim3(64), 0, stream, scene, camera, n, rays, throughput_bounces, hits, last_triangle, pixel_hash, accumulation, out);
}

static void launch_debug_light(hipStream_t stream, const HiprLight& light, const float* position3, const float* in_n3, int n, int mode, float* out_n8) {
    hipLaunchKernelGGL(HIPR_UNIT(k_debug_light), dim3((n + 63) / 64), dim3(64), 0, stream, light, position3, in_n3, n, mode, out_n8);
}

static void launch_debug_shading(hipStream_t stream, const DeviceTables& tables, int model, const float* params, const float* wo_n3, const float* in_n3, int n, int mode,
                          float* out_n7) {
    hipLaunchKernelGGL(HIPR_UNIT(k_debug_shading), dim3((n + 63) / 64), dim3(64), 0, stream, tables, model, params, wo_n3, in_n3, n, mode, out_n7);
}

// The transcendentals as this build of the unit evaluates them (device_shading.h sincos_ / pow_), over arrays: function 0 sin, 1 cos, 2 pow(x, y).
__global__ void HIPR_UNIT(k_debug_math)(int function, int n, const float* x, const float* y, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (function == 2) out[i] = pow_(x[i], y[i]);
    else { float s, c; sincos_(x[i], s, c); out[i] = function == 0 ? s : c; }
}
static void launch_debug_math(hipStream_t stream, int function, int n, const float* x, const float* y, float* out) {
    hipLaunchKernelGGL(HIPR_UNIT(k_debug_math), dim3((n + 255) / 256), dim3(256), 0, stream, function, n, x, y, out);
}

// (a function: a namespace-scope constant would be emitted for the device as well, where the launchers do not exist)
const ShadeUnit& HIPR_UNIT(shade_unit)() {
    static const ShadeUnit unit = {launch_shade, launch_debug_shade, launch_debug_light, launch_debug_shading, launch_debug_math, HIPR_SHADE_WAVES};
    return unit;
}

} // namespace hipr
