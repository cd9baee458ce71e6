// DeviceShadeHost.hip -- the DEVICE code of the shade stage, compiled for the HOST by hipcc's host pass (tests/native/libdevice_shade_host.so).
//
// Test infrastructure (tests/test_device_code_on_host_cpu.py); nothing here is linked into or loaded by the product, which has no CPU path.
//
// csrc/shade_kernel.h's shade_path -- attribute interpolation, material and textures, the RIS next event estimation, BSDF sampling, ray offsets: everything K3
// computes per hit -- with the headers it includes (device_shading.h, kernels.h' samplers and scene structures) is ordinary IEEE f32 arithmetic in a fixed order.
// With HD = __host__ __device__ (HIPR_HOST_DEVICE) clang compiles every such function for both sides; this file calls them from host code, so `hipcc
// --cuda-host-only -ffp-contract=off` yields an x86 build of exactly the statements the GPU runs. Built with HIPR_VERIFY_MATH = 1 it is the arithmetic of the
// verification build libhiprenderer_verify.so, operation for operation (the stage tests return the same mismatch counts on both, to the digit), which lets the CPU
// suite hold the device's K3 to the oracle's hit programs BIT for bit, entry by entry, without a GPU -- and say which word of which entry differs when it does not.
//
// Record written per entry (32 words; the words of parts that do not apply stay zero; the oracle writes the same: oracle/integrator.cpp shade_hit_for_test):
//   0 flags: 1 the path continues, 2 a shadow ray was emitted, 4 the hit was shaded        1-3 radiance added to the path's slot
//   4-7 next origin + tmin    8-11 next direction + BSDF PDF    12-15 throughput + bits(bounces)    16 bits(last accepted triangle)
//   17-20 shadow origin + tmax    21-23 direction to the light    24-26 radiance the shadow ray carries
#define HIPR_HOST_DEVICE 1
#define HIPR_SHADE_TU 1      // none of the kernels of the other translation unit: this build holds host code only
#define HIPR_FAST_MATH 0
#ifndef HIPR_VERIFY_MATH
#define HIPR_VERIFY_MATH 1
#endif
#include "../../bifrost3d_amd/csrc/shade_kernel.h"

#include <vector>

namespace {

using namespace hipr;

struct HostScene {
    DeviceScene scene = {};
    std::vector<float4> triangles, shade_triangles, geometry, sample_offsets, env_samples;
    std::vector<HiprInstance> instances;
    std::vector<uint32_t> indices, tints, sobol;
    std::vector<float2> texcoords;
    std::vector<float> emissions, env_pdf;
    std::vector<HiprMaterial> materials;
    std::vector<HiprLight> lights;
    std::vector<HiprTexture> textures;
    std::vector<uint8_t> texels;
    std::vector<ushort2> rho, dielectric;
    std::vector<unsigned short> alpha;
};

unsigned short to_unorm16(float v) { return (unsigned short)(v * 65535 + 0.5f); }      // hiprenderer.hip hipr_upload_tables

float reverse_halton(int prime, int i) {      // hiprenderer.hip (OR/Renderer.cpp:323-336): primes 2, 3, 5, 7, digits d -> p - d, f64 inside
    double h = 0.0, f = 1.0 / double(prime), fct = f;
    while (i > 0) {
        int digit = i % prime;
        h += (digit == 0 ? 0 : prime - digit) * fct;
        i /= prime;
        fct *= f;
    }
    return float(h);
}

template <typename T, typename S>
void copy_in(std::vector<T>& to, const S* from, size_t count) {
    to.resize(count);
    if (count && from) std::memcpy(to.data(), from, count * sizeof(T));
}

template <int MODELS, int TEXTURES>
void shade_entries(const HostScene& h, const HiprCameraState& cam, uint32_t n, const float* rays, const float* thr, const float* hits, const uint32_t* last_triangle,
                   const uint32_t* pixel_hash, const uint32_t* accumulation, float* out) {
    const DeviceScene& sc = h.scene;
    for (uint32_t i = 0; i < n; ++i) {
        // what k_shade fetches for a queue entry (shade_fetch_inputs / _geometry / _material), from the arrays of this call
        ShadeInputs in;
        in.entry = i;
        in.meta = make_uint2(0u, last_triangle[i]);
        in.o = make_float4(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2], rays[8 * i + 3]);
        in.d = make_float4(rays[8 * i + 4], rays[8 * i + 5], rays[8 * i + 6], rays[8 * i + 7]);
        in.t = make_float4(thr[4 * i], thr[4 * i + 1], thr[4 * i + 2], thr[4 * i + 3]);
        in.hit = make_float4(hits[4 * i], hits[4 * i + 1], hits[4 * i + 2], hits[4 * i + 3]);
        const ShadeGeometry geo = shade_fetch_geometry(sc, in);
        const HiprMaterial mat = shade_fetch_material(sc, in, geo);
        ShadeOutput so = {};
        shade_path<MODELS, false, SHADE_PART_ALL, TEXTURES>(sc, cam, HIPR_ENTRY_PATH_TRACING, sc.sobol_tables, mk3(in.o.x, in.o.y, in.o.z), &in.o, mk3(in.d.x, in.d.y, in.d.z), in.d.w,
                                                            mk3(in.t.x, in.t.y, in.t.z), __float_as_uint(in.t.w), in.meta.y, pixel_hash[i], accumulation[i], in.hit, geo, mat, false, so);
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
}

} // namespace

extern "C" {

// The scene as hipr_upload_scene + hipr_upload_tables + hipr_set_scene_state leave it on the device (csrc/hiprenderer.hip), for the parts the shade stage reads.
void* dsh_scene_create(const HiprSceneDesc* s, const HiprSceneState* state, const float* ggx_with_fresnel_rho, const float* ggx_rho, const float* dielectric_light_rho,
                       const float* dielectric_dense_rho, const float* alpha) {
    HostScene* h = new HostScene;
    DeviceScene& d = h->scene;
    copy_in(h->triangles, s->triangles, size_t(s->triangle_count) * 3);      // 48 B = 3 float4 each
    copy_in(h->instances, s->instances, s->instance_count);
    copy_in(h->indices, s->indices, s->index_count);
    copy_in(h->geometry, s->geometry, s->vertex_count);
    if (s->texcoords) copy_in(h->texcoords, s->texcoords, s->vertex_count);
    if (s->tints) copy_in(h->tints, s->tints, s->vertex_count);
    if (s->emissions) copy_in(h->emissions, s->emissions, size_t(s->vertex_count) * 3);
    copy_in(h->materials, s->materials, s->material_count);
    copy_in(h->lights, s->lights, s->light_count);
    copy_in(h->textures, s->textures, s->texture_count);
    copy_in(h->texels, s->texels, s->texel_bytes);
    d.triangles = h->triangles.data(); d.instances = h->instances.data(); d.indices = h->indices.data(); d.geometry = h->geometry.data();
    d.texcoords = h->texcoords.empty() ? nullptr : h->texcoords.data(); d.tints = h->tints.empty() ? nullptr : h->tints.data();
    d.emissions = h->emissions.empty() ? nullptr : h->emissions.data();
    d.materials = h->materials.data(); d.lights = h->lights.data(); d.textures = h->textures.data(); d.texels = h->texels.data();
    d.triangle_count = s->triangle_count; d.light_count = s->light_count;
    // tables: unorm16, as uploaded (hipr_upload_tables)
    h->rho.resize(32 * 32); h->alpha.resize(32 * 32);
    for (int i = 0; i < 32 * 32; ++i) { h->rho[i] = {to_unorm16(ggx_with_fresnel_rho[i]), to_unorm16(ggx_rho[i])}; h->alpha[i] = to_unorm16(alpha[i]); }
    const int per_medium = 16 * 16 * 16;
    h->dielectric.resize(2 * per_medium);
    for (int i = 0; i < per_medium; ++i) {
        h->dielectric[i] = {to_unorm16(dielectric_light_rho[2 * i]), to_unorm16(dielectric_light_rho[2 * i + 1])};
        h->dielectric[per_medium + i] = {to_unorm16(dielectric_dense_rho[2 * i]), to_unorm16(dielectric_dense_rho[2 * i + 1])};
    }
    d.tables = {h->rho.data(), h->dielectric.data(), h->alpha.data()};
    // reverse Halton offsets and the byte-indexed Sobol tables (hipr_create)
    h->sample_offsets.resize(256);
    const int primes[4] = {2, 3, 5, 7};
    for (int i = 0; i < 256; ++i) h->sample_offsets[i] = make_float4(reverse_halton(primes[0], i), reverse_halton(primes[1], i), reverse_halton(primes[2], i), reverse_halton(primes[3], i));
    d.sample_offsets = h->sample_offsets.data();
    h->sobol.resize(SOBOL_TABLE_WORDS);
    for (int dim = 0; dim < 3; ++dim)
        for (int k = 0; k < 4; ++k)
            for (int b = 0; b < 256; ++b) {
                uint32_t v = 0;
                for (int j = 0; j < 8; ++j)
                    if (b & (1 << j)) v ^= SOBOL_DIRECTIONS[dim][8 * k + j];
                h->sobol[(dim * 4 + k) * 256 + b] = v;
            }
    d.sobol_tables = h->sobol.data();
    // environment (hipr_upload_scene) and scene state (hipr_set_scene_state)
    const HiprEnvironment* env = s->environment;
    d.env_map_ID = env ? env->environment_map_ID : 0;
    if (env) {
        copy_in(h->env_pdf, env->per_pixel_PDF, size_t(env->pdf_width) * env->pdf_height);
        copy_in(h->env_samples, env->samples, size_t(env->sample_count) * 2);      // 32 B = 2 float4 each
        d.env_per_pixel_PDF = h->env_pdf.data(); d.env_samples = h->env_samples.data();
        d.env_pdf_width = env->pdf_width; d.env_pdf_height = env->pdf_height; d.env_sample_count = env->sample_count;
    }
    for (int i = 0; i < 3; ++i) d.env_tint[i] = state->environment_tint[i];
    d.next_event_sample_count = std::min(std::max(state->next_event_sample_count, 0), 256);
    // the shading records (k_build_shade_triangles)
    h->shade_triangles.resize(size_t(s->triangle_count) * SHADE_TRIANGLE_QUADS);
    for (uint32_t t = 0; t < s->triangle_count; ++t) build_shade_triangle_record(d, t, h->shade_triangles.data());
    d.shade_triangles = h->shade_triangles.data();
    return h;
}

void dsh_scene_destroy(void* scene) { delete static_cast<HostScene*>(scene); }

// k_debug_shading (csrc/shade.hip: hipr_debug_shading), statement for statement, on the tables of `scene`. terms 1: the forms the shade kernel calls (ShadingTerms:
// what depends on the outgoing direction computed once per hit); 0: the plain functions written after the reference's.
void dsh_shading(void* scene, int model, const float* params, const float* wo_n3, const float* in_n3, int n, int mode, int terms, float* out_n7) {
    const DeviceTables t = static_cast<HostScene*>(scene)->scene.tables;
    for (int i = 0; i < n; ++i) {
        const f3 wo = {wo_n3[3 * i], wo_n3[3 * i + 1], wo_n3[3 * i + 2]}, in = {in_n3[3 * i], in_n3[3 * i + 1], in_n3[3 * i + 2]};
        MaterialInputs m;
        m.tint = {params[0], params[1], params[2]};
        m.roughness = params[3]; m.specularity = params[4]; m.metallic = params[5]; m.coat = params[6]; m.coat_roughness = params[7];
        const float cos_theta = params[8] != params[8] ? wo.z : params[8];
        const float hint = params[9] != params[9] ? -1.0f : params[9];
        Shading s;
        if (model == HIPR_SHADING_DIFFUSE) s = make_diffuse(m.tint, m.roughness);
        else if (model == HIPR_SHADING_TRANSMISSIVE) s = make_transmissive(t, m, cos_theta, hint);
        else s = make_default(t, m, cos_theta, hint);
        float* o = out_n7 + 7 * i;
        const ShadingTerms st = shading_terms<7>(s, wo);
        if (mode == 0) {
            const Sample r = terms ? shading_sample<7>(s, st, wo, in) : shading_sample<7>(s, wo, in);
            o[0] = r.f.x; o[1] = r.f.y; o[2] = r.f.z; o[3] = r.pdf; o[4] = r.dir.x; o[5] = r.dir.y; o[6] = r.dir.z;
        } else {
            const Response r = terms ? shading_evaluate_with_PDF<7>(s, st, wo, in) : shading_evaluate_with_PDF<7>(s, wo, in);
            o[0] = r.f.x; o[1] = r.f.y; o[2] = r.f.z; o[3] = r.pdf; o[4] = o[5] = o[6] = 0.0f;
        }
    }
}

// The samplers' texel index arithmetic (csrc/kernels.h): out[2 k] = wrap_coord(i, n, repeat), out[2 k + 1] = wrap_next(out[2 k], i, n, repeat) = the neighbour texel of a bilinear tap.
void dsh_wrap(const int* coordinates, int count, int n, int repeat, int* out) {
    for (int k = 0; k < count; ++k) {
        out[2 * k] = wrap_coord(coordinates[k], n, repeat);
        out[2 * k + 1] = wrap_next(out[2 * k], coordinates[k], n, repeat);
    }
}

// k_debug_light (csrc/shade.hip: hipr_debug_light).
void dsh_light(const HiprLight* light, const float* position3, const float* in_n3, int n, int mode, float* out_n8) {
    const f3 position = {position3[0], position3[1], position3[2]};
    for (int i = 0; i < n; ++i) {
        const f3 in = {in_n3[3 * i], in_n3[3 * i + 1], in_n3[3 * i + 2]};
        float* o = out_n8 + 8 * i;
        if (mode == 0) {
            const LightSample s = light_sample_radiance(*light, position, mk2(in.x, in.y));
            o[0] = s.radiance.x; o[1] = s.radiance.y; o[2] = s.radiance.z; o[3] = s.pdf; o[4] = s.dir.x; o[5] = s.dir.y; o[6] = s.dir.z; o[7] = s.distance;
        } else {
            const f3 radiance = spot_evaluate(*light, position, in);
            o[0] = radiance.x; o[1] = radiance.y; o[2] = radiance.z; o[3] = spot_pdf(*light, position, in); o[4] = o[5] = o[6] = o[7] = 0.0f;
        }
    }
}

// shade_path for n queue entries. models / textures pick the instantiation k_shade would run (MODELS mask 1 Default, 2 Diffuse, 4 Transmissive, 7 all; TEXTURES 0 / 1 / 2):
// the template arguments only remove code, which this entry lets a test confirm. Returns 0, or -1 for a combination that is not instantiated here.
int dsh_shade(void* scene, const HiprCameraState* cam, uint32_t n, const float* rays_n8, const float* throughput_bounces_n4, const float* hits_n4, const uint32_t* last_triangle,
              const uint32_t* pixel_hash, const uint32_t* accumulation, int models, int textures, float* out_n32) {
    const HostScene& h = *static_cast<HostScene*>(scene);
#define HIPR_CASE(M, T) if (models == M && textures == T) { shade_entries<M, T>(h, *cam, n, rays_n8, throughput_bounces_n4, hits_n4, last_triangle, pixel_hash, accumulation, out_n32); return 0; }
    HIPR_CASE(7, 2) HIPR_CASE(7, 0) HIPR_CASE(1, 0) HIPR_CASE(1, 1) HIPR_CASE(2, 0) HIPR_CASE(4, 2)
#undef HIPR_CASE
    return -1;
}

// csrc/spec_math.h as compiled for the host, over arrays: function 0 sin, 1 cos, 2 pow(x, y) (k_debug_spec_math, csrc/shade.hip: hipr_debug_spec_math).
void dsh_spec_math(int function, int n, const float* x, const float* y, float* out) {
    for (int i = 0; i < n; ++i) {
        if (function == 2) out[i] = spec_pow(x[i], y[i]);
        else { float s, c; spec_sincos(x[i], s, c); out[i] = function == 0 ? s : c; }
    }
}

}
