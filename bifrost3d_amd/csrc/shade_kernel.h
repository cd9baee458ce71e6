// shade_kernel.h -- K3 of the wavefront path tracer: shade + next event estimation + BSDF sampling + stream compaction.
// Compiled in its own translation unit (shade.hip) with hardware-approximate division / sqrt / transcendentals, the
// way the reference builds its shading PTX (--use_fast_math, extensions/OptiXRenderer/CMakeLists.txt:82-83). The
// ray generation, traversal and accumulation kernels stay correctly rounded so they match the CPU oracle bit for bit.
#pragma once

#include "kernels.h"

namespace hipr {

// ---------------------------------------------------------------------------------------------
// K3: shade + next event estimation + BSDF sampling + stream compaction
// ---------------------------------------------------------------------------------------------
HD f3 fix_backfacing_shading_normal(f3 w, f3 n, float target) {
    float c = dot(w, n);
    return c < target ? normalize(n - (c - target) * w) : n;
}

HD f3 offset_ray_origin(f3 p, f3 n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int ox = int(int_scale * n.x), oy = int(int_scale * n.y), oz = int(int_scale * n.z);
    f3 pi = {__int_as_float(__float_as_int(p.x) + (p.x < 0 ? -ox : ox)), __int_as_float(__float_as_int(p.y) + (p.y < 0 ? -oy : oy)),
             __int_as_float(__float_as_int(p.z) + (p.z < 0 ? -oz : oz))};
    return {fabsf(p.x) < origin ? p.x + float_scale * n.x : pi.x, fabsf(p.y) < origin ? p.y + float_scale * n.y : pi.y,
            fabsf(p.z) < origin ? p.z + float_scale * n.z : pi.z};
}
HD f3 offset_ray_origin(f3 p, f3 direction, f3 geometric_normal) {
    return offset_ray_origin(p, dot(geometric_normal, direction) >= 0 ? geometric_normal : -geometric_normal);
}

struct ShadeOutput {
    bool continues;      // the path goes on (next bounce or retrace)
    bool shadow;         // a shadow ray was emitted
    bool shaded;         // an accepted surface hit
    f3 o, d; float tmin, bsdf_pdf; f3 throughput; uint32_t bounces, last_triangle;
    f3 so, sd; float stmax; f3 sradiance;
    uint32_t shadow_class;   // the light the shadow ray goes to, min(light index, SHADOW_CLASSES - 1): the block lists its shadow rays by it (k_shade)
    f3 add_radiance;
    bool nee_reached;    // SHADE_PART_NEE: the hit was accepted and next event estimation ran ...
    bool nee_valid;      // ... and kept a light sample with a valid PDF
};

// The shade kernel whole, or as two kernels run one after the other over the same queue (HIPR_SHADE_SPLIT=1): next event estimation (queues the shadow
// rays and leaves one flag per accepted hit: was a light sample kept -- the one thing the BSDF sample needs to know of it, MonteCarlo.cu:224) and the
// rest (emission, miss and light hits, rejected hits, BSDF sampling; queues the paths that continue). Both halves redo the hit's attributes, textures
// and shading setup; each needs fewer registers than the whole.
constexpr int SHADE_PART_ALL = 0, SHADE_PART_NEE = 1, SHADE_PART_BSDF = 2;
// HIPR_SHADOW_CLASSES > 1 (opt-in A/B of round 5): shadow rays listed BY LIGHT inside every block's stretch of the shadow queue -- the rays toward
// light 0, then light 1, then all others -- so that the trace waves, which take them 64 at a time, hold rays toward ONE light from a few neighbouring pixels
// (near-parallel for a directional light, converging on one small sphere): coherent by construction, no extra pass, no extra atomics, no gathered reads; frames and
// counters bit-identical. Measured on the atrium (directional + sphere light; profiles/r05_ab_shadow_by_light.txt): trace 79.9 -> 80.0 ms per step -- nothing --
// and the three ballots cost the shade kernel 1.6 ms (24.0 -> 25.6). The lanes the fused launches lack are those of the BSDF-sampled closest-hit rays, whose
// directions no listing makes alike; the shadow rays were not what held the waves back. Off (1 class: the code reduces to the single list).
#ifndef HIPR_SHADOW_CLASSES
#define HIPR_SHADOW_CLASSES 1
#endif
constexpr uint32_t SHADOW_CLASSES = HIPR_SHADOW_CLASSES;
static_assert(SHADE_BLOCK <= 1023 && 10u + 10u * SHADOW_CLASSES <= 50u, "k_shade packs a block's queue counts into 10-bit fields of one LDS word (paths that continue, shadow rays per class, arrivals from bit 50)");

// What shade_path reads of the hit triangle, fetched one loop iteration ahead by k_shade.
struct ShadeGeometry {
    float4 ta, tb, tc;                 // world-space positions + ids (the 48 B traversal triangle)
    float4 s0, s1, s2, s3, s4, s5;     // shading record (k_build_shade_triangles)
};

// TEXTURES: 0 = the scene holds neither textures nor an environment map, 1 = 8-bit material textures only (round 4: the samplers lose their float formats, and the environment lookups -- atan2 / asin, the PDF image, the presampled
// light -- are dead code for a textured scene under a plain sky, profiles/r04_ab_coverage_chain.txt), 2 = everything. Scenes with neither (decided at upload) run the instantiation without the samplers -- four inlined copies of
// sample_texture, the environment lookups -- which are never executed there but cost the kernel 43 spilled scalar registers (round 4, profiles/r04_ab_shade_without_textures.txt:
// atrium shade 32.9 -> 31.6 ms per 64 accumulations, Cornell all-Diffuse step -2.8 %).
template <int MODELS, bool AOV, int PART = SHADE_PART_ALL, int TEXTURES = 2>
HD void shade_path(const DeviceScene& sc, const HiprCameraState& cam, int entry, const uint32_t* sobol_lds, f3 ro, const float4* origin_of_entry, f3 rd, float bsdf_pdf, f3 throughput, uint32_t bounces,
                   uint32_t last_triangle, uint32_t pixel_hash, uint32_t accumulation, float4 hit, const ShadeGeometry& geo, const HiprMaterial& mp, bool nee_kept_a_sample,
                   ShadeOutput& out) {
    // `ro`: the ray's origin where k_shade fetched it -- rays that did not hit a triangle, and every ray of the AOV entries. A surface hit is shaded from its
    // barycentrics and never reads it, except the rare rejected hit that sends the same ray on: that one loads it here (origin_of_entry).
    out.continues = out.shadow = out.shaded = out.nee_reached = out.nee_valid = false;
    out.add_radiance = mk3(0.0f);
    const uint32_t id = __float_as_uint(hit.w);
    if (PART == SHADE_PART_NEE && (id == HIPR_HIT_MISS || (id & HIPR_HIT_LIGHT))) return;
    if (id == HIPR_HIT_MISS) {
        if (AOV) {   // material_index stays 0 -> black; the depth entry adds |origin - 1e30 * direction| (SimpleRGPs.cu:227-239, 349-362)
            if (entry == HIPR_ENTRY_DEPTH) out.add_radiance = mk3(length(ro - 1e30f * rd));
            return;
        }
        // miss program (SimpleRGPs.cu:349-362): the tint, or the environment map weighted against the BSDF sample that got here
        f3 environment = mk3(sc.env_tint[0], sc.env_tint[1], sc.env_tint[2]);
        if (TEXTURES >= 2 && sc.env_map_ID) {
            environment = environment_evaluate(sc, rd);
            if (pdf_valid_not_delta(bsdf_pdf)) environment *= balance_heuristic(pdf_value(bsdf_pdf), pdf_value(environment_pdf(sc, rd)));
        }
        out.add_radiance = throughput * environment;
        return;
    }
    if (id & HIPR_HIT_LIGHT) {
        if (AOV) {
            if (entry == HIPR_ENTRY_DEPTH) out.add_radiance = mk3(length(ro - (rd * hit.x + ro)));
            if (entry == HIPR_ENTRY_DENOISER_ALBEDO) {   // the light's radiance brought into [0, 1) (SimpleRGPs.cu:180-182)
                const f3 L = light_evaluate_intersection(sc.lights[id & ~HIPR_HIT_LIGHT], ro, rd, bsdf_pdf);
                out.add_radiance = L / (mk3(1.0f) + L);
            }
            return;
        }
        f3 L = light_evaluate_intersection(sc.lights[id & ~HIPR_HIT_LIGHT], ro, rd, bsdf_pdf);
        out.add_radiance = min3(throughput, mk3(4.0f)) * L;
        return;
    }

    // --- attributes of the accepted closest hit only (TriangleAttributes.cu:35-84), from the flattened shading record ---------
    const float4 ta = geo.ta, tb = geo.tb, tc = geo.tc;
    const float4 s0 = geo.s0, s1 = geo.s1, s2 = geo.s2, s3 = geo.s3, s4 = geo.s4, s5 = geo.s5;
    const f3 p0 = {ta.x, ta.y, ta.z}, p1 = {ta.w, tb.x, tb.y}, p2 = {tb.z, tb.w, tc.x};
    const uint32_t prim = __float_as_uint(tc.z);
    const uint32_t mesh_flags = __float_as_uint(s2.w);
    const int32_t instance_id = int32_t(__float_as_uint(s1.w));
    const float u = hit.y, v = hit.z, w = 1.0f - u - v;

    f3 geometric_normal = normalize(cross(p1 - p0, p2 - p0));
    const f2 texcoord = (mesh_flags & HIPR_MESH_TEXCOORDS) ? mk2(s3.z, s3.w) * u + mk2(s4.x, s4.y) * v + mk2(s3.x, s3.y) * w : mk2(0.0f, 0.0f);

    const bool thin_walled = (mp.flags & (HIPR_MATERIAL_CUTOUT | HIPR_MATERIAL_THIN_WALLED)) != 0;
    const bool transmissive = mp.shading_model == HIPR_SHADING_TRANSMISSIVE;
    const bool hit_from_front = dot(geometric_normal, rd) < 0.0f;
    const bool backside_cull = !hit_from_front && !thin_walled && !transmissive;

    const f4 bsdf_u = sobol4f_tables(accumulation, pixel_hash, 8u * bounces + 2u, sobol_lds);   // BSDF dimension, always drawn
    const float coverage = TEXTURES ? material_coverage<TEXTURES >= 2>(sc, mp, texcoord) : material_coverage_untextured(mp);
    if (backside_cull || coverage < bsdf_u.w) {
        // rejected hit: same ray, tmin bumped past it, counters untouched (MonteCarlo.cu:159-164)
        if (PART == SHADE_PART_NEE) return;
        out.continues = true;
        if (!AOV) { const float4 o4 = *origin_of_entry; ro = mk3(o4.x, o4.y, o4.z); }
        out.o = ro; out.d = rd; out.tmin = nextafterf(hit.x, __builtin_inff()); out.bsdf_pdf = bsdf_pdf;
        out.throughput = throughput; out.bounces = bounces; out.last_triangle = last_triangle;
        return;
    }
    out.shaded = PART != SHADE_PART_NEE;
    out.nee_reached = true;

    const f3 position = p1 * u + p2 * v + p0 * w;
    f3 shading_normal = geometric_normal;
    if (mesh_flags & HIPR_MESH_NORMALS) shading_normal = normalize(mk3(s1.x, s1.y, s1.z) * u + mk3(s2.x, s2.y, s2.z) * v + mk3(s0.x, s0.y, s0.z) * w);
    f4 tint_scale = {1, 1, 1, 1};
    if (mesh_flags & HIPR_MESH_TINTS) {
        const uint32_t t0 = __float_as_uint(s4.z), t1 = __float_as_uint(s4.w), t2 = __float_as_uint(s5.x);
        const float s = 1.0f / 255.0f;
        auto ch = [](uint32_t p, int c) { return float((p >> (8 * c)) & 0xFFu); };
        tint_scale = {(ch(t1, 0) * u + ch(t2, 0) * v + ch(t0, 0) * w) * s, (ch(t1, 1) * u + ch(t2, 1) * v + ch(t0, 1) * w) * s,
                      (ch(t1, 2) * u + ch(t2, 2) * v + ch(t0, 2) * w) * s, (ch(t1, 3) * u + ch(t2, 3) * v + ch(t0, 3) * w) * s};
    }
    f3 emission = {1, 1, 1};
    if (mesh_flags & HIPR_MESH_EMISSIVE) {   // rare: per-vertex emission stays behind the instance
        const HiprInstance& inst = sc.instances[__float_as_uint(tc.y)];
        const uint32_t* idx = sc.indices + 3 * size_t(inst.index_offset + prim);
        const uint32_t i0 = idx[0], i1 = idx[1], i2 = idx[2];
        const float* e = sc.emissions + 3 * size_t(inst.vertex_offset);
        auto em = [&](uint32_t i) { return mk3(e[3 * i], e[3 * i + 1], e[3 * i + 2]); };
        emission = em(i1) * u + em(i2) * v + em(i0) * w;
    }

    geometric_normal = hit_from_front ? geometric_normal : -geometric_normal;
    shading_normal = hit_from_front ? shading_normal : -shading_normal;
    shading_normal = fix_backfacing_shading_normal(-rd, shading_normal, 0.002f);
    const Frame tbn = make_frame(shading_normal);
    const f3 wo = to_local(tbn, -rd);
    const float cos_theta = (hit_from_front || thin_walled) ? wo.z : -wo.z;

    // --- material ---------------------------------------------------------------------------------
    f4 tr = {mp.tint[0], mp.tint[1], mp.tint[2], mp.roughness};
    if (TEXTURES && mp.tint_roughness_texture_ID) tr = tr * sample_texture<TEXTURES >= 2>(sc, mp.tint_roughness_texture_ID, texcoord);
    if (TEXTURES && mp.roughness_texture_ID) tr.w *= sample_texture<TEXTURES >= 2>(sc, mp.roughness_texture_ID, texcoord).x;
    tr = tr * tint_scale;
    MaterialInputs in;
    in.tint = {tr.x, tr.y, tr.z};
    in.roughness = tr.w;
    in.specularity = mp.specularity;
    in.metallic = (TEXTURES && mp.metallic_texture_ID) ? mp.metallic * sample_texture<TEXTURES >= 2>(sc, mp.metallic_texture_ID, texcoord).x : mp.metallic;
    in.coat = mp.coat / 65535.0f;
    in.coat_roughness = mp.coat_roughness / 65535.0f;
    // PathRegularizationSettings::PDF_scale_at_accumulation (OR/PublicTypes.h:44), per path: a pass may carry several accumulations
    const float max_PDF_hint = bsdf_pdf * (cam.path_regularization_PDF_scale * (1.0f + cam.path_regularization_scale_decay * float(int(accumulation))));
    Shading shading;
    if (HIPR_HAS_DIFFUSE(MODELS) && (MODELS == 2 || mp.shading_model == HIPR_SHADING_DIFFUSE)) shading = make_diffuse(in.tint, in.roughness);
    else if (HIPR_HAS_TRANSMISSIVE(MODELS) && (MODELS == 4 || transmissive)) shading = make_transmissive(sc.tables, in, cos_theta, max_PDF_hint);
    else if (HIPR_HAS_DEFAULT(MODELS)) shading = make_default(sc.tables, in, cos_theta, max_PDF_hint);
    else shading = make_diffuse(in.tint, in.roughness);

    if (AOV && entry != HIPR_ENTRY_DEPTH) {
        // First accepted hit of the AOV entry points (SimpleRGPs.cu:265-340). The vertex tint scale takes the
        // float_to_unorm8 round trip of the payload (MonteCarlo.cu:170) before it is used.
        auto q8 = [](float v) { return float((unsigned char)(saturate(v) * 255.0f + 0.5f)) * (1.0f / 255.0f); };
        const f4 qscale = {q8(tint_scale.x), q8(tint_scale.y), q8(tint_scale.z), q8(tint_scale.w)};
        f4 base = {mp.tint[0], mp.tint[1], mp.tint[2], mp.roughness};
        if (mp.tint_roughness_texture_ID) base = base * sample_texture(sc, mp.tint_roughness_texture_ID, texcoord);
        if (mp.roughness_texture_ID) base.w *= sample_texture(sc, mp.roughness_texture_ID, texcoord).x;
        f3 value = {0, 0, 0};
        if (entry == HIPR_ENTRY_TINT) value = mk3(base.x, base.y, base.z) * mk3(qscale.x, qscale.y, qscale.z);
        else if (entry == HIPR_ENTRY_ROUGHNESS) value = mk3(base.w * qscale.w);
        else if (entry == HIPR_ENTRY_SHADING_NORMAL) value = shading_normal * 0.5f + 0.5f;
        else if (entry == HIPR_ENTRY_PRIMITIVE_ID) {
            const uint32_t instance_encoding = uint32_t(instance_id) & 0x3FFFFFFu;
            const uint32_t primitive_encoding = __brev(prim + 1u) >> 2;
            const uint32_t code = instance_encoding ^ primitive_encoding;
            auto compact_by_2 = [](uint32_t v) {
                v &= 0x09249249u; v = (v ^ (v >> 2)) & 0x030c30c3u; v = (v ^ (v >> 4)) & 0x0300f00fu; v = (v ^ (v >> 8)) & 0xff0000ffu; v = (v ^ (v >> 16)) & 0x000003ffu;
                return v;
            };
            value = mk3(float(compact_by_2(code >> 2)), float(compact_by_2(code >> 1)), float(compact_by_2(code))) / 1023.0f;
        } else if (entry == HIPR_ENTRY_ALBEDO || entry == HIPR_ENTRY_DENOISER_ALBEDO) {
            const float abs_cos = fabsf(dot(rd, shading_normal));
            const f4 trq = base * qscale;
            MaterialInputs ai = in;
            ai.tint = {trq.x, trq.y, trq.z};
            ai.roughness = trq.w;
            // the denoiser's feature image takes every material as DefaultShading (SimpleRGPs.cu:171-178)
            const bool by_model = entry == HIPR_ENTRY_ALBEDO;
            if (by_model && mp.shading_model == HIPR_SHADING_DIFFUSE) value = ai.tint;
            else if (by_model && transmissive) {
                const Shading t = make_transmissive(sc.tables, ai, abs_cos, -1.0f);
                const f2 rho = fetch_dielectric_rho(sc.tables, abs_cos, sqrtf(t.s1), t.s2);
                const float reflection = rho.y / rho.x;
                value = reflection + (1 - reflection) * t.a;
            } else {
                const Shading d = make_default(sc.tables, ai, abs_cos, -1.0f);
                const f2 rho = fetch_specular_rho(sc.tables, abs_cos, d.s0);
                value = d.a + mk3(lerp(rho.x, rho.y, d.b.x), lerp(rho.x, rho.y, d.b.y), lerp(rho.x, rho.y, d.b.z)) * d.s1;
                if (d.s2 > 0.0f) {
                    const f2 crho = fetch_specular_rho(sc.tables, abs_cos, sqrtf(d.s3));
                    value = value + lerp(crho.x, crho.y, HIPR_COAT_SPECULARITY) * d.s2;
                }
            }
        }
        out.add_radiance = value;
        return;
    }

    if (PART != SHADE_PART_NEE) out.add_radiance = throughput * emission * mk3(mp.emission[0], mp.emission[1], mp.emission[2]);
#ifndef HIPR_SHADE_TERMS
#define HIPR_SHADE_TERMS 1      // 0: round 3's evaluation and sampling functions (A/B: profiles/r04_ab_shade_terms.txt)
#endif
#if HIPR_SHADE_TERMS
    const ShadingTerms terms = shading_terms<MODELS>(shading, wo);      // what the evaluations below share: the part that depends on wo only
#endif

    // --- next event estimation: streaming RIS over the light candidates (MonteCarlo.cu:91-123) ------
    LightSample kept = light_sample_none();
    uint32_t kept_light = 0;
    if (PART != SHADE_PART_BSDF && sc.light_count != 0) {
        const f4 base = sobol4f_tables(accumulation, pixel_hash, 8u * bounces + 1u, sobol_lds);
        const int n = sc.next_event_sample_count;
        for (int s = 0; s < n; ++s) {
            const ScalarFloat4 off = as_constant(sc.sample_offsets)[s];   // uniform index: scalar load
            f4 r = {base.x + off.x, base.y + off.y, base.z + off.z, base.w + off.w};
            r = {r.x - floorf(r.x), r.y - floorf(r.y), r.z - floorf(r.z), r.w - floorf(r.w)};
            const int light_count = int(sc.light_count);
            int li = int(r.z * light_count);
            li = li > light_count - 1 ? light_count - 1 : li;
            LightSample c;
            if (TEXTURES >= 2 && (sc.lights[li].flags & HIPR_LIGHT_TYPE_MASK) == HIPR_LIGHT_PRESAMPLED_ENVIRONMENT) {   // table lookup, PresampledEnvironmentLightImpl.h:22-27
                int index = int(r.x * float(sc.env_sample_count));
                index = index > int(sc.env_sample_count) - 1 ? int(sc.env_sample_count) - 1 : index;
                const float4 a = sc.env_samples[2 * index], b = sc.env_samples[2 * index + 1];
                c = {mk3(a.x, a.y, a.z) * mk3(sc.env_tint[0], sc.env_tint[1], sc.env_tint[2]), a.w, mk3(b.x, b.y, b.z), b.w};
            } else
                c = light_sample_radiance(sc.lights[li], position, mk2(r.x, r.y));
            c.radiance *= float(light_count);
            c.radiance *= fabsf(dot(tbn.n, c.dir)) / pdf_value(c.pdf);
#if HIPR_SHADE_TERMS
            Response f = shading_evaluate_with_PDF<MODELS>(shading, terms, wo, to_local(tbn, c.dir));
#else
            Response f = shading_evaluate_with_PDF<MODELS>(shading, wo, to_local(tbn, c.dir));
#endif
            if (!pdf_is_delta(c.pdf)) c.radiance *= balance_heuristic(pdf_value(c.pdf), pdf_value(f.pdf));
            else f.f = min3(f.f, mk3(32.0f));
            c.radiance *= f.f;
            const float w_old = sum(kept.radiance), w_new = sum(c.radiance);
            const float p_new = w_new / (w_old + w_new);
            if (r.w < p_new) { kept = c; kept.radiance /= p_new; kept_light = uint32_t(li); }
            else kept.radiance /= 1.0f - p_new;
        }
        kept.radiance /= float(n);
    }
    if (PART != SHADE_PART_BSDF) {
        const f3 light_origin = offset_ray_origin(position, kept.dir, geometric_normal);
        kept.radiance *= throughput;
        if (kept.radiance.x > 0 || kept.radiance.y > 0 || kept.radiance.z > 0) {
            out.shadow = true;
            out.so = light_origin; out.sd = kept.dir; out.stmax = kept.distance; out.sradiance = kept.radiance;
            out.shadow_class = min(kept_light, SHADOW_CLASSES - 1u);
        }
        out.nee_valid = pdf_is_valid(kept.pdf);
        if (PART == SHADE_PART_NEE) return;
    }
    const bool light_sample_kept = PART == SHADE_PART_BSDF ? nee_kept_a_sample : pdf_is_valid(kept.pdf);

    // --- BSDF sampling (MonteCarlo.cu:204-232) ---------------------------------------------------
#if HIPR_SHADE_TERMS
    const Sample bs = shading_sample<MODELS>(shading, terms, wo, mk3(bsdf_u.x, bsdf_u.y, bsdf_u.z));
#else
    const Sample bs = shading_sample<MODELS>(shading, wo, mk3(bsdf_u.x, bsdf_u.y, bsdf_u.z));
#endif
    const bool is_reflection = bs.dir.z >= 0;
    f3 direction = to_world(tbn, bs.dir);
    float new_pdf = bs.pdf;
    if (pdf_is_valid(bs.pdf)) throughput *= (bs.f * fabsf(bs.dir.z)) / pdf_value(bs.pdf);
    else throughput = mk3(0.0f);
    const float cos_geometric = dot(direction, geometric_normal);
    if (is_reflection ? cos_geometric < 0.0f : cos_geometric >= 0.0f)
        direction = reflect(direction, geometric_normal);
    if (!light_sample_kept) new_pdf = pdf_disable_MIS(new_pdf);
    bounces += 1u;

    out.o = offset_ray_origin(position, direction, geometric_normal);
    out.d = direction; out.tmin = 0.0f; out.bsdf_pdf = new_pdf; out.throughput = throughput; out.bounces = bounces; out.last_triangle = id;
    out.continues = bounces <= cam.max_bounce_count && !is_black(throughput);
    if (AOV) {   // depth entry: distance from the ray origin to the offset origin of the next ray (SimpleRGPs.cu:232-236)
        out.add_radiance = mk3(length(ro - out.o));
        out.continues = out.shadow = false;
    }
}

// Persistent blocks (the launch sizes the grid to what is resident) walk the queue with a grid stride and run one iteration
// AHEAD on their inputs: while a batch of 256 paths is shaded, the path state and hit of the next batch are in flight; its
// triangle, shading record and material are requested once the hit id has arrived, behind the compaction barriers. The four
// dependent gathers of a hit (path state -> hit -> triangle / record -> material) are thereby off the critical path.
struct ShadeInputs {
    uint32_t entry;        // the queue entry (HIPR_DEAD_SLOT: none)
    uint2 meta;            // slot, last accepted triangle
    float4 o, d, t, hit;   // origin + tmin (fetched with the geometry, and only where it is used: shade_fetch_origin), direction + pdf, throughput + bounces, (t, u, v, id)
};

// `i`: the queue entry, or HIPR_DEAD_SLOT for none (past the end of the queue).
HD ShadeInputs shade_fetch_inputs(const PathState& in, const float4* hits, uint32_t i) {
    ShadeInputs r;
    r.entry = i;
    r.meta = make_uint2(HIPR_DEAD_SLOT, 0u);
    r.o = r.d = r.t = make_float4(0, 0, 0, 0);
    r.hit = make_float4(0, 0, 0, __uint_as_float(HIPR_HIT_MISS));
    if (i != HIPR_DEAD_SLOT) {
        r.meta = in.meta[i]; r.d = in.d_pdf[i]; r.hit = hits[i];
        r.t = in.thr_bounces ? in.thr_bounces[i] : make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(0u));      // camera rays (bounce 0): k_generate leaves these 16 B unwritten
    }
    return r;
}
// The queue entry that place j of the shading order holds (k_classify_hits; the queue's own order without it).
// The queue entry the kernel takes j-th: queue order, or k_classify_hits' listing [plain surface hits | coated surface hits | everything else].
struct ShadeOrder { const uint32_t* order; const uint32_t* order_coat; uint32_t plain, coated; };
HD uint32_t shade_entry(const ShadeOrder& o, uint32_t j, uint32_t n) {
    if (j >= n) return HIPR_DEAD_SLOT;
    if (!o.order) return j;
    return (j >= o.plain && j - o.plain < o.coated) ? o.order_coat[j - o.plain] : o.order[j];
}
HD bool shade_hits_triangle(const ShadeInputs& in) {
    const uint32_t id = __float_as_uint(in.hit.w);
    return in.meta.x != HIPR_DEAD_SLOT && id != HIPR_HIT_MISS && !(id & HIPR_HIT_LIGHT);
}
// The ray's origin, one stage behind the other inputs: known to be needed once the hit id is there. Misses and light hits use it (environment lookup, light
// evaluation), surface hits do not -- in the listing's order whole batches skip the 16 B (profiles/r04_ab_shade_queue_bytes.txt).
template <bool AOV>
HD void shade_fetch_origin(const PathState& in, ShadeInputs& r) {
    if (r.entry != HIPR_DEAD_SLOT && r.meta.x != HIPR_DEAD_SLOT && (AOV || !shade_hits_triangle(r))) r.o = in.o_tmin[r.entry];
}
HD ShadeGeometry shade_fetch_geometry(const DeviceScene& sc, const ShadeInputs& in) {
    ShadeGeometry g;
    g.ta = g.tb = g.tc = g.s0 = g.s1 = g.s2 = g.s3 = g.s4 = g.s5 = make_float4(0, 0, 0, 0);
    if (shade_hits_triangle(in)) {
        const uint32_t id = __float_as_uint(in.hit.w);
        const float4* sp = sc.shade_triangles + SHADE_TRIANGLE_QUADS * size_t(id);      // one 128 B line: positions, normals, uvs, tints, ids
        g.ta = sp[0]; g.tb = sp[1]; g.tc = sp[2];
        g.s0 = sp[3]; g.s1 = sp[4]; g.s2 = sp[5]; g.s3 = sp[6]; g.s4 = sp[7];
        g.s5 = make_float4(g.tc.w, 0.0f, 0.0f, 0.0f);
    }
    return g;
}
HD HiprMaterial shade_fetch_material(const DeviceScene& sc, const ShadeInputs& in, const ShadeGeometry& g) {
    return sc.materials[shade_hits_triangle(in) ? __float_as_uint(g.s0.w) : 0u];   // slot 0 is the invalid material: always readable
}

// Three waves per SIMD (<= 168 VGPRs) and three resident blocks per CU: measured on the atrium against two (the prefetching loop
// below hides most, not all, of the gather latency): 29.4 -> 25.9 ms of shade time per step; four (128 VGPRs, spills) gives nothing.
#ifndef HIPR_SHADE_WAVES
#define HIPR_SHADE_WAVES 3
#endif
#ifndef HIPR_SHADE_LDS_TABLES
#define HIPR_SHADE_LDS_TABLES 1
#endif
// One block barrier per batch (arrival-order places + last-wave reservation) instead of three (scan by thread 0 between two barriers + one to reuse the
// LDS words): atrium shade 23.5 -> 23.3 ms, Cornell 18.2 -> 17.8 ms. What remains of the 10-14 % of the loop spent at barriers is the imbalance
// between the block's four waves, not the barrier count.
#ifndef HIPR_SHADE_ONE_BARRIER
#define HIPR_SHADE_ONE_BARRIER 1
#endif
constexpr uint32_t SHADE_LDS_LIGHTS = 32;   // light arrays up to this size are copied to LDS (1.5 KB); larger ones are read from global memory
constexpr int shade_waves_per_simd(int part) { return part == SHADE_PART_ALL ? HIPR_SHADE_WAVES : HIPR_SHADE_SPLIT_WAVES; }
// ARITHMETIC (HIPR_ARITHMETIC_*) names the build of the shade unit the instantiation belongs to: shade.hip is compiled twice into one library (fast and exact arithmetic,
// hipr_set_arithmetic), and two kernels of one name would share one host-side launch stub.
template <int MODELS, bool AOV, int PART, int TEXTURES, int ARITHMETIC>
__global__ __launch_bounds__(SHADE_BLOCK, shade_waves_per_simd(PART)) void k_shade(DeviceScene sc, HiprCameraState cam, FrameInfo frame, int entry, PathState in, const float4* hits, const uint32_t* order_list, const uint32_t* order_coat, const unsigned long long* listed, PathState out,
                                                        ShadowQueue shadows, float4* radiance, const uint32_t* count_ptr, unsigned long long* out_counts,
                                                        unsigned long long* zero_a, unsigned long long* zero_b, unsigned char* nee_flags, DeviceCounters* counters) {
#if HIPR_SHADE_ONE_BARRIER
    __shared__ unsigned long long s_arrivals[2], s_totals[2];
    __shared__ uint32_t s_base[4];
#else
    __shared__ uint32_t s_cont[SHADE_BLOCK / 64], s_shad[SHADE_BLOCK / 64], s_base[2];
#endif
    __shared__ uint32_t s_sobol[SOBOL_TABLE_WORDS];
#if HIPR_SHADE_LDS_TABLES
    // The two small lookup tables every Default / Transmissive hit reads with data-dependent indices (GGX rho 32 x 32 ushort2, alpha-from-PDF 32 x 32
    // ushort: 6 KB) and the light array (RIS picks a random light per candidate) live in LDS: three to six dependent global round trips per hit become
    // LDS reads. The 32 KB dielectric table (Transmissive only) stays in global memory.
    __shared__ ushort2 s_ggx_rho[32 * 32];
    __shared__ unsigned short s_alpha[32 * 32];
    __shared__ HiprLight s_lights[SHADE_LDS_LIGHTS];
#endif
    const uint32_t n = *count_ptr;
    ShadeOrder order = {order_list, order_coat, 0u, 0u};
    if (order_list && listed) { order.plain = uint32_t(listed[0]); order.coated = uint32_t(listed[1]); }      // k_classify_hits has run on this stream
    // The counters the NEXT bounce's kernels fill start from zero; nothing on the device reads or writes them while this kernel runs (hiprenderer.hip
    // enqueue_bounce). A fill command in the stream for each costs more than this whole kernel does on a bounce of a few thousand paths.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (zero_a) *zero_a = 0ull;
        if (zero_b) { zero_b[0] = 0ull; zero_b[1] = 0ull; }      // the listing pass's two counters of the next bounce
    }
    if (blockIdx.x * SHADE_BLOCK >= n) return;
    const uint32_t stride = gridDim.x * SHADE_BLOCK;
    uint32_t base = blockIdx.x * SHADE_BLOCK;
    ShadeInputs cur = shade_fetch_inputs(in, hits, shade_entry(order, base + threadIdx.x, n));
    uint32_t next_entry = shade_entry(order, base + stride + threadIdx.x, n);      // the entries are looked up two batches ahead, the inputs one
    for (uint32_t w = threadIdx.x; w < SOBOL_TABLE_WORDS; w += SHADE_BLOCK) s_sobol[w] = sc.sobol_tables[w];
#if HIPR_SHADE_LDS_TABLES
    for (uint32_t w = threadIdx.x; w < 32 * 32; w += SHADE_BLOCK) { s_ggx_rho[w] = sc.tables.ggx_rho[w]; s_alpha[w] = sc.tables.alpha[w]; }
    const bool lights_in_lds = sc.light_count <= SHADE_LDS_LIGHTS;
    if (lights_in_lds)
        for (uint32_t w = threadIdx.x; w < sc.light_count * 12u; w += SHADE_BLOCK) reinterpret_cast<uint32_t*>(s_lights)[w] = reinterpret_cast<const uint32_t*>(sc.lights)[w];
    sc.tables.ggx_rho = s_ggx_rho;
    sc.tables.alpha = s_alpha;
    if (lights_in_lds) sc.lights = s_lights;
#endif
    ShadeGeometry geo = shade_fetch_geometry(sc, cur);
    shade_fetch_origin<AOV>(in, cur);
    HiprMaterial mat = shade_fetch_material(sc, cur, geo);
#if HIPR_SHADE_ONE_BARRIER
    if (threadIdx.x < 2) s_arrivals[threadIdx.x] = 0ull;
#endif
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t shaded_total = 0;
#if HIPR_SHADE_ONE_BARRIER
    uint32_t batch = 0;
#else
    const uint32_t wave = threadIdx.x >> 6;
#endif
    for (; base < n; base += stride) {
        // inputs of the next batch: issued now, first used after this batch has been shaded
        ShadeInputs next = shade_fetch_inputs(in, hits, next_entry);
        next_entry = shade_entry(order, base + 2u * stride + threadIdx.x, n);

        ShadeOutput so;
        so.continues = so.shadow = so.shaded = false;
#ifdef HIPR_SHADE_EXTRA_READ      // experiment: 16 (or 32) more bytes read per queue entry, coalesced -- what does a byte of queue traffic cost this kernel?
        {
            const float4 extra = shadows.radiance[base + threadIdx.x];
            if (extra.w == 12345.678f) radiance[base + threadIdx.x] = extra;
#if HIPR_SHADE_EXTRA_READ > 1
            const float4 extra2 = shadows.d_slot[base + threadIdx.x];
            if (extra2.w == 12345.678f) radiance[base + threadIdx.x] = extra2;
#endif
        }
#endif
        const uint32_t slot = cur.meta.x;
        if (slot != HIPR_DEAD_SLOT) {
            uint32_t pixel_hash, accumulation;
            path_sample_of_slot(frame, cam, slot, pixel_hash, accumulation);
            bool nee_kept_a_sample = false;
            if (PART == SHADE_PART_BSDF && shade_hits_triangle(cur)) nee_kept_a_sample = nee_flags[cur.entry] != 0;   // written for every accepted hit; read by those only
            shade_path<MODELS, AOV, PART, TEXTURES>(sc, cam, entry, s_sobol, mk3(cur.o.x, cur.o.y, cur.o.z), in.o_tmin + cur.entry, mk3(cur.d.x, cur.d.y, cur.d.z), cur.d.w, mk3(cur.t.x, cur.t.y, cur.t.z),
                                          __float_as_uint(cur.t.w), cur.meta.y, pixel_hash, accumulation, cur.hit, geo, mat, nee_kept_a_sample, so);
            if (PART == SHADE_PART_NEE && so.nee_reached) nee_flags[cur.entry] = so.nee_valid ? 1 : 0;
            if (so.add_radiance.x != 0.0f || so.add_radiance.y != 0.0f || so.add_radiance.z != 0.0f) {
                float4 acc = radiance[slot];
                acc.x += so.add_radiance.x; acc.y += so.add_radiance.y; acc.z += so.add_radiance.z;
                radiance[slot] = acc;
            }
        }
        // the hit id of the next batch has arrived by now: request its triangle and shading record
        geo = shade_fetch_geometry(sc, next);
        shade_fetch_origin<AOV>(in, next);

#if HIPR_SHADE_ONE_BARRIER
        // ---- compaction: ballot + prefix popcount in the wave; the waves of the block take their places in the block's stretch of both queues in the
        // order they get here (one LDS atomic on a packed {paths, shadow rays, arrivals} counter), the last one to arrive reserves the stretch with
        // ONE 64-bit global atomic (low word = paths that continue, high word = shadow rays) and publishes its base: one block barrier per batch.
        // The counters alternate between two LDS words by batch parity; the last wave of a batch clears the word of the next one, which no wave
        // can have reached before the barrier below.
        const unsigned long long cont_mask = wave_ballot(so.continues), shad_mask = wave_ballot(so.shadow);
        const unsigned long long lt = (1ull << lane) - 1ull;
        shaded_total += so.shaded ? 1u : 0u;
        const uint32_t parity = batch & 1u;
        // shadow rays by light: class masks of this wave (the last class takes every light from index SHADOW_CLASSES - 1 on)
        unsigned long long class_mask[SHADOW_CLASSES];
#pragma unroll
        for (uint32_t k = 0; k < SHADOW_CLASSES; ++k) class_mask[k] = SHADOW_CLASSES == 1 ? shad_mask : wave_ballot(so.shadow && so.shadow_class == k);
        // packed per-wave counts: paths that continue (bits 0-9), shadow rays of class k (bits 10 + 10 k ...), arrivals (bits 50-52); a block holds at most 256 of each
        unsigned long long before_me = 0ull;
        if (lane == 0) {
            unsigned long long mine = (unsigned long long)__popcll(cont_mask) | (1ull << 50);
#pragma unroll
            for (uint32_t k = 0; k < SHADOW_CLASSES; ++k) mine |= (unsigned long long)__popcll(class_mask[k]) << (10u + 10u * k);
            before_me = atomicAdd(&s_arrivals[parity], mine);
            if ((before_me >> 50) == SHADE_BLOCK / 64 - 1) {   // the last wave of the batch
                const unsigned long long total = before_me + mine;
                const unsigned long long c = total & 0x3FFull;
                unsigned long long sh = 0ull;
#pragma unroll
                for (uint32_t k = 0; k < SHADOW_CLASSES; ++k) sh += (total >> (10u + 10u * k)) & 0x3FFull;
                const unsigned long long base = (c | sh) ? atomicAdd(out_counts, (sh << 32) | c) : 0ull;
                s_base[2 * parity] = uint32_t(base);
                s_base[2 * parity + 1] = uint32_t(base >> 32);
                s_totals[parity] = total;
                s_arrivals[1u - parity] = 0ull;
            }
        }
        before_me = __shfl(before_me, 0);
        mat = shade_fetch_material(sc, next, geo);
        __syncthreads();
        const uint32_t cont_slot = s_base[2 * parity] + uint32_t(before_me & 0x3FFull) + __popcll(cont_mask & lt);
        // this lane's shadow ray: after the block's rays of lower classes, after the rays of its class from the waves that arrived earlier, after the lanes below it
        uint32_t shad_slot = s_base[2 * parity + 1];
        {
            const unsigned long long total = s_totals[parity];
            const uint32_t mine = so.shadow ? so.shadow_class : 0u;
#pragma unroll
            for (uint32_t k = 0; k < SHADOW_CLASSES; ++k) {
                if (k < mine) shad_slot += uint32_t((total >> (10u + 10u * k)) & 0x3FFull);
                if (k == mine) shad_slot += uint32_t((before_me >> (10u + 10u * k)) & 0x3FFull) + __popcll(class_mask[k] & lt);
            }
        }
        ++batch;
#else
        // ---- compaction: ballot + prefix popcount in the wave, LDS scan over the block's waves, one atomic for both queues
        const unsigned long long cont_mask = wave_ballot(so.continues), shad_mask = wave_ballot(so.shadow);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lane == 0) { s_cont[wave] = __popcll(cont_mask); s_shad[wave] = __popcll(shad_mask); }
        shaded_total += so.shaded ? 1u : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t c = 0, s = 0;
            for (int wv = 0; wv < SHADE_BLOCK / 64; ++wv) { uint32_t t = s_cont[wv]; s_cont[wv] = c; c += t; t = s_shad[wv]; s_shad[wv] = s; s += t; }
            // one 64-bit atomic reserves space in both queues: low word = paths that continue, high word = shadow rays
            const unsigned long long before = (c | s) ? atomicAdd(out_counts, ((unsigned long long)s << 32) | c) : 0ull;
            s_base[0] = uint32_t(before);
            s_base[1] = uint32_t(before >> 32);
        }
        mat = shade_fetch_material(sc, next, geo);
        __syncthreads();
        const uint32_t cont_slot = s_base[0] + s_cont[wave] + __popcll(cont_mask & lt), shad_slot = s_base[1] + s_shad[wave] + __popcll(shad_mask & lt);
#endif
        if (so.continues) {
            const uint32_t j = cont_slot;
            out.o_tmin[j] = make_float4(so.o.x, so.o.y, so.o.z, so.tmin);
            out.d_pdf[j] = make_float4(so.d.x, so.d.y, so.d.z, so.bsdf_pdf);
            out.thr_bounces[j] = make_float4(so.throughput.x, so.throughput.y, so.throughput.z, __uint_as_float(so.bounces));
            out.meta[j] = make_uint2(slot, so.last_triangle);
        }
        if (so.shadow) {
            const uint32_t j = shad_slot;
            shadows.o_tmax[j] = make_float4(so.so.x, so.so.y, so.so.z, so.stmax);
            shadows.d_slot[j] = make_float4(so.sd.x, so.sd.y, so.sd.z, __uint_as_float(slot));
            shadows.radiance[j] = make_float4(so.sradiance.x, so.sradiance.y, so.sradiance.z, 0.0f);
        }
        cur = next;
#if !HIPR_SHADE_ONE_BARRIER
        __syncthreads();
#endif
    }
    wave_add(&counters->shaded_hits, shaded_total);
}

} // namespace hipr
