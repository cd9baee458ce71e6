// kernels.h -- the wavefront path tracing kernels (gfx950 / MI355X).
//
// Stage            kernel               reference semantics carried (OR = extensions/OptiXRenderer/OptiXRenderer)
//   K1 generate    k_generate           OR/Shading/SimpleRGPs.cu:44-72 camera ray + payload init
//   K2 closest     k_trace_closest      OptiX rtTrace(MonteCarlo) + OR/Shading/LightSources/LightSources.cu:31-70
//   K3 shade       k_shade              OR/Shading/MonteCarlo.cu:61-233,291-302 + SimpleRGPs.cu:349-362 + TriangleAttributes.cu:35-84
//   K4 shadow      k_trace_shadow       OptiX rtTrace(Shadow) + OR/Shading/MonteCarlo.cu:278-285
//   K6 accumulate  k_accumulate         OR/Shading/SimpleRGPs.cu:74-107
// Stream compaction (K5) is fused into K3: wave ballot + LDS prefix over the block's waves + one
// global atomic per block and queue.
//
// Data layout: every per-path quantity is a float4 / uint4 SoA array so that a wave reads and
// writes 1 KiB contiguous per instruction. Paths are physically compacted each bounce (double
// buffered), so trace and shade always read coalesced. BVH nodes are 64 B (one lane = four
// dwordx4 loads), triangles 48 B. The per-lane traversal stack lives in LDS, laid out
// [depth][lane] so lane l always hits bank l % 32 (conflict free).
#pragma once

#include "device_shading.h"

namespace hipr {

#define HIPR_DEAD_SLOT 0xFFFFFFFFu
#define HIPR_HIT_MISS 0xFFFFFFFFu
#define HIPR_HIT_LIGHT 0x80000000u
#define HIPR_NO_TRIANGLE 0xFFFFFFFFu

constexpr int TRACE_BLOCK = 128;   // 2 waves; LDS stack = STACK * 128 * 4 B
constexpr int SHADE_BLOCK = 256;

struct DeviceScene {
    const float4* nodes;
    const float4* triangles;
    const HiprInstance* instances;
    const uint32_t* indices;
    const float4* geometry;
    const float2* texcoords;
    const uint32_t* tints;
    const float* emissions;
    const HiprMaterial* materials;
    const HiprLight* lights;
    const HiprTexture* textures;
    const uint8_t* texels;
    const float4* sample_offsets;
    DeviceTables tables;
    uint32_t node_count, triangle_count, light_count;
    float env_tint[3];
    int next_event_sample_count;
};

struct PathState {
    float4* o_tmin;       // origin.xyz, tmin
    float4* d_pdf;        // direction.xyz, bsdf pdf
    float4* thr_bounces;  // throughput.xyz, bits(bounces)
    uint4* meta;          // slot, last accepted triangle, pixel hash, accumulation
};

struct ShadowQueue {
    float4* o_tmax;       // origin.xyz, tmax
    float4* d_slot;       // direction.xyz, bits(slot)
    float4* radiance;     // rgb, unused
};

struct FrameInfo {
    uint32_t width, height, tiles_x, tiles_total, tile_phase, tile_stride, owned_tiles, samples_per_pass;
};

struct DeviceCounters {
    unsigned long long shaded_hits, closest_nodes, closest_triangles, shadow_nodes, shadow_triangles;
};

// Blocks that share an XCD (blockIdx % 8) get a contiguous range of chunks so that spatially
// neighbouring rays reuse the same L2 (the BVH top and the local geometry).
HD uint32_t xcd_chunk(uint32_t block, uint32_t grid) {
    uint32_t per_xcd = (grid + 7u) >> 3;
    return (block & 7u) * per_xcd + (block >> 3);
}

// ---------------------------------------------------------------------------------------------
// K1: camera rays
// ---------------------------------------------------------------------------------------------
HD bool owned_pixel(const FrameInfo& f, uint32_t k, uint32_t& x, uint32_t& y) {
    uint32_t tile = (k >> 6) * f.tile_stride + f.tile_phase;
    uint32_t lane = k & 63u;
    x = (tile % f.tiles_x) * 8u + (lane & 7u);
    y = (tile / f.tiles_x) * 8u + (lane >> 3);
    return tile < f.tiles_total && x < f.width && y < f.height;
}

HD void camera_ray(const HiprCameraState& cam, uint32_t x, uint32_t y, uint32_t width, uint32_t height, uint32_t accumulation,
                   uint32_t pixel_hash, f3& origin, f3& direction) {
    float jx = 0.5f, jy = 0.5f;
    if (accumulation != 0) {
        f4 s = sobol4f(accumulation, pixel_hash, 0u);
        jx = s.x; jy = s.y;
    }
    float vx = (float(x) + jx) / float(width), vy = (float(y) + jy) / float(height);
    float nx = vx * 2.0f - 1.0f, ny = vy * 2.0f - 1.0f;
    const float* m = cam.inverse_view_projection_matrix;
    float wx = m[0] * nx + m[1] * ny + m[2] * -1.0f + m[3] * 1.0f;
    float wy = m[4] * nx + m[5] * ny + m[6] * -1.0f + m[7] * 1.0f;
    float wz = m[8] * nx + m[9] * ny + m[10] * -1.0f + m[11] * 1.0f;
    float ww = m[12] * nx + m[13] * ny + m[14] * -1.0f + m[15] * 1.0f;
    origin = mk3(wx, wy, wz) / ww;
    const float* p = cam.inverse_projection_matrix;
    f3 v = {p[0] * nx + p[1] * ny + p[2] * 1.0f + p[3] * 1.0f, p[4] * nx + p[5] * ny + p[6] * 1.0f + p[7] * 1.0f,
            p[8] * nx + p[9] * ny + p[10] * 1.0f + p[11] * 1.0f};
    const float* r = cam.view_to_world_rotation;
    direction = normalize(mk3(r[0] * v.x + r[1] * v.y + r[2] * v.z, r[3] * v.x + r[4] * v.y + r[5] * v.z, r[6] * v.x + r[7] * v.y + r[8] * v.z));
}

__global__ __launch_bounds__(256) void k_generate(FrameInfo frame, HiprCameraState cam, PathState out, float4* radiance, uint32_t n_paths) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_paths) return;
    uint32_t per_sample = frame.owned_tiles * 64u;
    uint32_t s = p / per_sample, k = p - s * per_sample;
    uint32_t x, y;
    bool valid = owned_pixel(frame, k, x, y);
    uint32_t accumulation = cam.accumulations + s;
    uint32_t pixel_hash = pcg2d_x(x, y);
    f3 o = {0, 0, 0}, d = {0, 0, 1};
    if (valid) camera_ray(cam, x, y, frame.width, frame.height, accumulation, pixel_hash, o, d);
    out.o_tmin[p] = make_float4(o.x, o.y, o.z, 0.0f);
    out.d_pdf[p] = make_float4(d.x, d.y, d.z, -1.0f);           // bsdf_PDF = delta_dirac(1)
    out.thr_bounces[p] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(0u));
    out.meta[p] = make_uint4(valid ? p : HIPR_DEAD_SLOT, HIPR_NO_TRIANGLE, pixel_hash, accumulation);
    radiance[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// BVH2 traversal shared by K2 and K4 (DESIGN.md "Traversal order" is the specification both this
// and the oracle implement; node / triangle visit counts therefore agree exactly).
// ---------------------------------------------------------------------------------------------
HD bool slab(f3 inv, f3 ood, float lox, float hix, float loy, float hiy, float loz, float hiz, float tmin, float tmax, float& tnear) {
    float x0 = fmaf(lox, inv.x, -ood.x), x1 = fmaf(hix, inv.x, -ood.x);
    float y0 = fmaf(loy, inv.y, -ood.y), y1 = fmaf(hiy, inv.y, -ood.y);
    float z0 = fmaf(loz, inv.z, -ood.z), z1 = fmaf(hiz, inv.z, -ood.z);
    tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
    float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    tfar = fminf(tfar, tmax) * 1.0000004f;   // ~3 ulp slack: boxes and hits that tie within rounding are still visited
    return tnear <= tfar;
}

HD bool intersect_triangle(f3 v0, f3 v1, f3 v2, f3 o, f3 d, float& t, float& u, float& v) {
    f3 e1 = v1 - v0, e2 = v2 - v0;
    f3 p = cross_fma(d, e2);
    float det = dot_fma(e1, p);
    if (!(det != 0.0f)) return false;
    float inv = 1.0f / det;
    f3 tv = o - v0;
    u = dot_fma(tv, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    f3 q = cross_fma(tv, e1);
    v = dot_fma(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = dot_fma(e2, q) * inv;
    return true;
}

// `stack` points at this lane's column of the LDS stack; entry k is stack[k * STRIDE].
template <int STRIDE, typename LeafFn>
HD void traverse(const DeviceScene& sc, f3 o, f3 d, float tmin, const float& tmax, int* stack, uint32_t& nodes_visited, LeafFn&& leaf) {
    if (sc.node_count == 0) return;
    f3 sd = {fabsf(d.x) > 1e-20f ? d.x : copysignf(1e-20f, d.x), fabsf(d.y) > 1e-20f ? d.y : copysignf(1e-20f, d.y),
             fabsf(d.z) > 1e-20f ? d.z : copysignf(1e-20f, d.z)};
    f3 inv = {1.0f / sd.x, 1.0f / sd.y, 1.0f / sd.z};
    f3 ood = o * inv;
    int sp = 0;
    int cur = 0;
    for (;;) {
        const float4* n = sc.nodes + 4 * size_t(cur);
        const float4 n0 = n[0], n1 = n[1], n2 = n[2];
        const float4 n3 = n[3];
        ++nodes_visited;
        float t0, t1;
        bool h0 = slab(inv, ood, n0.x, n0.y, n0.z, n0.w, n2.x, n2.y, tmin, tmax, t0);
        bool h1 = slab(inv, ood, n1.x, n1.y, n1.z, n1.w, n2.z, n2.w, tmin, tmax, t1);
        int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
        if (h0 && h1 && t1 < t0) { int tmp = c0; c0 = c1; c1 = tmp; }
        if (!h0 && h1) { c0 = c1; h0 = true; h1 = false; }
        int next = INT32_MIN;
        bool stop = false;
        if (h0) {
            if (c0 < 0) stop = leaf(uint32_t(~c0));
            else next = c0;
        }
        if (h1 && !stop) {
            if (c1 < 0) stop = leaf(uint32_t(~c1));
            else if (next == INT32_MIN) next = c1;
            else { stack[sp * STRIDE] = c1; ++sp; }
        }
        if (stop) return;
        if (next == INT32_MIN) {
            if (sp == 0) return;
            --sp;
            next = stack[sp * STRIDE];
        }
        cur = next;
    }
}

HD void wave_add(unsigned long long* dst, uint32_t v) {
    // one atomic per wave: butterfly reduction over the 64 lanes
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

// ---------------------------------------------------------------------------------------------
// K2: closest hit
// ---------------------------------------------------------------------------------------------
HD float4 closest_hit(const DeviceScene& sc, f3 o, f3 d, float tmin, uint32_t skip, int* stack_col, uint32_t& nodes, uint32_t& tris) {
    float best_t = __builtin_inff();
    float best_u = 0, best_v = 0;
    uint32_t best_id = HIPR_HIT_MISS;
    traverse<TRACE_BLOCK>(sc, o, d, tmin, best_t, stack_col, nodes, [&](uint32_t leaf) {
        uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + count; ++i) {
            ++tris;
            const float4* tp = sc.triangles + 3 * size_t(i);
            float4 a = tp[0], b = tp[1], c = tp[2];
            if (i == skip) continue;
            float t, u, v;
            if (!intersect_triangle(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, t, u, v)) continue;
            if (!(t > tmin)) continue;
            if (t < best_t || (t == best_t && i < best_id)) { best_t = t; best_u = u; best_v = v; best_id = i; }
        }
        return false;
    });
    // Analytic area lights take part in closest-hit selection (LightSources.cu:31-70).
    for (uint32_t li = 0; li < sc.light_count; ++li) {
        const HiprLight& l = sc.lights[li];
        uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
        float t = -1e30f;
        if (type == HIPR_LIGHT_SPHERE) {
            if (!(l.data[6] > 0.0f)) continue;
            t = ray_sphere(o, d, L3(l, 3), l.data[6]);
        } else if (type == HIPR_LIGHT_SPOT) {
            if (!(l.data[6] > 0.0f)) continue;
            t = ray_disk(o, d, L3(l, 3), L3(l, 7), l.data[6]);
        } else
            continue;
        if (t > tmin && t < best_t) { best_t = t; best_u = 0; best_v = 0; best_id = HIPR_HIT_LIGHT | li; }
    }
    return make_float4(best_t, best_u, best_v, __uint_as_float(best_id));
}

template <int STACK, bool INSTRUMENT>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_closest(DeviceScene sc, PathState in, float4* hits, const uint32_t* count_ptr,
                                                               DeviceCounters* counters) {
    __shared__ int s_stack[STACK * TRACE_BLOCK];
    const uint32_t n = *count_ptr;
    const uint32_t chunks = (n + TRACE_BLOCK - 1) / TRACE_BLOCK;
    uint32_t nodes = 0, tris = 0;
    for (uint32_t c = xcd_chunk(blockIdx.x, gridDim.x); c < chunks; c += ((gridDim.x + 7u) >> 3) * 8u) {
        uint32_t i = c * TRACE_BLOCK + threadIdx.x;
        if (i >= n) continue;
        uint4 meta = in.meta[i];
        if (meta.x == HIPR_DEAD_SLOT) { hits[i] = make_float4(0, 0, 0, __uint_as_float(HIPR_HIT_MISS)); continue; }
        float4 o = in.o_tmin[i], d = in.d_pdf[i];
        hits[i] = closest_hit(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), o.w, meta.y, s_stack + threadIdx.x, nodes, tris);
    }
    if (INSTRUMENT) {
        wave_add(&counters->closest_nodes, nodes);
        wave_add(&counters->closest_triangles, tris);
    }
}

// ---------------------------------------------------------------------------------------------
// Textures and materials (software samplers; OR/Renderer.cpp:703-751, OR/Types.h:389-414)
// ---------------------------------------------------------------------------------------------
HD float srgb_to_linear(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }

HD f4 fetch_texel(const DeviceScene& sc, const HiprTexture& tex, int x, int y) {
    const uint8_t* base = sc.texels + tex.texel_offset;
    size_t i = size_t(y) * tex.width + size_t(x);
    f4 r;
    if (tex.format == HIPR_TEXEL_R8) { r = {base[i] / 255.0f, 0, 0, 1}; }
    else if (tex.format == HIPR_TEXEL_RGBA8) {
        uint32_t p = reinterpret_cast<const uint32_t*>(base)[i];
        r = {(p & 0xFFu) / 255.0f, ((p >> 8) & 0xFFu) / 255.0f, ((p >> 16) & 0xFFu) / 255.0f, (p >> 24) / 255.0f};
    } else if (tex.format == HIPR_TEXEL_R32F) { r = {reinterpret_cast<const float*>(base)[i], 0, 0, 1}; }
    else { float4 v = reinterpret_cast<const float4*>(base)[i]; r = {v.x, v.y, v.z, v.w}; }
    if (tex.is_sRGB) {
        r.x = srgb_to_linear(r.x);
        if (tex.format == HIPR_TEXEL_RGBA8 || tex.format == HIPR_TEXEL_RGBA32F) { r.y = srgb_to_linear(r.y); r.z = srgb_to_linear(r.z); }
    }
    return r;
}
HD int wrap_coord(int i, int n, int repeat) {
    if (repeat) { i %= n; return i < 0 ? i + n : i; }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
HD f4 sample_texture(const DeviceScene& sc, int id, f2 uv) {
    const HiprTexture tex = sc.textures[id];
    int w = int(tex.width), h = int(tex.height);
    if (tex.filter & 1) {
        float xb = uv.x * w - 0.5f, yb = uv.y * h - 0.5f;
        float xf = floorf(xb), yf = floorf(yb);
        float fx = xb - xf, fy = yb - yf;
        int x0 = wrap_coord(int(xf), w, tex.wrap_u), x1 = wrap_coord(int(xf) + 1, w, tex.wrap_u);
        int y0 = wrap_coord(int(yf), h, tex.wrap_v), y1 = wrap_coord(int(yf) + 1, h, tex.wrap_v);
        f4 a = fetch_texel(sc, tex, x0, y0), b = fetch_texel(sc, tex, x1, y0);
        f4 c = fetch_texel(sc, tex, x0, y1), d = fetch_texel(sc, tex, x1, y1);
        f4 lo = a + (b - a) * fx, hi = c + (d - c) * fx;
        return lo + (hi - lo) * fy;
    }
    int x = wrap_coord(int(floorf(uv.x * w)), w, tex.wrap_u);
    int y = wrap_coord(int(floorf(uv.y * h)), h, tex.wrap_v);
    return fetch_texel(sc, tex, x, y);
}
HD float material_coverage(const DeviceScene& sc, const HiprMaterial& m, f2 uv) {
    float tex = 1.0f;
    if (m.coverage_texture_ID) tex = sample_texture(sc, m.coverage_texture_ID, uv).x;
    if (m.flags & HIPR_MATERIAL_CUTOUT) return tex < m.coverage ? 0.0f : 1.0f;
    return m.coverage * tex;
}

HD f2 triangle_texcoord(const DeviceScene& sc, const HiprInstance& inst, uint32_t prim, float u, float v) {
    if (!(inst.mesh_flags & HIPR_MESH_TEXCOORDS)) return {0, 0};
    const uint32_t* idx = sc.indices + 3 * size_t(inst.index_offset + prim);
    const float2* tc = sc.texcoords + inst.vertex_offset;
    float w = 1.0f - u - v;
    float2 t0 = tc[idx[0]], t1 = tc[idx[1]], t2 = tc[idx[2]];
    return mk2(t1.x, t1.y) * u + mk2(t2.x, t2.y) * v + mk2(t0.x, t0.y) * w;
}

HD f3 decode_octahedral(float packed) {
    uint32_t bits = __float_as_uint(packed);
    float fx = float(short(bits & 0xFFFFu)), fy = float(short(bits >> 16));
    f3 n = {fx, fy, 32767.0f - fabsf(fx) - fabsf(fy)};
    float t = fmaxf(-n.z, 0.0f);
    n.x += n.x >= 0 ? -t : t;
    n.y += n.y >= 0 ? -t : t;
    return normalize(n);
}

// ---------------------------------------------------------------------------------------------
// K4: shadow rays (any-hit accumulation, early out on opaque hits)
// ---------------------------------------------------------------------------------------------
HD f3 shadow_transmittance(const DeviceScene& sc, f3 o, f3 d, float tmin, float tmax, f3 radiance, int* stack_col, uint32_t& nodes, uint32_t& tris) {
    traverse<TRACE_BLOCK>(sc, o, d, tmin, tmax, stack_col, nodes, [&](uint32_t leaf) {
        uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + count; ++i) {
            ++tris;
            const float4* tp = sc.triangles + 3 * size_t(i);
            float4 a = tp[0], b = tp[1], c = tp[2];
            float t, u, v;
            if (!intersect_triangle(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, t, u, v)) continue;
            if (!(t > tmin && t < tmax)) continue;
            float coverage = 1.0f;
            if (!(__float_as_uint(c.w) & HIPR_TRIANGLE_OPAQUE)) {
                const HiprInstance& inst = sc.instances[__float_as_uint(c.y)];
                coverage = material_coverage(sc, sc.materials[inst.material_index], triangle_texcoord(sc, inst, __float_as_uint(c.z), u, v));
            }
            radiance *= 1.0f - coverage;
            if (radiance.x < 0.0000001f && radiance.y < 0.0000001f && radiance.z < 0.0000001f) {
                radiance = mk3(0.0f);
                return true;
            }
        }
        return false;
    });
    return radiance;
}

template <int STACK, bool INSTRUMENT>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_shadow(DeviceScene sc, ShadowQueue q, float4* radiance, const uint32_t* count_ptr,
                                                              DeviceCounters* counters) {
    __shared__ int s_stack[STACK * TRACE_BLOCK];
    const uint32_t n = *count_ptr;
    const uint32_t chunks = (n + TRACE_BLOCK - 1) / TRACE_BLOCK;
    uint32_t nodes = 0, tris = 0;
    for (uint32_t c = xcd_chunk(blockIdx.x, gridDim.x); c < chunks; c += ((gridDim.x + 7u) >> 3) * 8u) {
        uint32_t i = c * TRACE_BLOCK + threadIdx.x;
        if (i >= n) continue;
        float4 o = q.o_tmax[i], d = q.d_slot[i], r = q.radiance[i];
        f3 result = shadow_transmittance(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), 0.0f, o.w, mk3(r.x, r.y, r.z), s_stack + threadIdx.x, nodes, tris);
        uint32_t slot = __float_as_uint(d.w);
        float4 acc = radiance[slot];
        acc.x += result.x; acc.y += result.y; acc.z += result.z;
        radiance[slot] = acc;
    }
    if (INSTRUMENT) {
        wave_add(&counters->shadow_nodes, nodes);
        wave_add(&counters->shadow_triangles, tris);
    }
}

// ---------------------------------------------------------------------------------------------
// K2 / K4, persistent form. Same per-ray visiting order as traverse() above (so results and the
// node / triangle counters are identical), restructured for wave64 efficiency:
//   * persistent waves: a wave claims TRACE_CHUNK consecutive rays with one global atomic and
//     refills finished lanes from its private range, so a wave is never held hostage by its longest
//     ray (rays visit between a handful and a few hundred nodes);
//   * one work item per loop iteration and lane -- either one BVH node or ONE triangle. Leaves are
//     stack items like inner nodes, so a lane that reached a leaf does not make the other 63 lanes
//     wait for up to four sequential triangle tests.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t TRACE_CHUNK_MAX = 512; // rays a wave claims per global atomic (the launch passes min(this, fair share))
// The ray range of a launch is cut into TRACE_SHARDS contiguous shards, each with its own claim counter on its
// own 64 B line: 64 x the atomic throughput of a single word (one word saturates near 88 atomics/us), and blocks
// that start on the same shard work on neighbouring rays. A wave that drains its shard steals from the next.
constexpr uint32_t TRACE_SHARDS = 64;
constexpr uint32_t TRACE_SHARD_STRIDE = 16;   // uint32 words between shard counters

template <int STACK, bool SHADOW, bool INSTRUMENT>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_persistent(DeviceScene sc, PathState in, float4* hits, ShadowQueue q, float4* radiance,
                                                                  const uint32_t* count_ptr, uint32_t* work_counter, uint32_t chunk_size,
                                                                  int refill_below, DeviceCounters* counters) {
    __shared__ int s_stack[STACK * TRACE_BLOCK];
    int* stack = s_stack + threadIdx.x;
    const uint32_t n = *count_ptr;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;

    uint32_t chunk_next = 0, chunk_end = 0;   // wave uniform
    uint32_t shard = blockIdx.x % TRACE_SHARDS, shards_drained = 0;
    bool exhausted = false;

    bool active = false, finished = false;    // finished: traversal done, result still in registers
    uint32_t ray_index = 0;
    f3 o = {0, 0, 0}, d = {0, 0, 1}, inv = {0, 0, 0}, ood = {0, 0, 0};
    float tmin = 0.0f, tmax = 0.0f;                       // tmax: best distance so far (closest) or the ray extent (shadow)
    float best_u = 0.0f, best_v = 0.0f;
    uint32_t best_id = HIPR_HIT_MISS, skip = HIPR_NO_TRIANGLE;
    f3 rad = {0, 0, 0};
    uint32_t slot = 0;
    int cur = 0, sp = 0;
    uint32_t tri_cur = 0, tri_end = 0;
    uint32_t nodes = 0, tris = 0;

    auto set_item = [&](int item) {
        if (item >= 0) { cur = item; tri_cur = tri_end = 0; }
        else { const uint32_t leaf = uint32_t(~item); tri_cur = leaf >> 3; tri_end = tri_cur + (leaf & 7u) + 1u; }
    };
    auto pop_next = [&]() {
        if (sp == 0) { active = false; finished = true; return; }
        --sp;
        set_item(stack[sp * TRACE_BLOCK]);
    };

    for (;;) {
        // ---- retire finished lanes (converged: every lane of the wave is here) --------------------------------------
        if (finished) {
            if constexpr (SHADOW) {
                float4 acc = radiance[slot];
                acc.x += rad.x; acc.y += rad.y; acc.z += rad.z;
                radiance[slot] = acc;
            } else {
                for (uint32_t li = 0; li < sc.light_count; ++li) {   // analytic area lights, LightSources.cu:31-70
                    const HiprLight& l = sc.lights[li];
                    const uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
                    float t = -1e30f;
                    if (type == HIPR_LIGHT_SPHERE) { if (!(l.data[6] > 0.0f)) continue; t = ray_sphere(o, d, L3(l, 3), l.data[6]); }
                    else if (type == HIPR_LIGHT_SPOT) { if (!(l.data[6] > 0.0f)) continue; t = ray_disk(o, d, L3(l, 3), L3(l, 7), l.data[6]); }
                    else continue;
                    if (t > tmin && t < tmax) { tmax = t; best_u = 0; best_v = 0; best_id = HIPR_HIT_LIGHT | li; }
                }
                hits[ray_index] = make_float4(tmax, best_u, best_v, __uint_as_float(best_id));
            }
            finished = false;
        }
        // ---- refill idle lanes from the wave's private range -------------------------------------------------------
        const unsigned long long idle = __ballot(!active);
        if (idle && !exhausted) {
            while (chunk_next >= chunk_end && !exhausted) {
                const uint32_t shard_begin = uint32_t((unsigned long long)n * shard / TRACE_SHARDS);
                const uint32_t shard_end = uint32_t((unsigned long long)n * (shard + 1u) / TRACE_SHARDS);
                uint32_t base = 0;
                if (lane == 0) {
                    uint32_t* counter = work_counter + shard * TRACE_SHARD_STRIDE;
                    // probe with a plain L2 load first: at the end of a launch every wave walks the drained shards
                    base = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (base < shard_end - shard_begin) base = atomicAdd(counter, chunk_size);
                }
                base = __shfl(base, 0);
                if (base < shard_end - shard_begin) { chunk_next = shard_begin + base; chunk_end = min(chunk_next + chunk_size, shard_end); }
                else {
                    shard = (shard + 1u) % TRACE_SHARDS;
                    if (++shards_drained >= TRACE_SHARDS) exhausted = true;
                }
            }
            if (chunk_next < chunk_end) {
                const uint32_t idx = chunk_next + __popcll(idle & lt);
                if (!active && idx < chunk_end) {
                    ray_index = idx;
                    bool dead = false;
                    float4 ro, rdv;
                    if constexpr (SHADOW) {
                        ro = q.o_tmax[idx]; rdv = q.d_slot[idx];
                        const float4 rr = q.radiance[idx];
                        rad = mk3(rr.x, rr.y, rr.z);
                        slot = __float_as_uint(rdv.w);
                        tmin = 0.0f; tmax = ro.w;
                    } else {
                        const uint4 meta = in.meta[idx];
                        dead = meta.x == HIPR_DEAD_SLOT;
                        skip = meta.y;
                        ro = in.o_tmin[idx]; rdv = in.d_pdf[idx];
                        tmin = ro.w; tmax = __builtin_inff();
                        best_u = best_v = 0.0f; best_id = HIPR_HIT_MISS;
                    }
                    if (dead) hits[idx] = make_float4(0, 0, 0, __uint_as_float(HIPR_HIT_MISS));
                    else {
                        o = mk3(ro.x, ro.y, ro.z); d = mk3(rdv.x, rdv.y, rdv.z);
                        const f3 sd = {fabsf(d.x) > 1e-20f ? d.x : copysignf(1e-20f, d.x), fabsf(d.y) > 1e-20f ? d.y : copysignf(1e-20f, d.y),
                                       fabsf(d.z) > 1e-20f ? d.z : copysignf(1e-20f, d.z)};
                        inv = {1.0f / sd.x, 1.0f / sd.y, 1.0f / sd.z};
                        ood = o * inv;
                        sp = 0; cur = 0; tri_cur = tri_end = 0;
                        if (sc.node_count == 0) finished = true;
                        else active = true;
                    }
                }
                chunk_next = min(chunk_next + uint32_t(__popcll(idle)), chunk_end);
            }
        }
        unsigned long long busy = __ballot(active);
        if (!busy) {
            if (__ballot(finished)) continue;
            if (exhausted) break;
            continue;
        }

        // ---- traverse. Every iteration runs ONE of the two blocks -- the one more lanes are waiting for -- so the
        // ---- 64 lanes are not serialised through both; a lane's own visiting order is unchanged.
        do {
            const bool tri_mode = active && tri_cur < tri_end;
            const bool node_mode = active && !tri_mode;
            const unsigned long long tmask = __ballot(tri_mode), nmask = __ballot(node_mode);
            if (__popcll(tmask) > __popcll(nmask)) {
                if (tri_mode) {
                    const uint32_t i = tri_cur++;
                    ++tris;
                    const float4* tp = sc.triangles + 3 * size_t(i);
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    float t, u, v;
                    bool stop = false;
                    if ((SHADOW || i != skip) && intersect_triangle(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, t, u, v)) {
                        if constexpr (SHADOW) {
                            if (t > tmin && t < tmax) {
                                float coverage = 1.0f;
                                if (!(__float_as_uint(c.w) & HIPR_TRIANGLE_OPAQUE)) {
                                    const HiprInstance& inst = sc.instances[__float_as_uint(c.y)];
                                    coverage = material_coverage(sc, sc.materials[inst.material_index], triangle_texcoord(sc, inst, __float_as_uint(c.z), u, v));
                                }
                                rad *= 1.0f - coverage;
                                if (rad.x < 0.0000001f && rad.y < 0.0000001f && rad.z < 0.0000001f) { rad = mk3(0.0f); stop = true; }
                            }
                        } else {
                            if (t > tmin && (t < tmax || (t == tmax && i < best_id))) { tmax = t; best_u = u; best_v = v; best_id = i; }
                        }
                    }
                    if (stop) { active = false; finished = true; }
                    else if (tri_cur == tri_end) pop_next();
                }
            } else if (node_mode) {
                const float4* np = sc.nodes + 4 * size_t(cur);
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                ++nodes;
                float t0, t1;
                bool h0 = slab(inv, ood, n0.x, n0.y, n0.z, n0.w, n2.x, n2.y, tmin, tmax, t0);
                bool h1 = slab(inv, ood, n1.x, n1.y, n1.z, n1.w, n2.z, n2.w, tmin, tmax, t1);
                int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                if (h0 && h1 && t1 < t0) { const int tmp = c0; c0 = c1; c1 = tmp; }
                if (!h0 && h1) { c0 = c1; h0 = true; h1 = false; }
                if (h0 && h1) {
                    // Visiting order of the specification: a leaf child is intersected before descending into an inner sibling.
                    int first = c0, second = c1;
                    if (c0 >= 0 && c1 < 0) { first = c1; second = c0; }
                    stack[sp * TRACE_BLOCK] = second;
                    ++sp;
                    set_item(first);
                } else if (h0) set_item(c0);
                else pop_next();
            }
            busy = __ballot(active);
        } while (busy && (exhausted || __popcll(busy) >= refill_below));
    }

    if (INSTRUMENT) {
        wave_add(SHADOW ? &counters->shadow_nodes : &counters->closest_nodes, nodes);
        wave_add(SHADOW ? &counters->shadow_triangles : &counters->closest_triangles, tris);
    }
}

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
    f3 add_radiance;
};

HD void shade_path(const DeviceScene& sc, const HiprCameraState& cam, f3 ro, f3 rd, float bsdf_pdf, f3 throughput, uint32_t bounces,
                   uint32_t last_triangle, uint32_t pixel_hash, uint32_t accumulation, float4 hit, ShadeOutput& out) {
    out.continues = out.shadow = out.shaded = false;
    out.add_radiance = mk3(0.0f);
    const uint32_t id = __float_as_uint(hit.w);
    if (id == HIPR_HIT_MISS) {
        out.add_radiance = throughput * mk3(sc.env_tint[0], sc.env_tint[1], sc.env_tint[2]);
        return;
    }
    if (id & HIPR_HIT_LIGHT) {
        f3 L = light_evaluate_intersection(sc.lights[id & ~HIPR_HIT_LIGHT], ro, rd, bsdf_pdf);
        out.add_radiance = min3(throughput, mk3(4.0f)) * L;
        return;
    }

    // --- attributes of the accepted closest hit only (TriangleAttributes.cu:35-84) -------------
    const float4* tp = sc.triangles + 3 * size_t(id);
    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
    const f3 p0 = {ta.x, ta.y, ta.z}, p1 = {ta.w, tb.x, tb.y}, p2 = {tb.z, tb.w, tc.x};
    const HiprInstance& inst = sc.instances[__float_as_uint(tc.y)];
    const uint32_t prim = __float_as_uint(tc.z);
    const HiprMaterial mp = sc.materials[inst.material_index];
    const float u = hit.y, v = hit.z, w = 1.0f - u - v;
    const uint32_t* idx = sc.indices + 3 * size_t(inst.index_offset + prim);
    const uint32_t i0 = idx[0], i1 = idx[1], i2 = idx[2];

    f3 geometric_normal = normalize(cross(p1 - p0, p2 - p0));
    const f2 texcoord = triangle_texcoord(sc, inst, prim, u, v);

    const bool thin_walled = (mp.flags & (HIPR_MATERIAL_CUTOUT | HIPR_MATERIAL_THIN_WALLED)) != 0;
    const bool transmissive = mp.shading_model == HIPR_SHADING_TRANSMISSIVE;
    const bool hit_from_front = dot(geometric_normal, rd) < 0.0f;
    const bool backside_cull = !hit_from_front && !thin_walled && !transmissive;

    const f4 bsdf_u = sobol4f(accumulation, pixel_hash, 8u * bounces + 2u);   // BSDF dimension, always drawn
    const float coverage = material_coverage(sc, mp, texcoord);
    if (backside_cull || coverage < bsdf_u.w) {
        // rejected hit: same ray, tmin bumped past it, counters untouched (MonteCarlo.cu:159-164)
        out.continues = true;
        out.o = ro; out.d = rd; out.tmin = nextafterf(hit.x, __builtin_inff()); out.bsdf_pdf = bsdf_pdf;
        out.throughput = throughput; out.bounces = bounces; out.last_triangle = last_triangle;
        return;
    }
    out.shaded = true;

    const f3 position = p1 * u + p2 * v + p0 * w;
    f3 shading_normal = geometric_normal;
    if (inst.mesh_flags & HIPR_MESH_NORMALS) {
        const float4* g = sc.geometry + inst.vertex_offset;
        f3 n = decode_octahedral(g[i1].w) * u + decode_octahedral(g[i2].w) * v + decode_octahedral(g[i0].w) * w;
        n = normalize(n);
        const float* M = inst.object_to_world;
        shading_normal = normalize(mk3(M[0] * n.x + M[1] * n.y + M[2] * n.z, M[4] * n.x + M[5] * n.y + M[6] * n.z, M[8] * n.x + M[9] * n.y + M[10] * n.z));
    }
    f4 tint_scale = {1, 1, 1, 1};
    if (inst.mesh_flags & HIPR_MESH_TINTS) {
        const uint32_t* tints = sc.tints + inst.vertex_offset;
        const uint32_t t0 = tints[i0], t1 = tints[i1], t2 = tints[i2];
        const float s = 1.0f / 255.0f;
        auto ch = [](uint32_t p, int c) { return float((p >> (8 * c)) & 0xFFu); };
        tint_scale = {(ch(t1, 0) * u + ch(t2, 0) * v + ch(t0, 0) * w) * s, (ch(t1, 1) * u + ch(t2, 1) * v + ch(t0, 1) * w) * s,
                      (ch(t1, 2) * u + ch(t2, 2) * v + ch(t0, 2) * w) * s, (ch(t1, 3) * u + ch(t2, 3) * v + ch(t0, 3) * w) * s};
    }
    f3 emission = {1, 1, 1};
    if (inst.mesh_flags & HIPR_MESH_EMISSIVE) {
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
    if (mp.tint_roughness_texture_ID) tr = tr * sample_texture(sc, mp.tint_roughness_texture_ID, texcoord);
    if (mp.roughness_texture_ID) tr.w *= sample_texture(sc, mp.roughness_texture_ID, texcoord).x;
    tr = tr * tint_scale;
    MaterialInputs in;
    in.tint = {tr.x, tr.y, tr.z};
    in.roughness = tr.w;
    in.specularity = mp.specularity;
    in.metallic = mp.metallic_texture_ID ? mp.metallic * sample_texture(sc, mp.metallic_texture_ID, texcoord).x : mp.metallic;
    in.coat = mp.coat / 65535.0f;
    in.coat_roughness = mp.coat_roughness / 65535.0f;
    const float max_PDF_hint = bsdf_pdf * cam.path_regularization_PDF_scale;
    Shading shading;
    if (mp.shading_model == HIPR_SHADING_DIFFUSE) shading = make_diffuse(in.tint, in.roughness);
    else if (transmissive) shading = make_transmissive(sc.tables, in, cos_theta, max_PDF_hint);
    else shading = make_default(sc.tables, in, cos_theta, max_PDF_hint);

    out.add_radiance = throughput * emission * mk3(mp.emission[0], mp.emission[1], mp.emission[2]);

    // --- next event estimation: streaming RIS over the light candidates (MonteCarlo.cu:91-123) ------
    LightSample kept = light_sample_none();
    if (sc.light_count != 0) {
        const f4 base = sobol4f(accumulation, pixel_hash, 8u * bounces + 1u);
        const int n = sc.next_event_sample_count;
        for (int s = 0; s < n; ++s) {
            const float4 off = sc.sample_offsets[s];
            f4 r = {base.x + off.x, base.y + off.y, base.z + off.z, base.w + off.w};
            r = {r.x - floorf(r.x), r.y - floorf(r.y), r.z - floorf(r.z), r.w - floorf(r.w)};
            const int light_count = int(sc.light_count);
            int li = int(r.z * light_count);
            li = li > light_count - 1 ? light_count - 1 : li;
            LightSample c = light_sample_radiance(sc.lights[li], position, mk2(r.x, r.y));
            c.radiance *= float(light_count);
            c.radiance *= fabsf(dot(tbn.n, c.dir)) / pdf_value(c.pdf);
            Response f = shading_evaluate_with_PDF(shading, wo, to_local(tbn, c.dir));
            if (!pdf_is_delta(c.pdf)) c.radiance *= balance_heuristic(pdf_value(c.pdf), pdf_value(f.pdf));
            else f.f = min3(f.f, mk3(32.0f));
            c.radiance *= f.f;
            const float w_old = sum(kept.radiance), w_new = sum(c.radiance);
            const float p_new = w_new / (w_old + w_new);
            if (r.w < p_new) { kept = c; kept.radiance /= p_new; }
            else kept.radiance /= 1.0f - p_new;
        }
        kept.radiance /= float(n);
    }
    const f3 light_origin = offset_ray_origin(position, kept.dir, geometric_normal);
    kept.radiance *= throughput;
    if (kept.radiance.x > 0 || kept.radiance.y > 0 || kept.radiance.z > 0) {
        out.shadow = true;
        out.so = light_origin; out.sd = kept.dir; out.stmax = kept.distance; out.sradiance = kept.radiance;
    }

    // --- BSDF sampling (MonteCarlo.cu:204-232) ---------------------------------------------------
    const Sample bs = shading_sample(shading, wo, mk3(bsdf_u.x, bsdf_u.y, bsdf_u.z));
    const bool is_reflection = bs.dir.z >= 0;
    f3 direction = to_world(tbn, bs.dir);
    float new_pdf = bs.pdf;
    if (pdf_is_valid(bs.pdf)) throughput *= (bs.f * fabsf(bs.dir.z)) / pdf_value(bs.pdf);
    else throughput = mk3(0.0f);
    const float cos_geometric = dot(direction, geometric_normal);
    if (is_reflection ? cos_geometric < 0.0f : cos_geometric >= 0.0f)
        direction = reflect(direction, geometric_normal);
    if (!pdf_is_valid(kept.pdf)) new_pdf = pdf_disable_MIS(new_pdf);
    bounces += 1u;

    out.o = offset_ray_origin(position, direction, geometric_normal);
    out.d = direction; out.tmin = 0.0f; out.bsdf_pdf = new_pdf; out.throughput = throughput; out.bounces = bounces; out.last_triangle = id;
    out.continues = bounces <= cam.max_bounce_count && !is_black(throughput);
}

__global__ __launch_bounds__(SHADE_BLOCK) void k_shade(DeviceScene sc, HiprCameraState cam, PathState in, const float4* hits, PathState out,
                                                        ShadowQueue shadows, float4* radiance, const uint32_t* count_ptr, uint32_t* next_count,
                                                        uint32_t* shadow_count, DeviceCounters* counters) {
    __shared__ uint32_t s_cont[SHADE_BLOCK / 64], s_shad[SHADE_BLOCK / 64], s_base[2];
    const uint32_t n = *count_ptr;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t shaded_total = 0;
    for (uint32_t base = blockIdx.x * SHADE_BLOCK; base < n; base += gridDim.x * SHADE_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        ShadeOutput so;
        so.continues = so.shadow = so.shaded = false;
        uint32_t slot = HIPR_DEAD_SLOT, pixel_hash = 0, accumulation = 0;
        if (i < n) {
            const uint4 meta = in.meta[i];
            slot = meta.x; pixel_hash = meta.z; accumulation = meta.w;
            if (slot != HIPR_DEAD_SLOT) {
                const float4 o = in.o_tmin[i], d = in.d_pdf[i], t = in.thr_bounces[i];
                shade_path(sc, cam, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), d.w, mk3(t.x, t.y, t.z), __float_as_uint(t.w), meta.y, pixel_hash,
                           accumulation, hits[i], so);
                if (so.add_radiance.x != 0.0f || so.add_radiance.y != 0.0f || so.add_radiance.z != 0.0f) {
                    float4 acc = radiance[slot];
                    acc.x += so.add_radiance.x; acc.y += so.add_radiance.y; acc.z += so.add_radiance.z;
                    radiance[slot] = acc;
                }
            }
        }
        // ---- compaction: ballot + prefix popcount in the wave, LDS scan over the block's waves, one atomic per queue
        const unsigned long long cont_mask = __ballot(so.continues), shad_mask = __ballot(so.shadow);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lane == 0) { s_cont[wave] = __popcll(cont_mask); s_shad[wave] = __popcll(shad_mask); }
        shaded_total += so.shaded ? 1u : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t c = 0, s = 0;
            for (int wv = 0; wv < SHADE_BLOCK / 64; ++wv) { uint32_t t = s_cont[wv]; s_cont[wv] = c; c += t; t = s_shad[wv]; s_shad[wv] = s; s += t; }
            s_base[0] = c ? atomicAdd(next_count, c) : 0u;
            s_base[1] = s ? atomicAdd(shadow_count, s) : 0u;
        }
        __syncthreads();
        if (so.continues) {
            const uint32_t j = s_base[0] + s_cont[wave] + __popcll(cont_mask & lt);
            out.o_tmin[j] = make_float4(so.o.x, so.o.y, so.o.z, so.tmin);
            out.d_pdf[j] = make_float4(so.d.x, so.d.y, so.d.z, so.bsdf_pdf);
            out.thr_bounces[j] = make_float4(so.throughput.x, so.throughput.y, so.throughput.z, __uint_as_float(so.bounces));
            out.meta[j] = make_uint4(slot, so.last_triangle, pixel_hash, accumulation);
        }
        if (so.shadow) {
            const uint32_t j = s_base[1] + s_shad[wave] + __popcll(shad_mask & lt);
            shadows.o_tmax[j] = make_float4(so.so.x, so.so.y, so.so.z, so.stmax);
            shadows.d_slot[j] = make_float4(so.sd.x, so.sd.y, so.sd.z, __uint_as_float(slot));
            shadows.radiance[j] = make_float4(so.sradiance.x, so.sradiance.y, so.sradiance.z, 0.0f);
        }
        __syncthreads();
    }
    wave_add(&counters->shaded_hits, shaded_total);
}

// ---------------------------------------------------------------------------------------------
// K6: f64 running mean + half4 output
// ---------------------------------------------------------------------------------------------
HD unsigned short float_to_half_bits(float v) {
    _Float16 h = (_Float16)v;   // v_cvt_f16_f32, round to nearest even like __float2half_rn
    unsigned short bits;
    __builtin_memcpy(&bits, &h, 2);
    return bits;
}

__global__ __launch_bounds__(256) void k_accumulate(FrameInfo frame, uint32_t first_accumulation, const float4* radiance, double4* accumulation,
                                                     ushort4* out, uint32_t out_pitch) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    const uint32_t per_sample = frame.owned_tiles * 64u;
    if (k >= per_sample) return;
    uint32_t x, y;
    if (!owned_pixel(frame, k, x, y)) return;
    double4 acc = accumulation[k];
    for (uint32_t s = 0; s < frame.samples_per_pass; ++s) {
        const float4 r = radiance[s * per_sample + k];
        const uint32_t a = first_accumulation + s;
        if (a != 0) {
            const double t = 1.0 / (a + 1.0);
            acc.x = acc.x + (double(r.x) - acc.x) * t;
            acc.y = acc.y + (double(r.y) - acc.y) * t;
            acc.z = acc.z + (double(r.z) - acc.z) * t;
        } else { acc.x = r.x; acc.y = r.y; acc.z = r.z; }
        acc.w = 1.0;
    }
    accumulation[k] = acc;
    if (out) {
        const size_t dst = frame.tile_stride == 1 ? size_t(x) + size_t(y) * out_pitch : size_t(k);
        out[dst] = make_ushort4(float_to_half_bits(float(acc.x)), float_to_half_bits(float(acc.y)), float_to_half_bits(float(acc.z)), float_to_half_bits(1.0f));
    }
}

// Assemble a full frame from the compact per-rank buffers gathered over RCCL.
__global__ __launch_bounds__(256) void k_scatter_tiles(const ushort4* compact, unsigned long long rank_stride, uint32_t rank_count, uint32_t width,
                                                        uint32_t height, ushort4* out, uint32_t out_pitch) {
    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u), y = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tile = (y >> 3) * tiles_x + (x >> 3);
    const uint32_t rank = tile % rank_count, local_tile = tile / rank_count;
    const uint32_t lane = (x & 7u) + ((y & 7u) << 3);
    out[size_t(x) + size_t(y) * out_pitch] = compact[size_t(rank) * rank_stride + size_t(local_tile) * 64u + lane];
}

// Debug / parity helpers -------------------------------------------------------------------------
__global__ void k_debug_sobol(const uint32_t* triples, uint32_t n, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u4 s = sobol4ui(triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]);
    out[4 * i] = s.x; out[4 * i + 1] = s.y; out[4 * i + 2] = s.z; out[4 * i + 3] = s.w;
}

} // namespace hipr
