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
#include "fast_divide.h"

namespace hipr {

#define HIPR_DEAD_SLOT 0xFFFFFFFFu
#define HIPR_HIT_MISS 0xFFFFFFFFu
#define HIPR_HIT_LIGHT 0x80000000u
#define HIPR_NO_TRIANGLE 0xFFFFFFFFu

constexpr int TRACE_BLOCK = 128;   // 2 waves; LDS stack = STACK * 128 * 4 B
#ifndef HIPR_SHADE_BLOCK_THREADS
#define HIPR_SHADE_BLOCK_THREADS 256
#endif
constexpr int SHADE_BLOCK = HIPR_SHADE_BLOCK_THREADS;
#ifndef HIPR_SHADE_SPLIT_WAVES
#define HIPR_SHADE_SPLIT_WAVES 4    // waves per SIMD the two halves of the split shade kernel are compiled for (shade_kernel.h SHADE_PART_*)
#endif
constexpr int SHADE_TRIANGLE_QUADS = 8;   // float4 per shading record (128 B = one cache line), see k_build_shade_triangles

struct DeviceScene {
    const float4* nodes;
    const uint4* wide_nodes;         // HiprWideNode, 4 x uint4 each: what the persistent kernels traverse
    const float4* triangles;
    const float4* shade_triangles;   // SHADE_TRIANGLE_QUADS float4 per triangle, built on upload (k_build_shade_triangles)
    const float4* trace_triangles;   // 3 float4 per triangle: v0, e1 = v1 - v0, e2 = v2 - v0, then instance / primitive / flags as in `triangles` (k_build_trace_triangles)
    const float4* trace_items;       // exhaustive search only: 4 float4 per item, a triangle or two triangles merged into a parallelogram (build_trace_items)
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
    // presampled environment light (HiprEnvironment); env_map_ID == 0: the environment is the constant tint
    const float* env_per_pixel_PDF;
    const float4* env_samples;       // 2 float4 per HiprLightSample: radiance + PDF, direction + distance
    int env_map_ID;
    uint32_t env_pdf_width, env_pdf_height, env_sample_count;
    const uint32_t* sobol_tables;   // SOBOL_TABLE_WORDS words, see sobol4ui_tables
    DeviceTables tables;
    uint32_t node_count, wide_node_count, triangle_count, light_count, trace_item_count;
    float env_tint[3];
    int next_event_sample_count;
};

// Wave-uniform reads of scene arrays go through the constant address space: a uniform index then compiles to scalar loads
// (s_load_dwordx4 into SGPRs) instead of 64 identical vector loads. Valid because the kernels never write the scene.
typedef float ScalarFloat4 __attribute__((ext_vector_type(4)));
typedef const ScalarFloat4 __attribute__((address_space(4))) * ConstantFloat4Pointer;
HD ConstantFloat4Pointer as_constant(const float4* p) { return (ConstantFloat4Pointer)(p); }
HD HiprLight load_light_uniform(const HiprLight* lights, uint32_t i) {   // i must be wave uniform
    const ConstantFloat4Pointer p = (ConstantFloat4Pointer)(lights + i);
    const ScalarFloat4 a = p[0], b = p[1], c = p[2];
    HiprLight l;
    l.data[0] = a.x; l.data[1] = a.y; l.data[2] = a.z; l.data[3] = a.w; l.data[4] = b.x; l.data[5] = b.y; l.data[6] = b.z; l.data[7] = b.w;
    l.data[8] = c.x; l.data[9] = c.y; l.data[10] = c.z; l.flags = __float_as_uint(c.w);
    return l;
}

struct PathState {
    float4* o_tmin;       // origin.xyz, tmin
    float4* d_pdf;        // direction.xyz, bsdf pdf
    float4* thr_bounces;  // throughput.xyz, bits(bounces)
    uint2* meta;          // slot, last accepted triangle (the pixel's hash and the accumulation follow from the slot: path_sample_of_slot)
};

struct ShadowQueue {
    float4* o_tmax;       // origin.xyz, tmax
    float4* d_slot;       // direction.xyz, bits(slot)
    float4* radiance;     // rgb, unused
};

#ifndef HIPR_PIXEL_MAJOR_SLOTS
#define HIPR_PIXEL_MAJOR_SLOTS 1
#endif
struct FrameInfo {
    uint32_t width, height, tiles_x, tiles_total, tile_phase, tile_stride, owned_tiles, samples_per_pass;
    Divisor by_samples_per_pass, by_tiles_x;     // fast_divide.h; set with the fields they divide by (hiprenderer.hip set_frame_divisors)
};

struct DeviceCounters {
    unsigned long long shaded_hits, closest_nodes, closest_triangles, shadow_nodes, shadow_triangles;
    // diagnostics of the persistent kernels (instrumented builds): wave iterations by kind and the lanes that did work in them
    unsigned long long node_iterations, node_lanes, triangle_iterations, triangle_lanes, busy_lanes, refills;
    unsigned long long pushes, pushes_past_16, pushes_past_24;   // stack pushes of the persistent kernels, and those that landed on entry 16 / 24 or deeper
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
    const uint32_t row = divide(tile, f.by_tiles_x);
    x = (tile - row * f.tiles_x) * 8u + (lane & 7u);
    y = row * 8u + (lane >> 3);
    return tile < f.tiles_total && x < f.width && y < f.height;
}

// Path slot p <-> (owned pixel k, sample s of the pass), and what the samplers need of it: the pixel's hash and the accumulation the sample belongs to. The queue
// entries carried both until round 4 (16 B more read and written per entry and bounce); they cost ~25 integer instructions to state again.
HD void path_sample_of_slot(const FrameInfo& f, const HiprCameraState& cam, uint32_t slot, uint32_t& pixel_hash, uint32_t& accumulation) {
#if HIPR_PIXEL_MAJOR_SLOTS
    const uint32_t k = divide(slot, f.by_samples_per_pass), s = slot - k * f.samples_per_pass;
#else
    const uint32_t per_sample = f.owned_tiles * 64u;
    const uint32_t s = slot / per_sample, k = slot - s * per_sample;
#endif
    uint32_t x, y;
    (void)owned_pixel(f, k, x, y);
    pixel_hash = pcg2d_x(x, y);
    accumulation = cam.accumulations + s;
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

#ifndef HIPR_SHADE_TU
// Camera rays of one wavefront's path slots into its queue entries [0, n_paths): the slots are dealt to the wavefronts in groups of 64, round robin
// (wavefront `phase` of `wavefronts`).
__global__ __launch_bounds__(256) void k_generate(FrameInfo frame, HiprCameraState cam, PathState out, float4* radiance, uint32_t phase, uint32_t wavefronts, uint32_t n_paths) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_paths) return;
    const uint32_t p = ((i >> 6) * wavefronts + phase) * 64u + (i & 63u);
    // Path slot p <-> (owned pixel k, sample s of the pass). Pixel-major since round 4, p = k * samples_per_pass + s: the samples of a pixel sit side by
    // side in the queue, so a wave of camera rays is one or two pixels' worth of near-identical rays -- its lanes walk the tree in step (no lane waits for
    // the other kind of item), fetch the same nodes, and shade one material -- and the radiance slots of a pixel are one contiguous run for k_accumulate.
    // Round 3's sample-major order (p = s * owned pixels + k) put 64 neighbouring PIXELS of one sample in a wave. profiles/r04_ab_slot_order.txt.
#if HIPR_PIXEL_MAJOR_SLOTS
    uint32_t k = divide(p, frame.by_samples_per_pass), s = p - k * frame.samples_per_pass;
#else
    uint32_t per_sample = frame.owned_tiles * 64u;
    uint32_t s = p / per_sample, k = p - s * per_sample;
#endif
    uint32_t x, y;
    bool valid = owned_pixel(frame, k, x, y);
    uint32_t accumulation = cam.accumulations + s;
    uint32_t pixel_hash = pcg2d_x(x, y);
    f3 o = {0, 0, 0}, d = {0, 0, 1};
    if (valid) camera_ray(cam, x, y, frame.width, frame.height, accumulation, pixel_hash, o, d);
    out.o_tmin[i] = make_float4(o.x, o.y, o.z, 0.0f);
    out.d_pdf[i] = make_float4(d.x, d.y, d.z, -1.0f);           // bsdf_PDF = delta_dirac(1)
    // throughput (1, 1, 1) and bounce count 0 are not written: shade(0) is launched without the array and fills them in (shade_fetch_inputs)
    out.meta[i] = make_uint2(valid ? p : HIPR_DEAD_SLOT, HIPR_NO_TRIANGLE);
    radiance[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

#endif // HIPR_SHADE_TU

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

// Ray / triangle test (DESIGN.md "Triangle test"): Moeller-Trumbore on v0 and the two edges e1 = v1 - v0, e2 = v2 - v0, which
// the upload computes once per triangle (k_build_trace_triangles). The inside decision is taken on the UNNORMALISED barycentrics
// un = tv . (d x e2), vn = d . (tv x e1) against the determinant, with its sign folded in by flipping sign bits:
//     inside  <=>  det != 0  and  un * s >= 0  and  vn * s >= 0  and  un * s + vn * s <= |det|          (s = sign of det)
// so the division 1 / det and the three products that yield t, u, v are only evaluated for rays inside a triangle -- in a wave
// of coherent rays that is a small minority of the tests (triangle_hit_values). The oracle takes the same decision on the
// same values (oracle/integrator.cpp intersect_triangle), which keeps hits and counters bit-identical.
struct TriangleTest { float det, un, vn; f3 q; };
HD bool triangle_inside(f3 v0, f3 e1, f3 e2, f3 o, f3 d, TriangleTest& r) {
    const f3 p = cross_fma(d, e2);
    r.det = dot_fma(e1, p);
    const f3 tv = o - v0;
    r.un = dot_fma(tv, p);
    r.q = cross_fma(tv, e1);
    r.vn = dot_fma(d, r.q);
    const uint32_t sign = __float_as_uint(r.det) & 0x80000000u;
    const float us = __uint_as_float(__float_as_uint(r.un) ^ sign), vs = __uint_as_float(__float_as_uint(r.vn) ^ sign);
    return (r.det != 0.0f) & (us >= 0.0f) & (vs >= 0.0f) & (us + vs <= fabsf(r.det));
}
HD void triangle_hit_values(const TriangleTest& r, f3 e2, float& t, float& u, float& v) {
    const float inv = 1.0f / r.det;
    u = r.un * inv;
    v = r.vn * inv;
    t = dot_fma(e2, r.q) * inv;
}


// Exhaustive-search items (scenes of at most SMALL_SCENE_TRIANGLES triangles): a triangle, or TWO triangles that form a
// parallelogram (a, b, c) + (a, c, d) with d = a + (c - b), tested as one: the same Moeller-Trumbore solve against e1 = b - a and
// e2 = d - a yields (s, r) with the parallelogram being 0 <= s, r <= 1; s >= r is the half of (a, b, c), whose corner weights are
// (1 - s, s - r, r); the other half is (a, c, d) with (1 - r, s, r - s). A box side or a wall costs one test instead of two.
// Item layout (4 float4): {v0, e1.x} {e1.yz, e2.xy} {e2.z, instance, primitive A, flags} {triangle A, triangle B, primitive B, selectors};
// selectors: 2 bits each, which corner weight of the half is the triangle's u (weight of its vertex 1) and v (vertex 2): u of A,
// v of A, u of B, v of B. The host builds the items at upload (hiprenderer.hip build_trace_items); the oracle restates the pairing.
constexpr uint32_t HIPR_ITEM_QUAD = 2u;   // next to HIPR_TRIANGLE_OPAQUE in the item's flags
struct ItemTest { float det, un, vn, us, vs; f3 q; };
HD bool item_inside(f3 v0, f3 e1, f3 e2, bool quad, f3 o, f3 d, ItemTest& r) {
    const f3 p = cross_fma(d, e2);
    r.det = dot_fma(e1, p);
    const f3 tv = o - v0;
    r.un = dot_fma(tv, p);
    r.q = cross_fma(tv, e1);
    r.vn = dot_fma(d, r.q);
    const uint32_t sign = __float_as_uint(r.det) & 0x80000000u;
    r.us = __uint_as_float(__float_as_uint(r.un) ^ sign);
    r.vs = __uint_as_float(__float_as_uint(r.vn) ^ sign);
    const float limit = fabsf(r.det);
    const bool upper = quad ? ((r.us <= limit) & (r.vs <= limit)) : (r.us + r.vs <= limit);   // `quad` is wave uniform
    return (r.det != 0.0f) & (r.us >= 0.0f) & (r.vs >= 0.0f) & upper;
}
// Barycentrics of the hit triangle from the solve's (s, r): `second_half` = the hit is in (a, c, d); `selectors` as in the item.
HD void item_barycentrics(bool quad, bool second_half, uint32_t selectors, float s, float r, float& u, float& v) {
    if (!quad) { u = s; v = r; return; }
    const float w0 = second_half ? 1.0f - r : 1.0f - s, w1 = second_half ? s : s - r, w2 = second_half ? r - s : r;
    const uint32_t cu = (selectors >> (second_half ? 4 : 0)) & 3u, cv = (selectors >> (second_half ? 6 : 2)) & 3u;
    u = cu == 0 ? w0 : (cu == 1 ? w1 : w2);
    v = cv == 0 ? w0 : (cv == 1 ? w1 : w2);
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

// Wave votes on a predicate the compiler already holds as a lane mask: __builtin_amdgcn_ballot_w64 is one scalar AND with exec, where HIP's
// __ballot(int) / __any(int) first materialise the predicate as 0 / 1 in a VGPR and compare it again (two VALU instructions per vote).
__device__ __forceinline__ unsigned long long wave_ballot(bool predicate) { return __builtin_amdgcn_ballot_w64(predicate); }
__device__ __forceinline__ bool wave_any(bool predicate) { return __builtin_amdgcn_ballot_w64(predicate) != 0ull; }

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
            const float4* tp = sc.trace_triangles + 3 * size_t(i);
            float4 a = tp[0], b = tp[1], c = tp[2];
            if (i == skip) continue;
            TriangleTest test;
            if (!triangle_inside(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, test)) continue;
            float t, u, v;
            triangle_hit_values(test, mk3(b.z, b.w, c.x), t, u, v);
            if (!(t > tmin)) continue;
            if (t < best_t || (t == best_t && i < best_id)) { best_t = t; best_u = u; best_v = v; best_id = i; }
        }
        return false;
    });
    // Analytic area lights take part in closest-hit selection (LightSources.cu:31-70).
    for (uint32_t li = 0; li < sc.light_count; ++li) {
        const HiprLight l = load_light_uniform(sc.lights, li);
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
        uint2 meta = in.meta[i];
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
HD float srgb_to_linear(float c) { return c <= 0.04045f ? c / 12.92f : pow_((c + 0.055f) / 1.055f, 2.4f); }

// FLOAT_FORMATS = false: the caller knows the scene holds 8-bit textures only (k_shade<..., TEXTURES = 1>)
template <bool FLOAT_FORMATS = true>
HD f4 fetch_texel(const DeviceScene& sc, const HiprTexture& tex, int x, int y) {
    const uint8_t* base = sc.texels + tex.texel_offset;
    size_t i = size_t(y) * tex.width + size_t(x);
    f4 r;
    if (tex.format == HIPR_TEXEL_R8) { r = {base[i] / 255.0f, 0, 0, 1}; }
    else if (!FLOAT_FORMATS || tex.format == HIPR_TEXEL_RGBA8) {
        uint32_t p = reinterpret_cast<const uint32_t*>(base)[i];
        r = {(p & 0xFFu) / 255.0f, ((p >> 8) & 0xFFu) / 255.0f, ((p >> 16) & 0xFFu) / 255.0f, (p >> 24) / 255.0f};
    } else if (tex.format == HIPR_TEXEL_R32F) { r = {reinterpret_cast<const float*>(base)[i], 0, 0, 1}; }
    else { float4 v = reinterpret_cast<const float4*>(base)[i]; r = {v.x, v.y, v.z, v.w}; }
    if (tex.is_sRGB) {
        r.x = srgb_to_linear(r.x);
        if (tex.format == HIPR_TEXEL_RGBA8 || (FLOAT_FORMATS && tex.format == HIPR_TEXEL_RGBA32F)) { r.y = srgb_to_linear(r.y); r.z = srgb_to_linear(r.z); }
    }
    return r;
}
// The texel index of coordinate i in a texture of n texels. The same integers as `i %= n; i < 0 ? i + n : i` by two exact shortcuts (round 5): for a power-of-two size the
// mask IS the non-negative remainder, also of a negative i (two's complement), and most textures are powers of two -- the remainder by a per-lane n that the compiler
// expands into ~35 instructions is then skipped by the whole wave; and the neighbour texel of a bilinear tap follows from the wrapped first one (wrap_next) instead of a
// second remainder. Measured on the textured atrium (128 x 128 tint textures, 64 x 64 coverage): profiles/r05_ab_texture_wrap.txt.
HD int wrap_coord(int i, int n, int repeat) {
    if (repeat) {
        if ((n & (n - 1)) == 0) return i & (n - 1);
        i %= n;
        return i < 0 ? i + n : i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// wrap_coord(i + 1, n, repeat), given wrapped = wrap_coord(i, n, repeat): ((i + 1) mod n) = ((i mod n) + 1) mod n
HD int wrap_next(int wrapped, int i, int n, int repeat) {
    if (repeat) return wrapped + 1 == n ? 0 : wrapped + 1;
    return i + 1 < 0 ? 0 : (i + 1 >= n ? n - 1 : i + 1);
}
template <bool FLOAT_FORMATS = true>
HD f4 sample_texture(const DeviceScene& sc, int id, f2 uv) {
    const HiprTexture tex = sc.textures[id];
    int w = int(tex.width), h = int(tex.height);
    if (tex.filter & 1) {
        float xb = uv.x * w - 0.5f, yb = uv.y * h - 0.5f;
        float xf = floorf(xb), yf = floorf(yb);
        float fx = xb - xf, fy = yb - yf;
        int x0 = wrap_coord(int(xf), w, tex.wrap_u), x1 = wrap_next(x0, int(xf), w, tex.wrap_u);
        int y0 = wrap_coord(int(yf), h, tex.wrap_v), y1 = wrap_next(y0, int(yf), h, tex.wrap_v);
        f4 a = fetch_texel<FLOAT_FORMATS>(sc, tex, x0, y0), b = fetch_texel<FLOAT_FORMATS>(sc, tex, x1, y0);
        f4 c = fetch_texel<FLOAT_FORMATS>(sc, tex, x0, y1), d = fetch_texel<FLOAT_FORMATS>(sc, tex, x1, y1);
        f4 lo = a + (b - a) * fx, hi = c + (d - c) * fx;
        return lo + (hi - lo) * fy;
    }
    int x = wrap_coord(int(floorf(uv.x * w)), w, tex.wrap_u);
    int y = wrap_coord(int(floorf(uv.y * h)), h, tex.wrap_v);
    return fetch_texel<FLOAT_FORMATS>(sc, tex, x, y);
}
template <bool FLOAT_FORMATS = true>
HD float material_coverage(const DeviceScene& sc, const HiprMaterial& m, f2 uv) {
    float tex = 1.0f;
    if (m.coverage_texture_ID) tex = sample_texture<FLOAT_FORMATS>(sc, m.coverage_texture_ID, uv).x;
    if (m.flags & HIPR_MATERIAL_CUTOUT) return tex < m.coverage ? 0.0f : 1.0f;
    return m.coverage * tex;
}

// The same for scenes whose coverage textures are all single-channel 8-bit and linear (what the loaders make of an alpha channel; decided at upload): the sampler
// without its other three formats and the sRGB decode -- code a shadow ray never reaches there, and registers the traversal loop gets back
// (k_trace_wide8<..., COVERAGE_R8 = true>, profiles/r04_ab_coverage_chain.txt). Term for term what sample_texture / fetch_texel compute for such a texture.
HD float sample_texture_r8(const DeviceScene& sc, int id, f2 uv) {
    const HiprTexture tex = sc.textures[id];
    const uint8_t* base = sc.texels + tex.texel_offset;
    const int w = int(tex.width), h = int(tex.height);
    auto texel = [&](int x, int y) { return base[size_t(y) * tex.width + size_t(x)] / 255.0f; };
    if (tex.filter & 1) {
        const float xb = uv.x * w - 0.5f, yb = uv.y * h - 0.5f;
        const float xf = floorf(xb), yf = floorf(yb);
        const float fx = xb - xf, fy = yb - yf;
        const int x0 = wrap_coord(int(xf), w, tex.wrap_u), x1 = wrap_next(x0, int(xf), w, tex.wrap_u);
        const int y0 = wrap_coord(int(yf), h, tex.wrap_v), y1 = wrap_next(y0, int(yf), h, tex.wrap_v);
        const float a = texel(x0, y0), b = texel(x1, y0), c = texel(x0, y1), d = texel(x1, y1);
        const float lo = a + (b - a) * fx, hi = c + (d - c) * fx;
        return lo + (hi - lo) * fy;
    }
    return texel(wrap_coord(int(floorf(uv.x * w)), w, tex.wrap_u), wrap_coord(int(floorf(uv.y * h)), h, tex.wrap_v));
}
HD float material_coverage_r8(const DeviceScene& sc, const HiprMaterial& m, f2 uv) {
    float tex = 1.0f;
    if (m.coverage_texture_ID) tex = sample_texture_r8(sc, m.coverage_texture_ID, uv);
    if (m.flags & HIPR_MATERIAL_CUTOUT) return tex < m.coverage ? 0.0f : 1.0f;
    return m.coverage * tex;
}

// the same for a material known to carry no coverage texture (the shade kernel's instantiation for scenes without textures)
HD float material_coverage_untextured(const HiprMaterial& m) {
    if (m.flags & HIPR_MATERIAL_CUTOUT) return 1.0f < m.coverage ? 0.0f : 1.0f;
    return m.coverage;
}

// ---------------------------------------------------------------------------------------------
// Presampled environment light (ORS/LightSources/PresampledEnvironmentLightImpl.h:18-41, OR/Utils.h:288-292)
// ---------------------------------------------------------------------------------------------
HD f2 direction_to_latlong_texcoord(f3 direction) {
    const float u = (atan2_(direction.z, direction.x) + HIPR_PI) * 0.5f / HIPR_PI;
    const float v = (asin_(direction.y) + HIPR_PI * 0.5f) / HIPR_PI;
    return {u, v};
}
// Solid angle PDF of sampling `direction` from the environment: nearest, clamped lookup of the per-texel PDF over sin(theta).
HD float environment_pdf(const DeviceScene& sc, f3 direction) {
    const f2 uv = direction_to_latlong_texcoord(direction);
    const float sin_theta = sqrtf(1.0f - direction.y * direction.y);
    const int w = int(sc.env_pdf_width), h = int(sc.env_pdf_height);
    int x = int(floorf(uv.x * w)), y = int(floorf(uv.y * h));
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    const float pdf = sc.env_per_pixel_PDF[x + y * w] / sin_theta;
    return sin_theta == 0.0f ? -0.0f : pdf;   // PDF::delta_dirac(0) at the poles
}
HD f3 environment_evaluate(const DeviceScene& sc, f3 direction) {
    const f4 texel = sample_texture(sc, sc.env_map_ID, direction_to_latlong_texcoord(direction));
    return mk3(sc.env_tint[0], sc.env_tint[1], sc.env_tint[2]) * mk3(texel.x, texel.y, texel.z);
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
            const float4* tp = sc.trace_triangles + 3 * size_t(i);
            float4 a = tp[0], b = tp[1], c = tp[2];
            TriangleTest test;
            if (!triangle_inside(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, test)) continue;
            float t, u, v;
            triangle_hit_values(test, mk3(b.z, b.w, c.x), t, u, v);
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
// K2 / K4 for scenes of at most SMALL_SCENE_TRIANGLES triangles (the Cornell box has 34): exhaustive search. Every lane tests
// every triangle, so the loop is uniform -- no stack, no divergence, triangle data arrive through the scalar cache -- and that
// beats a BVH whose rays diverge after two or three nodes (measured: profiles/). The result is the closest hit under the same
// tie-break (order independent), i.e. what oracle::closest_hit_bruteforce returns; counters: 0 nodes, every triangle per ray.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_SCENE_TRIANGLES = 64;


template <bool INSTRUMENT>
__global__ __launch_bounds__(256) void k_trace_closest_small(DeviceScene sc, PathState in, float4* hits, const uint32_t* count_ptr, DeviceCounters* counters) {
    const uint32_t n = *count_ptr;
    uint32_t tris = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint2 meta = in.meta[i];
        if (meta.x == HIPR_DEAD_SLOT) { hits[i] = make_float4(0, 0, 0, __uint_as_float(HIPR_HIT_MISS)); continue; }
        const float4 ro = in.o_tmin[i], rd = in.d_pdf[i];
        const f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
        const float tmin = ro.w;
        const uint32_t skip = meta.y;
        float best_t = __builtin_inff(), best_s = 0.0f, best_r = 0.0f;
        uint32_t best_id = HIPR_HIT_MISS, best_code = 0;   // code: selectors | quad << 8 | second half << 9
        const ConstantFloat4Pointer items = as_constant(sc.trace_items);
        for (uint32_t t = 0; t < sc.trace_item_count; ++t) {   // uniform: scalar loads of the item
            const ScalarFloat4 a = items[4 * t], b = items[4 * t + 1], c = items[4 * t + 2], m = items[4 * t + 3];
            const bool quad = (__float_as_uint(c.w) & HIPR_ITEM_QUAD) != 0;
            ItemTest test;
            bool inside = item_inside(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), quad, o, d, test);
            const bool second_half = quad & (test.us < test.vs);
            const uint32_t id = second_half ? __float_as_uint(m.y) : __float_as_uint(m.x);
            inside &= id != skip;
            if (!wave_any(inside)) continue;       // wave-uniform: most items are missed by every ray of the wave
            const float inv = 1.0f / test.det;
            const float tt = dot_fma(mk3(b.z, b.w, c.x), test.q) * inv;
            const bool closer = inside & (tt > tmin) & ((tt < best_t) | ((tt == best_t) & (id < best_id)));
            best_t = closer ? tt : best_t; best_s = closer ? test.un * inv : best_s; best_r = closer ? test.vn * inv : best_r; best_id = closer ? id : best_id;
            best_code = closer ? ((__float_as_uint(m.w) & 0xFFu) | (quad ? 256u : 0u) | (second_half ? 512u : 0u)) : best_code;
        }
        float best_u, best_v;
        item_barycentrics((best_code & 256u) != 0, (best_code & 512u) != 0, best_code & 0xFFu, best_s, best_r, best_u, best_v);
        if (INSTRUMENT) tris += sc.trace_item_count;
        for (uint32_t li = 0; li < sc.light_count; ++li) {   // analytic area lights, LightSources.cu:31-70
            const HiprLight l = load_light_uniform(sc.lights, li);
            const uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
            float t = -1e30f;
            if (type == HIPR_LIGHT_SPHERE) { if (!(l.data[6] > 0.0f)) continue; t = ray_sphere(o, d, L3(l, 3), l.data[6]); }
            else if (type == HIPR_LIGHT_SPOT) { if (!(l.data[6] > 0.0f)) continue; t = ray_disk(o, d, L3(l, 3), L3(l, 7), l.data[6]); }
            else continue;
            if (t > tmin && t < best_t) { best_t = t; best_u = 0; best_v = 0; best_id = HIPR_HIT_LIGHT | li; }
        }
        hits[i] = make_float4(best_t, best_u, best_v, __uint_as_float(best_id));
    }
    if (INSTRUMENT) wave_add(&counters->closest_triangles, tris);
}

// Any-hit transmittance in triangle order (oracle::shadow_bruteforce); a lane stops at full occlusion, a wave when all its lanes did.
template <bool INSTRUMENT>
__global__ __launch_bounds__(256) void k_trace_shadow_small(DeviceScene sc, ShadowQueue q, float4* radiance, const uint32_t* count_ptr, DeviceCounters* counters) {
    const uint32_t n = *count_ptr;
    uint32_t tris = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 ro = q.o_tmax[i], rd = q.d_slot[i], rr = q.radiance[i];
        const f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
        const float tmax = ro.w;
        f3 rad = mk3(rr.x, rr.y, rr.z);
        bool blocked = false;
        const ConstantFloat4Pointer items = as_constant(sc.trace_items);
        for (uint32_t t = 0; t < sc.trace_item_count; ++t) {
            if (!wave_any(!blocked)) break;
            const ScalarFloat4 a = items[4 * t], b = items[4 * t + 1], c = items[4 * t + 2], m = items[4 * t + 3];
            if (INSTRUMENT) tris += blocked ? 0u : 1u;
            const bool quad = (__float_as_uint(c.w) & HIPR_ITEM_QUAD) != 0;
            ItemTest test;
            const bool inside = item_inside(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), quad, o, d, test) & !blocked;
            if (!wave_any(inside)) continue;
            const float inv = 1.0f / test.det;
            const float tt = dot_fma(mk3(b.z, b.w, c.x), test.q) * inv;
            if (inside && tt > 0.0f && tt < tmax) {
                float coverage = 1.0f;
                if (!(__float_as_uint(c.w) & HIPR_TRIANGLE_OPAQUE)) {
                    const bool second_half = quad & (test.us < test.vs);
                    float u, v;
                    item_barycentrics(quad, second_half, __float_as_uint(m.w) & 0xFFu, test.un * inv, test.vn * inv, u, v);
                    const HiprInstance& inst = sc.instances[__float_as_uint(c.y)];
                    coverage = material_coverage(sc, sc.materials[inst.material_index], triangle_texcoord(sc, inst, second_half ? __float_as_uint(m.z) : __float_as_uint(c.z), u, v));
                }
                rad *= 1.0f - coverage;
                if (rad.x < 0.0000001f && rad.y < 0.0000001f && rad.z < 0.0000001f) { rad = mk3(0.0f); blocked = true; }
            }
        }
        const uint32_t slot = __float_as_uint(rd.w);
        float4 acc = radiance[slot];
        acc.x += rad.x; acc.y += rad.y; acc.z += rad.z;
        radiance[slot] = acc;
    }
    if (INSTRUMENT) wave_add(&counters->shadow_triangles, tris);
}

// ---------------------------------------------------------------------------------------------
// K2 / K4, persistent form. Same per-ray visiting order as traverse() above (so results and the
// node / triangle counters are identical), restructured for wave64 efficiency:
//   * persistent waves: a wave claims TRACE_CHUNK consecutive rays with one global atomic and
//     refills finished lanes from its private range, so a wave is never held hostage by its longest
//     ray (rays visit between a handful and a few hundred nodes);
//   * one kind of work item per loop iteration and lane -- either one BVH node or (up to two) triangles of one leaf. Leaves are
//     stack items like inner nodes, so a lane that reached a leaf does not make the other 63 lanes
//     wait for up to four sequential triangle tests.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t TRACE_CHUNK_MAX = 512; // rays a wave claims per global atomic (the launch passes min(this, fair share))
// The ray range of a launch is cut into TRACE_SHARDS contiguous shards, each with its own claim counter on its
// own 64 B line: 64 x the atomic throughput of a single word (one word saturates near 88 atomics/us), and blocks
// that start on the same shard work on neighbouring rays. A wave that drains its shard steals from the next.
constexpr uint32_t TRACE_SHARDS = 64;   // == wave size: a drained wave probes all of them with one load per lane
constexpr uint32_t TRACE_SHARD_STRIDE = 16;   // uint32 words between shard counters
#ifndef HIPR_NODE_QUAD_GATHER
#define HIPR_NODE_QUAD_GATHER 0
#endif
constexpr int TRACE_SPILL_ENTRIES = 96;       // stack entries beyond the LDS stack, in scratch memory (OVERFLOW kernels only)

// MODE: which rays one launch serves.
//   TRACE_CLOSEST  the path queue (closest hit)                       -- bounce 0 and the stage-level parity entry point
//   TRACE_SHADOW   the shadow queue (any hit, transmittance)          -- the shadow rays of the last bounce
//   TRACE_FUSED    both: the closest-hit rays of bounce k and the shadow rays of bounce k - 1 are independent, so one launch
//                  takes them as one index space [0, n_closest + n_shadow). Every lane carries its own kind, waves refill
//                  from whatever is left, and the long-ray tail of one kind is filled with work of the other.
enum { TRACE_CLOSEST = 0, TRACE_SHADOW = 1, TRACE_FUSED = 2 };

// Waves per SIMD the persistent kernels are compiled for. With 16 LDS stack entries (4 KB per wave) the register file is the limit: measured on the
// atrium, 5 waves 60.4 ms of trace time per step, 6 waves (<= 80 VGPRs) 57.7, 7 waves (72 VGPRs) 57.8, 8 waves (64 VGPRs, spills) 60.3. With 32
// entries LDS holds five waves and the registers of a sixth are better spent (10 M triangle atrium: 116.7 ms at 93 VGPRs, 121.0 at 80).
#ifndef HIPR_TRACE_WAVES_SHALLOW
#define HIPR_TRACE_WAVES_SHALLOW 6
#endif
HD constexpr int trace_waves_per_simd(int stack_entries) { return stack_entries <= 24 ? HIPR_TRACE_WAVES_SHALLOW : 5; }
template <int STACK, int MODE, bool INSTRUMENT, bool OVERFLOW>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(trace_waves_per_simd(STACK)))) void k_trace_persistent(DeviceScene sc, PathState in, float4* hits, ShadowQueue q, float4* radiance,
                                                                  const uint32_t* closest_count_ptr, const uint32_t* shadow_count_ptr, uint32_t* work_counter,
                                                                  int refill_below, DeviceCounters* counters) {
    __shared__ int s_stack[STACK * TRACE_BLOCK];
    int* stack = s_stack + threadIdx.x;
#if HIPR_NODE_QUAD_GATHER
    // Node fetch by quads (see the node block): four planes of 64 x 16 B per wave, each padded by 16 B so that a lane's four b128 reads of its own node
    // fall on different banks than those of the lanes served in the same LDS cycle.
    constexpr int TILE_PLANE = 65;
    __shared__ uint4 s_tile[(TRACE_BLOCK / 64) * 4 * TILE_PLANE];
    uint4* tile = s_tile + (threadIdx.x >> 6) * 4 * TILE_PLANE;
#endif
    // Entries beyond the LDS stack (only trees whose worst case needs more than STACK entries are compiled with OVERFLOW) go
    // to a per-lane array in scratch memory; traversals rarely get that deep.
    int spill[OVERFLOW ? TRACE_SPILL_ENTRIES : 1];
    const uint32_t n_closest = MODE != TRACE_SHADOW ? *closest_count_ptr : 0u;
    const uint32_t n = n_closest + (MODE != TRACE_CLOSEST ? *shadow_count_ptr : 0u);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // every wave should get several chunks so the tail balances; small launches fall back to one wave-load per claim
    const uint32_t chunk_size = min(TRACE_CHUNK_MAX, max(64u, (n / (gridDim.x * (TRACE_BLOCK / 64u) * 2u)) & ~63u));

    uint32_t chunk_next = 0, chunk_end = 0;   // wave uniform
    // Blocks that share an XCD (blockIdx % 8: dispatch deals blocks round-robin over the XCDs) start on eight NEIGHBOURING shards, so one XCD's L2
    // serves one contiguous eighth of the ray queue -- rays of neighbouring pixels, which walk the same part of the BVH (HIPR_SHARD_MAP=0: blockIdx % 64).
#ifndef HIPR_XCD_SHARDS
#define HIPR_XCD_SHARDS 1
#endif
    uint32_t shard = HIPR_XCD_SHARDS ? ((blockIdx.x & 7u) * (TRACE_SHARDS / 8u) + ((blockIdx.x >> 3) % (TRACE_SHARDS / 8u))) : blockIdx.x % TRACE_SHARDS;
    bool exhausted = false;

    bool active = false, finished = false;    // finished: traversal done, result still in registers
    bool is_shadow = MODE == TRACE_SHADOW;     // per lane in the fused mode
    uint32_t ray_index = 0;
    f3 o = {0, 0, 0}, d = {0, 0, 1}, inv = {0, 0, 0}, ood = {0, 0, 0};
    float tmin = 0.0f, tmax = 0.0f;            // tmax: best distance so far (closest) or the ray extent (shadow)
    // Four registers with a meaning per ray kind (a lane is one or the other, never both):
    //   closest: pay_x, pay_y = barycentrics of the best hit, pay_z = bits of its id, pay_k = the triangle the ray left from
    //   shadow : pay_x, pay_y, pay_z = radiance carried by the ray,                  pay_k = radiance slot of the path
    float pay_x = 0.0f, pay_y = 0.0f, pay_z = __uint_as_float(HIPR_HIT_MISS);
    uint32_t pay_k = HIPR_NO_TRIANGLE;
    int cur = 0, sp = 0;
    uint32_t tri_cur = 0, tri_end = 0;
    uint32_t nodes = 0, tris = 0, shadow_nodes = 0, shadow_tris = 0;
    uint32_t diag_node_iterations = 0, diag_node_lanes = 0, diag_triangle_iterations = 0, diag_triangle_lanes = 0, diag_busy_lanes = 0, diag_refills = 0;   // lane 0 only
    uint32_t diag_pushes = 0, diag_pushes_16 = 0, diag_pushes_24 = 0;   // per lane

    // Work item encoding: >= 0 inner node index, < 0 leaf ~((first << 3) | (count - 1)), TRACE_DONE = nothing left.
    constexpr int TRACE_DONE = 0x7FFFFFFF;

    for (;;) {
        // ---- retire finished lanes (converged: every lane of the wave is here) --------------------------------------
        if (finished) {
            if (is_shadow) {
                if constexpr (MODE != TRACE_CLOSEST) {
                    float4 acc = radiance[pay_k];
                    acc.x += pay_x; acc.y += pay_y; acc.z += pay_z;
                    radiance[pay_k] = acc;
                }
            } else if constexpr (MODE != TRACE_SHADOW) {
                for (uint32_t li = 0; li < sc.light_count; ++li) {   // analytic area lights, LightSources.cu:31-70
                    const HiprLight l = load_light_uniform(sc.lights, li);
                    const uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
                    float t = -1e30f;
                    if (type == HIPR_LIGHT_SPHERE) { if (!(l.data[6] > 0.0f)) continue; t = ray_sphere(o, d, L3(l, 3), l.data[6]); }
                    else if (type == HIPR_LIGHT_SPOT) { if (!(l.data[6] > 0.0f)) continue; t = ray_disk(o, d, L3(l, 3), L3(l, 7), l.data[6]); }
                    else continue;
                    if (t > tmin && t < tmax) { tmax = t; pay_x = 0; pay_y = 0; pay_z = __uint_as_float(HIPR_HIT_LIGHT | li); }
                }
                hits[ray_index] = make_float4(tmax, pay_x, pay_y, pay_z);
            }
            finished = false;
        }
        // ---- refill idle lanes from the wave's private range -------------------------------------------------------
        const unsigned long long idle = wave_ballot(!active);
        if (INSTRUMENT && lane == 0 && idle && !exhausted) ++diag_refills;
        if (idle && !exhausted) {
            while (chunk_next >= chunk_end && !exhausted) {
                const uint32_t shard_begin = uint32_t((unsigned long long)n * shard / TRACE_SHARDS);
                const uint32_t shard_end = uint32_t((unsigned long long)n * (shard + 1u) / TRACE_SHARDS);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter + shard * TRACE_SHARD_STRIDE, chunk_size);
                base = __shfl(base, 0);
                if (base < shard_end - shard_begin) { chunk_next = shard_begin + base; chunk_end = min(chunk_next + chunk_size, shard_end); }
                else {
                    // This shard is drained: look at all 64 claim counters at once (lane k probes shard k; TRACE_SHARDS == wave size)
                    // and move to the next one that still has rays, instead of walking the shards one dependent load at a time.
                    const uint32_t my_size = uint32_t((unsigned long long)n * (lane + 1u) / TRACE_SHARDS) - uint32_t((unsigned long long)n * lane / TRACE_SHARDS);
                    const uint32_t claimed = __hip_atomic_load(work_counter + lane * TRACE_SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long open = wave_ballot(claimed < my_size);
                    if (!open) exhausted = true;
                    else {
                        const unsigned long long rotated = shard ? ((open >> shard) | (open << (64u - shard))) : open;
                        shard = (shard + uint32_t(__builtin_ctzll(rotated))) % TRACE_SHARDS;
                    }
                }
            }
            if (chunk_next < chunk_end) {
                const uint32_t idx = chunk_next + __popcll(idle & lt);
                if (!active && idx < chunk_end) {
                    bool dead = false;
                    float4 ro, rdv;
                    if (MODE == TRACE_FUSED) is_shadow = idx >= n_closest;
                    if (is_shadow) {
                        if constexpr (MODE != TRACE_CLOSEST) {
                            const uint32_t si = idx - n_closest;
                            ray_index = si;
                            ro = q.o_tmax[si]; rdv = q.d_slot[si];
                            const float4 rr = q.radiance[si];
                            pay_x = rr.x; pay_y = rr.y; pay_z = rr.z;
                            pay_k = __float_as_uint(rdv.w);
                            tmin = 0.0f; tmax = ro.w;
                        }
                    } else if constexpr (MODE != TRACE_SHADOW) {
                        ray_index = idx;
                        const uint2 meta = in.meta[idx];
                        dead = meta.x == HIPR_DEAD_SLOT;
                        pay_k = meta.y;
                        ro = in.o_tmin[idx]; rdv = in.d_pdf[idx];
                        tmin = ro.w; tmax = __builtin_inff();
                        pay_x = pay_y = 0.0f; pay_z = __uint_as_float(HIPR_HIT_MISS);
                    }
                    if (dead) hits[idx] = make_float4(0, 0, 0, __uint_as_float(HIPR_HIT_MISS));
                    else {
                        o = mk3(ro.x, ro.y, ro.z); d = mk3(rdv.x, rdv.y, rdv.z);
                        const f3 sd = {fabsf(d.x) > 1e-20f ? d.x : copysignf(1e-20f, d.x), fabsf(d.y) > 1e-20f ? d.y : copysignf(1e-20f, d.y),
                                       fabsf(d.z) > 1e-20f ? d.z : copysignf(1e-20f, d.z)};
                        inv = {1.0f / sd.x, 1.0f / sd.y, 1.0f / sd.z};
                        ood = o * inv;
                        sp = 0; cur = 0; tri_cur = tri_end = 0;
                        if (sc.wide_node_count == 0) finished = true;
                        else active = true;
                    }
                }
                chunk_next = min(chunk_next + uint32_t(__popcll(idle)), chunk_end);
            }
        }
        unsigned long long busy = wave_ballot(active);
        if (!busy) {
            if (wave_ballot(finished)) continue;
            if (exhausted) break;
            continue;
        }

        // ---- traverse. Every iteration runs ONE of the two blocks -- the one more lanes are waiting for -- so the
        // ---- 64 lanes are not serialised through both; a lane's own visiting order is unchanged.
        do {
            const bool tri_mode = active & (tri_cur < tri_end);
            const bool node_mode = active & !tri_mode;
            const unsigned long long tmask = wave_ballot(tri_mode), nmask = wave_ballot(node_mode);
            int next_item = TRACE_DONE;     // the item this lane continues with; selects only, no divergent state updates
            bool take_next = false, need_pop = false;
            if (INSTRUMENT && lane == 0) {
                const bool triangles = __popcll(tmask) > __popcll(nmask);
                diag_triangle_iterations += triangles; diag_triangle_lanes += triangles ? __popcll(tmask) : 0;
                diag_node_iterations += !triangles; diag_node_lanes += triangles ? 0 : __popcll(nmask);
                diag_busy_lanes += __popcll(tmask | nmask);
            }
            if (__popcll(tmask) > __popcll(nmask)) {
                // Up to two triangles of the leaf per triangle iteration, one after the other (same tests in the same order per ray): the fixed cost of
                // an iteration (vote, branch, pop / next-item logic) is paid once per pair. Measured on the atrium: 1 / 2 / 3 triangles 57.7 / 55.7 /
                // 56.3 ms of trace time per step; the same unrolling of the node step (a lane walking on into its nearest inner child) loses: 56.5 ms.
#ifndef HIPR_TRIANGLES_PER_ITERATION
#define HIPR_TRIANGLES_PER_ITERATION 2
#endif
                bool testing = tri_mode;
#pragma unroll
                for (int rep = 0; rep < HIPR_TRIANGLES_PER_ITERATION; ++rep)
                if (testing) {
                    const uint32_t i = tri_cur;
                    tri_cur = i + 1u;
                    if (INSTRUMENT) { tris += is_shadow ? 0u : 1u; shadow_tris += is_shadow ? 1u : 0u; }
                    const float4* tp = sc.trace_triangles + 3 * size_t(i);
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    TriangleTest test;
                    const bool hit = triangle_inside(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, test);
                    float t = 0.0f, u = 0.0f, v = 0.0f;
                    if (wave_any(hit)) triangle_hit_values(test, mk3(b.z, b.w, c.x), t, u, v);     // the lanes in this branch agree to skip the division
                    need_pop = tri_cur == tri_end;
                    if constexpr (MODE != TRACE_CLOSEST) {
                        if (is_shadow && hit && t > tmin && t < tmax) {
                            float coverage = 1.0f;
                            if (!(__float_as_uint(c.w) & HIPR_TRIANGLE_OPAQUE)) {
                                const HiprInstance& inst = sc.instances[__float_as_uint(c.y)];
                                coverage = material_coverage(sc, sc.materials[inst.material_index], triangle_texcoord(sc, inst, __float_as_uint(c.z), u, v));
                            }
                            pay_x *= 1.0f - coverage; pay_y *= 1.0f - coverage; pay_z *= 1.0f - coverage;
                            if (pay_x < 0.0000001f && pay_y < 0.0000001f && pay_z < 0.0000001f) {   // fully shadowed: the ray is done
                                pay_x = pay_y = pay_z = 0.0f;
                                need_pop = false; take_next = true; next_item = TRACE_DONE;
                            }
                        }
                    }
                    if constexpr (MODE != TRACE_SHADOW) {
                        const bool closer = !is_shadow & hit & (i != pay_k) & (t > tmin) & ((t < tmax) | ((t == tmax) & (i < __float_as_uint(pay_z))));
                        tmax = closer ? t : tmax; pay_x = closer ? u : pay_x; pay_y = closer ? v : pay_y; pay_z = closer ? __uint_as_float(i) : pay_z;
                    }
                    testing = !need_pop && !take_next;   // more triangles in this leaf, and the ray still wants them
                }
            }
#if HIPR_NODE_QUAD_GATHER
            else {
                // The 64 B nodes are fetched by QUADS of lanes: in instruction i the four lanes of a quad load the four 16 B quarters of the node of quad-mate i
                // (one contiguous 64 B read of the L1 instead of four 16 B reads in four instructions), straight into LDS (global_load_lds_dwordx4: lane l's
                // 16 B land at plane i + 16 l). Afterwards every lane reads its own node, which its quad assembled in plane (lane % 4). A divergent 16 B load
                // costs the CU's address / tag pipeline one cycle per lane whatever it returns; this way a node costs it one lane-load, not four.
                const uint32_t packed = (uint32_t(cur) << 1) | (node_mode ? 1u : 0u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t mate = i == 0 ? __builtin_amdgcn_mov_dpp(packed, 0x00, 0xF, 0xF, false) : i == 1 ? __builtin_amdgcn_mov_dpp(packed, 0x55, 0xF, 0xF, false)
                                        : i == 2 ? __builtin_amdgcn_mov_dpp(packed, 0xAA, 0xF, 0xF, false) : __builtin_amdgcn_mov_dpp(packed, 0xFF, 0xF, 0xF, false);
                    if (mate & 1u)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc.wide_nodes + 4 * size_t(mate >> 1) + (lane & 3u)),
                                                         (__attribute__((address_space(3))) void*)(tile + i * TILE_PLANE), 16, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the wave's loads have landed in its planes
            }
            if (!(__popcll(tmask) > __popcll(nmask)) && node_mode) {
                const uint4* np = tile + (lane & 3u) * TILE_PLANE + (lane & ~3u);
                const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
#else
            else if (node_mode) {
                // One compressed 4-wide node (HiprWideNode): 4 gathers of 16 B serve what ~2.1 BVH2 nodes (8.4 gathers) did.
                const uint4* np = sc.wide_nodes + 4 * size_t(cur);
                const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
#endif
#ifdef HIPR_EXTRA_NODE_LOADS
                // Sensitivity experiment (profiles/r03_ab_extra_node_loads.txt): N more 16 B loads per lane from the node's own line -- no new cache line, no new miss, only more
                // work for the address / tag pipeline. The pointer is laundered so that the loads are not merged with the four above.
                uint32_t extra_bits = 0u;
#pragma unroll
                for (int e = 0; e < HIPR_EXTRA_NODE_LOADS; ++e) {
                    unsigned long long laundered = (unsigned long long)(np + (e & 3));
                    asm volatile("" : "+v"(laundered));
                    const uint4 x = *reinterpret_cast<const uint4*>(laundered);
                    extra_bits |= (x.x == 0x7FC12345u) & (x.w == 0x7FC54321u);
                }
#endif
                if (INSTRUMENT) { nodes += is_shadow ? 0u : 1u; shadow_nodes += is_shadow ? 1u : 0u; }
                // slab distances of a quantised bound q: fma(float(q), A, B) with A = 2^(e - 127) * inv_d, B = fma(origin, inv_d, -ood)
                const float ax = __uint_as_float((w0.w & 0xFFu) << 23) * inv.x, ay = __uint_as_float(((w0.w >> 8) & 0xFFu) << 23) * inv.y,
                            az = __uint_as_float(((w0.w >> 16) & 0xFFu) << 23) * inv.z;
                const float bx = fmaf(__uint_as_float(w0.x), inv.x, -ood.x), by = fmaf(__uint_as_float(w0.y), inv.y, -ood.y), bz = fmaf(__uint_as_float(w0.z), inv.z, -ood.z);
                uint32_t key[4];
                int child[4] = {int(w3.x), int(w3.y), int(w3.z), int(w3.w)};
                // fma(q, A, B) is monotone in q with the sign of A, and qlo <= qhi: the entry bound of an axis is the low one where the
                // ray travels in +axis (A >= 0) and the high one otherwise -- the same values min / max of the pair would give.
                const uint32_t nx = ax >= 0.0f ? w1.x : w1.w, fx = ax >= 0.0f ? w1.w : w1.x;
                const uint32_t ny = ay >= 0.0f ? w1.y : w2.x, fy = ay >= 0.0f ? w2.x : w1.y;
                const uint32_t nz = az >= 0.0f ? w1.z : w2.y, fz = az >= 0.0f ? w2.y : w1.z;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float x0 = fmaf(float((nx >> (8 * k)) & 0xFFu), ax, bx), x1 = fmaf(float((fx >> (8 * k)) & 0xFFu), ax, bx);
                    const float y0 = fmaf(float((ny >> (8 * k)) & 0xFFu), ay, by), y1 = fmaf(float((fy >> (8 * k)) & 0xFFu), ay, by);
                    const float z0 = fmaf(float((nz >> (8 * k)) & 0xFFu), az, bz), z1 = fmaf(float((fz >> (8 * k)) & 0xFFu), az, bz);
                    const float tnear = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
                    float tfar = fminf(fminf(x1, y1), z1);
                    tfar = fminf(tfar, tmax) * 1.0000004f;
                    const bool hit = (child[k] != HIPR_WIDE_EMPTY) & (tnear <= tfar);
                    key[k] = hit ? ((__float_as_uint(tnear) & 0x7FFFFFFCu) | uint32_t(k)) : 0xFFFFFFFFu;   // distinct keys: a total order
                }
                // sorting network, nearest first
#define HIPR_ORDER(i, j) { const bool s_ = key[j] < key[i]; const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); \
                           const int ci_ = s_ ? child[j] : child[i], cj_ = s_ ? child[i] : child[j]; key[i] = lo_; key[j] = hi_; child[i] = ci_; child[j] = cj_; }
                HIPR_ORDER(0, 1) HIPR_ORDER(2, 3) HIPR_ORDER(0, 2) HIPR_ORDER(1, 3) HIPR_ORDER(1, 2)
#undef HIPR_ORDER
                // the nearest hit child is next; the others wait on the stack, farthest first
#pragma unroll
                for (int k = 3; k >= 1; --k) {
                    if (key[k] != 0xFFFFFFFFu) {
                        if (!OVERFLOW || sp < STACK) stack[sp * TRACE_BLOCK] = child[k];
                        else spill[sp - STACK] = child[k];
                        if (INSTRUMENT) { ++diag_pushes; diag_pushes_16 += sp >= 16; diag_pushes_24 += sp >= 24; }
                        ++sp;
                    }
                }
                take_next = key[0] != 0xFFFFFFFFu;
#ifdef HIPR_EXTRA_NODE_LOADS
                take_next = take_next | (extra_bits != 0u);
#endif
                next_item = child[0];
                need_pop = !take_next;
            }
            if (need_pop) {
                const int below = sp > 0 ? sp - 1 : 0;
                int popped = stack[(OVERFLOW && below >= STACK ? 0 : below) * TRACE_BLOCK];
                if (OVERFLOW && below >= STACK) popped = spill[below - STACK];
                next_item = sp > 0 ? popped : TRACE_DONE;
                sp = below;
                take_next = true;
            }
            if (take_next) {
                const bool done = next_item == TRACE_DONE, leaf = next_item < 0;
                const uint32_t code = uint32_t(~next_item), first_triangle = code >> 3;
                tri_cur = leaf ? first_triangle : 0u;
                tri_end = leaf ? first_triangle + (code & 7u) + 1u : 0u;
                cur = (leaf | done) ? cur : next_item;
                active = !done;
                finished = done;
            }
            busy = wave_ballot(active);
        } while (busy && (exhausted || __popcll(busy) >= refill_below));
    }

    if (INSTRUMENT) {
        wave_add(&counters->node_iterations, diag_node_iterations); wave_add(&counters->node_lanes, diag_node_lanes);
        wave_add(&counters->triangle_iterations, diag_triangle_iterations); wave_add(&counters->triangle_lanes, diag_triangle_lanes);
        wave_add(&counters->busy_lanes, diag_busy_lanes); wave_add(&counters->refills, diag_refills);
        wave_add(&counters->pushes, diag_pushes); wave_add(&counters->pushes_past_16, diag_pushes_16); wave_add(&counters->pushes_past_24, diag_pushes_24);
        if (MODE != TRACE_SHADOW) { wave_add(&counters->closest_nodes, nodes); wave_add(&counters->closest_triangles, tris); }
        if (MODE != TRACE_CLOSEST) { wave_add(&counters->shadow_nodes, shadow_nodes); wave_add(&counters->shadow_triangles, shadow_tris); }
    }
}

// The 128-byte shading record of triangle t (what k_shade fetches per hit; layout below): the body of k_build_shade_triangles, also run by the host build of the
// shade stage (tests/native/DeviceShadeHost.hip).
HD void build_shade_triangle_record(const DeviceScene& sc, uint32_t t, float4* out) {
    const float4 tc = sc.triangles[3 * size_t(t) + 2];
    const HiprInstance inst = sc.instances[__float_as_uint(tc.y)];
    const uint32_t prim = __float_as_uint(tc.z);
    const uint32_t* idx = sc.indices + 3 * size_t(inst.index_offset + prim);
    const uint32_t i[3] = {idx[0], idx[1], idx[2]};
    const float* M = inst.object_to_world;
    const uint32_t words[3] = {uint32_t(inst.material_index), uint32_t(inst.instance_id), inst.mesh_flags};
    float4* q = out + SHADE_TRIANGLE_QUADS * size_t(t) + 3;
    for (int k = 0; k < 3; ++k) {
        f3 n = {0, 0, 0};
        if (inst.mesh_flags & HIPR_MESH_NORMALS) {
            const f3 o = decode_octahedral(sc.geometry[inst.vertex_offset + i[k]].w);
            n = mk3(M[0] * o.x + M[1] * o.y + M[2] * o.z, M[4] * o.x + M[5] * o.y + M[6] * o.z, M[8] * o.x + M[9] * o.y + M[10] * o.z);
        }
        q[k] = make_float4(n.x, n.y, n.z, __uint_as_float(words[k]));
    }
    float2 uv[3] = {{0, 0}, {0, 0}, {0, 0}};
    if (inst.mesh_flags & HIPR_MESH_TEXCOORDS) for (int k = 0; k < 3; ++k) uv[k] = sc.texcoords[inst.vertex_offset + i[k]];
    uint32_t tint[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (inst.mesh_flags & HIPR_MESH_TINTS) for (int k = 0; k < 3; ++k) tint[k] = sc.tints[inst.vertex_offset + i[k]];
    q[3] = make_float4(uv[0].x, uv[0].y, uv[1].x, uv[1].y);
    q[4] = make_float4(uv[2].x, uv[2].y, __uint_as_float(tint[0]), __uint_as_float(tint[1]));
    q[-3] = sc.triangles[3 * size_t(t)];
    q[-2] = sc.triangles[3 * size_t(t) + 1];
    q[-1] = make_float4(tc.x, tc.y, tc.z, __uint_as_float(tint[2]));
}

#ifndef HIPR_SHADE_TU
// ---------------------------------------------------------------------------------------------
// Shading records. The reference resolves a hit through instance -> mesh buffers -> index -> vertex (TriangleAttributes.cu:35-84,
// four dependent fetches); here everything the shade kernel interpolates is flattened per world-space triangle when the scene
// is uploaded, so a hit costs ONE dependent fetch of ONE 128 B cache line (round 3: the world-space positions sit in the record too -- the shade kernel's
// gathers miss the L2, and a 48 B triangle plus a 96 B record used to touch three to four lines per hit):
//   quad 0..2: the triangle as in `triangles` (v0, v1, v2, instance index, primitive index), .w of quad 2 = bits(tint2)
//   quad 3..5: object-to-world transformed vertex normal i (not normalised: sum(w_i * M n_i) = M sum(w_i * n_i), so the
//              interpolated normal is the one the per-hit transform gives), .w = bits(material index | instance id | mesh flags)
//   quad 6   : uv0.xy, uv1.xy       quad 7: uv2.xy, bits(tint0), bits(tint1)
// Per-vertex emission (rare) still goes through the instance.
// ---------------------------------------------------------------------------------------------
// What the trace kernels read per triangle: the vertex and the two edges the test works on (the edges rounded once, here, as
// the oracle rounds them), ids and flags where the uploaded triangle has them.
__global__ __launch_bounds__(256) void k_build_trace_triangles(const float4* __restrict__ triangles, uint32_t triangle_count, float4* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= triangle_count) return;
    const float4 a = triangles[3 * size_t(t)], b = triangles[3 * size_t(t) + 1], c = triangles[3 * size_t(t) + 2];
    const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y) - v0, e2 = mk3(b.z, b.w, c.x) - v0;
    out[3 * size_t(t)] = make_float4(v0.x, v0.y, v0.z, e1.x);
    out[3 * size_t(t) + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    out[3 * size_t(t) + 2] = make_float4(e2.z, c.y, c.z, c.w);
}

__global__ __launch_bounds__(256) void k_build_shade_triangles(DeviceScene sc, float4* out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= sc.triangle_count) return;
    build_shade_triangle_record(sc, t, out);
}

// ---------------------------------------------------------------------------------------------
// K5c: the order in which the shade kernel takes the rays of a bounce. A wave of k_shade runs the surface code -- attributes, material, three RIS light
// candidates, BSDF sample: thousands of instructions -- as long as ONE of its 64 rays hit a triangle, and 42 % of the atrium's traced rays did not
// (they escaped, hit a light, or sit in a dead slot): measured 28 of 64 lanes live per instruction. This pass lists the indices of the rays that hit a
// triangle from the front of `order` and all others from its back (one 64-bit atomic per block of 256 rays reserves both ranges; inside a block the rays
// keep their order), so that all but one or two of the shade kernel's batches are of one kind. Which rays are shaded, and what each yields, is unchanged.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t CLASSIFY_ROUNDS = 16;      // rounds of 256 rays a block lists per reservation: 4096 rays per atomic (one 64-bit atomic on one address costs ~8 ns device-wide)
// Round 4: the surface hits are listed in two classes. The shade kernel's instruction stream forks on the material where a COAT is present -- a second
// specular layer: make_default prepares it, every BSDF evaluation (three light candidates and the sample) adds its lobe -- and a wave that holds one
// coated hit among its 64 walks that code for all of them (atrium: 4 of 25 materials are coated, i.e. nearly every wave of 64 mixed hits). `triangle_class`
// holds one byte per scene triangle (bit 0: its material has a coat; written at upload), the coated hits go to a list of their own (`order_coat`, counted in
// taken[1]) and the shade kernel takes the entries as [plain surface hits | coated surface hits | everything else]: order[j] for j < plain,
// order_coat[j - plain] for the next `coated`, order[j] again behind them (the others were listed from the back of `order`: they end at n - 1).
// MEASURED (profiles/r04_ab_shade_classes.txt): frames bit-identical (tests/test_gpu_coverage.py), atrium shade 17.1 -> 17.3 ms per step: only 4.6 % of the
// atrium's surface hits are coated (the four coated materials cover little area), so 95 % of the waves did skip the coat code -- and the byte gather per
// hit in this pass cost more than that code did. Off by default (HIPR_SHADE_CLASSES=1 turns it on); the listing is then round 3's two kinds.
template <bool CLASSES>
__global__ __launch_bounds__(256) void k_classify_hits(const float4* __restrict__ hits, const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ order,
                                                       unsigned long long* taken /* [0] low word: plain surface hits listed, high word: others listed; [1]: coated surface hits listed */,
                                                       const unsigned char* __restrict__ triangle_class, uint32_t* __restrict__ order_coat) {
    __shared__ uint32_t s_surface[4], s_other[4], s_coat[4];
    static_assert(CLASSIFY_ROUNDS * 4u == 64u, "the prefix of the (round, wave) table is one wave-wide scan");
    __shared__ uint32_t s_table[3][CLASSIFY_ROUNDS * 4];      // per class: a wave's rays of each round, then their exclusive prefix in (round, wave) order
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_coat_base;
    const uint32_t n = *count_ptr;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t chunk = blockIdx.x * (256u * CLASSIFY_ROUNDS); chunk < n; chunk += gridDim.x * (256u * CLASSIFY_ROUNDS)) {
        // pass 1: what the chunk holds, one bit per round and thread
        uint32_t surface_bits = 0u, valid_bits = 0u, coat_bits = 0u;
#pragma unroll
        for (uint32_t r = 0; r < CLASSIFY_ROUNDS; ++r) {
            const uint32_t i = chunk + r * 256u + threadIdx.x;
            const uint32_t id = i < n ? __float_as_uint(hits[i].w) : HIPR_HIT_MISS;
            const bool surface = i < n && id != HIPR_HIT_MISS && !(id & HIPR_HIT_LIGHT);
            const bool coated = CLASSES && surface && (triangle_class[id] & 1u);
            surface_bits |= (surface && !coated) ? (1u << r) : 0u;
            coat_bits |= coated ? (1u << r) : 0u;
            valid_bits |= i < n ? (1u << r) : 0u;
        }
        uint32_t surfaces = uint32_t(__popc(surface_bits)), coats = uint32_t(__popc(coat_bits)), others = uint32_t(__popc(valid_bits & ~(surface_bits | coat_bits)));
        for (int off = 32; off > 0; off >>= 1) { surfaces += __shfl_xor(surfaces, off); others += __shfl_xor(others, off); coats += __shfl_xor(coats, off); }
        if (lane == 0) { s_surface[wave] = surfaces; s_other[wave] = others; s_coat[wave] = coats; }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = atomicAdd(taken, (unsigned long long)(s_surface[0] + s_surface[1] + s_surface[2] + s_surface[3]) |
                                      (unsigned long long)(s_other[0] + s_other[1] + s_other[2] + s_other[3]) << 32);
            const uint32_t block_coats = CLASSES ? s_coat[0] + s_coat[1] + s_coat[2] + s_coat[3] : 0u;
            s_coat_base = block_coats ? uint32_t(atomicAdd(taken + 1, (unsigned long long)block_coats)) : 0u;
        }
        __syncthreads();
        // pass 2: places inside the block's ranges, rays in (round, thread) order. Every wave leaves its count of each class for each round in a table, the first wave
        // turns the table into exclusive prefixes in (round, wave) order -- one barrier on either side of that instead of the two per ROUND of a running sum (round 4:
        // the pass was the latency of its 34 barriers per chunk, profiles/r04_ab_knobs.txt) -- then every thread places its rays.
        const uint32_t front = uint32_t(s_base), back = uint32_t(s_base >> 32), coat_front = s_coat_base;
        unsigned long long surface_masks[CLASSIFY_ROUNDS], other_masks[CLASSIFY_ROUNDS], coat_masks[CLASSES ? CLASSIFY_ROUNDS : 1];
#pragma unroll
        for (uint32_t r = 0; r < CLASSIFY_ROUNDS; ++r) {
            const bool surface = (surface_bits >> r) & 1u, coated = (coat_bits >> r) & 1u, other = ((valid_bits & ~(surface_bits | coat_bits)) >> r) & 1u;
            surface_masks[r] = wave_ballot(surface); other_masks[r] = wave_ballot(other);
            if (CLASSES) coat_masks[r] = wave_ballot(coated);
            if (lane == 0) {
                s_table[0][r * 4u + wave] = uint32_t(__popcll(surface_masks[r])); s_table[1][r * 4u + wave] = uint32_t(__popcll(other_masks[r]));
                if (CLASSES) s_table[2][r * 4u + wave] = uint32_t(__popcll(coat_masks[r]));
            }
        }
        __syncthreads();
        if (wave == 0) {        // 16 rounds x 4 waves = 64 entries: one wave-wide exclusive scan per class
#pragma unroll
            for (int c = 0; c < (CLASSES ? 3 : 2); ++c) {
                const uint32_t mine = s_table[c][lane];
                uint32_t inclusive = mine;
                for (int off = 1; off < 64; off <<= 1) { const uint32_t up = __shfl_up(inclusive, off); inclusive += lane >= uint32_t(off) ? up : 0u; }
                s_table[c][lane] = inclusive - mine;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < CLASSIFY_ROUNDS; ++r) {
            const bool surface = (surface_bits >> r) & 1u, coated = (coat_bits >> r) & 1u, other = ((valid_bits & ~(surface_bits | coat_bits)) >> r) & 1u;
            const uint32_t i = chunk + r * 256u + threadIdx.x;
            if (surface) order[front + s_table[0][r * 4u + wave] + uint32_t(__popcll(surface_masks[r] & lt))] = i;
            if (CLASSES && coated) order_coat[coat_front + s_table[2][r * 4u + wave] + uint32_t(__popcll(coat_masks[r] & lt))] = i;
            if (other) order[n - 1u - (back + s_table[1][r * 4u + wave] + uint32_t(__popcll(other_masks[r] & lt)))] = i;
        }
        __syncthreads();        // the table is written again by the next chunk
    }
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

// Folds samples [first_sample, first_sample + sample_count) of the traced pass into the running mean as accumulations first_accumulation, + 1, ...
__global__ __launch_bounds__(256) void k_accumulate(FrameInfo frame, uint32_t first_sample, uint32_t sample_count, uint32_t first_accumulation, const float4* radiance,
                                                     double4* accumulation, ushort4* out, uint32_t out_pitch, float depth_normalizer) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    const uint32_t per_sample = frame.owned_tiles * 64u;
    uint32_t x = 0, y = 0;
    const bool mine = k < per_sample && owned_pixel(frame, k, x, y);
#if HIPR_PIXEL_MAJOR_SLOTS
    // The block's 256 pixels hold their samples in 256 runs of samples_per_pass float4 (pixel-major slots). A thread reading its own run 16 bytes at a time
    // makes every load of a wave touch 64 lines (measured: 1.0 ms per 32-sample pass at 1080p against 0.2 ms for the sample-major order); so the block brings
    // eight samples of all its pixels at a time through LDS -- consecutive threads load consecutive float4 of a run, 128 contiguous bytes per pixel -- and every
    // thread then folds its own eight IN ORDER (the running mean is order dependent: accumulation a before a + 1).
    constexpr uint32_t CHUNK = 8, ROW = CHUNK + 1;      // one float4 of padding per pixel row: the threads' reads of column c fall on different banks
    __shared__ float4 s_tile[256 * ROW];
    double4 acc = mine ? accumulation[k] : double4{0, 0, 0, 0};
    const size_t block_first = size_t(blockIdx.x) * 256u * frame.samples_per_pass + first_sample;
    for (uint32_t c0 = 0; c0 < sample_count; c0 += CHUNK) {
        const uint32_t columns = min(CHUNK, sample_count - c0);
        for (uint32_t e = threadIdx.x; e < 256u * columns; e += 256u) {
            const uint32_t row = e / columns, column = e - row * columns;
            if (blockIdx.x * 256u + row < per_sample) s_tile[row * ROW + column] = radiance[block_first + size_t(row) * frame.samples_per_pass + c0 + column];
        }
        __syncthreads();
        if (mine)
            for (uint32_t c = 0; c < columns; ++c) {
                const float4 r = s_tile[threadIdx.x * ROW + c];
                const uint32_t a = first_accumulation + c0 + c;
                if (a != 0) {
                    const double t = 1.0 / (a + 1.0);
                    acc.x = acc.x + (double(r.x) - acc.x) * t;
                    acc.y = acc.y + (double(r.y) - acc.y) * t;
                    acc.z = acc.z + (double(r.z) - acc.z) * t;
                } else { acc.x = r.x; acc.y = r.y; acc.z = r.z; }
                acc.w = 1.0;
            }
        __syncthreads();
    }
    if (!mine) return;
#else
    if (!mine) return;
    double4 acc = accumulation[k];
    for (uint32_t s = 0; s < sample_count; ++s) {
        const float4 r = radiance[(first_sample + s) * per_sample + k];
        const uint32_t a = first_accumulation + s;
        if (a != 0) {
            const double t = 1.0 / (a + 1.0);
            acc.x = acc.x + (double(r.x) - acc.x) * t;
            acc.y = acc.y + (double(r.y) - acc.y) * t;
            acc.z = acc.z + (double(r.z) - acc.z) * t;
        } else { acc.x = r.x; acc.y = r.y; acc.z = r.z; }
        acc.w = 1.0;
    }
#endif
    accumulation[k] = acc;
    if (out) {
        const size_t dst = frame.tile_stride == 1 ? size_t(x) + size_t(y) * out_pitch : size_t(k);
        if (depth_normalizer != 0.0f) {   // depth_RPG: output depth / max_depth (SimpleRGPs.cu:241-258)
            const unsigned short d = float_to_half_bits(float(acc.x) / depth_normalizer);
            out[dst] = make_ushort4(d, d, d, float_to_half_bits(1.0f));
        } else
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

// The presentation blit of the adaptor (DX11OptiXAdaptor/Adaptor.cpp:96-100 pixel shader): the renderer writes row 0 at
// the bottom, the back buffer has row 0 at the top.
__global__ __launch_bounds__(256) void k_present_flipped(const ushort4* pixels, uint32_t pitch, uint32_t width, uint32_t height, ushort4* backbuffer,
                                                          uint32_t backbuffer_pitch) {
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    backbuffer[size_t(x) + size_t(y) * backbuffer_pitch] = pixels[size_t(x) + size_t(height - y - 1u) * pitch];
}

// Debug / parity helpers -------------------------------------------------------------------------
// Evaluates BOTH forms of the sampler (the XOR loop used by k_generate and the LDS-table form used by k_shade) and
// poisons the output when they disagree, so the bit-exact test against the oracle covers the two.
__global__ void k_debug_sobol(const uint32_t* triples, uint32_t n, uint32_t* out, const uint32_t* sobol_tables) {
    __shared__ uint32_t s_sobol[SOBOL_TABLE_WORDS];
    for (uint32_t w = threadIdx.x; w < SOBOL_TABLE_WORDS; w += blockDim.x) s_sobol[w] = sobol_tables[w];
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u4 s = sobol4ui(triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]);
    u4 t = sobol4ui_tables(triples[3 * i], triples[3 * i + 1], triples[3 * i + 2], s_sobol);
    if (s.x != t.x || s.y != t.y || s.z != t.z || s.w != t.w) s.x = s.y = s.z = s.w = 0xDEADBEEFu;
    out[4 * i] = s.x; out[4 * i + 1] = s.y; out[4 * i + 2] = s.z; out[4 * i + 3] = s.w;
}

#endif // HIPR_SHADE_TU

} // namespace hipr
