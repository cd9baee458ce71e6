// oracle/integrator.cpp -- see integrator.h. TEST INFRASTRUCTURE ONLY.
#include "integrator.h"

#include <cmath>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------------------------
// Camera rays: fill_ray_info + initialize_monte_carlo_payload (ORS/SimpleRGPs.cu:44-72)
// ---------------------------------------------------------------------------------------------
static inline float4 mul4x4(const float* m, float4 v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w,
            m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w,
            m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w,
            m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w};
}
static inline float3 mul3x3(const float* m, float3 v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}

void generate_camera_ray(const HiprCameraState& cam, int x, int y, int width, int height, uint32_t accumulation,
                         float3& origin, float3& direction) {
    uint32_t pixel_hash = rng::pcg2d(uint32_t(x), uint32_t(y)).x;
    float2 jitter = {0.5f, 0.5f};
    if (accumulation != 0) {
        float4 s = rng::sample4f(accumulation, pixel_hash, 0u);   // dimension = 8 * bounces(0) + CAMERA_PARAMETERS(0)
        jitter = {s.x, s.y};
    }
    float2 screen_pos = {float(x) + jitter.x, float(y) + jitter.y};
    float2 viewport_pos = {screen_pos.x / float(width), screen_pos.y / float(height)};

    float4 ndc_near = {viewport_pos.x * 2.0f - 1.0f, viewport_pos.y * 2.0f - 1.0f, -1.0f, 1.0f};
    float4 scaled_near_world = mul4x4(cam.inverse_view_projection_matrix, ndc_near);
    origin = make_float3(scaled_near_world) / scaled_near_world.w;

    float4 ndc_far = {ndc_near.x, ndc_near.y, 1.0f, 1.0f};
    float4 scaled_view = mul4x4(cam.inverse_projection_matrix, ndc_far);
    direction = normalize(mul3x3(cam.view_to_world_rotation, make_float3(scaled_view)));
}

// ---------------------------------------------------------------------------------------------
// Intersection arithmetic shared (by specification, not by code) with the HIP kernels.
// ---------------------------------------------------------------------------------------------
static inline float dot_fma(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float3 cross_fma(float3 a, float3 b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}

// Moeller-Trumbore on v0, e1 = v1 - v0, e2 = v2 - v0 with the inside decision on the unnormalised barycentrics (the sign of the
// determinant folded in by flipping sign bits) and the division only for rays inside: kernels.h triangle_inside / triangle_hit_values.
bool intersect_triangle(const HiprTriangle& tri, float3 o, float3 d, float& t, float& u, float& v) {
    float3 v0 = {tri.v0[0], tri.v0[1], tri.v0[2]};
    float3 e1 = make_float3(tri.v1[0], tri.v1[1], tri.v1[2]) - v0;
    float3 e2 = make_float3(tri.v2[0], tri.v2[1], tri.v2[2]) - v0;
    float3 p = cross_fma(d, e2);
    float det = dot_fma(e1, p);
    float3 tv = o - v0;
    float un = dot_fma(tv, p);
    float3 q = cross_fma(tv, e1);
    float vn = dot_fma(d, q);
    float us = std::signbit(det) ? -un : un, vs = std::signbit(det) ? -vn : vn;
    if (!((det != 0.0f) && (us >= 0.0f) && (vs >= 0.0f) && (us + vs <= std::fabs(det))))
        return false;
    float inv = 1.0f / det;
    u = un * inv;
    v = vn * inv;
    t = dot_fma(e2, q) * inv;
    return true;
}

static inline void consider_triangle(const HiprSceneDesc& scene, uint32_t i, const Ray& ray, uint32_t skip, Hit& best) {
    if (i == skip)
        return;
    float t, u, v;
    if (!intersect_triangle(scene.triangles[i], ray.origin, ray.direction, t, u, v))
        return;
    if (!(t > ray.tmin))
        return;
    if (t < best.t || (t == best.t && i < best.id))
        best = {t, u, v, i};
}

// ---------------------------------------------------------------------------------------------
// Exhaustive search over ITEMS (kernels.h "Exhaustive-search items"): a triangle, or two triangles of one instance that form a
// parallelogram (a, b, c) + (a, c, d), tested with one solve against b - a and d - a. Restated here from the specification, not
// shared with the product: the pairing rule (first later triangle that fits, bit-identical shared corners, d = a + (c - b) within
// 1e-6 of the longest edge component, only in scenes of at most 64 triangles), the half decision on the unnormalised values and the corner
// weights (1 - s, s - r, r) / (1 - r, s, r - s) must all agree for hits to be bit-identical.
// ---------------------------------------------------------------------------------------------
struct SearchItem {
    float3 origin, edge1, edge2;
    bool parallelogram;
    uint32_t triangle[2];   // scene triangle of the half s >= r and of the half s < r
    int rotation[2];        // which stored vertex of each triangle is the shared corner a
};

static inline float3 vertex_of(const HiprTriangle& tri, int k) {
    const float* p = k % 3 == 0 ? tri.v0 : (k % 3 == 1 ? tri.v1 : tri.v2);
    return {p[0], p[1], p[2]};
}
static inline bool identical(float3 p, float3 q) { return std::memcmp(&p, &q, sizeof(float3)) == 0; }

static std::vector<SearchItem> make_search_items(const HiprSceneDesc& scene) {
    const uint32_t n = scene.triangle_count;
    std::vector<SearchItem> items;
    std::vector<char> taken(n, 0);
    for (uint32_t first = 0; first < n; ++first) {
        if (taken[first]) continue;
        const HiprTriangle& ta = scene.triangles[first];
        SearchItem item = {vertex_of(ta, 0), vertex_of(ta, 1) - vertex_of(ta, 0), vertex_of(ta, 2) - vertex_of(ta, 0), false, {first, first}, {0, 0}};
        for (uint32_t second = first + 1; n <= 64 && second < n && !item.parallelogram; ++second) {
            const HiprTriangle& tb = scene.triangles[second];
            if (taken[second] || tb.instance_index != ta.instance_index || tb.flags != ta.flags) continue;
            for (int turn_a = 0; turn_a < 3 && !item.parallelogram; ++turn_a)
                for (int turn_b = 0; turn_b < 3 && !item.parallelogram; ++turn_b) {
                    float3 a = vertex_of(ta, turn_a), b = vertex_of(ta, turn_a + 1), c = vertex_of(ta, turn_a + 2), d = vertex_of(tb, turn_b + 2);
                    if (!identical(a, vertex_of(tb, turn_b)) || !identical(c, vertex_of(tb, turn_b + 1))) continue;
                    float3 e1 = b - a, e2 = d - a, fourth = a + (c - b);
                    float longest = std::fmax(std::fmax(std::fmax(std::fabs(e1.x), std::fabs(e2.x)), std::fmax(std::fabs(e1.y), std::fabs(e2.y))), std::fmax(std::fabs(e1.z), std::fabs(e2.z)));
                    float off = std::fmax(std::fmax(std::fabs(d.x - fourth.x), std::fabs(d.y - fourth.y)), std::fabs(d.z - fourth.z));
                    if (off <= 1e-6f * longest) {
                        item = {a, e1, e2, true, {first, second}, {turn_a, turn_b}};
                        taken[second] = 1;
                    }
                }
        }
        items.push_back(item);
    }
    return items;
}

// The items of a scene are a function of its triangles. They are built once per API call: every extern "C" entry that searches
// exhaustively starts with reset_search_items() (scene memory is reused between calls, a pointer is no identity across them), and
// within a call each thread remembers the list it looked up last.
static std::mutex items_guard;
static std::map<std::pair<const HiprTriangle*, uint32_t>, std::vector<SearchItem>> items_cache;
static std::atomic<uint64_t> items_generation{1};
void reset_search_items() {
    std::lock_guard<std::mutex> lock(items_guard);
    items_cache.clear();
    ++items_generation;
}
static const std::vector<SearchItem>& search_items(const HiprSceneDesc& scene) {
    thread_local const HiprTriangle* last_triangles = nullptr;
    thread_local uint32_t last_count = 0;
    thread_local uint64_t last_generation = 0;
    thread_local const std::vector<SearchItem>* last_items = nullptr;
    if (last_items && last_triangles == scene.triangles && last_count == scene.triangle_count && last_generation == items_generation.load())
        return *last_items;
    std::lock_guard<std::mutex> lock(items_guard);
    auto key = std::make_pair(scene.triangles, scene.triangle_count);
    auto found = items_cache.find(key);
    if (found == items_cache.end()) found = items_cache.emplace(key, make_search_items(scene)).first;
    last_triangles = scene.triangles; last_count = scene.triangle_count; last_generation = items_generation.load(); last_items = &found->second;
    return *last_items;
}

uint32_t search_item_count(const HiprSceneDesc& scene) { return uint32_t(search_items(scene).size()); }

struct ItemHit { bool inside; int half; float t, u, v; };
static inline ItemHit intersect_item(const SearchItem& item, float3 o, float3 d) {
    float3 p = cross_fma(d, item.edge2);
    float det = dot_fma(item.edge1, p);
    float3 tv = o - item.origin;
    float un = dot_fma(tv, p);
    float3 q = cross_fma(tv, item.edge1);
    float vn = dot_fma(d, q);
    float us = std::signbit(det) ? -un : un, vs = std::signbit(det) ? -vn : vn, limit = std::fabs(det);
    bool upper = item.parallelogram ? (us <= limit && vs <= limit) : (us + vs <= limit);
    if (!((det != 0.0f) && (us >= 0.0f) && (vs >= 0.0f) && upper))
        return {false, 0, 0, 0, 0};
    int half = item.parallelogram && us < vs ? 1 : 0;
    float inv = 1.0f / det, s = un * inv, r = vn * inv;
    ItemHit hit = {true, half, dot_fma(item.edge2, q) * inv, s, r};
    if (item.parallelogram) {
        // corner weights of the half in the order (a, b, c) or (a, c, d); the triangle's stored vertex k is corner (k - rotation) mod 3
        float weights[3] = {half ? 1.0f - r : 1.0f - s, half ? s : s - r, half ? r - s : r};
        int rotation = item.rotation[half];
        hit.u = weights[(1 - rotation + 3) % 3];
        hit.v = weights[(2 - rotation + 3) % 3];
    }
    return hit;
}

Hit closest_hit_bruteforce(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip, TraversalCounters* counters) {
    Hit best = {ray.tmax, 0, 0, HIT_MISS};
    const std::vector<SearchItem>& items = search_items(scene);
    for (const SearchItem& item : items) {
        ItemHit hit = intersect_item(item, ray.origin, ray.direction);
        uint32_t id = item.triangle[hit.half];
        if (!hit.inside || id == skip || !(hit.t > ray.tmin))
            continue;
        if (hit.t < best.t || (hit.t == best.t && id < best.id))
            best = {hit.t, hit.u, hit.v, id};
    }
    if (counters) counters->triangles += items.size();   // the exhaustive kernels count every item of every ray
    return best;
}

struct SlabRay { float3 inv_d, ood; };
static inline SlabRay make_slab_ray(const Ray& ray) {
    auto safe = [](float d) { return fabsf(d) > 1e-20f ? d : copysignf(1e-20f, d); };
    float3 inv = {1.0f / safe(ray.direction.x), 1.0f / safe(ray.direction.y), 1.0f / safe(ray.direction.z)};
    return {inv, ray.origin * inv};
}
// Returns entry distance or -1 if the child box is missed within [tmin, tmax].
static inline bool slab(const SlabRay& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float tmin, float tmax, float& tnear) {
    float x0 = fmaf(lox, r.inv_d.x, -r.ood.x), x1 = fmaf(hix, r.inv_d.x, -r.ood.x);
    float y0 = fmaf(loy, r.inv_d.y, -r.ood.y), y1 = fmaf(hiy, r.inv_d.y, -r.ood.y);
    float z0 = fmaf(loz, r.inv_d.z, -r.ood.z), z1 = fmaf(hiz, r.inv_d.z, -r.ood.z);
    tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
    float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    tfar = fminf(tfar, tmax) * 1.0000004f;   // ~3 ulp slack: boxes and hits that tie within rounding are still visited
    return tnear <= tfar;
}

// Ordered BVH2 traversal; the visiting order is part of the specification so that the node /
// triangle counters of the oracle and the HIP kernels agree ("Traversal order").
template <typename LeafFn>
static inline void traverse(const HiprSceneDesc& scene, const Ray& ray, float& tmax, TraversalCounters* counters, LeafFn&& leaf) {
    if (scene.node_count == 0)
        return;
    SlabRay sr = make_slab_ray(ray);
    int32_t stack[128];
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        const HiprBvhNode& n = scene.nodes[cur];
        if (counters) counters->nodes++;
        float t0, t1;
        bool h0 = slab(sr, n.c0xy[0], n.c0xy[1], n.c0xy[2], n.c0xy[3], n.cz[0], n.cz[1], ray.tmin, tmax, t0);
        bool h1 = slab(sr, n.c1xy[0], n.c1xy[1], n.c1xy[2], n.c1xy[3], n.cz[2], n.cz[3], ray.tmin, tmax, t1);
        int32_t c0 = n.child[0], c1 = n.child[1];
        if (h0 && h1 && t1 < t0) { int32_t tmp = c0; c0 = c1; c1 = tmp; }
        if (!h0 && h1) { c0 = c1; h0 = true; h1 = false; }
        int32_t next = INT32_MIN;
        bool stop = false;
        if (h0) {
            if (c0 < 0) stop = leaf(uint32_t(~c0));
            else next = c0;
        }
        if (h1 && !stop) {
            if (c1 < 0) stop = leaf(uint32_t(~c1));
            else if (next == INT32_MIN) next = c1;
            else stack[sp++] = c1;
        }
        if (stop)
            return;
        if (next == INT32_MIN) {
            if (sp == 0)
                return;
            next = stack[--sp];
        }
        cur = next;
    }
}

// Traversal of the compressed 4-wide BVH (HiprWideNode, include/hiprenderer_c.h). Specification shared with
// k_trace_persistent (csrc/kernels.h), so that results AND node / triangle counters agree exactly:
//   * child bounds are never reconstructed: per axis A = 2^(e - 127) * inv_d and B = fma(origin, inv_d, -ood); the slab
//     distances of a bound q (0..255) are fma(float(q), A, B); tnear / tfar as in slab() above (same 3 ulp slack);
//   * the hit children are visited in ascending order of key = (bits(tnear) & 0x7FFFFFFC) | slot (distinct keys: a total
//     order); the first becomes the current item, the others go on the stack, farthest first;
//   * a leaf is an item like a node: its triangles are tested in storage order when it is taken.
// Most entries any traverse_wide call has had on its stack since the last reset: lets a test prove that its rays drive the
// traversal past the 32 entries the device keeps in LDS (the scratch-backed OVERFLOW kernels).
static std::atomic<int> g_wide_stack_high_water{0};
int wide_stack_high_water(bool reset) { const int v = g_wide_stack_high_water.load(); if (reset) g_wide_stack_high_water.store(0); return v; }

template <typename LeafFn>
static inline void traverse_wide(const HiprSceneDesc& scene, const Ray& ray, float& tmax, TraversalCounters* counters, LeafFn&& leaf) {
    if (scene.wide_node_count == 0)
        return;
    const SlabRay sr = make_slab_ray(ray);
    const float inv[3] = {sr.inv_d.x, sr.inv_d.y, sr.inv_d.z}, ood[3] = {sr.ood.x, sr.ood.y, sr.ood.z};
    int32_t stack[256];
    int sp = 0;
    int32_t item = 0;
    for (;;) {
        bool descended = false;
        if (item < 0) {
            if (leaf(uint32_t(~item)))
                return;
        } else {
            const HiprWideNode& n = scene.wide_nodes[item];
            if (counters) counters->nodes++;
            float A[3], B[3];
            for (int a = 0; a < 3; ++a) {
                A[a] = int_as_float(int32_t(((n.exponents >> (8 * a)) & 0xFFu) << 23)) * inv[a];
                B[a] = fmaf(n.origin[a], inv[a], -ood[a]);
            }
            uint32_t key[4];
            int32_t child[4];
            for (int k = 0; k < 4; ++k) {
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = fmaf(float((n.qlo[a] >> (8 * k)) & 0xFFu), A[a], B[a]);
                    hi[a] = fmaf(float((n.qhi[a] >> (8 * k)) & 0xFFu), A[a], B[a]);
                }
                const float tnear = fmaxf(fmaxf(fminf(lo[0], hi[0]), fminf(lo[1], hi[1])), fmaxf(fminf(lo[2], hi[2]), ray.tmin));
                float tfar = fminf(fminf(fmaxf(lo[0], hi[0]), fmaxf(lo[1], hi[1])), fmaxf(lo[2], hi[2]));
                tfar = fminf(tfar, tmax) * 1.0000004f;
                const bool hit = n.child[k] != HIPR_WIDE_EMPTY && tnear <= tfar;
                key[k] = hit ? ((uint32_t(float_as_int(tnear)) & 0x7FFFFFFCu) | uint32_t(k)) : 0xFFFFFFFFu;
                child[k] = n.child[k];
            }
            auto order = [&](int i, int j) { if (key[j] < key[i]) { std::swap(key[i], key[j]); std::swap(child[i], child[j]); } };
            order(0, 1); order(2, 3); order(0, 2); order(1, 3); order(1, 2);
            int hits = 0;
            while (hits < 4 && key[hits] != 0xFFFFFFFFu) ++hits;
            for (int k = hits - 1; k >= 1; --k) stack[sp++] = child[k];
            if (sp > g_wide_stack_high_water.load(std::memory_order_relaxed)) g_wide_stack_high_water.store(sp, std::memory_order_relaxed);   // diagnostic, racy max is fine
            if (hits > 0) { item = child[0]; descended = true; }
        }
        if (!descended) {
            if (sp == 0)
                return;
            item = stack[--sp];
        }
    }
}

Hit closest_hit_wide(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip, TraversalCounters* counters) {
    Hit best = {ray.tmax, 0, 0, HIT_MISS};
    traverse_wide(scene, ray, best.t, counters, [&](uint32_t leaf) {
        uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + count; ++i) {
            if (counters) counters->triangles++;
            consider_triangle(scene, i, ray, skip, best);
        }
        return false;
    });
    return best;
}

// ---------------------------------------------------------------------------------------------
// Traversal of the 8-wide tree (include/hiprenderer_c.h "wide8"; csrc/wide8_kernels.h is the device side of the same specification, so that hits AND
// node / triangle counters agree exactly):
//   * a ray carries its octant, bit a = (1 / direction[a] < 0); a GROUP is (base | valid << 24, pending | inner_mask << 8): the hit children of one
//     node that are still to be visited, `pending` bit p standing for the child in position p ^ octant. Children are taken from the current group lowest
//     pending bit first -- the nearest octant first -- whether they are inner nodes or leaf records;
//   * visiting an inner node tests its up to eight quantised child boxes (slab distances fma(float(q), A, B) with A = 2^(e - 127) * inv_d and
//     B = fma(origin, inv_d, -ood), origin = fma(float(m), grid_cell, grid_min); tnear / tfar as for the other trees; a child is hit unless
//     fma(tfar, 1.0000004f, -tnear) is negative -- the other trees' 3 ulp of slack inside the one rounding of the fma, round 4); if any is hit the
//     current group goes on the stack (when it still has pending children) and the node's hits become the current group;
//   * visiting a leaf record tests triangle A = (a; e1, e2), then B = (a; e2, e3) if there is one, with the solve of intersect_triangle on the stored
//     edges; the weights (w, u, v) = (1 - u - v, u, v) of the record's corners are mapped to the scene triangle's (u, v) by the record's selectors.
// ---------------------------------------------------------------------------------------------
static inline bool intersect_edges(float3 v0, float3 e1, float3 e2, float3 o, float3 d, float& t, float& u, float& v, float* determinant = nullptr) {
    float3 p = cross_fma(d, e2);
    float det = dot_fma(e1, p);
    if (determinant) *determinant = det;
    float3 tv = o - v0;
    float un = dot_fma(tv, p);
    float3 q = cross_fma(tv, e1);
    float vn = dot_fma(d, q);
    float us = std::signbit(det) ? -un : un, vs = std::signbit(det) ? -vn : vn;
    if (!((det != 0.0f) && (us >= 0.0f) && (vs >= 0.0f) && (us + vs <= std::fabs(det))))
        return false;
    float inv = 1.0f / det;
    u = un * inv;
    v = vn * inv;
    t = dot_fma(e2, q) * inv;
    return true;
}
// Triangle `which` (0 = A, 1 = B) of a leaf record: hit distance and the scene triangle's barycentrics.
// `facing`: the solve's determinant signed by the SCENE triangle's winding (positive: the ray arrives at its front), HiprLeaf8::flags bit 4.
static inline bool intersect_record(const HiprLeaf8& r, int which, float3 o, float3 d, float& t, float& u, float& v, float* facing = nullptr) {
    const float3 a = {r.a[0], r.a[1], r.a[2]};
    const float3 first = which == 0 ? make_float3(r.e1[0], r.e1[1], r.e1[2]) : make_float3(r.e2[0], r.e2[1], r.e2[2]);
    const float3 second = which == 0 ? make_float3(r.e2[0], r.e2[1], r.e2[2]) : make_float3(r.e3[0], r.e3[1], r.e3[2]);
    float ru, rv, det;
    if (!intersect_edges(a, first, second, o, d, t, ru, rv, &det))
        return false;
    if (facing) *facing = (which == 1 && (r.flags & 16u)) ? -det : det;
    const float weights[3] = {1.0f - ru - rv, ru, rv};
    const uint32_t selectors = (r.flags >> (which == 0 ? 8 : 12)) & 15u;
    u = weights[selectors & 3u];
    v = weights[selectors >> 2];
    return true;
}

static std::atomic<int> g_wide8_stack_high_water{0};
int wide8_stack_high_water(bool reset) { const int v = g_wide8_stack_high_water.load(); if (reset) g_wide8_stack_high_water.store(0); return v; }

template <typename LeafFn>
static inline void traverse_wide8(const HiprSceneDesc& scene, const Ray& ray, float& tmax, TraversalCounters* counters, LeafFn&& leaf) {
    if (scene.wide8_slot_count == 0)
        return;
    const SlabRay sr = make_slab_ray(ray);
    const float inv[3] = {sr.inv_d.x, sr.inv_d.y, sr.inv_d.z}, ood[3] = {sr.ood.x, sr.ood.y, sr.ood.z};
    const uint32_t octant = (inv[0] < 0.0f ? 1u : 0u) | (inv[1] < 0.0f ? 2u : 0u) | (inv[2] < 0.0f ? 4u : 0u);
    uint32_t stack_x[64], stack_y[64];
    int sp = 0;
    // the root as a group of its own: one inner child in position 0 of base 0
    uint32_t gx = 0u | 1u << 24, gy = (1u << (0u ^ octant)) | 1u << 8;
    for (;;) {
        if ((gy & 0xFFu) == 0u) {
            if (sp == 0)
                return;
            --sp;
            gx = stack_x[sp]; gy = stack_y[sp];
        }
        const uint32_t p = uint32_t(__builtin_ctz(gy & 0xFFu));
        gy &= gy - 1u;
        const uint32_t position = p ^ octant;
        const bool inner = (gy >> (8u + position)) & 1u;
        const uint32_t slot = (gx & 0xFFFFFFu) + uint32_t(__builtin_popcount((gx >> 24) & ((1u << position) - 1u)));
        if (!inner) {
            if (leaf(scene.wide8_slots[slot].leaf))
                return;
            continue;
        }
        const HiprNode8& n = scene.wide8_slots[slot].node;
        if (counters) counters->nodes++;
        const uint64_t packed = uint64_t(n.origin[0]) | uint64_t(n.origin[1]) << 32;
        float A[3], B[3];
        for (int a = 0; a < 3; ++a) {
            const float origin = fmaf(float(uint32_t(packed >> (21 * a)) & 0x1FFFFFu), scene.wide8_grid_cell[a], scene.wide8_grid_min[a]);
            A[a] = int_as_float(int32_t(uint32_t(n.exponent[a]) << 23)) * inv[a];
            B[a] = fmaf(origin, inv[a], -ood[a]);
        }
        const uint32_t valid = n.base_valid >> 24;
        uint32_t hits = 0;
        for (uint32_t k = 0; k < 8; ++k) {
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                lo[a] = fmaf(float(n.qlo[a][k]), A[a], B[a]);
                hi[a] = fmaf(float(n.qhi[a][k]), A[a], B[a]);
            }
            const float tnear = fmaxf(fmaxf(fminf(lo[0], hi[0]), fminf(lo[1], hi[1])), fmaxf(fminf(lo[2], hi[2]), ray.tmin));
            const float tfar = fminf(fminf(fminf(fmaxf(lo[0], hi[0]), fmaxf(lo[1], hi[1])), fmaxf(lo[2], hi[2])), tmax);
            // missed when tfar * (1 + 3 ulp) - tnear, rounded ONCE, is negative (its sign bit is what the device collects: csrc/wide8_kernels.h)
            if ((valid >> k & 1u) && !std::signbit(fmaf(tfar, 1.0000004f, -tnear)))
                hits |= 1u << (k ^ octant);
        }
        if (hits) {
            if (gy & 0xFFu) {
                stack_x[sp] = gx; stack_y[sp] = gy;
                ++sp;
                if (sp > g_wide8_stack_high_water.load(std::memory_order_relaxed)) g_wide8_stack_high_water.store(sp, std::memory_order_relaxed);
            }
            gx = n.base_valid;
            gy = hits | uint32_t(n.inner_mask) << 8;
        }
    }
}

// hipr_set_backface_culling (include/hiprenderer_c.h), the oracle's side: on by default like the device's. A closest hit on a record triangle flagged
// one-sided whose facing is below -facing_margin is stepped over -- what the hit program would refuse (MonteCarlo.cu:147-164) and retrace past.
static std::atomic<bool> g_cull_backfaces{true};
void set_backface_culling(bool enable) { g_cull_backfaces.store(enable); }
bool backface_culling() { return g_cull_backfaces.load(); }
Hit closest_hit_wide8(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip, TraversalCounters* counters) {
    Hit best = {ray.tmax, 0, 0, HIT_MISS};
    const bool cull = backface_culling();
    traverse_wide8(scene, ray, best.t, counters, [&](const HiprLeaf8& record) {
        for (int which = 0; which < 2; ++which) {
            const uint32_t id = record.triangle[which];
            if (id == HIPR_LEAF8_NONE)
                break;
            if (counters) counters->triangles++;
            float t, u, v, facing;
            if (id == skip || !intersect_record(record, which, ray.origin, ray.direction, t, u, v, &facing) || !(t > ray.tmin))
                continue;
            if (cull && (record.flags >> (2 + which) & 1u) && facing < -record.facing_margin)
                continue;
            if (t < best.t || (t == best.t && id < best.id))
                best = {t, u, v, id};
        }
        return false;
    });
    return best;
}

Hit closest_hit_bvh(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip, TraversalCounters* counters) {
    Hit best = {ray.tmax, 0, 0, HIT_MISS};
    traverse(scene, ray, best.t, counters, [&](uint32_t leaf) {
        uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + count; ++i) {
            if (counters) counters->triangles++;
            consider_triangle(scene, i, ray, skip, best);
        }
        return false;
    });
    return best;
}

// LightSources.cu:31-70: sphere and disk lights take part in closest-hit selection.
void intersect_lights(const HiprSceneDesc& scene, const Ray& ray, Hit& hit) {
    for (uint32_t li = 0; li < scene.light_count; ++li) {
        const HiprLight& light = scene.lights[li];
        uint32_t type = light.flags & HIPR_LIGHT_TYPE_MASK;
        float t = -1e30f;
        if (type == HIPR_LIGHT_SPHERE) {
            SphereLight s = as_sphere(light);
            if (!(s.radius > 0.0f)) continue;   // bounds program invalidates the AABB (LightSources.cu:83-90)
            t = ray_sphere(ray.origin, ray.direction, s.position, s.radius);
        } else if (type == HIPR_LIGHT_SPOT) {
            SpotLight s = as_spot(light);
            if (!(s.radius > 0.0f)) continue;
            t = ray_disk(ray.origin, ray.direction, s.position, s.direction, s.radius);
        } else
            continue;
        if (t > ray.tmin && t < hit.t)
            hit = {t, 0, 0, HIT_LIGHT_BIT | li};
    }
}

// ---------------------------------------------------------------------------------------------
// Textures: software replacement of the samplers configured at OR/Renderer.cpp:703-751.
// ---------------------------------------------------------------------------------------------
static inline float srgb_to_linear(float c) {
    return c <= 0.04045f ? c / 12.92f : exact_powf((c + 0.055f) / 1.055f, 2.4f);
}

static inline float4 fetch_texel(const HiprSceneDesc& scene, const HiprTexture& tex, int x, int y) {
    const uint8_t* base = scene.texels + tex.texel_offset;
    size_t i = size_t(y) * tex.width + size_t(x);
    float4 r;
    switch (tex.format) {
    case HIPR_TEXEL_R8: { float v = base[i] / 255.0f; r = {v, 0, 0, 1}; break; }
    case HIPR_TEXEL_RGBA8: r = {base[4 * i] / 255.0f, base[4 * i + 1] / 255.0f, base[4 * i + 2] / 255.0f, base[4 * i + 3] / 255.0f}; break;
    case HIPR_TEXEL_R32F: { float v; std::memcpy(&v, base + 4 * i, 4); r = {v, 0, 0, 1}; break; }
    default: std::memcpy(&r, base + 16 * i, 16); break;
    }
    if (tex.is_sRGB) {
        r.x = srgb_to_linear(r.x);
        if (tex.format == HIPR_TEXEL_RGBA8 || tex.format == HIPR_TEXEL_RGBA32F) { r.y = srgb_to_linear(r.y); r.z = srgb_to_linear(r.z); }
    }
    return r;
}

static inline int wrap_coord(int i, int n, int repeat) {
    if (repeat) { i %= n; return i < 0 ? i + n : i; }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

float4 sample_texture(const HiprSceneDesc& scene, int texture_ID, float2 uv) {
    const HiprTexture& tex = scene.textures[texture_ID];
    int w = int(tex.width), h = int(tex.height);
    if (tex.filter & 1) {   // linear (magnification filter at LOD 0)
        float xb = uv.x * w - 0.5f, yb = uv.y * h - 0.5f;
        float xf = floorf(xb), yf = floorf(yb);
        float fx = xb - xf, fy = yb - yf;
        int x0 = wrap_coord(int(xf), w, tex.wrap_u), x1 = wrap_coord(int(xf) + 1, w, tex.wrap_u);
        int y0 = wrap_coord(int(yf), h, tex.wrap_v), y1 = wrap_coord(int(yf) + 1, h, tex.wrap_v);
        float4 a = fetch_texel(scene, tex, x0, y0), b = fetch_texel(scene, tex, x1, y0);
        float4 c = fetch_texel(scene, tex, x0, y1), d = fetch_texel(scene, tex, x1, y1);
        float4 lo = a + (b - a) * fx, hi = c + (d - c) * fx;
        return lo + (hi - lo) * fy;
    }
    int x = wrap_coord(int(floorf(uv.x * w)), w, tex.wrap_u);
    int y = wrap_coord(int(floorf(uv.y * h)), h, tex.wrap_v);
    return fetch_texel(scene, tex, x, y);
}

static inline float4 material_tint_roughness(const HiprSceneDesc& scene, const HiprMaterial& m, float2 uv) {
    float4 tr = {m.tint[0], m.tint[1], m.tint[2], m.roughness};
    if (m.tint_roughness_texture_ID)
        tr = tr * sample_texture(scene, m.tint_roughness_texture_ID, uv);
    if (m.roughness_texture_ID)
        tr.w *= sample_texture(scene, m.roughness_texture_ID, uv).x;
    return tr;
}
static inline float material_metallic(const HiprSceneDesc& scene, const HiprMaterial& m, float2 uv) {
    return m.metallic_texture_ID ? m.metallic * sample_texture(scene, m.metallic_texture_ID, uv).x : m.metallic;
}
float material_coverage(const HiprSceneDesc& scene, const HiprMaterial& m, float2 uv) {
    float tex = 1.0f;
    if (m.coverage_texture_ID)
        tex = sample_texture(scene, m.coverage_texture_ID, uv).x;
    if (m.flags & HIPR_MATERIAL_CUTOUT)
        return tex < m.coverage ? 0.0f : 1.0f;
    return m.coverage * tex;
}

// ---------------------------------------------------------------------------------------------
// Attribute interpolation (ORS/TriangleAttributes.cu:35-84) on the flattened scene.
// ---------------------------------------------------------------------------------------------
static inline float3 decode_octahedral(const int16_t e[2]) {
    float2 f = {float(e[0]), float(e[1])};
    float3 n = {f.x, f.y, 32767.0f - fabsf(f.x) - fabsf(f.y)};
    float t = fmaxf(-n.z, 0.0f);
    n.x += n.x >= 0 ? -t : t;
    n.y += n.y >= 0 ? -t : t;
    return normalize(n);
}

struct SurfaceAttributes {
    float3 position, geometric_normal, shading_normal;   // world space
    float2 texcoord;
    float4 tint_and_roughness_scale;
    float3 emission;
};

static inline float2 triangle_texcoord(const HiprSceneDesc& scene, const HiprTriangle& tri, float u, float v) {
    const HiprInstance& inst = scene.instances[tri.instance_index];
    if (!(inst.mesh_flags & HIPR_MESH_TEXCOORDS))
        return {0, 0};   // undefined in the reference (TriangleAttributes.cu:63-64), defined as 0 here
    const uint32_t* idx = scene.indices + 3 * size_t(inst.index_offset + tri.primitive_index);
    const float* tc = scene.texcoords + 2 * size_t(inst.vertex_offset);
    float w = 1.0f - u - v;
    float2 t0 = {tc[2 * idx[0]], tc[2 * idx[0] + 1]}, t1 = {tc[2 * idx[1]], tc[2 * idx[1] + 1]}, t2 = {tc[2 * idx[2]], tc[2 * idx[2] + 1]};
    return t1 * u + t2 * v + t0 * w;
}

static SurfaceAttributes interpolate_attributes(const HiprSceneDesc& scene, const HiprTriangle& tri, float u, float v) {
    const HiprInstance& inst = scene.instances[tri.instance_index];
    const uint32_t* idx = scene.indices + 3 * size_t(inst.index_offset + tri.primitive_index);
    float w = 1.0f - u - v;
    float3 p0 = {tri.v0[0], tri.v0[1], tri.v0[2]}, p1 = {tri.v1[0], tri.v1[1], tri.v1[2]}, p2 = {tri.v2[0], tri.v2[1], tri.v2[2]};

    SurfaceAttributes a;
    a.geometric_normal = normalize(cross(p1 - p0, p2 - p0));
    a.position = p1 * u + p2 * v + p0 * w;

    if (inst.mesh_flags & HIPR_MESH_NORMALS) {
        // The reference interpolates the decoded vertex normals in object space (TriangleAttributes.cu:58-61) and carries the result to world space with
        // rtTransformNormal(RT_OBJECT_TO_WORLD) + normalize (MonteCarlo.cu:176). Instance transforms are rotation + uniform scale + translation
        // (BF/Math/Transform.h:28-34), for which M (sum w_i n_i) = sum w_i (M n_i): the specification shared with the device is the
        // right-hand side -- every vertex normal taken to world space first (what the device's per-triangle shading record holds, k_build_shade_triangles),
        // interpolated there, normalised once. The two orders differ in the last bits only; stating one of them is what lets K3 be compared bit for bit.
        const HiprVertexGeometry* g = scene.geometry + inst.vertex_offset;
        const float* M = inst.object_to_world;
        auto world_normal = [&](uint32_t i) {
            const float3 o = decode_octahedral(g[i].oct_normal);
            return make_float3(M[0] * o.x + M[1] * o.y + M[2] * o.z, M[4] * o.x + M[5] * o.y + M[6] * o.z, M[8] * o.x + M[9] * o.y + M[10] * o.z);
        };
        a.shading_normal = normalize(world_normal(idx[1]) * u + world_normal(idx[2]) * v + world_normal(idx[0]) * w);
    } else
        a.shading_normal = a.geometric_normal;

    a.texcoord = triangle_texcoord(scene, tri, u, v);

    if (inst.mesh_flags & HIPR_MESH_TINTS) {
        const uint32_t* tints = scene.tints + inst.vertex_offset;
        auto ch = [&](uint32_t packed, int c) { return float((packed >> (8 * c)) & 0xFFu); };
        uint32_t t0 = tints[idx[0]], t1 = tints[idx[1]], t2 = tints[idx[2]];
        const float s = 1.0f / 255.0f;
        a.tint_and_roughness_scale = {(ch(t1, 0) * u + ch(t2, 0) * v + ch(t0, 0) * w) * s, (ch(t1, 1) * u + ch(t2, 1) * v + ch(t0, 1) * w) * s,
                                      (ch(t1, 2) * u + ch(t2, 2) * v + ch(t0, 2) * w) * s, (ch(t1, 3) * u + ch(t2, 3) * v + ch(t0, 3) * w) * s};
    } else
        a.tint_and_roughness_scale = {1, 1, 1, 1};

    if (inst.mesh_flags & HIPR_MESH_EMISSIVE) {
        const float* e = scene.emissions + 3 * size_t(inst.vertex_offset);
        auto em = [&](uint32_t i) { return make_float3(e[3 * i], e[3 * i + 1], e[3 * i + 2]); };
        a.emission = em(idx[1]) * u + em(idx[2]) * v + em(idx[0]) * w;
    } else
        a.emission = {1, 1, 1};
    return a;
}

// ---------------------------------------------------------------------------------------------
// Shadow rays: shadow_any_hit (ORS/MonteCarlo.cu:278-285) over every hit in (tmin, tmax).
// ---------------------------------------------------------------------------------------------
static inline bool shadow_triangle(const HiprSceneDesc& scene, uint32_t i, const Ray& ray, float3& radiance) {
    float t, u, v;
    const HiprTriangle& tri = scene.triangles[i];
    if (!intersect_triangle(tri, ray.origin, ray.direction, t, u, v))
        return false;
    if (!(t > ray.tmin && t < ray.tmax))
        return false;
    float coverage;
    if (tri.flags & HIPR_TRIANGLE_OPAQUE)
        coverage = 1.0f;
    else {
        const HiprInstance& inst = scene.instances[tri.instance_index];
        coverage = material_coverage(scene, scene.materials[inst.material_index], triangle_texcoord(scene, tri, u, v));
    }
    radiance *= 1.0f - coverage;
    if (radiance.x < 0.0000001f && radiance.y < 0.0000001f && radiance.z < 0.0000001f) {
        radiance = {0, 0, 0};
        return true;   // rtTerminateRay
    }
    return false;
}

float3 shadow_bruteforce(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters) {
    for (const SearchItem& item : search_items(scene)) {
        if (counters) counters->triangles++;
        ItemHit hit = intersect_item(item, ray.origin, ray.direction);
        if (!hit.inside || !(hit.t > ray.tmin && hit.t < ray.tmax))
            continue;
        const HiprTriangle& tri = scene.triangles[item.triangle[hit.half]];
        float coverage = 1.0f;
        if (!(tri.flags & HIPR_TRIANGLE_OPAQUE)) {
            const HiprInstance& inst = scene.instances[tri.instance_index];
            coverage = material_coverage(scene, scene.materials[inst.material_index], triangle_texcoord(scene, tri, hit.u, hit.v));
        }
        radiance *= 1.0f - coverage;
        if (radiance.x < 0.0000001f && radiance.y < 0.0000001f && radiance.z < 0.0000001f)
            return {0, 0, 0};   // rtTerminateRay
    }
    return radiance;
}

float3 shadow_wide(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters) {
    float tmax = ray.tmax;
    traverse_wide(scene, ray, tmax, counters, [&](uint32_t leaf) {
        uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + count; ++i) {
            if (counters) counters->triangles++;
            if (shadow_triangle(scene, i, ray, radiance))
                return true;
        }
        return false;
    });
    return radiance;
}

float3 shadow_wide8(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters) {
    float tmax = ray.tmax;
    traverse_wide8(scene, ray, tmax, counters, [&](const HiprLeaf8& record) {
        for (int which = 0; which < 2; ++which) {
            const uint32_t id = record.triangle[which];
            if (id == HIPR_LEAF8_NONE)
                break;
            if (counters) counters->triangles++;
            float t, u, v;
            if (!intersect_record(record, which, ray.origin, ray.direction, t, u, v) || !(t > ray.tmin && t < ray.tmax))
                continue;
            float coverage = 1.0f;
            if (!(record.flags >> which & 1u)) {
                const HiprTriangle& tri = scene.triangles[id];
                const HiprInstance& inst = scene.instances[tri.instance_index];
                coverage = material_coverage(scene, scene.materials[inst.material_index], triangle_texcoord(scene, tri, u, v));
            }
            radiance *= 1.0f - coverage;
            if (radiance.x < 0.0000001f && radiance.y < 0.0000001f && radiance.z < 0.0000001f) {
                radiance = {0, 0, 0};
                return true;   // rtTerminateRay
            }
        }
        return false;
    });
    return radiance;
}

float3 shadow_bvh(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters) {
    float tmax = ray.tmax;
    traverse(scene, ray, tmax, counters, [&](uint32_t leaf) {
        uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + count; ++i) {
            if (counters) counters->triangles++;
            if (shadow_triangle(scene, i, ray, radiance))
                return true;
        }
        return false;
    });
    return radiance;
}

// ---------------------------------------------------------------------------------------------
// The path tracer
// ---------------------------------------------------------------------------------------------
static inline float3 fix_backfacing_shading_normal(float3 w, float3 n, float target_cos_theta) {
    float cos_theta = dot(w, n);
    if (cos_theta < target_cos_theta)
        return normalize(n - (cos_theta - target_cos_theta) * w);
    return n;
}

static inline float3 offset_ray_origin(float3 p, float3 n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int ox = int(int_scale * n.x), oy = int(int_scale * n.y), oz = int(int_scale * n.z);
    float3 p_i = {int_as_float(float_as_int(p.x) + (p.x < 0 ? -ox : ox)),
                  int_as_float(float_as_int(p.y) + (p.y < 0 ? -oy : oy)),
                  int_as_float(float_as_int(p.z) + (p.z < 0 ? -oz : oz))};
    return {fabsf(p.x) < origin ? p.x + float_scale * n.x : p_i.x,
            fabsf(p.y) < origin ? p.y + float_scale * n.y : p_i.y,
            fabsf(p.z) < origin ? p.z + float_scale * n.z : p_i.z};
}
static inline float3 offset_ray_origin(float3 p, float3 direction, float3 geometric_normal) {
    float cos_theta = dot(geometric_normal, direction);
    return offset_ray_origin(p, cos_theta >= 0 ? geometric_normal : -geometric_normal);
}

static inline float4 toroidal_shift(float4 base, float4 shift) {
    float4 s = base + shift;
    return {s.x - floorf(s.x), s.y - floorf(s.y), s.z - floorf(s.z), s.w - floorf(s.w)};
}

struct Payload {
    float3 radiance = {0, 0, 0};
    uint32_t last_triangle = HIT_MISS;   // PrimitiveID of the last accepted hit, as a global triangle index
    float3 throughput = {1, 1, 1};
    uint32_t bounces = 0;
    float3 position, direction;
    float ray_min_t = 0.0f;
    PDF bsdf_PDF = PDF::delta_dirac(1);
    LightSample light_sample = LightSample::none();
    float3 light_sample_origin = {0, 0, 0};
    uint32_t pixel_hash = 0, accumulation = 0;
    // recorded at the accepted hit for the AOV entry points (MonteCarlo.cu:166-179)
    int material_index = 0;
    float2 texcoord = {0, 0};
    float4 tint_and_roughness_scale = {1, 1, 1, 1};   // after the float_to_unorm8 round trip
    float3 shading_normal = {0, 0, 0};
    int instance_id = 0, primitive_index = 0;
    float4 sample4f(uint32_t d) const { return rng::sample4f(accumulation, pixel_hash, 8u * bounces + d); }
};

// ---------------------------------------------------------------------------------------------
// Presampled environment light (ORS/LightSources/PresampledEnvironmentLightImpl.h:18-41, OR/Utils.h:288-292). The samples and
// the per-texel PDF image arrive in HiprSceneDesc::environment, built by the host as OR/PresampledEnvironmentMap.cpp:19-101 does.
// ---------------------------------------------------------------------------------------------
static inline float2 direction_to_latlong_texcoord(float3 direction) {
    const float PI = 3.14159265358979323846f;
    float u = (exact_atan2f(direction.z, direction.x) + PI) * 0.5f / PI;
    float v = (exact_asinf(direction.y) + PI * 0.5f) / PI;
    return {u, v};
}

static PDF environment_pdf(const HiprEnvironment& env, float3 direction) {
    float2 uv = direction_to_latlong_texcoord(direction);
    float sin_theta = sqrtf(1.0f - direction.y * direction.y);
    int w = int(env.pdf_width), h = int(env.pdf_height);
    int x = int(floorf(uv.x * w)), y = int(floorf(uv.y * h));   // nearest filtering, clamp to edge
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    float pdf = env.per_pixel_PDF[x + y * w] / sin_theta;
    return sin_theta == 0.0f ? PDF::delta_dirac(0) : PDF(pdf);
}

static float3 environment_evaluate(const HiprSceneDesc& scene, const HiprSceneState& state, float3 direction) {
    float4 texel = sample_texture(scene, scene.environment->environment_map_ID, direction_to_latlong_texcoord(direction));
    return make_float3(state.environment_tint[0], state.environment_tint[1], state.environment_tint[2]) * make_float3(texel.x, texel.y, texel.z);
}

static LightSample environment_sample(const HiprEnvironment& env, const HiprSceneState& state, float random_x) {
    int index = int(random_x * float(env.sample_count));
    if (index > int(env.sample_count) - 1) index = int(env.sample_count) - 1;
    const HiprLightSample& s = env.samples[index];
    LightSample ls;
    ls.radiance = make_float3(s.radiance[0], s.radiance[1], s.radiance[2]) * make_float3(state.environment_tint[0], state.environment_tint[1], state.environment_tint[2]);
    ls.pdf = PDF(s.PDF);
    ls.direction_to_light = {s.direction_to_light[0], s.direction_to_light[1], s.direction_to_light[2]};
    ls.distance = s.distance;
    return ls;
}

template <typename Model>
static LightSample sample_single_light(const HiprSceneDesc& scene, const HiprSceneState& state, const Model& material, float3 p, float3 wo, const TBN& tbn, float3 u) {
    int light_count = int(scene.light_count);
    int li = int(u.z * light_count);
    if (li > light_count - 1) li = light_count - 1;
    LightSample ls = (scene.lights[li].flags & HIPR_LIGHT_TYPE_MASK) == HIPR_LIGHT_PRESAMPLED_ENVIRONMENT ? environment_sample(*scene.environment, state, u.x)
                                                                                                     : Lights::sample_radiance(scene.lights[li], p, make_float2(u.x, u.y));
    ls.radiance *= float(light_count);
    float N_dot_L = dot(tbn.normal, ls.direction_to_light);
    ls.radiance *= fabsf(N_dot_L) / ls.pdf.value();
    BSDFResponse f = material.evaluate_with_PDF(wo, tbn.to_local(ls.direction_to_light));
    if (!ls.pdf.is_delta_dirac())
        ls.radiance *= MIS_weight(ls.pdf, f.pdf);
    else
        f.reflectance = fminf3(f.reflectance, make_float3(32.0f));
    ls.radiance *= f.reflectance;
    return ls;
}

template <typename Model>
static LightSample reestimated_light_samples(const HiprSceneDesc& scene, const HiprSceneState& state, const float4* offsets,
                                             const Payload& payload, const Model& material, float3 p, float3 wo, const TBN& tbn) {
    if (scene.light_count == 0)
        return LightSample::none();
    float4 base = payload.sample4f(1);   // NEXT_EVENT_ESTIMATION
    LightSample kept = LightSample::none();
    int n = state.next_event_sample_count;
    for (int s = 0; s < n; ++s) {
        float4 r = toroidal_shift(base, offsets[s]);
        LightSample candidate = sample_single_light(scene, state, material, p, wo, tbn, make_float3(r.x, r.y, r.z));
        float w_old = sum(kept.radiance), w_new = sum(candidate.radiance);
        float p_new = w_new / (w_old + w_new);
        if (r.w < p_new) {
            kept = candidate;
            kept.radiance /= p_new;
        } else
            kept.radiance /= 1.0f - p_new;
    }
    kept.radiance /= float(n);
    return kept;
}

template <typename Model>
static void finish_closest_hit(const HiprSceneDesc& scene, const HiprSceneState& state, const float4* offsets, Payload& payload,
                               const Model& material, const HiprMaterial& mp, const SurfaceAttributes& a, float3 ray_direction,
                               float3 geometric_normal, const TBN& tbn, float3 wo, float3 bsdf_u) {
    payload.radiance += payload.throughput * a.emission * make_float3(mp.emission[0], mp.emission[1], mp.emission[2]);

    payload.light_sample = reestimated_light_samples(scene, state, offsets, payload, material, a.position, wo, tbn);
    payload.light_sample_origin = offset_ray_origin(a.position, payload.light_sample.direction_to_light, geometric_normal);
    payload.light_sample.radiance *= payload.throughput;

    BSDFSample s = material.sample(wo, bsdf_u);
    bool is_reflection = s.direction.z >= 0;
    payload.direction = tbn.to_world(s.direction);
    payload.bsdf_PDF = s.pdf;
    if (s.pdf.is_valid())
        payload.throughput *= (s.reflectance * fabsf(s.direction.z)) / s.pdf.value();   // f * |cos| / pdf, optix float3 / float
    else
        payload.throughput = {0, 0, 0};

    float cos_geometric = dot(payload.direction, geometric_normal);
    if (is_reflection ? cos_geometric < 0.0f : cos_geometric >= 0.0f)
        payload.direction = reflect(payload.direction, geometric_normal);

    payload.position = offset_ray_origin(a.position, payload.direction, geometric_normal);
    payload.ray_min_t = 0.0f;
    payload.bounces += 1u;
    if (!payload.light_sample.pdf.is_valid())
        payload.bsdf_PDF.disable_MIS();
    (void)ray_direction;
}

// path_tracing_closest_hit<> (ORS/MonteCarlo.cu:129-233). Returns true when the hit was accepted.
static bool closest_hit_program(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam, const float4* offsets,
                                Payload& payload, const Hit& hit, float3 ray_direction) {
    payload.light_sample = LightSample::none();
    const HiprTriangle& tri = scene.triangles[hit.id];
    const HiprInstance& inst = scene.instances[tri.instance_index];
    const HiprMaterial& mp = scene.materials[inst.material_index];
    SurfaceAttributes a = interpolate_attributes(scene, tri, hit.u, hit.v);

    bool thin_walled = (mp.flags & (HIPR_MATERIAL_CUTOUT | HIPR_MATERIAL_THIN_WALLED)) != 0;
    bool transmissive = mp.shading_model == HIPR_SHADING_TRANSMISSIVE;
    float3 geometric_normal = a.geometric_normal;
    bool hit_from_front = dot(geometric_normal, ray_direction) < 0.0f;
    bool backside_cull = !hit_from_front && !thin_walled && !transmissive;

    float4 bsdf_coverage_u = payload.sample4f(2);   // BSDF dimension, always drawn
    float coverage = material_coverage(scene, mp, a.texcoord);
    bool discard_from_coverage = coverage < bsdf_coverage_u.w;
    if (backside_cull || discard_from_coverage) {
        payload.ray_min_t = nextafterf(hit.t, INFINITY);
        return false;
    }

    payload.last_triangle = hit.id;
    payload.material_index = inst.material_index;
    payload.texcoord = a.texcoord;
    {
        auto q8 = [](float v) { return float((unsigned char)(saturate(v) * 255.0f + 0.5f)) * (1.0f / 255.0f); };
        const float4 t = a.tint_and_roughness_scale;
        payload.tint_and_roughness_scale = {q8(t.x), q8(t.y), q8(t.z), q8(t.w)};
    }
    payload.instance_id = inst.instance_id;
    payload.primitive_index = int(tri.primitive_index);
    // float_to_unorm8 round trip of the vertex tint scale (MonteCarlo.cu:170 stores it, AOVs read it);
    // the shading model itself receives the unquantised attribute (MonteCarlo.cu:242,252,264).
    geometric_normal = hit_from_front ? geometric_normal : -geometric_normal;
    float3 shading_normal = hit_from_front ? a.shading_normal : -a.shading_normal;
    shading_normal = fix_backfacing_shading_normal(-ray_direction, shading_normal, 0.002f);
    payload.shading_normal = shading_normal;
    const TBN tbn(shading_normal);
    float3 wo = tbn.to_local(-ray_direction);
    float cos_theta = (hit_from_front || thin_walled) ? wo.z : -wo.z;

    float4 tr = material_tint_roughness(scene, mp, a.texcoord) * a.tint_and_roughness_scale;
    MaterialInputs in;
    in.tint = make_float3(tr);
    in.roughness = tr.w;
    in.specularity = mp.specularity;
    in.metallic = material_metallic(scene, mp, a.texcoord);
    in.coat = unorm16(mp.coat);
    in.coat_roughness = unorm16(mp.coat_roughness);
    float3 bsdf_u = {bsdf_coverage_u.x, bsdf_coverage_u.y, bsdf_coverage_u.z};
    // OR/PublicTypes.h:44 PDF_scale_at_accumulation, evaluated for the accumulation of this path
    PDF max_PDF_hint = payload.bsdf_PDF * (cam.path_regularization_PDF_scale * (1.0f + cam.path_regularization_scale_decay * float(int(payload.accumulation))));

    if (mp.shading_model == HIPR_SHADING_DIFFUSE) {
        DiffuseShading m = {in.tint, in.roughness};
        finish_closest_hit(scene, state, offsets, payload, m, mp, a, ray_direction, geometric_normal, tbn, wo, bsdf_u);
    } else if (transmissive) {
        TransmissiveShading m = TransmissiveShading::with_max_PDF_hint(in, cos_theta, max_PDF_hint);
        finish_closest_hit(scene, state, offsets, payload, m, mp, a, ray_direction, geometric_normal, tbn, wo, bsdf_u);
    } else {
        DefaultShading m = DefaultShading::with_max_PDF_hint(in, cos_theta, max_PDF_hint);
        finish_closest_hit(scene, state, offsets, payload, m, mp, a, ray_direction, geometric_normal, tbn, wo, bsdf_u);
    }
    return true;
}

// What runs for a traced ray once its closest hit is known: the miss program, light_closest_hit or path_tracing_closest_hit<> -- path_trace_pixel's loop body
// between the closest-hit query and the shadow query. Its own function so that shade_hit_for_test below can run exactly this for a given ray and hit.
static void hit_programs(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam, const float4* offsets, Payload& payload, const Ray& ray,
                         const Hit& hit, RenderCounters* counters) {
    if (hit.id == HIT_MISS) {
        // miss program (SimpleRGPs.cu:349-362) + evaluate_intersection (LightImpl.h:86-97) for an environment map
        float3 env = {state.environment_tint[0], state.environment_tint[1], state.environment_tint[2]};
        if (scene.environment && scene.environment->environment_map_ID) {
            env = environment_evaluate(scene, state, ray.direction);
            if (payload.bsdf_PDF.use_for_MIS()) env *= MIS_weight(payload.bsdf_PDF, environment_pdf(*scene.environment, ray.direction));
        }
        payload.radiance += payload.throughput * env;
        payload.throughput = {0, 0, 0};
    } else if (hit.id & HIT_LIGHT_BIT) {
        // light_closest_hit (MonteCarlo.cu:291-302)
        const HiprLight& light = scene.lights[hit.id & ~HIT_LIGHT_BIT];
        float3 L = Lights::evaluate_intersection(light, ray.origin, ray.direction, payload.bsdf_PDF);
        payload.throughput = fminf3(payload.throughput, make_float3(4));
        payload.radiance += payload.throughput * L;
        payload.throughput = {0, 0, 0};
    } else {
        bool accepted = closest_hit_program(scene, state, cam, offsets, payload, hit, ray.direction);
        if (accepted && counters) counters->shaded_hits++;
        if (!accepted && counters) counters->rejected_hits++;
    }
}

// Stage-level checker of K3 (tests/test_device_code_on_host_cpu.py, tests/test_gpu_verify_build.py): one trip through the hit programs for a ray whose closest hit is
// given, with everything the device's shade stage reports for it -- radiance added, the continuing ray and path state, the shadow ray. Same record as
// tests/native/DeviceShadeHost.hip writes for the device code (32 words per entry, the words of parts that do not apply left zero).
void shade_hit_for_test(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam, const float4* offsets, const float* ray8, const float* thr_bounces4,
                        const float* hit4, uint32_t last_triangle, uint32_t pixel_hash, uint32_t accumulation, float* out32) {
    auto bits = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    auto word = [](float f) { uint32_t v; std::memcpy(&v, &f, 4); return v; };
    Payload payload;
    payload.position = {ray8[0], ray8[1], ray8[2]}; payload.ray_min_t = ray8[3];
    payload.direction = {ray8[4], ray8[5], ray8[6]}; payload.bsdf_PDF = PDF(ray8[7]);
    payload.throughput = {thr_bounces4[0], thr_bounces4[1], thr_bounces4[2]}; payload.bounces = word(thr_bounces4[3]);
    payload.last_triangle = last_triangle; payload.pixel_hash = pixel_hash; payload.accumulation = accumulation;
    const Ray ray = {payload.position, payload.ray_min_t, payload.direction, INFINITY};
    const Hit hit = {hit4[0], hit4[1], hit4[2], word(hit4[3])};
    const uint32_t bounces_before = payload.bounces;
    hit_programs(scene, state, cam, offsets, payload, ray, hit, nullptr);
    for (int k = 0; k < 32; ++k) out32[k] = 0.0f;
    const bool surface = hit.id != HIT_MISS && !(hit.id & HIT_LIGHT_BIT);
    const bool rejected = surface && payload.bounces == bounces_before;      // a refused hit sends the same ray on (ray_min_t bumped), nothing else changes
    const bool continues = surface && (rejected || (payload.bounces <= cam.max_bounce_count && !is_black(payload.throughput)));
    const LightSample& ls = payload.light_sample;
    const bool shadow = surface && !rejected && (ls.radiance.x > 0 || ls.radiance.y > 0 || ls.radiance.z > 0);
    out32[0] = bits((continues ? 1u : 0u) | (shadow ? 2u : 0u) | (surface && !rejected ? 4u : 0u));
    out32[1] = payload.radiance.x; out32[2] = payload.radiance.y; out32[3] = payload.radiance.z;
    if (continues) {
        out32[4] = payload.position.x; out32[5] = payload.position.y; out32[6] = payload.position.z; out32[7] = payload.ray_min_t;
        out32[8] = payload.direction.x; out32[9] = payload.direction.y; out32[10] = payload.direction.z; out32[11] = payload.bsdf_PDF.v;
        out32[12] = payload.throughput.x; out32[13] = payload.throughput.y; out32[14] = payload.throughput.z; out32[15] = bits(payload.bounces);
        out32[16] = bits(payload.last_triangle);
    }
    if (shadow) {
        out32[17] = payload.light_sample_origin.x; out32[18] = payload.light_sample_origin.y; out32[19] = payload.light_sample_origin.z; out32[20] = ls.distance;
        out32[21] = ls.direction_to_light.x; out32[22] = ls.direction_to_light.y; out32[23] = ls.direction_to_light.z;
        out32[24] = ls.radiance.x; out32[25] = ls.radiance.y; out32[26] = ls.radiance.z;
    }
}

float3 path_trace_pixel(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam,
                        const float4* offsets, int x, int y, int width, int height, uint32_t accumulation,
                        const RenderSettings& settings, RenderCounters* counters) {
    Payload payload;
    payload.pixel_hash = rng::pcg2d(uint32_t(x), uint32_t(y)).x;
    payload.accumulation = accumulation;
    generate_camera_ray(cam, x, y, width, height, accumulation, payload.position, payload.direction);
    if (counters) counters->camera_rays++;

    do {
        Ray ray = {payload.position, payload.ray_min_t, payload.direction, INFINITY};
        Hit hit = settings.use_wide8 ? closest_hit_wide8(scene, ray, payload.last_triangle, counters ? &counters->closest : nullptr)
                  : settings.use_wide ? closest_hit_wide(scene, ray, payload.last_triangle, counters ? &counters->closest : nullptr)
                  : settings.use_bvh ? closest_hit_bvh(scene, ray, payload.last_triangle, counters ? &counters->closest : nullptr)
                                   : closest_hit_bruteforce(scene, ray, payload.last_triangle, counters ? &counters->closest : nullptr);
        intersect_lights(scene, ray, hit);
        if (counters) counters->closest_rays++;

        hit_programs(scene, state, cam, offsets, payload, ray, hit, counters);

        const LightSample& ls = payload.light_sample;
        if (ls.radiance.x > 0 || ls.radiance.y > 0 || ls.radiance.z > 0) {
            Ray shadow = {payload.light_sample_origin, 0.0f, ls.direction_to_light, ls.distance};
            float3 r = settings.use_wide8 ? shadow_wide8(scene, shadow, ls.radiance, counters ? &counters->shadow : nullptr)
                       : settings.use_wide ? shadow_wide(scene, shadow, ls.radiance, counters ? &counters->shadow : nullptr)
                       : settings.use_bvh ? shadow_bvh(scene, shadow, ls.radiance, counters ? &counters->shadow : nullptr)
                                        : shadow_bruteforce(scene, shadow, ls.radiance, counters ? &counters->shadow : nullptr);
            if (counters) counters->shadow_rays++;
            payload.radiance += r;
        }
        payload.light_sample = LightSample::none();
    } while (payload.bounces <= cam.max_bounce_count && !is_black(payload.throughput));

    return payload.radiance;
}

// ---------------------------------------------------------------------------------------------
// AOV entry points: depth_RPG and process_material_intersection (ORS/SimpleRGPs.cu:227-340).
// ---------------------------------------------------------------------------------------------
static inline uint32_t compact_by_2(uint32_t v) {
    v &= 0x09249249u;
    v = (v ^ (v >> 2)) & 0x030c30c3u;
    v = (v ^ (v >> 4)) & 0x0300f00fu;
    v = (v ^ (v >> 8)) & 0xff0000ffu;
    v = (v ^ (v >> 16)) & 0x000003ffu;
    return v;
}

float3 aov_pixel(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam, const float4* offsets, int x, int y,
                 int width, int height, uint32_t accumulation, int entry, const RenderSettings& settings) {
    Payload payload;
    payload.pixel_hash = rng::pcg2d(uint32_t(x), uint32_t(y)).x;
    payload.accumulation = accumulation;
    generate_camera_ray(cam, x, y, width, height, accumulation, payload.position, payload.direction);
    float depth = 0.0f;
    float3 last_ray_direction = payload.direction;
    do {
        last_ray_direction = payload.direction;
        float3 last_position = payload.position;
        Ray ray = {payload.position, payload.ray_min_t, payload.direction, INFINITY};
        Hit hit = settings.use_wide8 ? closest_hit_wide8(scene, ray, payload.last_triangle, nullptr)
                  : settings.use_wide ? closest_hit_wide(scene, ray, payload.last_triangle, nullptr)
                  : settings.use_bvh ? closest_hit_bvh(scene, ray, payload.last_triangle, nullptr) : closest_hit_bruteforce(scene, ray, payload.last_triangle, nullptr);
        intersect_lights(scene, ray, hit);
        if (hit.id == HIT_MISS) {
            payload.throughput = {0, 0, 0};
            payload.position = 1e30f * payload.direction;
        } else if (hit.id & HIT_LIGHT_BIT) {
            if (entry == HIPR_ENTRY_DENOISER_ALBEDO) {   // AIDenoiser::path_tracing_RPG, SimpleRGPs.cu:180-182: radiance / (1 + radiance) of the light the path hit
                const float3 L = Lights::evaluate_intersection(scene.lights[hit.id & ~HIT_LIGHT_BIT], ray.origin, ray.direction, payload.bsdf_PDF);
                return L / (make_float3(1) + L);
            }
            payload.throughput = {0, 0, 0};
            payload.position = ray.direction * hit.t + ray.origin;
        } else
            closest_hit_program(scene, state, cam, offsets, payload, hit, ray.direction);
        depth += length(last_position - payload.position);
    } while (payload.material_index == 0 && !is_black(payload.throughput));

    if (entry == HIPR_ENTRY_DEPTH)
        return make_float3(depth);
    if (payload.material_index == 0)
        return {0, 0, 0};
    const HiprMaterial& mp = scene.materials[payload.material_index];
    const float4 scale = payload.tint_and_roughness_scale;
    const float4 tr = material_tint_roughness(scene, mp, payload.texcoord);
    switch (entry) {
    case HIPR_ENTRY_TINT: return make_float3(tr) * make_float3(scale);
    case HIPR_ENTRY_ROUGHNESS: return make_float3(tr.w * scale.w);
    case HIPR_ENTRY_SHADING_NORMAL: return payload.shading_normal * 0.5f + 0.5f;
    case HIPR_ENTRY_PRIMITIVE_ID: {
        uint32_t instance_encoding = uint32_t(payload.instance_id) & 0x3FFFFFFu;
        uint32_t primitive_encoding = rng::reverse_bits(uint32_t(payload.primitive_index) + 1u) >> 2;
        uint32_t code = instance_encoding ^ primitive_encoding;
        return make_float3(float(compact_by_2(code >> 2)), float(compact_by_2(code >> 1)), float(compact_by_2(code))) / 1023.0f;
    }
    case HIPR_ENTRY_ALBEDO: case HIPR_ENTRY_DENOISER_ALBEDO: {
        float abs_cos_theta = fabsf(dot(last_ray_direction, payload.shading_normal));
        float4 trq = tr * scale;
        MaterialInputs in = {make_float3(trq), trq.w, mp.specularity, material_metallic(scene, mp, payload.texcoord), unorm16(mp.coat), unorm16(mp.coat_roughness)};
        // the denoiser's feature image takes every material as DefaultShading (SimpleRGPs.cu:171-178)
        if (entry == HIPR_ENTRY_ALBEDO && mp.shading_model == HIPR_SHADING_DIFFUSE) return in.tint;
        if (entry == HIPR_ENTRY_ALBEDO && mp.shading_model == HIPR_SHADING_TRANSMISSIVE) return TransmissiveShading(in, abs_cos_theta).rho(abs_cos_theta);
        return DefaultShading(in, abs_cos_theta).rho(abs_cos_theta);
    }
    }
    return {0, 0, 0};
}

} // namespace oracle
