// wide8_kernels.h -- K2 / K4 over the 8-wide compressed BVH with leaf records (include/hiprenderer_c.h "wide8"): the traversal kernel of every scene
// with more than 64 BVH2 nodes since round 3.
//
// Why this tree. Measured on the MI355X (profiles/r03_*): the 4-wide kernel kept the CU's texture addresser 82 % busy -- every 16-byte load a lane issues
// to its own address costs that pipeline about a cycle, whatever it returns, and one more load per node visit (+20 % lane-loads) cost +9 %, two +27 % --
// while its waves spent half their time waiting on the chain pop -> fetch -> test -> push. So the tree is built for few lane-loads and few dependent
// fetches per ray: an inner node holds EIGHT quantised child boxes in 64 bytes (four loads; a visit used to buy four boxes), a leaf is one 64-byte
// record of two triangles that share an edge (four loads for two triangles; they used to cost six), inner nodes and records share one slot array so that
// one base index addresses all children, and the visiting order comes from the ray's octant, not from sorted distances: the children sit in positions
// that say on which side of the node's centre they lie, and a ray takes the hit ones in ascending (position XOR octant).
//   atrium, per closest-hit ray: 17.8 node + 2.6 leaf visits and 83 lane-loads before, 13.1 + 2.4 visits and 62 lane-loads now; per shadow ray 20.1 + 7.0
//   visits and 121 lane-loads before, 11.4 + 5.5 visits and 68 now.
//
// Same specification as oracle/integrator.cpp traverse_wide8 (hits, transmittance and the node / triangle counters are bit-identical):
//   * GROUP = (base | valid << 24, pending | inner_mask << 8): the hit children of one node still to be visited; `pending` bit p stands for the child in
//     position p ^ octant, octant bit a = (1 / direction[a] < 0). A ray takes children from its current group lowest pending bit first, inner nodes and
//     leaf records alike; when the group is empty it pops one from its stack (LDS, one 8-byte entry per lane and depth).
//   * visiting a node: slab distances of the 8 quantised boxes, fma(float(q), A, B) with A = 2^(e - 127) * inv_d, B = fma(origin, inv_d, -ood) and
//     origin = fma(float(m), grid_cell, grid_min); tnear = max(entry distances, tmin), tfar = min(exit distances, tmax); a child is hit unless
//     fma(tfar, 1.0000004f, -tnear) has its sign bit set (the 3 ulp of slack of the other searches, folded into the one rounding of the fma). If any
//     child is hit, the current group is pushed when it still has pending children, and the node's hits become the current group.
//   * visiting a record: triangle A = (a; e1, e2), then B = (a; e2, e3), each with the Moeller-Trumbore solve of kernels.h on the stored edges; the
//     weights (1 - u - v, u, v) of the record's corners become the scene triangle's (u, v) through the record's selectors.
// Tried and not adopted: TWO rays per lane, the lane bringing forward whichever of its rays has an item of the kind the wave is working on
// (tools/experiments/wide8_dual_kernels.h.txt; per-ray order unchanged, the parity tests pass with it). The wave's instructions find more lanes with work,
// but the second set of ray registers (26 per lane) does not fit next to the node block's temporaries: at four waves per SIMD the compiler spills 114-128
// VGPRs (round 3: 50.2 -> 83.0 ms per atrium step), and at three waves, where nothing spills, three instruction streams per SIMD do not cover the chain
// vote -> exchange -> gather -> test (round 4: 45.9 -> 63.1 ms; profiles/r04_ab_two_rays_three_waves.txt).
// The wave-level machinery (persistent waves claiming 64-ray chunks from sharded counters, refilling idle lanes, one KIND of item per iteration chosen by
// a wave vote, fused closest-hit + shadow launches) is that of k_trace_persistent (kernels.h), which remains for the 4-wide tree.
#pragma once

#include "kernels.h"

namespace hipr {

struct Wide8Scene {
    const uint4* slots;          // HiprSlot8, 4 x uint4 each
    uint32_t slot_count;
    float grid_min[3], grid_cell[3];
    uint32_t cull_backfaces;     // hipr_set_backface_culling: closest hits clearly on the back of a one-sided record triangle are stepped over
};

// LDS stack entries (8 bytes each) by tree height: the stack holds at most one group per level above the current one.
constexpr int WIDE8_STACK_SHALLOW = 12;      // 12 KB per block of two waves: twelve blocks per CU, six waves per SIMD
#ifndef HIPR_WIDE8_WAVES
#define HIPR_WIDE8_WAVES 6
#endif
#ifndef HIPR_WIDE8_SIGN_HITS
#define HIPR_WIDE8_SIGN_HITS 1      // the hit children of a node from the sign bits of fma(tfar, slack, -tnear): the specification since round 4 (0: round 3's multiply + compare)
#endif
// The wave runs the record (leaf) block when leaf lanes x DEN > node lanes x NUM. Rounds 3-5 ran the kind MORE lanes wait for (1 / 1). A record iteration is the
// cheaper of the two, so the wave's progress per instruction is best when records are taken at about half the node lanes' count already (1 / 2): atrium
// 5 551 -> 5 618 Mrays/s (trace 79.8 -> 78.7 ms per step), material scene +0.6 %; 3 / 4 gives half of that, 1 / 3 nothing, 1 / 4 and anything above 1 / 1 lose
// (profiles/r06_ab_leaf_vote.txt). A lane's own order of items does not depend on the vote: hits and counters are unchanged.
#ifndef HIPR_WIDE8_LEAF_VOTE_NUM
#define HIPR_WIDE8_LEAF_VOTE_NUM 1
#define HIPR_WIDE8_LEAF_VOTE_DEN 2
#endif
#ifndef HIPR_WIDE8_WAVES_LOW
#define HIPR_WIDE8_WAVES_LOW 6
#endif
HD constexpr int wide8_waves_per_simd(int stack_entries) { return stack_entries <= 8 ? HIPR_WIDE8_WAVES_LOW : (stack_entries <= 12 ? HIPR_WIDE8_WAVES : (stack_entries <= 16 ? 5 : 4)); }

// COVERAGE_R8: ... and every coverage texture is single-channel 8-bit, linear (kernels.h material_coverage_r8).
// COVERAGE: the scene holds triangles that are not statically opaque (coverage textures, cut-outs, partial coverage): a shadow ray that hits one samples its material's
// coverage. Scenes without any (the common case; decided at upload) run the instantiation without that code: it is a fifth of the kernel's instructions and, though never
// executed there, it weighs on the register allocation of the loop around it (profiles/r04_ab_trace_without_coverage.txt).
// SORTED: the rays are taken in the order of `sorted` (ray_sort.hip: closest-hit rays first, each kind by origin cell and direction octant) instead of queue order;
// every result is stored where it is in queue order, so nothing downstream sees the difference.
template <int STACK, int MODE, bool INSTRUMENT, bool COVERAGE = true, bool SORTED = false, bool COVERAGE_R8 = false>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(wide8_waves_per_simd(STACK)))) void k_trace_wide8(DeviceScene sc, Wide8Scene tree, PathState in, float4* hits, ShadowQueue q,
        float4* radiance, const uint32_t* closest_count_ptr, const uint32_t* shadow_count_ptr, uint32_t* work_counter, int refill_below, DeviceCounters* counters,
        const uint32_t* sorted = nullptr) {
    __shared__ uint2 s_stack[STACK * TRACE_BLOCK];
    uint2* stack = s_stack + threadIdx.x;
    const uint32_t n_closest = MODE != TRACE_SHADOW ? *closest_count_ptr : 0u;
    const uint32_t n = n_closest + (MODE != TRACE_CLOSEST ? *shadow_count_ptr : 0u);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t chunk_size = min(TRACE_CHUNK_MAX, max(64u, (n / (gridDim.x * (TRACE_BLOCK / 64u) * 2u)) & ~63u));

    uint32_t chunk_next = 0, chunk_end = 0;   // wave uniform
    uint32_t shard = (blockIdx.x & 7u) * (TRACE_SHARDS / 8u) + ((blockIdx.x >> 3) % (TRACE_SHARDS / 8u));     // blocks of one XCD start on neighbouring shards
    bool exhausted = false;

    bool active = false, finished = false;     // finished: traversal done, result still in registers
    bool is_shadow = MODE == TRACE_SHADOW;     // per lane in the fused mode
    uint32_t ray_index = 0;
    f3 o = {0, 0, 0}, d = {0, 0, 1}, inv = {0, 0, 0}, ood = {0, 0, 0};
    float tmin = 0.0f, tmax = 0.0f;            // tmax: best distance so far (closest) or the ray extent (shadow)
    // closest: pay_x, pay_y = barycentrics of the best hit, pay_z = bits of its id, pay_k = the triangle the ray left from
    // shadow : pay_x, pay_y, pay_z = radiance carried by the ray,                  pay_k = radiance slot of the path
    float pay_x = 0.0f, pay_y = 0.0f, pay_z = __uint_as_float(HIPR_HIT_MISS);
    uint32_t pay_k = HIPR_NO_TRIANGLE;
    uint32_t octant = 0;
    uint32_t gx = 0, gy = 0;                   // the current group
    uint32_t item = 0;                         // slot of the item this lane works on next
    bool item_is_leaf = false;
    int sp = 0;
    uint32_t nodes = 0, tris = 0, shadow_nodes = 0, shadow_tris = 0;
    uint32_t diag_node_iterations = 0, diag_node_lanes = 0, diag_triangle_iterations = 0, diag_triangle_lanes = 0, diag_busy_lanes = 0, diag_refills = 0;   // lane 0 only
    uint32_t diag_pushes = 0, diag_pushes_deep = 0;

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
                    const uint32_t entry = SORTED ? sorted[idx] : idx;
                    if (is_shadow) {
                        if constexpr (MODE != TRACE_CLOSEST) {
                            const uint32_t si = entry - n_closest;
                            ray_index = si;
                            ro = q.o_tmax[si]; rdv = q.d_slot[si];
                            const float4 rr = q.radiance[si];
                            pay_x = rr.x; pay_y = rr.y; pay_z = rr.z;
                            pay_k = __float_as_uint(rdv.w);
                            tmin = 0.0f; tmax = ro.w;
                        }
                    } else if constexpr (MODE != TRACE_SHADOW) {
                        ray_index = entry;
                        const uint2 meta = in.meta[entry];
                        dead = meta.x == HIPR_DEAD_SLOT;
                        pay_k = meta.y;
                        ro = in.o_tmin[entry]; rdv = in.d_pdf[entry];
                        tmin = ro.w; tmax = __builtin_inff();
                        pay_x = pay_y = 0.0f; pay_z = __uint_as_float(HIPR_HIT_MISS);
                    }
                    if (dead) hits[entry] = make_float4(0, 0, 0, __uint_as_float(HIPR_HIT_MISS));
                    else {
                        o = mk3(ro.x, ro.y, ro.z); d = mk3(rdv.x, rdv.y, rdv.z);
                        const f3 sd = {fabsf(d.x) > 1e-20f ? d.x : copysignf(1e-20f, d.x), fabsf(d.y) > 1e-20f ? d.y : copysignf(1e-20f, d.y),
                                       fabsf(d.z) > 1e-20f ? d.z : copysignf(1e-20f, d.z)};
                        inv = {1.0f / sd.x, 1.0f / sd.y, 1.0f / sd.z};
                        ood = o * inv;
                        octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
                        sp = 0;
                        gx = 1u << 24; gy = 1u << 8;       // the root as a group of its own (one inner child in position 0 of base 0), already taken:
                        item = 0u; item_is_leaf = false;   // the ray starts on the root node
                        if (tree.slot_count == 0) finished = true;
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

        // ---- traverse. Every iteration runs ONE of the two blocks -- the one more lanes are waiting for.
        do {
            const bool leaf_mode = active & item_is_leaf;
            const bool node_mode = active & !item_is_leaf;
            const unsigned long long lmask = wave_ballot(leaf_mode), nmask = wave_ballot(node_mode);
            bool advance = false;       // this lane finished its item in this iteration and takes the next one
            if (INSTRUMENT && lane == 0) {
                const bool leaves = __popcll(lmask) * HIPR_WIDE8_LEAF_VOTE_DEN > __popcll(nmask) * HIPR_WIDE8_LEAF_VOTE_NUM;
                diag_triangle_iterations += leaves; diag_triangle_lanes += leaves ? __popcll(lmask) : 0;
                diag_node_iterations += !leaves; diag_node_lanes += leaves ? 0 : __popcll(nmask);
                diag_busy_lanes += __popcll(lmask | nmask);
            }
            if (__popcll(lmask) * HIPR_WIDE8_LEAF_VOTE_DEN > __popcll(nmask) * HIPR_WIDE8_LEAF_VOTE_NUM) {
                if (leaf_mode) {
                    const uint4* rp = tree.slots + 4 * size_t(item);
                    const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
                    const f3 a = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                    const f3 e1 = mk3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                    const f3 e2 = mk3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                    const f3 e3 = mk3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
                    const uint32_t flags = r3.z;
                    // Closest hits the hit program would refuse -- a one-sided surface reached from behind, MonteCarlo.cu:147-164 -- and have the ray traced again from
                    // just past: stepped over here. The solve's determinant carries the facing, positive from the front of the record's winding (B's may be the
                    // other way round: flags bit 4); refused below -facing_margin, never (limit -inf) for a two-sided triangle or with the culling off.
#ifndef HIPR_WIDE8_CULL_CODE
#define HIPR_WIDE8_CULL_CODE 1      // 0: compiled out (A/B of what the test itself costs)
#endif
                    const uint32_t cull = HIPR_WIDE8_CULL_CODE && tree.cull_backfaces != 0u ? flags : 0u;
                    const float refuse_a_below = (cull & 4u) ? -__uint_as_float(r3.w) : -__builtin_inff();
                    const float refuse_b_below = (cull & 8u) ? -__uint_as_float(r3.w) : -__builtin_inff();
                    bool testing = true;
                    advance = true;
#pragma unroll
                    for (int which = 0; which < 2; ++which) {
                        const uint32_t id = which == 0 ? r3.x : r3.y;
                        testing = testing & (id != HIPR_LEAF8_NONE);
                        if (testing) {
                            if (INSTRUMENT) { tris += is_shadow ? 0u : 1u; shadow_tris += is_shadow ? 1u : 0u; }
                            TriangleTest test;
                            const f3 first = which == 0 ? e1 : e2, second = which == 0 ? e2 : e3;
                            const bool hit = triangle_inside(a, first, second, o, d, test);
                            float t = 0.0f, ru = 0.0f, rv = 0.0f;
                            if (wave_any(hit)) triangle_hit_values(test, second, t, ru, rv);     // the lanes in this branch agree to skip the division
                            const float rw = 1.0f - ru - rv;
                            const uint32_t su = (flags >> (which == 0 ? 8 : 12)) & 3u, sv = (flags >> (which == 0 ? 10 : 14)) & 3u;
                            const float u = su == 0u ? rw : (su == 1u ? ru : rv), v = sv == 0u ? rw : (sv == 1u ? ru : rv);
                            if constexpr (MODE != TRACE_CLOSEST) {
                                if (is_shadow && hit && t > tmin && t < tmax) {
                                    float coverage = 1.0f;
                                    if (COVERAGE && !(flags >> which & 1u)) {
                                        // The triangle's texture coordinates and material index from its SHADING record (k_build_shade_triangles: the same floats the
                                        // vertex arrays hold, zero without texture coordinates): three loads side by side where triangle -> instance -> indices -> texture
                                        // coordinates were four one after the other (round 4: a wave waits on this chain for every lane that meets a cut-out;
                                        // profiles/r04_ab_coverage_chain.txt). triangle_texcoord's expression, term for term.
                                        const float4* record = sc.shade_triangles + SHADE_TRIANGLE_QUADS * size_t(id);
                                        const float4 r3 = record[3], r6 = record[6], r7 = record[7];
                                        const float w = 1.0f - u - v;
                                        const f2 uv = mk2(r6.z, r6.w) * u + mk2(r7.x, r7.y) * v + mk2(r6.x, r6.y) * w;
                                        coverage = COVERAGE_R8 ? material_coverage_r8(sc, sc.materials[__float_as_uint(r3.w)], uv) : material_coverage(sc, sc.materials[__float_as_uint(r3.w)], uv);
                                    }
                                    pay_x *= 1.0f - coverage; pay_y *= 1.0f - coverage; pay_z *= 1.0f - coverage;
                                    if (pay_x < 0.0000001f && pay_y < 0.0000001f && pay_z < 0.0000001f) {   // fully shadowed: the ray is done
                                        pay_x = pay_y = pay_z = 0.0f;
                                        testing = false;
                                        gy = 0u; sp = 0;        // nothing left to visit: the common tail below retires the ray
                                    }
                                }
                            }
                            if constexpr (MODE != TRACE_SHADOW) {
                                const float facing = which == 0 ? test.det : __uint_as_float(__float_as_uint(test.det) ^ ((flags & 16u) << 27));
                                const bool refused = facing < (which == 0 ? refuse_a_below : refuse_b_below);
                                const bool closer = !is_shadow & hit & !refused & (id != pay_k) & (t > tmin) & ((t < tmax) | ((t == tmax) & (id < __float_as_uint(pay_z))));
                                tmax = closer ? t : tmax; pay_x = closer ? u : pay_x; pay_y = closer ? v : pay_y; pay_z = closer ? __uint_as_float(id) : pay_z;
                            }
                        }
                    }
                }
            } else if (node_mode) {
                const uint4* np = tree.slots + 4 * size_t(item);
                const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
                if (INSTRUMENT) { nodes += is_shadow ? 0u : 1u; shadow_nodes += is_shadow ? 1u : 0u; }
                advance = true;
                // origin on the scene grid: three 21-bit fields
                const uint32_t mx = w0.x & 0x1FFFFFu, my = (w0.x >> 21) | ((w0.y & 0x3FFu) << 11), mz = (w0.y >> 10) & 0x1FFFFFu;
                const float ox = fmaf(float(mx), tree.grid_cell[0], tree.grid_min[0]), oy = fmaf(float(my), tree.grid_cell[1], tree.grid_min[1]),
                            oz = fmaf(float(mz), tree.grid_cell[2], tree.grid_min[2]);
                const float ax = __uint_as_float((w0.z & 0xFFu) << 23) * inv.x, ay = __uint_as_float(((w0.z >> 8) & 0xFFu) << 23) * inv.y,
                            az = __uint_as_float(((w0.z >> 16) & 0xFFu) << 23) * inv.z;
                const float bx = fmaf(ox, inv.x, -ood.x), by = fmaf(oy, inv.y, -ood.y), bz = fmaf(oz, inv.z, -ood.z);
                // fma(q, A, B) is monotone in q with the sign of A and qlo <= qhi: the entry bound of an axis is the low one where the ray travels in +axis
                // (an empty position has qlo > qhi and is masked by `valid` below)
                const bool px = ax >= 0.0f, py = ay >= 0.0f, pz = az >= 0.0f;
                const uint32_t nx[2] = {px ? w1.x : w2.z, px ? w1.y : w2.w}, fx[2] = {px ? w2.z : w1.x, px ? w2.w : w1.y};
                const uint32_t ny[2] = {py ? w1.z : w3.x, py ? w1.w : w3.y}, fy[2] = {py ? w3.x : w1.z, py ? w3.y : w1.w};
                const uint32_t nz[2] = {pz ? w2.x : w3.z, pz ? w2.y : w3.w}, fz[2] = {pz ? w3.z : w2.x, pz ? w3.w : w2.y};
                uint32_t h = 0u;
#if HIPR_WIDE8_SIGN_HITS
                // A child is missed when fma(tfar, 1 + 3 ulp, -tnear) is negative: the sign bits of the eight differences are shifted into one word
                // (v_alignbit_b32: (h << 1) | (difference >> 31)), last child first, so that bit k stands for child k. One fma and one funnel shift
                // per child where a multiply, a compare, a select and a third of an or used to be: 231 -> 219 VALU instructions per node visit,
                // atrium trace 46.4 -> 45.5 ms per step (profiles/r04_ab_node_block.txt; the same table through an LDS byte table for the octant
                // permutation: 46.1, not adopted).
#pragma unroll
                for (int k = 7; k >= 0; --k) {
                    const int word = k >> 2, shift = 8 * (k & 3);
                    const float x0 = fmaf(float((nx[word] >> shift) & 0xFFu), ax, bx), x1 = fmaf(float((fx[word] >> shift) & 0xFFu), ax, bx);
                    const float y0 = fmaf(float((ny[word] >> shift) & 0xFFu), ay, by), y1 = fmaf(float((fy[word] >> shift) & 0xFFu), ay, by);
                    const float z0 = fmaf(float((nz[word] >> shift) & 0xFFu), az, bz), z1 = fmaf(float((fz[word] >> shift) & 0xFFu), az, bz);
                    const float tnear = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
                    const float tfar = fminf(fminf(fminf(x1, y1), z1), tmax);
                    h = __builtin_amdgcn_alignbit(h, __float_as_uint(fmaf(tfar, 1.0000004f, -tnear)), 31u);
                }
                h = ~h & (w0.w >> 24);
#else
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int word = k >> 2, shift = 8 * (k & 3);
                    const float x0 = fmaf(float((nx[word] >> shift) & 0xFFu), ax, bx), x1 = fmaf(float((fx[word] >> shift) & 0xFFu), ax, bx);
                    const float y0 = fmaf(float((ny[word] >> shift) & 0xFFu), ay, by), y1 = fmaf(float((fy[word] >> shift) & 0xFFu), ay, by);
                    const float z0 = fmaf(float((nz[word] >> shift) & 0xFFu), az, bz), z1 = fmaf(float((fz[word] >> shift) & 0xFFu), az, bz);
                    const float tnear = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
                    float tfar = fminf(fminf(x1, y1), z1);
                    tfar = fminf(tfar, tmax) * 1.0000004f;
                    h |= tnear <= tfar ? (1u << k) : 0u;
                }
                h &= w0.w >> 24;
#endif
                // bit k -> bit k ^ octant: three conditional swaps of neighbouring bits, pairs and nibbles
                h = (octant & 1u) ? (((h & 0x55u) << 1) | ((h >> 1) & 0x55u)) : h;
                h = (octant & 2u) ? (((h & 0x33u) << 2) | ((h >> 2) & 0x33u)) : h;
                h = (octant & 4u) ? (((h & 0x0Fu) << 4) | (h >> 4)) : h;
                if (h) {
                    if (gy & 0xFFu) {
                        stack[sp * TRACE_BLOCK] = make_uint2(gx, gy);
                        if (INSTRUMENT) { ++diag_pushes; diag_pushes_deep += sp >= 8; }
                        ++sp;
                    }
                    gx = w0.w;
                    gy = h | (w0.z >> 24) << 8;
                }
            }
            if (advance) {
                // the next item: the nearest-octant pending child of the current group, or of the group on top of the stack
                if ((gy & 0xFFu) == 0u && sp > 0) {
                    --sp;
                    const uint2 popped = stack[sp * TRACE_BLOCK];
                    gx = popped.x; gy = popped.y;
                }
                const bool done = (gy & 0xFFu) == 0u;
                const uint32_t p = uint32_t(__builtin_ctz(gy | 0x100u));      // the low byte is non-zero unless the ray is done
                gy = done ? gy : (gy & (gy - 1u));
                const uint32_t position = (p ^ octant) & 7u;
                item_is_leaf = !((gy >> (8u + position)) & 1u);
                item = (gx & 0xFFFFFFu) + uint32_t(__popc((gx >> 24) & ((1u << position) - 1u)));
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
        wave_add(&counters->pushes, diag_pushes); wave_add(&counters->pushes_past_16, diag_pushes_deep);
        if (MODE != TRACE_SHADOW) { wave_add(&counters->closest_nodes, nodes); wave_add(&counters->closest_triangles, tris); }
        if (MODE != TRACE_CLOSEST) { wave_add(&counters->shadow_nodes, shadow_nodes); wave_add(&counters->shadow_triangles, shadow_tris); }
    }
}

} // namespace hipr
