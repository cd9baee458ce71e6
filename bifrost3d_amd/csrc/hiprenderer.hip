// hiprenderer.hip -- implementation of the C-ABI declared in include/hiprenderer_c.h.
//
// Host side of the MI355X wavefront path tracer: owns the device copies of the scene, the SoA
// path / hit / shadow queues and the f64 accumulation buffer, and drives the per-pass kernel loop
//   generate -> { trace_closest -> shade(+compact) -> trace_shadow } until no path is alive -> accumulate.
// Replaces the OptiX context + context->launch() of OR/Renderer.cpp:273-574,1250-1265.
// There is no CPU fallback: every entry point that needs the GPU fails with a status code.
#include "kernels.h"
#ifndef HIPR_WIDE8_LOW_BUCKET
#define HIPR_WIDE8_LOW_BUCKET 0
#endif
#include "wide8_kernels.h"
#include "launch.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace hipr;

namespace {

thread_local std::string g_last_error;

int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list args;
    va_start(args, fmt);
    vsnprintf(buf, sizeof(buf), fmt, args);
    va_end(args);
    g_last_error = buf;
    fprintf(stderr, "hiprenderer: %s\n", buf);
    return status;
}

#define HIP_TRY(call)                                                                                          \
    do {                                                                                                       \
        hipError_t err__ = (call);                                                                             \
        if (err__ != hipSuccess)                                                                               \
            return fail(err__ == hipErrorOutOfMemory ? HIPR_ERROR_OUT_OF_MEMORY : HIPR_ERROR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(err__), __FILE__, __LINE__);                                         \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int resize(size_t new_bytes) {
        if (new_bytes <= bytes && ptr) return HIPR_OK;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
        if (new_bytes == 0) return HIPR_OK;
        HIP_TRY(hipMalloc(&ptr, new_bytes));
        bytes = new_bytes;
        return HIPR_OK;
    }
    int upload(const void* src, size_t n, hipStream_t stream) {
        if (n == 0) return HIPR_OK;
        if (int s = resize(n)) return s;
        HIP_TRY(hipMemcpyAsync(ptr, src, n, hipMemcpyHostToDevice, stream));
        return HIPR_OK;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
    template <typename T> T* as() const { return static_cast<T*>(ptr); }
};

struct TimedLaunch { int kernel; hipEvent_t start, stop; };

// One independent wavefront of paths: its own queues, queue sizes and stream. A pass splits its path slots over the
// context's wavefronts; they advance through their bounces independently, so while one of them shades (few resident
// waves, waiting on gathers) the other one traces (many waves, latency bound) on the same CUs.
struct Wavefront {
    hipStream_t stream = nullptr;       // wavefront 0 runs on the context stream, the others on their own
    hipStream_t own_stream = nullptr;
    hipEvent_t shade_done[2] = {nullptr, nullptr}, counts_copied[2] = {nullptr, nullptr}, finished = nullptr;   // by bounce parity: two bounces are in flight
    DeviceBuffer path[2][4], hits, shadow[3], queue_counts, order, order_coat, nee_flags, sort_keys[2], sort_order, sort_temp;     // sort_*: ray_sort.hip (HIPR_COHERENCE_SORT=1)   // queue_counts: COUNT_LINES 64 B lines (see there); order: k_classify_hits' listing of a bounce's rays
    uint32_t* host_counts = nullptr;    // pinned: {continuing paths, shadow rays} per bounce parity, [4] staging word
    uint32_t first_slot = 0, n_slots = 0;      // first_slot: the wavefront's phase in the round-robin deal of 64-slot groups (partition_path_slots)

    PathState path_state(int which) const {
        return {path[which][0].as<float4>(), path[which][1].as<float4>(), path[which][2].as<float4>(), path[which][3].as<uint2>()};
    }
    ShadowQueue shadow_queue() const { return {shadow[0].as<float4>(), shadow[1].as<float4>(), shadow[2].as<float4>()}; }
    void release() {
        for (auto& buffers : path) for (DeviceBuffer& b : buffers) b.release();
        hits.release(); queue_counts.release(); order.release(); order_coat.release(); nee_flags.release();
        sort_keys[0].release(); sort_keys[1].release(); sort_order.release(); sort_temp.release();
        for (DeviceBuffer& b : shadow) b.release();
        for (hipEvent_t e : shade_done) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : counts_copied) if (e) (void)hipEventDestroy(e);
        if (finished) (void)hipEventDestroy(finished);
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (host_counts) (void)hipHostFree(host_counts);
        *this = Wavefront();
    }
};
constexpr int MAX_WAVEFRONTS = 4;
constexpr int COUNT_PAIR_STRIDE = 16;   // uint32 words between the counters of a wavefront (one 64 B line each)
// A wavefront's counters: lines 0-2 the queue-size pairs {paths, shadow rays}, bounce k reads pair k % 3 and fills pair (k + 1) % 3; lines 3-4
// k_classify_hits' 8-byte counter of bounce parity 0 and 1. Nothing in the stream zeroes them between bounces: the shade kernel of bounce k zeroes
// pair (k + 2) % 3 -- read last by bounce k - 1, whose sizes the host has read back before it queues bounce k -- and the listing counter of parity (k + 1) & 1.
constexpr int COUNT_PAIRS = 3, COUNT_LINES = 5;
constexpr int HOST_COUNT_WORDS = 8 + COUNT_LINES * COUNT_PAIR_STRIDE;   // pinned: 2 x {paths, shadow rays} read back, [4..8) spare, then the image of the counters at the start of a pass

} // namespace

struct HiprContext {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr;
    hipEvent_t pass_start = nullptr;
    Wavefront wavefronts[MAX_WAVEFRONTS];
    int wavefront_limit = 0;                // hipr_set_wavefront_count / HIPR_WAVEFRONTS; 0 = by scene: see wavefronts_wanted()
    int wavefront_count = 1;                // set by partition_path_slots: small frames run as one wavefront
    int partitioned_for = 0;                // the wavefronts_wanted() the current partition was made for

    // scene
    DeviceBuffer shade_triangles, trace_triangles, trace_items, wide_nodes, wide8_slots, environment_PDF, environment_samples;
    Wide8Scene wide8 = {};              // the 8-wide tree with leaf records: what the persistent kernels walk when the scene brings one
    uint32_t wide8_height = 0;
    DeviceBuffer nodes, triangles, instances, indices, geometry, texcoords, tints, emissions, materials, lights, textures, texels;
    bool coverage_textures_r8 = false;  // every coverage texture of the uploaded scene is HIPR_TEXEL_R8 and linear: k_trace_wide8<..., COVERAGE_R8 = true>
    bool all_triangles_opaque = false;  // no triangle of the uploaded scene needs its material's coverage sampled (HIPR_TRIANGLE_OPAQUE on all): k_trace_wide8<..., COVERAGE = false>
    bool lean_trace = true;             // HIPR_LEAN_TRACE=0: always the full kernel
    bool scene_has_environment = true;  // ... an environment map, a presampled environment light or a float texture: k_shade<..., TEXTURES = 2>; 8-bit textures without those: TEXTURES = 1
    bool scene_has_textures = true;     // a material references a texture, or the scene brings an environment map / presampled environment light: k_shade<..., TEXTURES = true>
    bool lean_shade = true;             // HIPR_LEAN_SHADE=0: always the full kernel
    DeviceBuffer triangle_class;        // one byte per triangle for the listing pass (k_classify_hits): bit 0 = its material is coated
    int arithmetic = HIPR_ARITHMETIC_FAST;     // hipr_set_arithmetic: which build of the shade unit the context launches (launch.h ShadeUnit)
    const hipr::ShadeUnit& shade_unit() const { return arithmetic == HIPR_ARITHMETIC_EXACT ? hipr::shade_unit_exact() : hipr::shade_unit_fast(); }
    bool coherence_sort = false;        // HIPR_COHERENCE_SORT=1: the rays of a fused trace launch are taken by (origin cell, octant), ray_sort.hip (built and measured in round 4: profiles/r04_ab_coherence_sort.txt)
    bool any_coated_triangle = false, shade_classes = false;     // HIPR_SHADE_CLASSES=1: coated surface hits listed apart (built and measured in round 4: no gain, profiles/r04_ab_shade_classes.txt)
    DeviceBuffer ggx_rho, dielectric_rho, alpha, sample_offsets, sobol_tables;
    DeviceScene scene = {};
    bool tables_ready = false, scene_ready = false;
    size_t uploaded_instance_bytes = 0;
    uint32_t uploaded_material_count = 0, uploaded_texture_count = 0, uploaded_vertex_count = 0, uploaded_index_count = 0;   // pools a geometry update leaves in place
    uint64_t uploaded_texel_bytes = 0;
    int stack_size = 16;
    int shading_models = 7;             // bit mask of the shading models the scene's instances reference
    std::vector<HiprMaterial> uploaded_materials;      // host copies of what hipr_upload_scene put on the device: a refit (hipr_update_scene_geometry) leaves the material pool
    std::vector<uint32_t> uploaded_light_types;        // and the environment data as they are, so what is derived from them must come from THESE, not from the refit's description

    // frame
    FrameInfo frame = {};
    bool frame_ready = false;
    uint32_t n_slots = 0;   // owned_tiles * 64 * samples_per_pass
    // Pass pipelining (scenes of the persistent kernels, one wavefront per pass): consecutive passes alternate between two SLOTS -- wavefronts[0] / [1] with
    // their queues and streams, `radiance` / `radiance_other`, two halves of the work-counter ring -- and a pass whose live paths have dwindled to a
    // sliver hands its remaining bounces to the GPU blindly (the kernels read their sizes on the device) and returns: the tail, a chain of launches that
    // each run as long as ONE traversal takes, drains on its stream while the next pass's full-size launches run on the other one.
    struct PassSlot {
        bool pending = false;           // a detached tail is (or may be) still running on the slot's stream
        uint32_t alive_at_detach = 0;   // paths that entered the first bounce of the tail
        uint32_t first_parity = 0;      // the first tail bounce's sizes come back through the wavefront's regular read-back pair of this parity
        uint32_t blind_bounces = 0;     // bounces after it, read back into `tail_counts`
        uint32_t next_bounce = 0;       // the bounce that would follow the tail
        uint32_t* tail_counts = nullptr;   // pinned: {paths that continue, shadow rays queued} per blind bounce
        uint32_t work_index = 0;        // next unused claim-counter set of the slot's half of the ring
        HiprCameraState camera = {};    // of the pass that left the tail (finish_slot may have to queue more bounces)
    } pass_slots[2];
    int active_slot = 0;                // the slot whose pass is being queued (next_work_counter)
    int traced_slot = 0;                // the slot of the last hipr_trace_pass: what hipr_accumulate_samples folds
    int next_slot = 0;
    bool pipeline_passes = false;       // hipr_set_pass_pipelining / HIPR_PIPELINE_PASSES=1; off by default: measured slower
    bool pipelining_now = false;        // the pass being queued is a pipelined one
    int pipeline_spare_blocks = 1;      // HIPR_PIPELINE_SPARE_BLOCKS
    hipEvent_t accumulated = nullptr;   // the last hipr_accumulate_samples: the next one (on the other slot's stream) folds after it
    bool accumulated_valid = false;
    DeviceBuffer radiance_other;        // the other slot's radiance (the two swap when the slots do)
    uint32_t traced_samples = 0;        // samples the radiance buffer holds since the last hipr_trace_pass
    float pass_depth_normalizer = 0.0f; // depth entry: far - near of the traced pass's camera
    DeviceBuffer radiance, accumulation, scratch_accumulation, counters, work_counters;
    bool use_scratch = false;
    int entry = HIPR_ENTRY_PATH_TRACING;
    DeviceBuffer& active_accumulation() { return use_scratch ? scratch_accumulation : accumulation; }
    int trace_variant = -1;             // 1: persistent kernels, 0: one ray per lane, -1: pick by BVH size (HIPR_TRACE_VARIANT)
    uint32_t wide_stack_entries = 0;
    // persistent kernels walk the compressed 4-wide BVH; without one (HiprSceneDesc::wide_nodes == NULL) the plain BVH2 kernels serve every scene
    // Two half-frame wavefronts on two streams overlap one half's shading with the other's tracing. That pays where the trace kernels are short
    // and light (exhaustive search / BVH2: Cornell +9 % ... +27 %); the persistent wide-BVH kernels fill the register file on their own, the
    // other wavefront's blocks then wait for residency and nothing was gained with round 2's kernels (atrium 88.4 vs 88.8 ms, material 24.9 vs 25.1 ms
    // per step). With round 4's (profiles/r04_ab_wavefronts.txt) two wavefronts pay on every pass of tens of millions of paths -- atrium 63.4 -> 60.6 ms
    // per step, material 37.0 -> 35.7, 10 M triangles at 4K 91.8 -> 87.9: where one wavefront's launch drains, the other's blocks move in -- three and four do not
    // (62.0, 64.9), and a pass of 2 M paths (one accumulation per pass) loses 8 % to the halved launches: two from 2^24 path slots per pass on, else one.
    static constexpr uint64_t TWO_WAVEFRONTS_FROM_SLOTS = 1ull << 24;
    // `frame_is_set`: `frame` holds a valid description. hipr_set_frame partitions BEFORE it sets frame_ready (the first partition after an upload was
    // therefore always one wavefront with full-size queues, the first pass re-partitioned into two, and -- buffers never shrink -- wavefront 0 kept its full-size
    // queues next to wavefront 1's half: about 1.5x the queue memory of a 64-accumulation 1080p pass, plus a redundant allocation and synchronisation).
    int wavefronts_wanted(bool frame_is_set) const {
        if (wavefront_limit > 0) return wavefront_limit;
        if (scene_ready && use_persistent()) return frame_is_set && uint64_t(frame.owned_tiles) * 64u * frame.samples_per_pass >= TWO_WAVEFRONTS_FROM_SLOTS ? 2 : 1;
        return 2;
    }
    int wavefronts_wanted() const { return wavefronts_wanted(frame_ready); }
    // The search of the uploaded scene, fixed when it is uploaded (hipr_set_trace_variant / HIPR_TRACE_VARIANT name a request for the NEXT upload).
    int chosen_variant = HIPR_TRACE_BVH2;
    void choose_variant() {
        const bool large = scene.node_count > 64;
        int v = trace_variant;
        if (v < 0) v = large ? HIPR_TRACE_WIDE8_PERSISTENT : (scene.triangle_count <= SMALL_SCENE_TRIANGLES ? HIPR_TRACE_EXHAUSTIVE : HIPR_TRACE_BVH2);
        if (v == HIPR_TRACE_WIDE8_PERSISTENT && wide8.slot_count == 0) v = HIPR_TRACE_WIDE_PERSISTENT;      // no 8-wide tree (none given, or higher than the LDS stacks)
        if (v == HIPR_TRACE_WIDE_PERSISTENT && scene.wide_node_count == 0) v = HIPR_TRACE_BVH2;
        if (v == HIPR_TRACE_EXHAUSTIVE && scene.trace_item_count == 0 && scene.triangle_count != 0) v = HIPR_TRACE_BVH2;
        chosen_variant = v;
    }
    bool use_wide8() const { return chosen_variant == HIPR_TRACE_WIDE8_PERSISTENT; }
    bool use_wide4() const { return chosen_variant == HIPR_TRACE_WIDE_PERSISTENT; }
    bool use_persistent() const { return use_wide8() || use_wide4(); }      // fused launches, one wavefront
    // tiny scenes: exhaustive search over the triangles (k_trace_*_small)
    bool use_exhaustive() const { return chosen_variant == HIPR_TRACE_EXHAUSTIVE; }
    int active_trace_variant() const { return chosen_variant; }
    int cu_count = 256;
    int blocks_per_cu_override = 0;     // HIPR_BLOCKS_PER_CU
    int shade_blocks_per_cu = 0;        // persistent shade blocks per CU = waves per SIMD; 0: 3 (what the kernel is compiled for), 2 for all-Diffuse scenes (HIPR_SHADE_BLOCKS_PER_CU)
    uint32_t shade_ordered_from = 1u << 18;   // bounces with fewer paths than this are shaded in queue order (HIPR_SHADE_ORDERED_FROM)
    int refill_below = 40;              // persistent kernels refill a wave once fewer lanes than this are busy (HIPR_REFILL_BELOW)
    bool cull_backfaces = true;         // hipr_set_backface_culling / HIPR_BACKFACE_CULLING=0
    bool shade_split = false;           // HIPR_SHADE_SPLIT=1: the shade kernel as two, next event estimation and the rest (shade_kernel.h; measured)
    bool shade_ordered = true;          // k_classify_hits before k_shade (HIPR_SHADE_ORDERED=0: shade in queue order)
    bool shade_ordered_camera = false;  // ... also before shade(0) (HIPR_SHADE_ORDERED_CAMERA=1; measured: profiles/r04_ab_knobs.txt)
    int persistent_blocks_per_cu[3][3] = {{0, 0, 0}, {0, 0, 0}};   // [shadow][stack bucket]
    int wide8_blocks_per_cu[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // [mode][stack bucket]

    // bookkeeping
    HiprCounters total = {};   // since hipr_reset_counters
    bool instrument = false;
    bool trace_log = false;
    DeviceCounters trace_log_previous = {};
    bool timing = true;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<TimedLaunch> timed;
    HiprKernelTimes times = {};

    DeviceBuffer debug_a, debug_b, debug_c;

    hipEvent_t next_event() {
        if (events_used == event_pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
    // An event in a stream costs the command processor about 6 us between two kernels -- a tenth of a bounce of a few thousand paths -- so launches that
    // follow each other on a stream share the event between them: the stop of the one is the start of the next (until break_chain()).
    struct Chain { hipStream_t stream; hipEvent_t event; } chains[MAX_WAVEFRONTS + 1] = {};
    Chain& chain_of(hipStream_t on) {
        for (Chain& ch : chains) if (ch.stream == on) return ch;
        for (Chain& ch : chains) if (!ch.event) { ch.stream = on; return ch; }
        chains[0] = {on, nullptr};
        return chains[0];
    }
    void break_chain(hipStream_t on) { chain_of(on).event = nullptr; }     // something other than a timed launch was queued on the stream
    void begin_timed(int kernel, hipStream_t on) {
        if (!timing) return;
        Chain& ch = chain_of(on);
        TimedLaunch t = {kernel, ch.event ? ch.event : next_event(), next_event()};
        if (!t.start || !t.stop) return;
        if (!ch.event) (void)hipEventRecord(t.start, on);
        ch.event = nullptr;
        timed.push_back(t);
    }
    // Returns the event recorded behind the launch (nullptr with timing off).
    hipEvent_t end_timed(hipStream_t on) {
        if (!timing || timed.empty()) return nullptr;
        (void)hipEventRecord(timed.back().stop, on);
        chain_of(on).event = timed.back().stop;
        return timed.back().stop;
    }
    // Requires the stream to be idle (called after a synchronize).
    void collect_times() {
        for (const TimedLaunch& t : timed) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
                times.milliseconds[t.kernel] += ms;
                times.launches[t.kernel] += 1;
            }
        }
        timed.clear();
        events_used = 0;
        for (Chain& ch : chains) ch.event = nullptr;
    }
};

namespace {

uint32_t grid_for(uint32_t items, uint32_t block, uint32_t max_blocks) {
    uint32_t blocks = (items + block - 1) / block;
    blocks = std::max(1u, std::min(blocks, max_blocks));
    return (blocks + 7u) & ~7u;   // multiple of 8: xcd_chunk() needs every XCD to own the same number of chunks
}

// Items of the exhaustive search (kernels.h "Exhaustive-search items"): every triangle in order, except that a triangle (a, b, c)
// and a LATER one of the same instance and flags that is (a, c, d) in some rotation, with bit-identical shared corners and
// d = a + (c - b) within 1e-6 of the longest edge component, are merged into the parallelogram item (a, b - a, d - a) at the first
// one's place. Only scenes of at most SMALL_SCENE_TRIANGLES triangles -- the ones searched exhaustively by default -- are paired: above
// that an exhaustive search (HIPR_TRACE_VARIANT) stays per triangle and bit-identical to the BVH searches.
constexpr uint32_t PAIRING_LIMIT = SMALL_SCENE_TRIANGLES;
void build_trace_items(const HiprTriangle* triangles, uint32_t count, std::vector<float>& items) {
    auto corner = [&](uint32_t t, int k) -> const float* { const HiprTriangle& tri = triangles[t]; return k % 3 == 0 ? tri.v0 : (k % 3 == 1 ? tri.v1 : tri.v2); };
    auto same = [](const float* p, const float* q) { return std::memcmp(p, q, 12) == 0; };
    auto bits = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    std::vector<bool> merged(count, false);
    items.clear();
    for (uint32_t i = 0; i < count; ++i) {
        if (merged[i]) continue;
        int ra = 0, rb = 0;
        uint32_t partner = UINT32_MAX;
        for (uint32_t j = i + 1; count <= PAIRING_LIMIT && j < count && partner == UINT32_MAX; ++j) {
            if (merged[j] || triangles[j].instance_index != triangles[i].instance_index || triangles[j].flags != triangles[i].flags) continue;
            for (int x = 0; x < 3 && partner == UINT32_MAX; ++x)
                for (int y = 0; y < 3 && partner == UINT32_MAX; ++y) {
                    const float *a = corner(i, x), *b = corner(i, x + 1), *c = corner(i, x + 2), *a2 = corner(j, y), *c2 = corner(j, y + 1), *d = corner(j, y + 2);
                    if (!same(a, a2) || !same(c, c2)) continue;
                    float longest = 0.0f, off = 0.0f;
                    for (int k = 0; k < 3; ++k) {
                        longest = std::fmax(longest, std::fmax(std::fabs(b[k] - a[k]), std::fabs(d[k] - a[k])));
                        off = std::fmax(off, std::fabs(d[k] - (a[k] + (c[k] - b[k]))));
                    }
                    if (off <= 1e-6f * longest) { partner = j; ra = x; rb = y; }
                }
        }
        const float *a = corner(i, ra), *b = corner(i, ra + 1), *d = partner == UINT32_MAX ? corner(i, ra + 2) : corner(partner, rb + 2);
        // stored vertex k of a triangle is corner (k - rotation) mod 3 of its half: u = weight of vertex 1, v = weight of vertex 2
        const uint32_t selectors = partner == UINT32_MAX ? 0u : uint32_t((4 - ra) % 3) | uint32_t((5 - ra) % 3) << 2 | uint32_t((4 - rb) % 3) << 4 | uint32_t((5 - rb) % 3) << 6;
        const float item[16] = {a[0], a[1], a[2], b[0] - a[0], b[1] - a[1], b[2] - a[2], d[0] - a[0], d[1] - a[1], d[2] - a[2],
                                bits(triangles[i].instance_index), bits(triangles[i].primitive_index), bits(triangles[i].flags | (partner == UINT32_MAX ? 0u : HIPR_ITEM_QUAD)),
                                bits(i), bits(partner == UINT32_MAX ? i : partner), bits(partner == UINT32_MAX ? triangles[i].primitive_index : triangles[partner].primitive_index),
                                bits(selectors)};
        items.insert(items.end(), item, item + 16);
        if (partner != UINT32_MAX) merged[partner] = true;
    }
}

constexpr uint32_t WORK_SETS = 256;                                       // launches served before the ring is re-zeroed
constexpr uint32_t WORK_SET_WORDS = TRACE_SHARDS * TRACE_SHARD_STRIDE;   // one claim counter per shard, 64 B apart
constexpr uint32_t WORK_COUNTERS = WORK_SETS * WORK_SET_WORDS;

// Hands out a zeroed work counter for one persistent launch; re-zeroes the ring when it wraps (stream ordered).
// Each pass slot has its own half of the buffer (two passes may be in flight, each re-zeroing its own sets at its start on its own stream).
uint32_t* next_work_counter(HiprContext* c) {
    uint32_t& index = c->pass_slots[c->active_slot].work_index;
    uint32_t* ring = c->work_counters.as<uint32_t>() + size_t(c->active_slot) * WORK_COUNTERS;
    if (index >= WORK_SETS) {   // out of sets inside a pass (the used prefix is re-zeroed at every pass start): drain and re-zero
        for (int g = 0; g < MAX_WAVEFRONTS; ++g) if (c->wavefronts[g].stream) (void)hipStreamSynchronize(c->wavefronts[g].stream);
        (void)hipMemsetAsync(ring, 0, WORK_COUNTERS * sizeof(uint32_t), c->stream);
        (void)hipStreamSynchronize(c->stream);
        index = 0;
    }
    return ring + size_t(index++) * WORK_SET_WORDS;
}

// One persistent launch over the path queue (closest_count != nullptr), the shadow queue (shadow_count != nullptr) or both.
template <int STACK, int MODE, bool INSTRUMENT, bool OVERFLOW>
void launch_persistent(HiprContext* c, const Wavefront& w, const PathState& in, const uint32_t* closest_count, const uint32_t* shadow_count, uint32_t upper_bound, int bucket) {
    int& per_cu = c->persistent_blocks_per_cu[MODE][bucket];
    if (per_cu == 0) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_trace_persistent<STACK, MODE, INSTRUMENT, OVERFLOW>, TRACE_BLOCK, 0) != hipSuccess || blocks <= 0) blocks = 4;
        if (c->trace_log) fprintf(stderr, "[hipr] k_trace_persistent<%d, %d>: occupancy query says %d blocks of %d threads per CU\n", STACK, MODE, blocks, TRACE_BLOCK);
        if (c->blocks_per_cu_override > 0) blocks = c->blocks_per_cu_override;
        per_cu = blocks;
    }
    const uint32_t waves_per_block = TRACE_BLOCK / 64;
    uint32_t grid = uint32_t(c->cu_count) * uint32_t(per_cu);
    grid = std::max(1u, std::min(grid, (upper_bound + 63u) / 64u / waves_per_block + 1u));
    hipLaunchKernelGGL((k_trace_persistent<STACK, MODE, INSTRUMENT, OVERFLOW>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, in, w.hits.as<float4>(), w.shadow_queue(),
                       c->radiance.as<float4>(), closest_count, shadow_count, next_work_counter(c), c->refill_below, c->counters.as<DeviceCounters>());
}

// One persistent launch over the 8-wide tree (wide8_kernels.h). The LDS stack is sized by the tree's height: a ray keeps at most one group per level.
template <int STACK, int MODE, bool INSTRUMENT>
void launch_wide8(HiprContext* c, const Wavefront& w, const PathState& in, const uint32_t* closest_count, const uint32_t* shadow_count, uint32_t upper_bound, int bucket, const uint32_t* sorted = nullptr) {
    int& per_cu = c->wide8_blocks_per_cu[MODE][bucket];
    if (per_cu == 0) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_trace_wide8<STACK, MODE, INSTRUMENT>, TRACE_BLOCK, 0) != hipSuccess || blocks <= 0) blocks = 4;
        if (c->trace_log) fprintf(stderr, "[hipr] k_trace_wide8<%d, %d>: occupancy query says %d blocks of %d threads per CU\n", STACK, MODE, blocks, TRACE_BLOCK);
        if (c->blocks_per_cu_override > 0) blocks = c->blocks_per_cu_override;
        per_cu = blocks;
    }
    const uint32_t waves_per_block = TRACE_BLOCK / 64;
    // pipelined passes: one block slot per CU stays free, so that the other slot's tail launches find room next to this pass's persistent blocks
    uint32_t grid = uint32_t(c->cu_count) * uint32_t(c->pipelining_now && per_cu > 2 ? per_cu - c->pipeline_spare_blocks : per_cu);
    grid = std::max(1u, std::min(grid, (upper_bound + 63u) / 64u / waves_per_block + 1u));
#if HIPR_RAY_SORT
    if constexpr (MODE == TRACE_FUSED && !INSTRUMENT) if (sorted) {     // ray_sort.hip listed the launch's rays
        if (c->all_triangles_opaque && c->lean_trace)
            hipLaunchKernelGGL((k_trace_wide8<STACK, MODE, INSTRUMENT, false, true>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, c->wide8, in, w.hits.as<float4>(), w.shadow_queue(),
                               c->radiance.as<float4>(), closest_count, shadow_count, next_work_counter(c), c->refill_below, c->counters.as<DeviceCounters>(), sorted);
        else
            hipLaunchKernelGGL((k_trace_wide8<STACK, MODE, INSTRUMENT, true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, c->wide8, in, w.hits.as<float4>(), w.shadow_queue(),
                               c->radiance.as<float4>(), closest_count, shadow_count, next_work_counter(c), c->refill_below, c->counters.as<DeviceCounters>(), sorted);
        return;
    }
#endif
    if constexpr (!INSTRUMENT && MODE != TRACE_CLOSEST) if (!sorted && !c->all_triangles_opaque && c->coverage_textures_r8 && c->lean_trace) {     // the coverage sampler for 8-bit single-channel textures only
        hipLaunchKernelGGL((k_trace_wide8<STACK, MODE, INSTRUMENT, true, false, true>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, c->wide8, in, w.hits.as<float4>(), w.shadow_queue(),
                           c->radiance.as<float4>(), closest_count, shadow_count, next_work_counter(c), c->refill_below, c->counters.as<DeviceCounters>());
        return;
    }
    // scenes whose triangles are all statically opaque run the kernel without the coverage code (closest-only launches never reach it anyway)
    if (!INSTRUMENT && MODE != TRACE_CLOSEST && c->all_triangles_opaque && c->lean_trace)
        hipLaunchKernelGGL((k_trace_wide8<STACK, MODE, INSTRUMENT, false>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, c->wide8, in, w.hits.as<float4>(), w.shadow_queue(),
                           c->radiance.as<float4>(), closest_count, shadow_count, next_work_counter(c), c->refill_below, c->counters.as<DeviceCounters>());
    else
    hipLaunchKernelGGL((k_trace_wide8<STACK, MODE, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, c->wide8, in, w.hits.as<float4>(), w.shadow_queue(),
                       c->radiance.as<float4>(), closest_count, shadow_count, next_work_counter(c), c->refill_below, c->counters.as<DeviceCounters>());
}

template <int MODE, bool INSTRUMENT>
void launch_persistent_for_stack(HiprContext* c, const Wavefront& w, const PathState& in, const uint32_t* closest_count, const uint32_t* shadow_count, uint32_t upper_bound, const uint32_t* sorted = nullptr) {
    // LDS stack entries by the worst case of the wide tree. Trees that need up to 32 entries run with 16 in LDS and the rest in a per-lane scratch
    // array (traversals rarely get past 16): 4 KB of LDS per wave instead of 8 lets a sixth wave per SIMD stay resident, and the kernel is bound by
    // the latency of its dependent gathers (atrium, 260 k triangles: 61.0 -> 57.7 ms of trace time per step). Deeper trees (the 10 M triangle
    // atrium) spill often enough that 32 LDS entries + scratch is the faster split (116.7 vs 119.9 ms).
    if (c->use_wide8()) {       // height h: at most h - 1 groups wait on the stack
#if HIPR_WIDE8_LOW_BUCKET
        if (c->wide8_height <= 9u) launch_wide8<8, MODE, INSTRUMENT>(c, w, in, closest_count, shadow_count, upper_bound, 3, sorted);
        else
#endif
        if (c->wide8_height <= uint32_t(WIDE8_STACK_SHALLOW) + 1u) launch_wide8<WIDE8_STACK_SHALLOW, MODE, INSTRUMENT>(c, w, in, closest_count, shadow_count, upper_bound, 0, sorted);
        else if (c->wide8_height <= 17u) launch_wide8<16, MODE, INSTRUMENT>(c, w, in, closest_count, shadow_count, upper_bound, 1, sorted);
        else launch_wide8<32, MODE, INSTRUMENT>(c, w, in, closest_count, shadow_count, upper_bound, 2, sorted);
        return;
    }
#ifndef HIPR_STACK_MID
#define HIPR_STACK_MID 16
#endif
    if (c->wide_stack_entries <= 16) launch_persistent<16, MODE, INSTRUMENT, false>(c, w, in, closest_count, shadow_count, upper_bound, 0);
    else if (c->wide_stack_entries <= 32) launch_persistent<HIPR_STACK_MID, MODE, INSTRUMENT, true>(c, w, in, closest_count, shadow_count, upper_bound, 1);
    else launch_persistent<32, MODE, INSTRUMENT, true>(c, w, in, closest_count, shadow_count, upper_bound, 2);
}

template <bool INSTRUMENT>
void launch_trace_closest(HiprContext* c, const Wavefront& w, const PathState& in, const uint32_t* count_ptr, uint32_t upper_bound) {
    float4* hits = w.hits.as<float4>();
    // Scenes whose whole BVH sits in the L1 / scalar cache (a few dozen nodes) are VALU-issue bound and run fastest with
    // the plain one-ray-per-lane kernel; everything larger wants the persistent kernel (measured: profiles/).
    if (c->use_persistent()) { launch_persistent_for_stack<TRACE_CLOSEST, INSTRUMENT>(c, w, in, count_ptr, nullptr, upper_bound); return; }
    DeviceCounters* dc = c->counters.as<DeviceCounters>();
    if (c->use_exhaustive()) {
        hipLaunchKernelGGL((k_trace_closest_small<INSTRUMENT>), dim3(grid_for(upper_bound, 256, 256u * 16u)), dim3(256), 0, w.stream, c->scene, in, hits, count_ptr, dc);
        return;
    }
    const uint32_t grid = grid_for(upper_bound, TRACE_BLOCK, 256u * 16u);
    switch (c->stack_size) {
    case 16: hipLaunchKernelGGL((k_trace_closest<16, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, in, hits, count_ptr, dc); break;
    case 32: hipLaunchKernelGGL((k_trace_closest<32, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, in, hits, count_ptr, dc); break;
    default: hipLaunchKernelGGL((k_trace_closest<64, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, in, hits, count_ptr, dc); break;
    }
}

template <bool INSTRUMENT>
void launch_trace_shadow(HiprContext* c, const Wavefront& w, const uint32_t* count_ptr, uint32_t upper_bound) {
    if (c->use_persistent()) { launch_persistent_for_stack<TRACE_SHADOW, INSTRUMENT>(c, w, PathState{}, nullptr, count_ptr, upper_bound); return; }
    DeviceCounters* dc = c->counters.as<DeviceCounters>();
    float4* rad = c->radiance.as<float4>();
    ShadowQueue q = w.shadow_queue();
    if (c->use_exhaustive()) {
        hipLaunchKernelGGL((k_trace_shadow_small<INSTRUMENT>), dim3(grid_for(upper_bound, 256, 256u * 16u)), dim3(256), 0, w.stream, c->scene, q, rad, count_ptr, dc);
        return;
    }
    const uint32_t grid = grid_for(upper_bound, TRACE_BLOCK, 256u * 16u);
    switch (c->stack_size) {
    case 16: hipLaunchKernelGGL((k_trace_shadow<16, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, q, rad, count_ptr, dc); break;
    case 32: hipLaunchKernelGGL((k_trace_shadow<32, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, q, rad, count_ptr, dc); break;
    default: hipLaunchKernelGGL((k_trace_shadow<64, INSTRUMENT>), dim3(grid), dim3(TRACE_BLOCK), 0, w.stream, c->scene, q, rad, count_ptr, dc); break;
    }
}

// The closest-hit rays of this bounce and the shadow rays the previous bounce queued, as one persistent launch.
template <bool INSTRUMENT>
void launch_trace_fused(HiprContext* c, const Wavefront& w, const PathState& in, const uint32_t* closest_count, const uint32_t* shadow_count, uint32_t upper_bound) {
    const uint32_t* sorted = nullptr;
#if HIPR_RAY_SORT
    if (!INSTRUMENT && c->coherence_sort && c->use_wide8() && w.sort_order.ptr) {
        const ShadowQueue q = w.shadow_queue();
        hipr::RaySortLaunch a = {w.stream, in.o_tmin, in.d_pdf, q.o_tmax, q.d_slot, closest_count, shadow_count, std::min(upper_bound, 2u * std::max(w.n_slots, 64u)), {}, {},
                                 w.sort_keys[0].as<uint16_t>(), w.sort_keys[1].as<uint16_t>(), w.sort_order.as<uint32_t>(), w.sort_temp.ptr, w.sort_temp.bytes};
        for (int k = 0; k < 3; ++k) { a.grid_min[k] = c->wide8.grid_min[k]; a.cells_per_unit[k] = 16.0f / (c->wide8.grid_cell[k] * 2097152.0f); }
        if (hipr::launch_ray_sort(a) == 0) sorted = w.sort_order.as<uint32_t>();
    }
#endif
    launch_persistent_for_stack<TRACE_FUSED, INSTRUMENT>(c, w, in, closest_count, shadow_count, upper_bound, sorted);
}

void launch_shade(HiprContext* c, const Wavefront& w, const HiprCameraState& camera, int cur, uint32_t alive, const uint32_t* in_count, uint32_t* out_counts, uint32_t* zero_pair, bool camera_rays) {
    // The rays of the bounce listed by kind (kernels.h k_classify_hits): the shade kernel's batches then hold surface hits only, or none.
    const uint32_t* order = nullptr;
    const unsigned long long* listed = nullptr;
    // Pays where a good share of a bounce's rays did not hit a surface (the atrium's open roof: shade 23.3 -> 21.4 ms per step) and costs a pass over the hits
    // where nearly all did (the closed Cornell box: +7 %): on for the scenes of the persistent kernels, which are the large ones.
    // A bounce of a few thousand paths runs one wave per SIMD at most: the order they are taken in changes nothing, the listing pass would cost a launch.
    uint32_t* taken_words = w.queue_counts.as<uint32_t>() + COUNT_PAIR_STRIDE * COUNT_PAIRS;
    // Camera rays need no listing: a wave of them is one pixel's samples (or an 8 x 8 tile's pixels) -- they hit a surface, or miss, together (HIPR_SHADE_ORDERED_CAMERA=1 lists them anyway).
    if (c->shade_ordered && c->use_persistent() && c->entry == HIPR_ENTRY_PATH_TRACING && alive >= c->shade_ordered_from && (!camera_rays || c->shade_ordered_camera)) {
        unsigned long long* taken = reinterpret_cast<unsigned long long*>(taken_words + COUNT_PAIR_STRIDE * cur);
        if (c->shade_classes && c->any_coated_triangle)
            hipLaunchKernelGGL(k_classify_hits<true>, dim3(grid_for(alive, 256u * CLASSIFY_ROUNDS, uint32_t(c->cu_count) * 8u)), dim3(256), 0, w.stream, w.hits.as<float4>(), in_count, w.order.as<uint32_t>(), taken,
                               c->triangle_class.as<unsigned char>(), w.order_coat.as<uint32_t>());
        else
            hipLaunchKernelGGL(k_classify_hits<false>, dim3(grid_for(alive, 256u * CLASSIFY_ROUNDS, uint32_t(c->cu_count) * 8u)), dim3(256), 0, w.stream, w.hits.as<float4>(), in_count, w.order.as<uint32_t>(), taken,
                               (const unsigned char*)nullptr, w.order_coat.as<uint32_t>());
        order = w.order.as<uint32_t>();
        listed = taken;
    }
    // persistent blocks: three per CU stay resident (3 waves per SIMD), each walks the queue with a grid stride, one batch ahead on its inputs
    // measured: the Default / Transmissive kernels gain from a third wave per SIMD (atrium 29.4 -> 25.9 ms of shading per step), the lighter all-Diffuse
    // kernel loses (Cornell 18 390 -> 17 194 Mrays/s)
    const bool split = c->shade_split && c->entry == HIPR_ENTRY_PATH_TRACING && c->scene.light_count != 0 && w.nee_flags.ptr;
    const uint32_t blocks_per_cu = c->shade_blocks_per_cu > 0 ? uint32_t(c->shade_blocks_per_cu) : (split ? uint32_t(HIPR_SHADE_SPLIT_WAVES) : (c->shading_models == 2 ? 2u : uint32_t(c->shade_unit().waves_per_simd)));
    PathState shaded = w.path_state(cur);
    if (camera_rays) shaded.thr_bounces = nullptr;      // k_generate's queue: throughput 1, no bounce yet -- not stored
    ShadeLaunch a = {grid_for(alive, SHADE_BLOCK, uint32_t(c->cu_count) * blocks_per_cu), w.stream, c->scene, camera, c->frame, c->entry, shaded, w.hits.as<float4>(), order, w.order_coat.as<uint32_t>(), listed, w.path_state(1 - cur),
                     w.shadow_queue(), c->radiance.as<float4>(), in_count, reinterpret_cast<unsigned long long*>(out_counts), reinterpret_cast<unsigned long long*>(zero_pair),
                     reinterpret_cast<unsigned long long*>(taken_words + COUNT_PAIR_STRIDE * (1 - cur)), split ? w.nee_flags.as<unsigned char>() : nullptr,
                     c->counters.as<DeviceCounters>(), c->scene_has_textures || !c->lean_shade, c->scene_has_environment || !c->lean_shade};
    c->shade_unit().shade(c->shading_models, a);
}

// Splits the path slots of a pass (owned tiles x 64 x samples_per_pass) over the wavefronts on a tile (= wave) boundary and sizes the
// queues and the per-sample radiance buffer; small passes stay one wavefront. Buffers only ever grow. The accumulation is not touched.
int partition_path_slots(HiprContext* c) {
    const FrameInfo& fi = c->frame;
    const uint64_t slots = uint64_t(fi.owned_tiles) * 64u * fi.samples_per_pass;
    c->n_slots = uint32_t(slots);
    c->traced_samples = 0;   // the radiance buffer may move and its sample layout changes: nothing traced before can be folded any more
    int r = 0;
    c->partitioned_for = c->wavefronts_wanted(true);      // every caller has put a valid description into c->frame
    c->wavefront_count = int(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(c->partitioned_for), slots / 65536u)));
    // The slots are dealt to the wavefronts in groups of 64 (one wave of camera rays), round robin: with the pixel-major slot order every wavefront then
    // covers the whole frame evenly -- contiguous halves would be the top and the bottom of the image, one of them done with its deep bounces long before
    // the other (material scene: +2.3 % step time with halves). Queue entry i of wavefront g is slot ((i / 64) * G + g) * 64 + i % 64 (k_generate).
    const uint64_t groups = slots / 64u;      // owned_tiles * 64 * samples: a multiple of 64
    for (int g = 0; g < MAX_WAVEFRONTS; ++g) {
        Wavefront& w = c->wavefronts[g];
        w.first_slot = uint32_t(g);           // the wavefront's phase in the deal
        w.n_slots = g >= c->wavefront_count ? 0u : uint32_t((groups + uint64_t(c->wavefront_count) - 1u - uint64_t(g)) / uint64_t(c->wavefront_count) * 64u);
        const bool second_slot = g == 1 && c->wavefront_count == 1 && c->pipeline_passes && c->partitioned_for == 1;    // the other pass slot: a copy of wavefront 0's share
        if (second_slot) { w.first_slot = c->wavefronts[0].first_slot; w.n_slots = c->wavefronts[0].n_slots; }
        const size_t bytes = size_t(std::max(w.n_slots, 64u)) * 16;
        if (g >= c->wavefront_count && !second_slot) {   // queues of wavefronts this pass size does not use go back to the allocator
            for (auto& buffers : w.path) for (DeviceBuffer& b : buffers) b.release();
            w.hits.release(); w.order.release(); w.order_coat.release(); w.nee_flags.release();
            w.sort_keys[0].release(); w.sort_keys[1].release(); w.sort_order.release(); w.sort_temp.release();
            for (DeviceBuffer& b : w.shadow) b.release();
            continue;
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) r |= w.path[i][j].resize(j == 3 ? bytes / 2 : bytes);      // [3]: the 8-byte meta records
        r |= w.hits.resize(bytes);
        r |= w.order.resize(bytes / 4);
        r |= w.order_coat.resize(bytes / 4);
        if (c->shade_split) r |= w.nee_flags.resize(bytes / 16);
#if HIPR_RAY_SORT
        if (c->coherence_sort) {      // both queues of a fused launch: 2 x n_slots entries
            const size_t entries = bytes / 16 * 2;
            r |= w.sort_keys[0].resize(entries * 2); r |= w.sort_keys[1].resize(entries * 2); r |= w.sort_order.resize(entries * 4);
            r |= w.sort_temp.resize(hipr::ray_sort_temp_bytes(uint32_t(entries)));
        }
#endif
        for (int j = 0; j < 3; ++j) r |= w.shadow[j].resize(bytes);
    }
    const bool two_slots = c->wavefront_count == 1 && c->pipeline_passes && c->partitioned_for == 1;
    if (!two_slots) {
        if (c->traced_slot == 1 && c->radiance_other.ptr) std::swap(c->radiance, c->radiance_other);      // the slots are gone: `radiance` is wavefront 0's again
        c->radiance_other.release();
        c->traced_slot = c->next_slot = c->active_slot = 0;
    }
    r |= c->radiance.resize(slots * 16);
    if (two_slots) r |= c->radiance_other.resize(slots * 16);
    return r;
}

void set_frame_divisors(FrameInfo& f) { f.by_samples_per_pass = make_divisor(f.samples_per_pass); f.by_tiles_x = make_divisor(f.tiles_x); }

int check_context(HiprContext* c) {
    if (!c) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return HIPR_OK;
}

// Bounce k of a wavefront = trace stage + shade(k). The trace stage serves the closest-hit rays of bounce k AND the shadow rays
// shade(k - 1) queued (they are independent; radiance slots are still updated in the order shade(0), shadow(0), shade(1), ...):
// one fused persistent launch for big scenes, two plain launches otherwise. All queue sizes stay on the device; the host only
// needs them to stop. `bound`: an upper bound of the bounce's rays (sizes the launches); `host_sizes`: where the sizes the bounce
// produced -- {paths that continue, shadow rays queued} -- are read back to (nullptr: the wavefront's regular pair of the bounce's parity).
int enqueue_bounce(HiprContext* c, Wavefront& w, const HiprCameraState& camera, uint32_t k, uint32_t bound, uint32_t* host_sizes = nullptr) {
    const bool fused = c->use_persistent();
    if (!host_sizes) host_sizes = w.host_counts + 2 * (k & 1u);

    // Queue sizes live in three 8-byte pairs, 64 B apart (COUNT_PAIRS): pair[k % 3] = {paths of bounce k, shadow rays shade(k - 1) queued}.
    // shade(k) fills pair[(k + 1) % 3] with one 64-bit atomic per block and zeroes pair[(k + 2) % 3] for shade(k + 1).
    uint32_t* counts = w.queue_counts.as<uint32_t>();
    const int parity = int(k & 1u);
    uint32_t* in_count = counts + COUNT_PAIR_STRIDE * (k % COUNT_PAIRS);
    uint32_t* shadow_in = in_count + 1;
    uint32_t* out_count = counts + COUNT_PAIR_STRIDE * ((k + 1u) % COUNT_PAIRS);
    uint32_t* zero_pair = counts + COUNT_PAIR_STRIDE * ((k + 2u) % COUNT_PAIRS);
    const size_t first_timed = c->timed.size();
    // a bounce queued blindly (pipelined passes) is not behind the host's read of the sizes of bounce k - 2, which sit in the pair its shade kernel zeroes
    if (host_sizes != w.host_counts + 2 * (k & 1u) && k >= 2) {
        HIP_TRY(hipStreamWaitEvent(w.stream, w.counts_copied[parity], 0));
        c->break_chain(w.stream);
    }

    c->begin_timed(HIPR_KERNEL_TRACE_CLOSEST, w.stream);
    if (k > 0 && fused) {
        if (c->instrument) launch_trace_fused<true>(c, w, w.path_state(parity), in_count, shadow_in, 2u * bound);
        else launch_trace_fused<false>(c, w, w.path_state(parity), in_count, shadow_in, 2u * bound);
    } else {
        if (c->instrument) launch_trace_closest<true>(c, w, w.path_state(parity), in_count, bound);
        else launch_trace_closest<false>(c, w, w.path_state(parity), in_count, bound);
    }
    c->end_timed(w.stream);
    if (k > 0 && !fused) {
        c->begin_timed(HIPR_KERNEL_TRACE_SHADOW, w.stream);
        if (c->instrument) launch_trace_shadow<true>(c, w, shadow_in, bound);
        else launch_trace_shadow<false>(c, w, shadow_in, bound);
        c->end_timed(w.stream);
    }

    c->begin_timed(HIPR_KERNEL_SHADE, w.stream);
    launch_shade(c, w, camera, parity, bound, in_count, out_count, zero_pair, k == 0);
    hipEvent_t shaded = c->end_timed(w.stream);
    if (!shaded) {
        shaded = w.shade_done[parity];
        HIP_TRY(hipEventRecord(shaded, w.stream));
    }
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, shaded, 0));
    HIP_TRY(hipMemcpyAsync(host_sizes, out_count, 8, hipMemcpyDeviceToHost, c->copy_stream));     // {paths that continue, shadow rays}
    HIP_TRY(hipEventRecord(w.counts_copied[parity], c->copy_stream));

    if (c->instrument && c->trace_log) {   // diagnostic (HIPR_TRACE_LOG=1): per-bounce counters and kernel times; serialises the pass
        HIP_TRY(hipStreamSynchronize(w.stream));
        HIP_TRY(hipStreamSynchronize(c->copy_stream));
        DeviceCounters dc;
        HIP_TRY(hipMemcpy(&dc, c->counters.ptr, sizeof(dc), hipMemcpyDeviceToHost));
        const DeviceCounters& p = c->trace_log_previous;
        fprintf(stderr, "[hipr] wavefront %d bounce %u: <= %u closest rays: nodes %llu tris %llu | shadow rays of the previous bounce: nodes %llu tris %llu | kernels",
                int(&w - c->wavefronts), k, bound, dc.closest_nodes - p.closest_nodes, dc.closest_triangles - p.closest_triangles, dc.shadow_nodes - p.shadow_nodes,
                dc.shadow_triangles - p.shadow_triangles);
        for (size_t i = first_timed; i < c->timed.size(); ++i) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->timed[i].start, c->timed[i].stop);
            fprintf(stderr, " %s %.1f us", c->timed[i].kernel == HIPR_KERNEL_SHADE ? "shade" : (c->timed[i].kernel == HIPR_KERNEL_TRACE_SHADOW ? "shadow" : "trace"), ms * 1e3f);
        }
        fprintf(stderr, " -> %u paths continue, %u shadow rays\n", w.host_counts[2 * parity], w.host_counts[2 * parity + 1]);
        if (dc.node_iterations != p.node_iterations) {
            const double ni = double(dc.node_iterations - p.node_iterations), ti = double(dc.triangle_iterations - p.triangle_iterations);
            fprintf(stderr, "[hipr]     wave iterations: %.0f node (%.1f lanes working) + %.0f triangle (%.1f lanes working); %.1f lanes busy on average; %llu refills\n", ni,
                    double(dc.node_lanes - p.node_lanes) / ni, ti, ti > 0 ? double(dc.triangle_lanes - p.triangle_lanes) / ti : 0.0,
                    double(dc.busy_lanes - p.busy_lanes) / (ni + ti), dc.refills - p.refills);
            const double pushes = double(dc.pushes - p.pushes);
            if (pushes > 0)
                fprintf(stderr, "[hipr]     stack pushes: %.0f, of which %.3f %% onto entry 16 or deeper and %.3f %% onto entry 24 or deeper\n", pushes,
                        100.0 * double(dc.pushes_past_16 - p.pushes_past_16) / pushes, 100.0 * double(dc.pushes_past_24 - p.pushes_past_24) / pushes);
        }
        c->trace_log_previous = dc;
    }
    return HIPR_OK;
}

// Brings the books of a pass slot up to date: waits for the tail its last pass left on the slot's stream and adds the rays of the tail's bounces to the
// totals (their queue sizes were read back bounce by bounce into pinned memory). Should paths have outlived the blind bounces (hits that keep being
// rejected and retraced do not count as bounces), the pass is finished here bounce by bounce.
int finish_slot(HiprContext* c, int slot_index) {
    HiprContext::PassSlot& slot = c->pass_slots[slot_index];
    if (!slot.pending) return HIPR_OK;
    Wavefront& w = c->wavefronts[slot_index];
    HIP_TRY(hipStreamSynchronize(w.stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    slot.pending = false;
    uint32_t in = slot.alive_at_detach, shadows = 0;
    for (uint32_t t = 0; t <= slot.blind_bounces && in > 0; ++t) {
        const uint32_t* sizes = t == 0 ? w.host_counts + 2 * slot.first_parity : slot.tail_counts + 2 * (t - 1);
        c->total.closest_rays += in;
        c->total.shadow_rays += sizes[1];
        c->total.iterations += 1;
        in = sizes[0];
        shadows = sizes[1];
    }
    if (in == 0 && shadows == 0) return HIPR_OK;
    // the last blind bounce left paths (or only shadow rays): carry on the ordinary way
    const int saved_slot = c->active_slot;
    c->active_slot = slot_index;
    const bool swap = slot_index != c->traced_slot;     // launches write the radiance of the pass being finished
    if (swap) std::swap(c->radiance, c->radiance_other);
    int status = HIPR_OK;
    for (uint32_t k = slot.next_bounce; status == HIPR_OK; ++k) {
        status = enqueue_bounce(c, w, slot.camera, k, std::max(in, 1u));
        if (status == HIPR_OK && hipEventSynchronize(w.counts_copied[k & 1u]) != hipSuccess) status = fail(HIPR_ERROR_HIP, "hipEventSynchronize failed while a pass was being finished");
        if (status != HIPR_OK || in == 0) break;     // in == 0: that bounce only traced the last shadow rays
        c->total.closest_rays += in;
        c->total.shadow_rays += w.host_counts[2 * (k & 1u) + 1];
        c->total.iterations += 1;
        in = w.host_counts[2 * (k & 1u)];
        if (k > 8192) status = fail(HIPR_ERROR_HIP, "wavefront loop did not terminate");
    }
    if (swap) std::swap(c->radiance, c->radiance_other);
    c->active_slot = saved_slot;
    if (status == HIPR_OK) HIP_TRY(hipStreamSynchronize(w.stream));
    return status;
}

// Every pass queued so far is complete on the device and in the books (the entry points that read results, change the frame or the scene, or collect
// timings start here).
int finish_all(HiprContext* c) {
    for (int slot = 0; slot < 2; ++slot)
        if (int s = finish_slot(c, slot)) return s;
    for (int g = 0; g < MAX_WAVEFRONTS; ++g) if (c->wavefronts[g].stream) HIP_TRY(hipStreamSynchronize(c->wavefronts[g].stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    return HIPR_OK;
}

// O(n) check of everything the kernels dereference through an index of the description (the header is public: a bad ID must be an
// error code, not an out-of-bounds read on the GPU). Also derives the worst-case stack need of the wide tree instead of trusting
// the caller's figure. Returns nullptr when the scene is sound, a message otherwise.
const char* validate_scene(const HiprSceneDesc* s, uint32_t& wide_stack_need, char* message, size_t message_size) {
#define INVALID(...) do { snprintf(message, message_size, __VA_ARGS__); return message; } while (0)
    wide_stack_need = 0;
    auto texel_size = [](uint8_t format) -> uint64_t {
        return format == HIPR_TEXEL_R8 ? 1u : (format == HIPR_TEXEL_RGBA8 ? 4u : (format == HIPR_TEXEL_R32F ? 4u : (format == HIPR_TEXEL_RGBA32F ? 16u : 0u)));
    };
    if (s->texture_count && !s->textures) INVALID("texture_count is %u but textures is null", s->texture_count);
    if (s->light_count && !s->lights) INVALID("light_count is %u but lights is null", s->light_count);
    if (s->texel_bytes && !s->texels) INVALID("texel_bytes is %llu but texels is null", (unsigned long long)s->texel_bytes);
    for (uint32_t i = 1; i < s->texture_count; ++i) {   // slot 0 = none
        const HiprTexture& t = s->textures[i];
        const uint64_t size = texel_size(t.format);
        if (size == 0) INVALID("texture %u has the unknown texel format %u", i, unsigned(t.format));
        if (t.width == 0 || t.height == 0) INVALID("texture %u is %u x %u", i, t.width, t.height);
        const uint64_t bytes = uint64_t(t.width) * t.height * size;
        if (t.texel_offset > s->texel_bytes || bytes > s->texel_bytes - t.texel_offset || t.texel_offset % (size == 16 ? 16 : 4) != 0)
            INVALID("texture %u (%u x %u, offset %llu) does not fit the %llu byte texel pool", i, t.width, t.height, (unsigned long long)t.texel_offset, (unsigned long long)s->texel_bytes);
    }
    for (uint32_t i = 0; i < s->material_count; ++i) {
        const HiprMaterial& m = s->materials[i];
        const int32_t ids[4] = {m.tint_roughness_texture_ID, m.roughness_texture_ID, m.metallic_texture_ID, m.coverage_texture_ID};
        for (int32_t id : ids)
            if (id < 0 || (id > 0 && uint32_t(id) >= s->texture_count)) INVALID("material %u references texture %d of %u", i, id, s->texture_count);
        if (m.shading_model > HIPR_SHADING_TRANSMISSIVE) INVALID("material %u has the unknown shading model %u", i, unsigned(m.shading_model));
    }
    const uint32_t primitive_total = s->index_count / 3;
    for (uint32_t i = 0; i < s->instance_count; ++i) {
        const HiprInstance& inst = s->instances[i];
        if (inst.material_index < 0 || uint32_t(inst.material_index) >= s->material_count) INVALID("instance %u references material %d of %u", i, inst.material_index, s->material_count);
        if (inst.index_offset > primitive_total || inst.vertex_offset > s->vertex_count) INVALID("instance %u starts outside the index / vertex pools", i);
    }
    for (uint32_t t = 0; t < s->triangle_count; ++t) {
        const HiprTriangle& tri = s->triangles[t];
        if (tri.instance_index >= s->instance_count) INVALID("triangle %u references instance %u of %u", t, tri.instance_index, s->instance_count);
        const HiprInstance& inst = s->instances[tri.instance_index];
        if (tri.primitive_index >= primitive_total - inst.index_offset) INVALID("triangle %u references primitive %u beyond the index pool", t, tri.primitive_index);
        if (tri.flags & HIPR_TRIANGLE_ONE_SIDED) {      // the traversal may step over hits on its back: only where the hit program would refuse them
            const HiprMaterial& m = s->materials[inst.material_index];
            if ((m.flags & (HIPR_MATERIAL_CUTOUT | HIPR_MATERIAL_THIN_WALLED)) || m.shading_model == HIPR_SHADING_TRANSMISSIVE)
                INVALID("triangle %u is flagged HIPR_TRIANGLE_ONE_SIDED but material %d is thin-walled, a cut-out or transmissive", t, inst.material_index);
        }
        const uint32_t* idx = s->indices + 3 * size_t(inst.index_offset + tri.primitive_index);
        for (int k = 0; k < 3; ++k)
            if (idx[k] >= s->vertex_count - inst.vertex_offset) INVALID("triangle %u: vertex index %u is beyond the vertex pool", t, idx[k]);
    }
    auto leaf_ok = [&](int32_t ref) { const uint32_t code = uint32_t(~ref); return uint64_t(code >> 3) + (code & 7u) + 1u <= s->triangle_count; };
    if (s->triangle_count && s->node_count == 0) INVALID("a scene with triangles needs a BVH");
    for (uint32_t n = 0; n < s->node_count; ++n)
        for (int k = 0; k < 2; ++k) {
            const int32_t ref = s->nodes[n].child[k];
            if (ref >= 0 ? uint32_t(ref) >= s->node_count : !leaf_ok(ref)) INVALID("BVH node %u: child %d is out of range", n, ref);
        }
    if (s->wide_nodes && s->wide_node_count) {
        // need(node) = (children - 1) + max need(child): iterative post-order; a node reached twice or a cycle is an error
        std::vector<uint32_t> need(s->wide_node_count, 0u);
        std::vector<uint8_t> state(s->wide_node_count, 0);   // 0 unseen, 1 open, 2 done
        std::vector<uint32_t> stack = {0u};
        while (!stack.empty()) {
            const uint32_t n = stack.back();
            const HiprWideNode& w = s->wide_nodes[n];
            if (state[n] == 0) {
                state[n] = 1;
                for (int k = 0; k < 4; ++k) {
                    const int32_t ref = w.child[k];
                    if (ref == HIPR_WIDE_EMPTY) continue;
                    if (ref < 0) { if (!leaf_ok(ref)) INVALID("wide BVH node %u: leaf %d is out of range", n, ref); continue; }
                    if (uint32_t(ref) >= s->wide_node_count || state[ref] != 0) INVALID("wide BVH node %u: child %d is out of range or shared", n, ref);
                    stack.push_back(uint32_t(ref));
                }
            } else {
                uint32_t children = 0, below = 0;
                for (int k = 0; k < 4; ++k) {
                    const int32_t ref = w.child[k];
                    if (ref == HIPR_WIDE_EMPTY) continue;
                    ++children;
                    if (ref >= 0) below = std::max(below, need[ref]);
                }
                need[n] = (children ? children - 1u : 0u) + below;
                state[n] = 2;
                stack.pop_back();
            }
        }
        wide_stack_need = need[0];
    }
    return nullptr;
#undef INVALID
}

// The 8-wide tree of a description: every slot reached at most once from the root, child ranges inside the array, leaf records referencing triangles
// of the description, finite grid. Derives the height (the stack need of the traversal). Returns nullptr when sound.
const char* validate_wide8(const HiprSceneDesc* s, uint32_t& height, char* message, size_t message_size) {
#define INVALID(...) do { snprintf(message, message_size, __VA_ARGS__); return message; } while (0)
    height = 0;
    if (!s->wide8_slots || s->wide8_slot_count == 0) return nullptr;
    if (s->wide8_slot_count > 0x1000000u) INVALID("the 8-wide tree has %u slots, more than 2^24", s->wide8_slot_count);
    for (int a = 0; a < 3; ++a)
        if (!(s->wide8_grid_cell[a] > 0.0f) || !std::isfinite(s->wide8_grid_cell[a]) || !std::isfinite(s->wide8_grid_min[a])) INVALID("the 8-wide tree's grid is not finite");
    std::vector<uint8_t> seen(s->wide8_slot_count, 0);
    struct Visit { uint32_t slot, depth; };
    std::vector<Visit> stack = {{0u, 1u}};
    seen[0] = 1;
    while (!stack.empty()) {
        const Visit v = stack.back();
        stack.pop_back();
        height = std::max(height, v.depth);
        const HiprNode8& n = s->wide8_slots[v.slot].node;
        const uint32_t base = n.base_valid & 0xFFFFFFu, valid = n.base_valid >> 24;
        const uint32_t children = uint32_t(__builtin_popcount(valid));
        if (children == 0 || uint64_t(base) + children > s->wide8_slot_count) INVALID("8-wide node in slot %u: children [%u, %u) outside the %u slots", v.slot, base, base + children, s->wide8_slot_count);
        if (n.inner_mask & ~valid) INVALID("8-wide node in slot %u marks an empty position as an inner node", v.slot);
        uint32_t rank = 0;
        for (uint32_t position = 0; position < 8; ++position) {
            if (!(valid >> position & 1u)) continue;
            const uint32_t child = base + rank++;
            if (seen[child]) INVALID("slot %u of the 8-wide tree is reached twice", child);
            seen[child] = 1;
            if (n.inner_mask >> position & 1u) stack.push_back({child, v.depth + 1u});
            else {
                const HiprLeaf8& leaf = s->wide8_slots[child].leaf;
                if (leaf.triangle[0] >= s->triangle_count || (leaf.triangle[1] != HIPR_LEAF8_NONE && leaf.triangle[1] >= s->triangle_count))
                    INVALID("leaf record in slot %u references triangles %u / %u of %u", child, leaf.triangle[0], leaf.triangle[1], s->triangle_count);
                for (int shift = 8; shift < 16; shift += 2)
                    if (((leaf.flags >> shift) & 3u) == 3u) INVALID("leaf record in slot %u has an invalid corner selector", child);
                for (int which = 0; which < 2; ++which) {
                    if (leaf.triangle[which] == HIPR_LEAF8_NONE) continue;
                    const bool one_sided = (s->triangles[leaf.triangle[which]].flags & HIPR_TRIANGLE_ONE_SIDED) != 0;
                    if ((leaf.flags >> (2 + which) & 1u) && !one_sided) INVALID("leaf record in slot %u marks triangle %u one-sided, the triangle is not", child, leaf.triangle[which]);
                }
                if ((leaf.flags & 12u) && !(leaf.facing_margin >= 0.0f && std::isfinite(leaf.facing_margin))) INVALID("leaf record in slot %u has no valid facing margin", child);
            }
        }
    }
    return nullptr;
#undef INVALID
}

// Uploads (or re-uploads, after a refit) the 8-wide tree of a validated description; trees higher than the largest LDS stack are left to the 4-wide kernels.
int upload_wide8(HiprContext* c, const HiprSceneDesc* s, uint32_t height) {
    c->wide8 = {};
    c->wide8_height = 0;
    if (!s->wide8_slots || s->wide8_slot_count == 0 || height > 33u) return HIPR_OK;
    if (int r = c->wide8_slots.upload(s->wide8_slots, size_t(s->wide8_slot_count) * sizeof(HiprSlot8), c->stream)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->wide8.slots = c->wide8_slots.as<uint4>();
    c->wide8.slot_count = s->wide8_slot_count;
    for (int a = 0; a < 3; ++a) { c->wide8.grid_min[a] = s->wide8_grid_min[a]; c->wide8.grid_cell[a] = s->wide8_grid_cell[a]; }
    c->wide8.cull_backfaces = c->cull_backfaces ? 1u : 0u;
    c->wide8_height = height;
    return HIPR_OK;
}

unsigned short to_unorm16(float v) { return (unsigned short)(v * 65535 + 0.5f); }

// Reverse Halton offsets of OR/Renderer.cpp:323-336 (OR/RNG.h:196-231): primes 2, 3, 5, 7, digits d -> p - d, f64 inside.
float reverse_halton(int prime, int i) {
    double h = 0.0, f = 1.0 / double(prime), fct = f;
    while (i > 0) {
        int digit = i % prime;
        h += (digit == 0 ? 0 : prime - digit) * fct;
        i /= prime;
        fct *= f;
    }
    return float(h);
}

// What the kernels read per triangle is derived on the device from the uploaded pools: the shading records (k_build_shade_triangles), the
// vertex + edges form the trace kernels test (k_build_trace_triangles) and, for scenes searched exhaustively, the items (build_trace_items).
// `pools_uploaded`: the call follows an upload of materials, textures and texels (hipr_upload_scene). hipr_update_scene_geometry leaves those pools on the device as they are,
// so the kernel instantiations chosen from them (textures or not, environment code, 8-bit coverage sampler) must keep what the UPLOAD decided (a refit
// description with other materials or texture formats would otherwise switch the kernels over pools that still hold the old data).
int build_derived_geometry(HiprContext* c, const HiprSceneDesc* s, bool pools_uploaded) {
    DeviceScene& d = c->scene;
    if (pools_uploaded) {
        c->uploaded_materials.assign(s->materials, s->materials + s->material_count);
        c->uploaded_light_types.resize(s->light_count);
        for (uint32_t l = 0; l < s->light_count; ++l) c->uploaded_light_types[l] = s->lights[l].flags & HIPR_LIGHT_TYPE_MASK;
    }
    hipStream_t st = c->stream;
    if (s->triangle_count) {   // flatten the per-hit attribute chain into one record per triangle
        if (c->shade_triangles.resize(size_t(s->triangle_count) * SHADE_TRIANGLE_QUADS * sizeof(float4))) return HIPR_ERROR_OUT_OF_MEMORY;
        d.shade_triangles = c->shade_triangles.as<float4>();
        hipLaunchKernelGGL(k_build_shade_triangles, dim3((s->triangle_count + 255) / 256), dim3(256), 0, st, d, c->shade_triangles.as<float4>());
        if (c->trace_triangles.resize(size_t(s->triangle_count) * 3 * sizeof(float4))) return HIPR_ERROR_OUT_OF_MEMORY;
        d.trace_triangles = c->trace_triangles.as<float4>();
        hipLaunchKernelGGL(k_build_trace_triangles, dim3((s->triangle_count + 255) / 256), dim3(256), 0, st, d.triangles, s->triangle_count, c->trace_triangles.as<float4>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    }
    // the listing pass's class of every triangle (k_classify_hits): bit 0 = the material of its instance carries a coat
    if (pools_uploaded) {
    c->scene_has_textures = s->environment != nullptr;
    c->scene_has_environment = s->environment != nullptr;
    for (uint32_t m = 0; m < s->material_count; ++m) {
        const HiprMaterial& material = s->materials[m];
        c->scene_has_textures = c->scene_has_textures || material.tint_roughness_texture_ID || material.roughness_texture_ID || material.metallic_texture_ID || material.coverage_texture_ID;
    }
    for (uint32_t l = 0; l < s->light_count; ++l) c->scene_has_environment = c->scene_has_environment || (s->lights[l].flags & HIPR_LIGHT_TYPE_MASK) == HIPR_LIGHT_PRESAMPLED_ENVIRONMENT;
    for (uint32_t t = 1; t < s->texture_count; ++t)       // float textures take the generic samplers: TEXTURES = 2 as well
        c->scene_has_environment = c->scene_has_environment || s->textures[t].format == HIPR_TEXEL_R32F || s->textures[t].format == HIPR_TEXEL_RGBA32F;
    c->scene_has_textures = c->scene_has_textures || c->scene_has_environment;
    c->coverage_textures_r8 = true;
    for (uint32_t m = 0; m < s->material_count; ++m)
        if (const int32_t id = s->materials[m].coverage_texture_ID)
            c->coverage_textures_r8 = c->coverage_textures_r8 && uint32_t(id) < s->texture_count && s->textures[id].format == HIPR_TEXEL_R8 && !s->textures[id].is_sRGB;
    }
    c->any_coated_triangle = false;
    c->all_triangles_opaque = true;
    for (uint32_t t = 0; t < s->triangle_count; ++t) c->all_triangles_opaque = c->all_triangles_opaque && (s->triangles[t].flags & HIPR_TRIANGLE_OPAQUE) != 0;
    if (s->triangle_count) {
        std::vector<unsigned char> classes(s->triangle_count, 0);
        for (uint32_t t = 0; t < s->triangle_count; ++t) {
            const HiprMaterial& m = c->uploaded_materials[s->instances[s->triangles[t].instance_index].material_index];      // the pool on the device (see uploaded_materials)
            classes[t] = m.coat != 0 ? 1 : 0;
            c->any_coated_triangle = c->any_coated_triangle || classes[t] != 0;
        }
        if (c->triangle_class.upload(classes.data(), classes.size(), st)) return HIPR_ERROR_OUT_OF_MEMORY;
        HIP_TRY(hipStreamSynchronize(st));
    }
    d.trace_items = nullptr;
    d.trace_item_count = 0;
    if (s->triangle_count && (s->triangle_count <= SMALL_SCENE_TRIANGLES || c->trace_variant == HIPR_TRACE_EXHAUSTIVE)) {
        std::vector<float> items;
        build_trace_items(s->triangles, s->triangle_count, items);
        if (c->trace_items.upload(items.data(), items.size() * sizeof(float), st)) return HIPR_ERROR_OUT_OF_MEMORY;
        HIP_TRY(hipStreamSynchronize(st));
        d.trace_items = c->trace_items.as<float4>();
        d.trace_item_count = uint32_t(items.size() / 16);
    }
    return HIPR_OK;
}

} // namespace

extern "C" {

const char* hipr_last_error(void) { return g_last_error.c_str(); }
// Not part of the public header: group.hip hands the message of a member that failed on a worker thread to the calling thread.
void hipr_internal_set_last_error(const char* message) { g_last_error = message ? message : ""; }

int hipr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int hipr_create(int device_id, HiprContext** out_context) {
    if (!out_context) return fail(HIPR_ERROR_INVALID_ARGUMENT, "out_context is null");
    *out_context = nullptr;
    int n = hipr_device_count();
    if (n == 0) return fail(HIPR_ERROR_NO_DEVICE, "no HIP device available");
    if (device_id < 0 || device_id >= n) return fail(HIPR_ERROR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    HiprContext* c = new HiprContext();
    c->device = device_id;
    bool ok = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->pass_start, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->accumulated, hipEventDisableTiming) == hipSuccess &&
              hipHostMalloc((void**)&c->pass_slots[0].tail_counts, 2 * 128 * sizeof(uint32_t)) == hipSuccess && hipHostMalloc((void**)&c->pass_slots[1].tail_counts, 2 * 128 * sizeof(uint32_t)) == hipSuccess;
    c->stream = c->own_stream;
    for (int g = 0; ok && g < MAX_WAVEFRONTS; ++g) {
        Wavefront& w = c->wavefronts[g];
        if (g > 0) ok = ok && hipStreamCreateWithFlags(&w.own_stream, hipStreamNonBlocking) == hipSuccess;
        w.stream = g == 0 ? c->stream : w.own_stream;
        for (int i = 0; i < 2; ++i)
            ok = ok && hipEventCreateWithFlags(&w.shade_done[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&w.counts_copied[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&w.finished, hipEventDisableTiming) == hipSuccess && hipHostMalloc((void**)&w.host_counts, HOST_COUNT_WORDS * sizeof(uint32_t)) == hipSuccess &&
             w.queue_counts.resize(COUNT_LINES * COUNT_PAIR_STRIDE * sizeof(uint32_t)) == 0;
    }
    if (!ok) {
        hipr_destroy(c);
        return fail(HIPR_ERROR_HIP, "stream / event / pinned allocation failed");
    }
    if (c->counters.resize(sizeof(DeviceCounters)) || c->work_counters.resize(2 * WORK_COUNTERS * sizeof(uint32_t))) {
        hipr_destroy(c);
        return HIPR_ERROR_OUT_OF_MEMORY;
    }
    c->pass_slots[0].work_index = c->pass_slots[1].work_index = WORK_SETS;   // forces the first launch of either slot to zero its ring
    hipDeviceProp_t props;
    if (hipGetDeviceProperties(&props, device_id) == hipSuccess && props.multiProcessorCount > 0) c->cu_count = props.multiProcessorCount;
    if (const char* v = getenv("HIPR_TRACE_VARIANT")) c->trace_variant = atoi(v);
    if (const char* v = getenv("HIPR_REFILL_BELOW")) c->refill_below = atoi(v);
    if (const char* v = getenv("HIPR_SHADE_ORDERED_FROM")) c->shade_ordered_from = uint32_t(atoll(v));
    if (const char* v = getenv("HIPR_SHADE_BLOCKS_PER_CU")) c->shade_blocks_per_cu = std::max(1, atoi(v));
    if (const char* v = getenv("HIPR_BLOCKS_PER_CU")) c->blocks_per_cu_override = atoi(v);
    if (const char* v = getenv("HIPR_WAVEFRONTS")) c->wavefront_limit = std::max(0, std::min(MAX_WAVEFRONTS, atoi(v)));
    if (const char* v = getenv("HIPR_TRACE_LOG")) c->trace_log = atoi(v) != 0;
    if (const char* v = getenv("HIPR_SHADE_CLASSES")) c->shade_classes = atoi(v) != 0;
    if (const char* v = getenv("HIPR_COHERENCE_SORT")) c->coherence_sort = HIPR_RAY_SORT && atoi(v) != 0;      // only in a build with the experiment linked in (tools/experiments/ray_sort.hip)
    if (const char* v = getenv("HIPR_LEAN_TRACE")) c->lean_trace = atoi(v) != 0;
    if (const char* v = getenv("HIPR_LEAN_SHADE")) c->lean_shade = atoi(v) != 0;
    if (const char* v = getenv("HIPR_SHADE_ORDERED")) c->shade_ordered = atoi(v) != 0;
    if (const char* v = getenv("HIPR_SHADE_ORDERED_CAMERA")) c->shade_ordered_camera = atoi(v) != 0;
    if (const char* v = getenv("HIPR_SHADE_SPLIT")) c->shade_split = atoi(v) != 0;
    if (const char* v = getenv("HIPR_ARITHMETIC")) c->arithmetic = (v[0] == 'e' || v[0] == 'E' || v[0] == '1') ? HIPR_ARITHMETIC_EXACT : HIPR_ARITHMETIC_FAST;
    if (const char* v = getenv("HIPR_BACKFACE_CULLING")) c->cull_backfaces = atoi(v) != 0;
    if (const char* v = getenv("HIPR_PIPELINE_PASSES")) c->pipeline_passes = atoi(v) != 0;
    if (const char* v = getenv("HIPR_PIPELINE_SPARE_BLOCKS")) c->pipeline_spare_blocks = std::max(0, atoi(v));
    HIP_TRY(hipMemsetAsync(c->counters.ptr, 0, sizeof(DeviceCounters), c->stream));

    float offsets[256 * 4];
    const int primes[4] = {2, 3, 5, 7};
    for (int i = 0; i < 256; ++i)
        for (int d = 0; d < 4; ++d) offsets[4 * i + d] = reverse_halton(primes[d], i);
    if (int s = c->sample_offsets.upload(offsets, sizeof(offsets), c->stream)) { delete c; return s; }
    // Byte-indexed Sobol tables: entry [d][k][b] = XOR of the direction numbers of dimension d + 1 selected by byte k = b.
    std::vector<uint32_t> sobol(SOBOL_TABLE_WORDS);
    for (int d = 0; d < 3; ++d)
        for (int k = 0; k < 4; ++k)
            for (int b = 0; b < 256; ++b) {
                uint32_t v = 0;
                for (int j = 0; j < 8; ++j)
                    if (b & (1 << j)) v ^= SOBOL_DIRECTIONS[d][8 * k + j];
                sobol[(d * 4 + k) * 256 + b] = v;
            }
    if (int s = c->sobol_tables.upload(sobol.data(), sobol.size() * 4, c->stream)) { delete c; return s; }
    if (int finish_status = finish_all(c)) return finish_status;
    c->scene.sobol_tables = c->sobol_tables.as<uint32_t>();
    c->scene.sample_offsets = c->sample_offsets.as<float4>();
    c->scene.next_event_sample_count = 3;   // OR/Renderer.cpp:479
    *out_context = c;
    return HIPR_OK;
}

int hipr_destroy(HiprContext* c) {
    if (!c) return HIPR_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    DeviceBuffer* all[] = {&c->shade_triangles, &c->trace_triangles, &c->trace_items, &c->wide_nodes, &c->wide8_slots, &c->environment_PDF, &c->environment_samples, &c->nodes, &c->triangles, &c->instances, &c->indices, &c->geometry, &c->texcoords, &c->tints, &c->emissions, &c->materials,
                           &c->lights, &c->textures, &c->texels, &c->ggx_rho, &c->dielectric_rho, &c->alpha, &c->sample_offsets, &c->sobol_tables, &c->radiance, &c->radiance_other,
                           &c->accumulation, &c->scratch_accumulation, &c->counters, &c->work_counters, &c->debug_a, &c->debug_b, &c->debug_c};
    for (DeviceBuffer* b : all) b->release();
    for (Wavefront& w : c->wavefronts) w.release();
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->pass_start) (void)hipEventDestroy(c->pass_start);
    if (c->accumulated) (void)hipEventDestroy(c->accumulated);
    for (auto& slot : c->pass_slots) if (slot.tail_counts) (void)hipHostFree(slot.tail_counts);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
    return HIPR_OK;
}

int hipr_set_stream(HiprContext* c, void* hip_stream) {
    if (int s = check_context(c)) return s;
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->wavefronts[0].stream = c->stream;
    return HIPR_OK;
}

int hipr_upload_tables(HiprContext* c, const HiprTables* t) {
    if (int s = check_context(c)) return s;
    if (!t || !t->ggx_with_fresnel_rho || !t->ggx_rho || !t->dielectric_light_rho || !t->dielectric_dense_rho || !t->ggx_alpha_from_max_PDF)
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_upload_tables: null table");
    std::vector<unsigned short> rho(2 * 32 * 32), diel(2 * 32 * 16 * 16), alpha(32 * 32);
    for (int i = 0; i < 32 * 32; ++i) {
        rho[2 * i] = to_unorm16(t->ggx_with_fresnel_rho[i]);   // F0 = 0
        rho[2 * i + 1] = to_unorm16(t->ggx_rho[i]);            // F0 = 1
        alpha[i] = to_unorm16(t->ggx_alpha_from_max_PDF[i]);
    }
    const int per_medium = 16 * 16 * 16;
    for (int i = 0; i < 2 * per_medium; ++i) {
        diel[i] = to_unorm16(t->dielectric_light_rho[i]);
        diel[2 * per_medium + i] = to_unorm16(t->dielectric_dense_rho[i]);
    }
    if (int s = c->ggx_rho.upload(rho.data(), rho.size() * 2, c->stream)) return s;
    if (int s = c->dielectric_rho.upload(diel.data(), diel.size() * 2, c->stream)) return s;
    if (int s = c->alpha.upload(alpha.data(), alpha.size() * 2, c->stream)) return s;
    if (int finish_status = finish_all(c)) return finish_status;
    c->scene.tables = {c->ggx_rho.as<ushort2>(), c->dielectric_rho.as<ushort2>(), c->alpha.as<unsigned short>()};
    c->tables_ready = true;
    return HIPR_OK;
}

int hipr_validate_scene(const HiprSceneDesc* s) {
    if (!s) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_validate_scene: null scene");
    if (s->triangle_count && (!s->nodes || !s->triangles || !s->instances || !s->indices || !s->geometry || !s->materials))
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_validate_scene: missing geometry arrays");
    uint32_t wide_stack_need = 0;
    char invalid[256];
    if (validate_scene(s, wide_stack_need, invalid, sizeof(invalid))) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_validate_scene: %s", invalid);
    uint32_t wide8_height = 0;
    if (validate_wide8(s, wide8_height, invalid, sizeof(invalid))) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_validate_scene: %s", invalid);
    return HIPR_OK;
}

int hipr_upload_scene(HiprContext* c, const HiprSceneDesc* s) {
    if (int st = check_context(c)) return st;
    if (!s) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_upload_scene: null scene");
    if (s->triangle_count && (!s->nodes || !s->triangles || !s->instances || !s->indices || !s->geometry || !s->materials))
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_upload_scene: missing geometry arrays");
    if (s->bvh_max_depth > 64) return fail(HIPR_ERROR_UNSUPPORTED, "BVH depth %u exceeds the 64 entry LDS stack", s->bvh_max_depth);
    uint32_t wide_stack_need = 0;
    char invalid[256];
    if (validate_scene(s, wide_stack_need, invalid, sizeof(invalid))) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_upload_scene: %s", invalid);
    if (wide_stack_need > 32u + uint32_t(TRACE_SPILL_ENTRIES))
        return fail(HIPR_ERROR_UNSUPPORTED, "the wide BVH needs %u stack entries, more than the %u the traversal kernels provide", wide_stack_need, 32u + uint32_t(TRACE_SPILL_ENTRIES));
    uint32_t wide8_height = 0;
    if (validate_wide8(s, wide8_height, invalid, sizeof(invalid))) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_upload_scene: %s", invalid);
    for (uint32_t i = 0; i < s->instance_count; ++i)
        if ((s->instances[i].mesh_flags & HIPR_MESH_TEXCOORDS && !s->texcoords) || (s->instances[i].mesh_flags & HIPR_MESH_TINTS && !s->tints) ||
            (s->instances[i].mesh_flags & HIPR_MESH_EMISSIVE && !s->emissions))
            return fail(HIPR_ERROR_INVALID_ARGUMENT, "instance %u flags an attribute whose pool is null", i);
    if (int finish_status = finish_all(c)) return finish_status;
    hipStream_t st = c->stream;
    int r = 0;
    r |= c->nodes.upload(s->nodes, size_t(s->node_count) * sizeof(HiprBvhNode), st);
    if (s->wide_nodes && s->wide_node_count) r |= c->wide_nodes.upload(s->wide_nodes, size_t(s->wide_node_count) * sizeof(HiprWideNode), st);
    r |= c->triangles.upload(s->triangles, size_t(s->triangle_count) * sizeof(HiprTriangle), st);
    r |= c->instances.upload(s->instances, size_t(s->instance_count) * sizeof(HiprInstance), st);
    r |= c->indices.upload(s->indices, size_t(s->index_count) * 4, st);
    r |= c->geometry.upload(s->geometry, size_t(s->vertex_count) * sizeof(HiprVertexGeometry), st);
    if (s->texcoords) r |= c->texcoords.upload(s->texcoords, size_t(s->vertex_count) * 8, st);
    if (s->tints) r |= c->tints.upload(s->tints, size_t(s->vertex_count) * 4, st);
    if (s->emissions) r |= c->emissions.upload(s->emissions, size_t(s->vertex_count) * 12, st);
    r |= c->materials.upload(s->materials, size_t(s->material_count) * sizeof(HiprMaterial), st);
    r |= c->lights.upload(s->lights, size_t(s->light_count) * sizeof(HiprLight), st);
    r |= c->textures.upload(s->textures, size_t(s->texture_count) * sizeof(HiprTexture), st);
    r |= c->texels.upload(s->texels, s->texel_bytes, st);
    const HiprEnvironment* env = s->environment;
    if (env) {
        if (env->environment_map_ID <= 0 || uint32_t(env->environment_map_ID) >= s->texture_count || !env->per_pixel_PDF || !env->samples || env->sample_count == 0 ||
            env->pdf_width == 0 || env->pdf_height == 0)
            return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_upload_scene: incomplete environment description");
        const uint8_t format = s->textures[env->environment_map_ID].format;
        if (format != HIPR_TEXEL_RGBA8 && format != HIPR_TEXEL_RGBA32F)
            return fail(HIPR_ERROR_UNSUPPORTED, "only environments with 4 channels are supported (OptiXRenderer/Renderer.cpp:1141-1158)");
        r |= c->environment_PDF.upload(env->per_pixel_PDF, size_t(env->pdf_width) * env->pdf_height * sizeof(float), st);
        r |= c->environment_samples.upload(env->samples, size_t(env->sample_count) * sizeof(HiprLightSample), st);
    }
    if (r) return r < 0 ? r : HIPR_ERROR_HIP;
    HIP_TRY(hipStreamSynchronize(st));
    DeviceScene& d = c->scene;
    d.nodes = c->nodes.as<float4>();
    d.wide_nodes = s->wide_nodes && s->wide_node_count ? c->wide_nodes.as<uint4>() : nullptr;
    d.wide_node_count = s->wide_nodes ? s->wide_node_count : 0u;
    c->wide_stack_entries = wide_stack_need;   // derived from the tree (validate_scene), not taken from the caller
    d.triangles = c->triangles.as<float4>();
    d.instances = c->instances.as<HiprInstance>();
    d.indices = c->indices.as<uint32_t>();
    d.geometry = c->geometry.as<float4>();
    d.texcoords = c->texcoords.as<float2>();
    d.tints = c->tints.as<uint32_t>();
    d.emissions = c->emissions.as<float>();
    d.materials = c->materials.as<HiprMaterial>();
    d.lights = c->lights.as<HiprLight>();
    d.textures = c->textures.as<HiprTexture>();
    d.texels = c->texels.as<uint8_t>();
    d.env_map_ID = env ? env->environment_map_ID : 0;
    d.env_per_pixel_PDF = env ? c->environment_PDF.as<float>() : nullptr;
    d.env_samples = env ? c->environment_samples.as<float4>() : nullptr;
    d.env_pdf_width = env ? env->pdf_width : 0u; d.env_pdf_height = env ? env->pdf_height : 0u; d.env_sample_count = env ? env->sample_count : 0u;
    d.node_count = s->node_count;
    d.triangle_count = s->triangle_count;
    d.light_count = s->light_count;
    if (int status = build_derived_geometry(c, s, true)) return status;
    if (int status = upload_wide8(c, s, wide8_height)) return status;
    c->choose_variant();
    c->stack_size = s->bvh_max_depth <= 16 ? 16 : (s->bvh_max_depth <= 32 ? 32 : 64);
    int models = 0;
    for (uint32_t i = 0; i < s->instance_count; ++i) {
        const int32_t m = s->instances[i].material_index;
        if (m < 0 || uint32_t(m) >= s->material_count) return fail(HIPR_ERROR_INVALID_ARGUMENT, "instance %u references material %d of %u", i, m, s->material_count);
        models |= 1 << std::min<int>(s->materials[m].shading_model, 2);
    }
    c->shading_models = models ? models : 7;
    c->uploaded_instance_bytes = size_t(s->instance_count) * sizeof(HiprInstance);
    c->uploaded_material_count = s->material_count; c->uploaded_texture_count = s->texture_count; c->uploaded_vertex_count = s->vertex_count;
    c->uploaded_index_count = s->index_count; c->uploaded_texel_bytes = s->texel_bytes;
    c->scene_ready = true;
    return HIPR_OK;
}

int hipr_update_scene_geometry(HiprContext* c, const HiprSceneDesc* s) {
    if (int st = check_context(c)) return st;
    if (!s) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: null scene");
    if (!c->scene_ready) return fail(HIPR_ERROR_NOT_READY, "hipr_update_scene_geometry: no scene uploaded");
    if (s->triangle_count && (!s->nodes || !s->triangles || !s->instances || !s->indices || !s->geometry || !s->materials))
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: missing geometry arrays");
    // Meshes, materials and textures stay on the device as uploaded: the description is validated against ITS pools below, so those must be the
    // uploaded ones in size -- an index that is in range for a larger pool of the description would read out of bounds on the device.
    if (s->material_count != c->uploaded_material_count || s->texture_count != c->uploaded_texture_count || s->vertex_count != c->uploaded_vertex_count ||
        s->index_count != c->uploaded_index_count || s->texel_bytes != c->uploaded_texel_bytes)
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: material, texture, vertex, index and texel pool sizes must equal the uploaded scene's (those pools are not re-uploaded)");
    const DeviceScene& d = c->scene;
    if (s->node_count != d.node_count || s->triangle_count != d.triangle_count || (s->wide_nodes ? s->wide_node_count : 0u) != d.wide_node_count || s->light_count != d.light_count ||
        size_t(s->instance_count) * sizeof(HiprInstance) != c->uploaded_instance_bytes)
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: node, triangle, instance and light counts must equal the uploaded scene's (a refit keeps the topology)");
    uint32_t wide_stack_need = 0;
    char invalid[256];
    if (validate_scene(s, wide_stack_need, invalid, sizeof(invalid))) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: %s", invalid);
    if (wide_stack_need != c->wide_stack_entries) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: the wide BVH's topology changed");
    uint32_t wide8_height = 0;
    if (validate_wide8(s, wide8_height, invalid, sizeof(invalid))) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: %s", invalid);
    if (c->wide8.slot_count && (s->wide8_slot_count != c->wide8.slot_count || wide8_height != c->wide8_height))
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: the 8-wide tree's topology changed");
    // The lights ARE uploaded again (they move), but the kernels were instantiated for the kinds of light the upload brought -- environment code or not -- and the
    // environment's own data is not part of a refit: a light may move, not change its type.
    for (uint32_t l = 0; l < s->light_count; ++l)
        if ((s->lights[l].flags & HIPR_LIGHT_TYPE_MASK) != c->uploaded_light_types[l])
            return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_update_scene_geometry: light %u changes its type (%u -> %u); upload the scene instead", l, c->uploaded_light_types[l], s->lights[l].flags & HIPR_LIGHT_TYPE_MASK);
    if (int finish_status = finish_all(c)) return finish_status;      // pipelined passes included: no pending pass may go on over the new geometry
    hipStream_t st = c->stream;
    int r = 0;
    r |= c->nodes.upload(s->nodes, size_t(s->node_count) * sizeof(HiprBvhNode), st);
    if (d.wide_node_count) r |= c->wide_nodes.upload(s->wide_nodes, size_t(s->wide_node_count) * sizeof(HiprWideNode), st);
    r |= c->triangles.upload(s->triangles, size_t(s->triangle_count) * sizeof(HiprTriangle), st);
    r |= c->instances.upload(s->instances, size_t(s->instance_count) * sizeof(HiprInstance), st);
    r |= c->lights.upload(s->lights, size_t(s->light_count) * sizeof(HiprLight), st);
    if (r) return r < 0 ? r : HIPR_ERROR_HIP;
    HIP_TRY(hipStreamSynchronize(st));
    if (c->wide8.slot_count)
        if (int status = upload_wide8(c, s, wide8_height)) return status;
    int models = 0;   // the instances were re-uploaded: a changed material_index may reference another shading model -- of the material pool the DEVICE holds
    for (uint32_t i = 0; i < s->instance_count; ++i) models |= 1 << std::min<int>(c->uploaded_materials[s->instances[i].material_index].shading_model, 2);
    c->shading_models = models ? models : 7;
    return build_derived_geometry(c, s, false);
}

int hipr_set_scene_state(HiprContext* c, const HiprSceneState* state) {
    if (int s = check_context(c)) return s;
    if (!state) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null scene state");
    for (int i = 0; i < 3; ++i) c->scene.env_tint[i] = state->environment_tint[i];
    c->scene.next_event_sample_count = std::min(std::max(state->next_event_sample_count, 0), 256);
    return HIPR_OK;
}

int hipr_set_frame(HiprContext* c, const HiprFrameDesc* f) {
    if (int s = check_context(c)) return s;
    if (!f || f->width == 0 || f->height == 0 || f->tile_stride == 0 || f->tile_phase >= f->tile_stride || f->samples_per_pass == 0)
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_set_frame: bad frame description");
    if (int finish_status = finish_all(c)) return finish_status;
    FrameInfo fi;
    fi.width = f->width; fi.height = f->height;
    fi.tiles_x = (f->width + 7) / 8;
    fi.tiles_total = fi.tiles_x * ((f->height + 7) / 8);
    fi.tile_phase = f->tile_phase; fi.tile_stride = f->tile_stride;
    fi.owned_tiles = (fi.tiles_total + f->tile_stride - 1 - f->tile_phase) / f->tile_stride;
    fi.samples_per_pass = f->samples_per_pass;
    set_frame_divisors(fi);
    const uint64_t slots = uint64_t(fi.owned_tiles) * 64u * fi.samples_per_pass;
    if (slots == 0 || slots > 0x7FFFFFFFull) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_set_frame: %llu path slots per pass", (unsigned long long)slots);
    c->frame = fi;
    int r = partition_path_slots(c);
    const size_t acc_bytes = size_t(fi.owned_tiles) * 64 * sizeof(double4);
    c->accumulation.release();
    c->scratch_accumulation.release();
    c->use_scratch = false;
    r |= c->accumulation.resize(acc_bytes);
    if (r) return HIPR_ERROR_OUT_OF_MEMORY;
    HIP_TRY(hipMemsetAsync(c->accumulation.ptr, 0, acc_bytes, c->stream));
    if (int finish_status = finish_all(c)) return finish_status;
    c->frame_ready = true;
    c->traced_samples = 0;
    return HIPR_OK;
}

int hipr_set_entry_point(HiprContext* c, int entry) {
    if (!c) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null context");
    switch (entry) {
    case HIPR_ENTRY_PATH_TRACING: case HIPR_ENTRY_DEPTH: case HIPR_ENTRY_ALBEDO: case HIPR_ENTRY_TINT: case HIPR_ENTRY_ROUGHNESS:
    case HIPR_ENTRY_SHADING_NORMAL: case HIPR_ENTRY_PRIMITIVE_ID: case HIPR_ENTRY_DENOISER_ALBEDO:
        c->entry = entry;
        return HIPR_OK;
    case 1: case 2:
        return fail(HIPR_ERROR_UNSUPPORTED, "entry point %d is one of the two launches of the reference's AIDenoisedBackend command list; here that backend is a path tracing pass, "
                    "a HIPR_ENTRY_DENOISER_ALBEDO pass and hipr_denoiser_process (include/hipr_denoiser_c.h)", entry);
    }
    return fail(HIPR_ERROR_INVALID_ARGUMENT, "unknown entry point %d", entry);
}

int hipr_use_scratch_accumulation(HiprContext* c, int enable) {
    if (int s = check_context(c)) return s;
    if (!c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "no frame set");
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    if (enable) {
        const size_t bytes = size_t(c->frame.owned_tiles) * 64 * sizeof(double4);
        const bool keep = enable == 2 && c->scratch_accumulation.ptr && c->scratch_accumulation.bytes >= bytes;   // a running mean kept across calls
        if (int s = c->scratch_accumulation.resize(bytes)) return s;
        if (!keep) HIP_TRY(hipMemset(c->scratch_accumulation.ptr, 0, bytes));
    }
    c->use_scratch = enable != 0;
    return HIPR_OK;
}

int hipr_owned_pixel_count(HiprContext* c, uint32_t* out_count) {
    if (!c || !out_count) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null argument");
    if (!c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "no frame set");
    *out_count = c->frame.owned_tiles * 64u;
    return HIPR_OK;
}

int hipr_set_samples_per_pass(HiprContext* c, uint32_t samples_per_pass) {
    if (int s = check_context(c)) return s;
    if (!c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "no frame set");
    const uint64_t slots = uint64_t(c->frame.owned_tiles) * 64u * samples_per_pass;
    if (samples_per_pass == 0 || slots > 0x7FFFFFFFull) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_set_samples_per_pass: %llu path slots per pass", (unsigned long long)slots);
    if (samples_per_pass == c->frame.samples_per_pass) return HIPR_OK;
    if (int finish_status = finish_all(c)) return finish_status;      // the queues may move
    c->collect_times();
    c->frame.samples_per_pass = samples_per_pass;
    set_frame_divisors(c->frame);
    if (partition_path_slots(c)) return HIPR_ERROR_OUT_OF_MEMORY;
    return HIPR_OK;
}

int hipr_render_pass(HiprContext* c, const HiprCameraState* camera, void* out_half4_device, uint32_t out_pitch_pixels, int synchronize) {
    if (int s = hipr_trace_pass(c, camera)) return s;
    return hipr_accumulate_samples(c, 0, c->frame.samples_per_pass, camera->accumulations, out_half4_device, out_pitch_pixels, synchronize);
}

int hipr_trace_pass(HiprContext* c, const HiprCameraState* camera) {
    if (int s = check_context(c)) return s;
    if (!camera) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null camera");
    if (!c->tables_ready || !c->scene_ready || !c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "tables, scene and frame must be set before rendering");
    if (c->partitioned_for != c->wavefronts_wanted()) {      // the scene uploaded since hipr_set_frame wants another split of the path slots
        if (int s = finish_all(c)) return s;
        for (int g = 0; g < MAX_WAVEFRONTS; ++g) if (c->wavefronts[g].stream) HIP_TRY(hipStreamSynchronize(c->wavefronts[g].stream));
        c->collect_times();
        if (partition_path_slots(c)) return HIPR_ERROR_OUT_OF_MEMORY;
    }

    const FrameInfo& f = c->frame;
    const uint32_t n = c->n_slots;
    HiprCounters pass = {};

    // valid pixel-samples (edge tiles may hang over the frame)
    uint64_t valid_pixels = 0;
    if (f.tile_stride == 1) valid_pixels = uint64_t(f.width) * f.height;
    else {
        for (uint32_t t = f.tile_phase; t < f.tiles_total; t += f.tile_stride) {
            uint32_t tx = t % f.tiles_x, ty = t / f.tiles_x;
            uint32_t w = std::min(8u, f.width - tx * 8), h = std::min(8u, f.height - ty * 8);
            valid_pixels += uint64_t(w) * h;
        }
    }
    pass.camera_rays = valid_pixels * f.samples_per_pass;

    // Pipelined passes (HiprContext::PassSlot): this pass takes the slot the pass before the last one used, whose tail has long drained.
    const bool pipelined = c->pipeline_passes && c->use_persistent() && c->wavefront_count == 1 && c->radiance_other.ptr && !c->instrument;
    const int slot = pipelined ? c->next_slot : 0;
    c->pipelining_now = pipelined;
    if (pipelined) {
        if (int s = finish_slot(c, slot)) return s;
        c->next_slot = 1 - slot;
    } else if (int s = finish_all(c)) return s;
    if (slot != c->traced_slot) std::swap(c->radiance, c->radiance_other);      // `radiance` is the buffer of the pass being traced / last traced
    c->traced_slot = slot;
    c->active_slot = slot;
    const int first_wavefront = pipelined ? slot : 0, end_wavefront = pipelined ? slot + 1 : c->wavefront_count;
    hipStream_t lead = c->wavefronts[first_wavefront].stream;

    HiprContext::PassSlot& ps = c->pass_slots[slot];
    if (ps.work_index > 0) {   // claim counters used by the slot's previous pass (the slot's stream is ordered behind it)
        HIP_TRY(hipMemsetAsync(c->work_counters.as<uint32_t>() + size_t(slot) * WORK_COUNTERS, 0, size_t(std::min(ps.work_index, WORK_SETS)) * WORK_SET_WORDS * sizeof(uint32_t), lead));
        ps.work_index = 0;
    }
    if (!pipelined) HIP_TRY(hipEventRecord(c->pass_start, c->stream));

    // Start every wavefront: camera rays + bounce 0.
    uint32_t alive[MAX_WAVEFRONTS] = {}, bounce[MAX_WAVEFRONTS] = {};
    bool running[MAX_WAVEFRONTS] = {};
    for (int g = first_wavefront; g < end_wavefront; ++g) {
        Wavefront& w = c->wavefronts[g];
        if (!pipelined && g > 0) HIP_TRY(hipStreamWaitEvent(w.stream, c->pass_start, 0));
        // the counters at the start of a pass: pair 0 = {paths, 0 shadow rays}, all others zero (pinned image, rewritten only after the syncs of the next pass)
        memset(w.host_counts + 8, 0, COUNT_LINES * COUNT_PAIR_STRIDE * sizeof(uint32_t));
        w.host_counts[8] = w.n_slots;
        HIP_TRY(hipMemcpyAsync(w.queue_counts.as<uint32_t>(), w.host_counts + 8, COUNT_LINES * COUNT_PAIR_STRIDE * sizeof(uint32_t), hipMemcpyHostToDevice, w.stream));
        c->break_chain(w.stream);
        c->begin_timed(HIPR_KERNEL_GENERATE, w.stream);
        hipLaunchKernelGGL(k_generate, dim3((w.n_slots + 255) / 256), dim3(256), 0, w.stream, f, *camera, w.path_state(0), c->radiance.as<float4>(), w.first_slot, uint32_t(pipelined ? 1 : c->wavefront_count), w.n_slots);
        c->end_timed(w.stream);
        alive[g] = w.n_slots;
        running[g] = true;
        if (int s = enqueue_bounce(c, w, *camera, 0, alive[g])) return s;
    }
    // Round robin over the wavefronts: queue the next bounce speculatively (at most `alive` paths continue), then read the sizes
    // the current one produced -- the GPU never idles on the read-back. A wavefront whose paths all ended has, with that last speculative
    // bounce, also traced its last shadow rays.
    pass.closest_rays -= n - uint32_t(std::min<uint64_t>(n, pass.camera_rays));   // dead lanes of partial tiles are queued but never traced
    bool detached = false;
    for (int remaining = end_wavefront - first_wavefront; remaining > 0 && !detached;) {
        for (int g = first_wavefront; g < end_wavefront; ++g) {
            if (!running[g]) continue;
            Wavefront& w = c->wavefronts[g];
            const uint32_t k = bounce[g];
            if (int s = enqueue_bounce(c, w, *camera, k + 1, alive[g])) return s;
            HIP_TRY(hipEventSynchronize(w.counts_copied[k & 1u]));
            pass.closest_rays += alive[g];
            pass.shadow_rays += w.host_counts[2 * (k & 1u) + 1];
            pass.iterations += 1;
            alive[g] = w.host_counts[2 * (k & 1u)];
            bounce[g] = k + 1;
            if (alive[g] == 0) {
                running[g] = false;
                --remaining;
                if (!pipelined && g > 0) {
                    HIP_TRY(hipEventRecord(w.finished, w.stream));
                    HIP_TRY(hipStreamWaitEvent(c->stream, w.finished, 0));
                }
            } else if (pipelined && k >= 1 && uint64_t(alive[g]) * 64u < w.n_slots) {
                // A sliver of the paths is left (bounce k + 1 is queued for them already). Every bounce that can follow is queued now, sized by this
                // count and reading its own on the device: those the camera's bounce limit allows, one more for the last shadow rays, and a reserve
                // for hits that get rejected and retraced without counting as a bounce (finish_slot() takes over if that reserve runs out).
                const uint32_t reserve = 8u;
                const uint32_t blind = std::min(120u, (camera->max_bounce_count + 2u > k ? camera->max_bounce_count + 2u - k : 1u) + reserve);
                for (uint32_t t = 0; t < blind; ++t)
                    if (int s = enqueue_bounce(c, w, *camera, k + 2 + t, alive[g], ps.tail_counts + 2 * t)) return s;
                ps.pending = true;
                ps.alive_at_detach = alive[g];
                ps.first_parity = (k + 1) & 1u;
                ps.blind_bounces = blind;
                ps.next_bounce = k + 2 + blind;
                ps.camera = *camera;
                detached = true;
                break;
            }
            if (k > 4096) return fail(HIPR_ERROR_HIP, "wavefront loop did not terminate");
        }
    }

    c->pass_depth_normalizer = 0.0f;
    if (c->entry == HIPR_ENTRY_DEPTH) {   // max depth = distance between the near and far plane centres (SimpleRGPs.cu:247-255)
        const float* ip = camera->inverse_projection_matrix;
        const float near_z = (ip[8] * 0.0f + ip[9] * 0.0f + ip[10] * -1.0f + ip[11]) / (ip[12] * 0.0f + ip[13] * 0.0f + ip[14] * -1.0f + ip[15]);
        const float far_z = (ip[8] * 0.0f + ip[9] * 0.0f + ip[10] * 1.0f + ip[11]) / (ip[12] * 0.0f + ip[13] * 0.0f + ip[14] * 1.0f + ip[15]);
        c->pass_depth_normalizer = far_z - near_z;
    }
    HIP_TRY(hipGetLastError());
    c->traced_samples = f.samples_per_pass;
    c->total.camera_rays += pass.camera_rays;
    c->total.closest_rays += pass.closest_rays;
    c->total.shadow_rays += pass.shadow_rays;
    c->total.iterations += pass.iterations;
    if (c->instrument || c->timed.size() > 2048) {
        if (int s = finish_all(c)) return s;
        c->collect_times();
    }
    return HIPR_OK;
}

int hipr_accumulate_samples(HiprContext* c, uint32_t first_sample, uint32_t sample_count, uint32_t first_accumulation, void* out_half4_device, uint32_t out_pitch_pixels,
                            int synchronize) {
    if (int s = check_context(c)) return s;
    if (!c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "no frame set");
    if (sample_count == 0 || uint64_t(first_sample) + sample_count > c->traced_samples)
        return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_accumulate_samples: samples [%u, %u) of a traced pass of %u", first_sample, first_sample + sample_count, c->traced_samples);
    if (out_half4_device && c->frame.tile_stride == 1 && out_pitch_pixels < c->frame.width) return fail(HIPR_ERROR_INVALID_ARGUMENT, "output pitch smaller than the frame width");
    const FrameInfo& f = c->frame;
    // The fold runs on the stream of the slot that traced the samples, behind that pass's tail; two consecutive folds may be on different streams and
    // both update the running mean and the frame: the later one waits for the earlier one.
    hipStream_t stream = c->wavefronts[c->traced_slot].stream;
    if (c->accumulated_valid) HIP_TRY(hipStreamWaitEvent(stream, c->accumulated, 0));
    c->break_chain(stream);
    c->begin_timed(HIPR_KERNEL_ACCUMULATE, stream);
    hipLaunchKernelGGL(k_accumulate, dim3((f.owned_tiles * 64 + 255) / 256), dim3(256), 0, stream, f, first_sample, sample_count, first_accumulation, c->radiance.as<float4>(),
                       c->active_accumulation().as<double4>(), static_cast<ushort4*>(out_half4_device), out_pitch_pixels, c->pass_depth_normalizer);
    c->end_timed(stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->accumulated, stream));
    c->accumulated_valid = true;
    if (synchronize || c->instrument || c->timed.size() > 2048) {
        if (int s = finish_all(c)) return s;
        c->collect_times();
    }
    return HIPR_OK;
}

int hipr_synchronize(HiprContext* c) {
    if (int s = check_context(c)) return s;
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    return HIPR_OK;
}

int hipr_get_counters(HiprContext* c, HiprCounters* out) {
    if (int s = check_context(c)) return s;
    if (!out) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null counters");
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    DeviceCounters dc;
    HIP_TRY(hipMemcpy(&dc, c->counters.ptr, sizeof(dc), hipMemcpyDeviceToHost));
    *out = c->total;
    out->shaded_hits = dc.shaded_hits;
    out->closest_nodes = dc.closest_nodes;
    out->closest_triangles = dc.closest_triangles;
    out->shadow_nodes = dc.shadow_nodes;
    out->shadow_triangles = dc.shadow_triangles;
    return HIPR_OK;
}

int hipr_reset_counters(HiprContext* c) {
    if (int s = check_context(c)) return s;
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    HIP_TRY(hipMemset(c->counters.ptr, 0, sizeof(DeviceCounters)));
    c->total = {};
    c->trace_log_previous = {};
    return HIPR_OK;
}

int hipr_set_wavefront_count(HiprContext* c, int count) {
    if (!c) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null context");
    if (count < 0 || count > MAX_WAVEFRONTS) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_set_wavefront_count: %d is outside [0, %d]", count, MAX_WAVEFRONTS);
    c->wavefront_limit = count;   // 0: by scene; takes effect with the next pass
    return HIPR_OK;
}

int hipr_get_wavefront_count(HiprContext* c, int* out_count) {
    if (!c || !out_count) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_get_wavefront_count: null argument");
    // what the NEXT pass runs as: the partition is redone at its start when the rule's answer changed (scene uploaded, limit set since hipr_set_frame)
    const uint64_t slots = uint64_t(c->frame.owned_tiles) * 64u * c->frame.samples_per_pass;
    *out_count = c->frame_ready ? int(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(c->wavefronts_wanted()), slots / 65536u))) : 0;
    return HIPR_OK;
}

int hipr_get_trace_variant(HiprContext* c, int* out_variant) {
    if (!c || !out_variant) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_get_trace_variant: null argument");
    *out_variant = c->scene_ready ? c->active_trace_variant() : HIPR_TRACE_BVH2;
    return HIPR_OK;
}

int hipr_set_trace_variant(HiprContext* c, int variant) {
    if (!c) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null context");
    if (variant < -1 || variant > HIPR_TRACE_WIDE8_PERSISTENT) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_set_trace_variant: unknown variant %d", variant);
    c->trace_variant = variant;     // the exhaustive search's items are built at upload: upload the scene after this call
    return HIPR_OK;
}

int hipr_set_backface_culling(HiprContext* c, int enable) {
    if (int s = check_context(c)) return s;
    if (int s = finish_all(c)) return s;
    c->cull_backfaces = enable != 0;
    c->wide8.cull_backfaces = c->cull_backfaces ? 1u : 0u;
    return HIPR_OK;
}

int hipr_set_arithmetic(HiprContext* c, int arithmetic) {
    if (int s = check_context(c)) return s;
    if (arithmetic != HIPR_ARITHMETIC_FAST && arithmetic != HIPR_ARITHMETIC_EXACT) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_set_arithmetic: unknown mode %d", arithmetic);
    if (int s = finish_all(c)) return s;
    c->arithmetic = arithmetic;
    return HIPR_OK;
}

int hipr_get_arithmetic(HiprContext* c) {
    if (int s = check_context(c)) return s;      // statuses are negative
    return c->arithmetic;
}

int hipr_set_pass_pipelining(HiprContext* c, int enable) {
    if (int s = check_context(c)) return s;
    if (int s = finish_all(c)) return s;
    c->pipeline_passes = enable != 0;
    if (c->frame_ready && partition_path_slots(c)) return HIPR_ERROR_OUT_OF_MEMORY;      // the second slot's queues come and go with the setting
    return HIPR_OK;
}

int hipr_set_instrumentation(HiprContext* c, int count_traversal_steps) {
    if (!c) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null context");
    c->instrument = count_traversal_steps != 0;
    return HIPR_OK;
}

int hipr_reset_timers(HiprContext* c) {
    if (int s = check_context(c)) return s;
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    c->times = {};
    return HIPR_OK;
}

int hipr_get_kernel_times(HiprContext* c, HiprKernelTimes* out) {
    if (int s = check_context(c)) return s;
    if (!out) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null output");
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    *out = c->times;
    return HIPR_OK;
}

int hipr_read_accumulation(HiprContext* c, double* out_rgba, uint64_t capacity_pixels) {
    if (int s = check_context(c)) return s;
    if (!c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "no frame set");
    const FrameInfo& f = c->frame;
    const uint64_t owned = uint64_t(f.owned_tiles) * 64;
    const uint64_t needed = f.tile_stride == 1 ? uint64_t(f.width) * f.height : owned;
    if (!out_rgba || capacity_pixels < needed) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_read_accumulation: need room for %llu pixels", (unsigned long long)needed);
    if (int finish_status = finish_all(c)) return finish_status;
    c->collect_times();
    if (f.tile_stride != 1) {
        HIP_TRY(hipMemcpy(out_rgba, c->active_accumulation().ptr, owned * 32, hipMemcpyDeviceToHost));
        return HIPR_OK;
    }
    std::vector<double> compact(owned * 4);
    HIP_TRY(hipMemcpy(compact.data(), c->active_accumulation().ptr, owned * 32, hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < f.height; ++y)
        for (uint32_t x = 0; x < f.width; ++x) {
            const uint64_t k = (uint64_t(y >> 3) * f.tiles_x + (x >> 3)) * 64 + ((x & 7) + ((y & 7) << 3));
            std::memcpy(out_rgba + 4 * (uint64_t(y) * f.width + x), compact.data() + 4 * k, 32);
        }
    return HIPR_OK;
}

int hipr_scatter_tiles(HiprContext* c, const void* compact, uint64_t rank_stride, uint32_t rank_count, uint32_t width, uint32_t height, void* out,
                       uint32_t out_pitch) {
    if (int s = check_context(c)) return s;
    if (!compact || !out || rank_count == 0 || out_pitch < width) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_scatter_tiles: bad argument");
    hipLaunchKernelGGL(k_scatter_tiles, dim3((width + 15) / 16, (height + 15) / 16), dim3(256), 0, c->stream, static_cast<const ushort4*>(compact),
                       (unsigned long long)rank_stride, rank_count, width, height, static_cast<ushort4*>(out), out_pitch);
    HIP_TRY(hipGetLastError());
    return HIPR_OK;
}

// ------------------------------------------------------------------------------------------- presentation side
int hipr_device_malloc(HiprContext* c, uint64_t bytes, void** out_device_pointer) {
    if (int s = check_context(c)) return s;
    if (!out_device_pointer || bytes == 0) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_device_malloc: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(out_device_pointer, bytes));
    return HIPR_OK;
}

int hipr_device_free(HiprContext* c, void* device_pointer) {
    if (int s = check_context(c)) return s;
    if (!device_pointer) return HIPR_OK;
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipFree(device_pointer));
    return HIPR_OK;
}

int hipr_device_memset(HiprContext* c, void* device_pointer, int byte_value, uint64_t bytes) {
    if (int s = check_context(c)) return s;
    if (!device_pointer) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_device_memset: null pointer");
    HIP_TRY(hipMemsetAsync(device_pointer, byte_value, bytes, c->stream));
    return HIPR_OK;
}

int hipr_copy_to_host(HiprContext* c, void* host, const void* device_pointer, uint64_t bytes) {
    if (int s = check_context(c)) return s;
    if (!host || !device_pointer) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_copy_to_host: bad argument");
    HIP_TRY(hipMemcpyAsync(host, device_pointer, bytes, hipMemcpyDeviceToHost, c->stream));
    if (int finish_status = finish_all(c)) return finish_status;
    return HIPR_OK;
}

int hipr_present_flipped(HiprContext* c, const void* pixels, uint32_t pitch, uint32_t width, uint32_t height, void* backbuffer, uint32_t backbuffer_pitch) {
    if (int s = check_context(c)) return s;
    if (!pixels || !backbuffer || pitch < width || backbuffer_pitch < width) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_present_flipped: bad argument");
    if (width == 0 || height == 0) return HIPR_OK;
    hipLaunchKernelGGL(k_present_flipped, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, c->stream, static_cast<const ushort4*>(pixels), pitch, width,
                       height, static_cast<ushort4*>(backbuffer), backbuffer_pitch);
    HIP_TRY(hipGetLastError());
    return HIPR_OK;
}

// ------------------------------------------------------------------------------------------- debug / parity entry points
int hipr_debug_generate(HiprContext* c, const HiprCameraState* camera, uint32_t accumulation, float* out_origin_tmin, float* out_direction, uint32_t* out_pixel) {
    if (int s = check_context(c)) return s;
    if (!camera || !c->frame_ready) return fail(HIPR_ERROR_NOT_READY, "hipr_debug_generate needs a camera and a frame");
    FrameInfo f = c->frame;
    f.samples_per_pass = 1;
    set_frame_divisors(f);
    const uint32_t n = f.owned_tiles * 64;
    HiprCameraState cam = *camera;
    cam.accumulations = accumulation;
    DeviceBuffer bo, bd, bt, bm, br;
    if (bo.resize(size_t(n) * 16) | bd.resize(size_t(n) * 16) | bt.resize(size_t(n) * 16) | bm.resize(size_t(n) * 16) | br.resize(size_t(n) * 16)) return HIPR_ERROR_OUT_OF_MEMORY;
    const PathState out = {bo.as<float4>(), bd.as<float4>(), bt.as<float4>(), bm.as<uint2>()};
    hipLaunchKernelGGL(k_generate, dim3((n + 255) / 256), dim3(256), 0, c->stream, f, cam, out, br.as<float4>(), 0u, 1u, n);
    if (int finish_status = finish_all(c)) return finish_status;
    if (out_origin_tmin) HIP_TRY(hipMemcpy(out_origin_tmin, bo.ptr, size_t(n) * 16, hipMemcpyDeviceToHost));
    if (out_direction) HIP_TRY(hipMemcpy(out_direction, bd.ptr, size_t(n) * 16, hipMemcpyDeviceToHost));
    bo.release(); bd.release(); bt.release(); bm.release(); br.release();
    if (out_pixel) {
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t tile = (k >> 6) * f.tile_stride + f.tile_phase, lane = k & 63;
            uint32_t x = (tile % f.tiles_x) * 8 + (lane & 7), y = (tile / f.tiles_x) * 8 + (lane >> 3);
            out_pixel[k] = (tile < f.tiles_total && x < f.width && y < f.height) ? (x | (y << 16)) : 0xFFFFFFFFu;
        }
    }
    return HIPR_OK;
}

int hipr_debug_shading(HiprContext* c, int shading_model, const float* params10, const float* wo_n3, const float* in_n3, uint32_t n, int mode, float* out_n7) {
    if (int s = check_context(c)) return s;
    if (!c->tables_ready) return fail(HIPR_ERROR_NOT_READY, "hipr_debug_shading needs the tables");
    if (!params10 || !wo_n3 || !in_n3 || !out_n7 || shading_model < 0 || shading_model > 2 || mode < 0 || mode > 1) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_debug_shading: bad argument");
    if (n == 0) return HIPR_OK;
    DeviceBuffer bp, bw, bi, bo;
    int r = bp.upload(params10, 10 * 4, c->stream) | bw.upload(wo_n3, size_t(n) * 12, c->stream) | bi.upload(in_n3, size_t(n) * 12, c->stream) | bo.resize(size_t(n) * 28);
    if (r) return HIPR_ERROR_OUT_OF_MEMORY;
    c->shade_unit().debug_shading(c->stream, c->scene.tables, shading_model, bp.as<float>(), bw.as<float>(), bi.as<float>(), int(n), mode, bo.as<float>());
    HIP_TRY(hipGetLastError());
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out_n7, bo.ptr, size_t(n) * 28, hipMemcpyDeviceToHost));
    bp.release(); bw.release(); bi.release(); bo.release();
    return HIPR_OK;
}

int hipr_debug_shade(HiprContext* c, const HiprCameraState* camera, uint32_t n, const float* rays_n8, const float* throughput_bounces_n4, const float* hits_n4, const uint32_t* last_triangle,
                     const uint32_t* pixel_hash, const uint32_t* accumulation, float* out_n32) {
    if (int s = check_context(c)) return s;
    if (!c->tables_ready || !c->scene_ready) return fail(HIPR_ERROR_NOT_READY, "hipr_debug_shade needs the tables and a scene");
    if (!camera || !rays_n8 || !throughput_bounces_n4 || !hits_n4 || !last_triangle || !pixel_hash || !accumulation || !out_n32) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_debug_shade: null argument");
    if (n == 0) return HIPR_OK;
    for (uint32_t i = 0; i < n; ++i) {      // the entries name triangles and lights of the uploaded scene
        uint32_t id;
        std::memcpy(&id, hits_n4 + 4 * size_t(i) + 3, 4);
        if (id == HIPR_HIT_MISS) continue;
        if ((id & HIPR_HIT_LIGHT) ? (id & ~HIPR_HIT_LIGHT) >= c->scene.light_count : id >= c->scene.triangle_count) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_debug_shade: entry %u names hit %u outside the scene", i, id);
    }
    DeviceBuffer br, bt, bh, bl, bp, ba, bo;
    if (br.upload(rays_n8, size_t(n) * 32, c->stream) | bt.upload(throughput_bounces_n4, size_t(n) * 16, c->stream) | bh.upload(hits_n4, size_t(n) * 16, c->stream) |
        bl.upload(last_triangle, size_t(n) * 4, c->stream) | bp.upload(pixel_hash, size_t(n) * 4, c->stream) | ba.upload(accumulation, size_t(n) * 4, c->stream) | bo.resize(size_t(n) * 128))
        return HIPR_ERROR_OUT_OF_MEMORY;
    c->shade_unit().debug_shade(c->stream, c->scene, *camera, n, br.as<float4>(), bt.as<float4>(), bh.as<float4>(), bl.as<uint32_t>(), bp.as<uint32_t>(), ba.as<uint32_t>(), bo.as<float>());
    HIP_TRY(hipGetLastError());
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out_n32, bo.ptr, size_t(n) * 128, hipMemcpyDeviceToHost));
    br.release(); bt.release(); bh.release(); bl.release(); bp.release(); ba.release(); bo.release();
    return HIPR_OK;
}

int hipr_debug_math(HiprContext* c, int function, uint32_t n, const float* x, const float* y, float* out) {
    if (int s = check_context(c)) return s;
    if (!x || !out || function < 0 || function > 2 || (function == 2 && !y)) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_debug_math: bad argument");
    if (n == 0) return HIPR_OK;
    DeviceBuffer bx, by, bo;
    if (bx.upload(x, size_t(n) * 4, c->stream) | by.upload(y ? y : x, size_t(n) * 4, c->stream) | bo.resize(size_t(n) * 4)) return HIPR_ERROR_OUT_OF_MEMORY;
    c->shade_unit().debug_math(c->stream, function, int(n), bx.as<float>(), by.as<float>(), bo.as<float>());
    HIP_TRY(hipGetLastError());
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out, bo.ptr, size_t(n) * 4, hipMemcpyDeviceToHost));
    bx.release(); by.release(); bo.release();
    return HIPR_OK;
}

int hipr_debug_light(HiprContext* c, const HiprLight* light, const float* position3, const float* in_n3, uint32_t n, int mode, float* out_n8) {
    if (int s = check_context(c)) return s;
    if (!light || !position3 || !in_n3 || !out_n8 || mode < 0 || mode > 1) return fail(HIPR_ERROR_INVALID_ARGUMENT, "hipr_debug_light: bad argument");
    if (mode == 1 && (light->flags & HIPR_LIGHT_TYPE_MASK) != HIPR_LIGHT_SPOT) return fail(HIPR_ERROR_UNSUPPORTED, "hipr_debug_light: evaluate / pdf by direction is exposed for spot lights only");
    if (n == 0) return HIPR_OK;
    DeviceBuffer bp, bi, bo;
    if (bp.upload(position3, 3 * 4, c->stream) | bi.upload(in_n3, size_t(n) * 12, c->stream) | bo.resize(size_t(n) * 32)) return HIPR_ERROR_OUT_OF_MEMORY;
    c->shade_unit().debug_light(c->stream, *light, bp.as<float>(), bi.as<float>(), int(n), mode, bo.as<float>());
    HIP_TRY(hipGetLastError());
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out_n8, bo.ptr, size_t(n) * 32, hipMemcpyDeviceToHost));
    bp.release(); bi.release(); bo.release();
    return HIPR_OK;
}

int hipr_debug_sobol(HiprContext* c, const uint32_t* triples, uint32_t n, uint32_t* out_uint4) {
    if (int s = check_context(c)) return s;
    if (!triples || !out_uint4) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return HIPR_OK;
    if (int s = c->debug_a.upload(triples, size_t(n) * 12, c->stream)) return s;
    if (int s = c->debug_b.resize(size_t(n) * 16)) return s;
    hipLaunchKernelGGL(k_debug_sobol, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->debug_a.as<uint32_t>(), n, c->debug_b.as<uint32_t>(), c->sobol_tables.as<uint32_t>());
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out_uint4, c->debug_b.ptr, size_t(n) * 16, hipMemcpyDeviceToHost));
    return HIPR_OK;
}

// ---- the VALU roof, measured -----------------------------------------------------------------------------------------------------------------------
#define HIPR_RATE_KERNEL(NAME, INSTR)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, const uint32_t* in, int iterations) {                                \
        uint32_t a0 = in[threadIdx.x & 7], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const uint32_t m = in[8 + (threadIdx.x & 1)], k = in[10 + (threadIdx.x & 1)];                                               \
        for (int it = 0; it < iterations; ++it)                                                                                     \
            asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(k));        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                         \
    }
#define HIPR_RATE_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define HIPR_RATE_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define HIPR_RATE_CVT(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
HIPR_RATE_KERNEL(k_rate_fma, HIPR_RATE_FMA)
HIPR_RATE_KERNEL(k_rate_max, HIPR_RATE_MAX)
HIPR_RATE_KERNEL(k_rate_cvt, HIPR_RATE_CVT)

int hipr_debug_valu_issue_rates(HiprContext* c, double* out3) {
    if (int s = check_context(c)) return s;
    if (!out3) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null argument");
    if (int finish_status = finish_all(c)) return finish_status;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iterations = 4096;      // 8 blocks of 4 waves per CU: 8 waves per SIMD
    const uint32_t seed[12] = {0x3f800000u, 0x3f810000u, 0x3f820000u, 0x3f830000u, 0x3f840000u, 0x3f850000u, 0x3f860000u, 0x3f870000u, 0x3f7fff00u, 0x3f7ffe00u, 0x33800000u, 0x33900000u};
    if (int s = c->debug_a.upload(seed, sizeof(seed), c->stream)) return s;
    if (int s = c->debug_b.resize(size_t(blocks) * threads * 4)) return s;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    typedef void (*RateKernel)(uint32_t*, const uint32_t*, int);
    const RateKernel kernels[3] = {k_rate_fma, k_rate_max, k_rate_cvt};
    for (int which = 0; which < 3; ++which) {
        float best = 1e30f;
        for (int repeat = 0; repeat < 4; ++repeat) {      // the first launch warms the clocks up
            HIP_TRY(hipEventRecord(e0, c->stream));
            hipLaunchKernelGGL(kernels[which], dim3(blocks), dim3(threads), 0, c->stream, c->debug_b.as<uint32_t>(), c->debug_a.as<uint32_t>(), iterations);
            HIP_TRY(hipEventRecord(e1, c->stream));
            HIP_TRY(hipEventSynchronize(e1));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (repeat > 0) best = std::min(best, ms);
        }
        out3[which] = double(blocks) * (threads / 64) * double(iterations) * 8.0 / (double(best) * 1e-3);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return HIPR_OK;
}

int hipr_debug_sample_offsets(HiprContext* c, float* out_256x4) {
    if (int s = check_context(c)) return s;
    if (!out_256x4) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null argument");
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out_256x4, c->sample_offsets.ptr, 256 * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return HIPR_OK;
}

static int debug_prepare_rays(HiprContext* c, const float* rays, uint32_t n, std::vector<float>& o, std::vector<float>& d) {
    o.resize(size_t(n) * 4);
    d.resize(size_t(n) * 4);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(&o[4 * i], rays + 8 * size_t(i), 16);
        std::memcpy(&d[4 * i], rays + 8 * size_t(i) + 4, 16);
    }
    return HIPR_OK;
}

int hipr_debug_trace_closest(HiprContext* c, const float* rays, const uint32_t* skip, uint32_t n, float* out_hits) {
    if (int s = check_context(c)) return s;
    if (!c->scene_ready) return fail(HIPR_ERROR_NOT_READY, "no scene uploaded");
    if (!rays || !out_hits) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return HIPR_OK;
    std::vector<float> o, d;
    debug_prepare_rays(c, rays, n, o, d);
    std::vector<uint32_t> meta(size_t(n) * 2);
    for (uint32_t i = 0; i < n; ++i) { meta[2 * i] = i; meta[2 * i + 1] = skip ? skip[i] : HIPR_NO_TRIANGLE; }
    DeviceBuffer bo, bd, bm, bh, bc;
    int r = bo.upload(o.data(), o.size() * 4, c->stream) | bd.upload(d.data(), d.size() * 4, c->stream) | bm.upload(meta.data(), meta.size() * 4, c->stream) |
            bh.resize(size_t(n) * 16) | bc.upload(&n, 4, c->stream);
    if (r) return HIPR_ERROR_OUT_OF_MEMORY;
    HIP_TRY(hipMemsetAsync(c->counters.ptr, 0, sizeof(DeviceCounters), c->stream));
    PathState in = {bo.as<float4>(), bd.as<float4>(), nullptr, bm.as<uint2>()};
    Wavefront w;   // borrows the buffers above; never released
    w.stream = c->stream;
    w.hits = bh;
    if (c->instrument) launch_trace_closest<true>(c, w, in, bc.as<uint32_t>(), n);
    else launch_trace_closest<false>(c, w, in, bc.as<uint32_t>(), n);
    if (int finish_status = finish_all(c)) return finish_status;
    HIP_TRY(hipMemcpy(out_hits, bh.ptr, size_t(n) * 16, hipMemcpyDeviceToHost));
    c->total = {};
    c->total.closest_rays = n;
    bo.release(); bd.release(); bm.release(); bh.release(); bc.release();
    return HIPR_OK;
}

int hipr_debug_trace_shadow(HiprContext* c, const float* rays, uint32_t n, float* out_transmittance) {
    if (int s = check_context(c)) return s;
    if (!c->scene_ready) return fail(HIPR_ERROR_NOT_READY, "no scene uploaded");
    if (!rays || !out_transmittance) return fail(HIPR_ERROR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return HIPR_OK;
    std::vector<float> o(size_t(n) * 4), d(size_t(n) * 4), rad(size_t(n) * 4, 0.0f), ones(size_t(n) * 4, 1.0f);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(&o[4 * i], rays + 8 * size_t(i), 12);
        o[4 * i + 3] = rays[8 * size_t(i) + 7];               // tmax
        std::memcpy(&d[4 * i], rays + 8 * size_t(i) + 4, 12);
        std::memcpy(&d[4 * i + 3], &i, 4);                    // slot
    }
    DeviceBuffer bo, bd, br, bacc, bc;
    int r = bo.upload(o.data(), o.size() * 4, c->stream) | bd.upload(d.data(), d.size() * 4, c->stream) | br.upload(ones.data(), ones.size() * 4, c->stream) |
            bacc.upload(rad.data(), rad.size() * 4, c->stream) | bc.upload(&n, 4, c->stream);
    if (r) return HIPR_ERROR_OUT_OF_MEMORY;
    HIP_TRY(hipMemsetAsync(c->counters.ptr, 0, sizeof(DeviceCounters), c->stream));
    Wavefront w;   // borrows the buffers above; never released
    w.stream = c->stream;
    w.shadow[0] = bo; w.shadow[1] = bd; w.shadow[2] = br;
    const DeviceBuffer saved_radiance = c->radiance;
    c->radiance = bacc;
    if (c->instrument) launch_trace_shadow<true>(c, w, bc.as<uint32_t>(), n);
    else launch_trace_shadow<false>(c, w, bc.as<uint32_t>(), n);
    c->radiance = saved_radiance;
    if (int finish_status = finish_all(c)) return finish_status;
    std::vector<float> result(size_t(n) * 4);
    HIP_TRY(hipMemcpy(result.data(), bacc.ptr, result.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) out_transmittance[i] = result[4 * i];
    c->total = {};
    c->total.shadow_rays = n;
    bo.release(); bd.release(); br.release(); bacc.release(); bc.release();
    return HIPR_OK;
}

} // extern "C"
