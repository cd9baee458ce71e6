// launch.h -- host-side launch interface between hiprenderer.hip and shade.hip.
#pragma once

#include "kernels.h"

namespace hipr {

struct ShadeLaunch {
    uint32_t grid;
    hipStream_t stream;
    DeviceScene scene;
    HiprCameraState camera;
    FrameInfo frame;           // path slot -> pixel and sample (path_sample_of_slot)
    int entry;                 // HIPR_ENTRY_*
    PathState in;
    const float4* hits;
    const uint32_t* order;     // k_classify_hits: the queue entries in the order the kernel takes them (surface hits first); nullptr = queue order
    const uint32_t* order_coat;               // the coated surface hits, listed apart
    const unsigned long long* listed;         // k_classify_hits' counters of this bounce: [0] low word = plain surface hits, [1] = coated surface hits
    PathState out;
    ShadowQueue shadows;
    float4* radiance;
    const uint32_t* in_count;
    unsigned long long* out_counts;   // {paths that continue, shadow rays queued}: one 8-byte word so that a block reserves both with one atomic
    unsigned long long* zero_a;       // two 8-byte counters the kernel zeroes for the kernels of the next bounce (nullptr: none)
    unsigned long long* zero_b;
    unsigned char* nee_flags;         // one byte per queue entry: non-null runs the kernel as its two halves (shade_kernel.h SHADE_PART_*)
    DeviceCounters* counters;
    bool textures;                    // the scene holds textures or an environment map (false: k_shade<..., TEXTURES = 0>)
    bool environment;                 // ... an environment map or a presampled environment light (false with textures: TEXTURES = 1)
};


#ifndef HIPR_RAY_SORT
#define HIPR_RAY_SORT 0     // 1: tools/experiments/ray_sort.hip is linked in (tools/build_variant.sh ... "-DHIPR_RAY_SORT=1"); the product is built without (measured -20 %, rocPRIM dependency)
#endif
#if HIPR_RAY_SORT
// tools/experiments/ray_sort.hip: the rays of one fused trace launch listed by (kind, origin cell, direction octant). All pointers are device pointers.
struct RaySortLaunch {
    hipStream_t stream;
    const float4 *closest_o, *closest_d, *shadow_o, *shadow_d;     // the two queues' origins and directions
    const uint32_t *closest_count, *shadow_count;
    uint32_t capacity;                       // entries listed: an upper bound of closest + shadow rays known to the host
    float grid_min[3], cells_per_unit[3];    // 16 cells over the scene's bounds per axis
    uint16_t *keys, *keys_sorted;            // capacity entries each
    uint32_t* order;                         // capacity entries: order[i] = index into the closest queue, or closest count + index into the shadow queue
    void* temp;
    size_t temp_bytes;                       // ray_sort_temp_bytes(capacity) or more
};
size_t ray_sort_temp_bytes(uint32_t capacity);
int launch_ray_sort(const RaySortLaunch& args);      // 0, or the hipError_t of the sort
#endif

// The host entry points of ONE build of shade.hip. The library holds two (shade.hip's header): the fast unit and the exact one; a context launches through
// the unit of its arithmetic mode (hipr_set_arithmetic). All pointers are device pointers; see k_debug_shading etc. in shade.hip.
struct ShadeUnit {
    void (*shade)(int shading_models, const ShadeLaunch& args);
    void (*debug_shade)(hipStream_t stream, const DeviceScene& scene, const HiprCameraState& camera, uint32_t n, const float4* rays, const float4* throughput_bounces, const float4* hits,
                        const uint32_t* last_triangle, const uint32_t* pixel_hash, const uint32_t* accumulation, float* out);
    void (*debug_light)(hipStream_t stream, const HiprLight& light, const float* position3, const float* in_n3, int n, int mode, float* out_n8);
    void (*debug_shading)(hipStream_t stream, const DeviceTables& tables, int model, const float* params10, const float* wo_n3, const float* in_n3, int n, int mode, float* out_n7);
    void (*debug_math)(hipStream_t stream, int function, int n, const float* x, const float* y, float* out);
    int waves_per_simd;     // what this build's k_shade is compiled for (HIPR_SHADE_WAVES): the persistent grid is that many blocks of 256 threads per CU, all resident
};
const ShadeUnit& shade_unit_fast();
const ShadeUnit& shade_unit_exact();

} // namespace hipr
