// oracle/integrator.h -- CPU restatement of the reference's path tracing integrator, structured
// like the reference's megakernel (one loop per pixel), NOT like the wavefront kernels it checks.
// TEST INFRASTRUCTURE ONLY (see oracle/vecmath.h).
//
// Reference files followed (relative to /root/reference/extensions/OptiXRenderer/OptiXRenderer/):
//   Shading/SimpleRGPs.cu:44-140 (camera rays, accumulate, path loop), :349-362 (miss)
//   Shading/MonteCarlo.cu:61-302 (RIS next event estimation, closest hit, shadow any hit, light hit)
//   Shading/TriangleAttributes.cu:35-84 (attribute interpolation)
//   Shading/LightSources/LightSources.cu:31-70 (analytic light intersection)
//   Types.h:389-414 (material texture lookups), Utils.h:67-74, 372-397
//
// Parity status: closest-hit selection, BVH traversal and triangle intersection happen inside
// NVIDIA OptiX 6.5 in the reference (closed, not under /root/reference) -- "parity unpinned" for
// those; this file states the arithmetic the HIP kernels and the oracle agree on.
#pragma once

#include "rng.h"
#include "shading.h"

#include <vector>

namespace oracle {

struct Ray { float3 origin; float tmin; float3 direction; float tmax; };

struct Hit {
    float t, u, v;
    uint32_t id;   // 0xFFFFFFFF miss, 0x80000000 | light index, else global triangle index
};
static const uint32_t HIT_MISS = 0xFFFFFFFFu;
static const uint32_t HIT_LIGHT_BIT = 0x80000000u;

struct TraversalCounters { uint64_t nodes = 0, triangles = 0; };

// --- camera -----------------------------------------------------------------------------------
void generate_camera_ray(const HiprCameraState& cam, int x, int y, int width, int height, uint32_t accumulation,
                         float3& origin, float3& direction);

// --- intersection -----------------------------------------------------------------------------
bool intersect_triangle(const HiprTriangle& tri, float3 o, float3 d, float& t, float& u, float& v);
uint32_t search_item_count(const HiprSceneDesc& scene);
void reset_search_items();   // call at the start of every API entry that may search exhaustively (integrator.cpp search_items)
Hit closest_hit_bruteforce(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip_triangle, TraversalCounters* counters = nullptr);
Hit closest_hit_bvh(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip_triangle, TraversalCounters* counters);
Hit closest_hit_wide(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip_triangle, TraversalCounters* counters);   // compressed 4-wide BVH
Hit closest_hit_wide8(const HiprSceneDesc& scene, const Ray& ray, uint32_t skip_triangle, TraversalCounters* counters);  // compressed 8-wide BVH with leaf records
void intersect_lights(const HiprSceneDesc& scene, const Ray& ray, Hit& hit);
// Shadow any-hit accumulation over all triangles in (tmin, tmax); returns the attenuated radiance.
float3 shadow_bruteforce(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters = nullptr);
float3 shadow_bvh(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters);
float3 shadow_wide(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters);
float3 shadow_wide8(const HiprSceneDesc& scene, const Ray& ray, float3 radiance, TraversalCounters* counters);
int wide8_stack_high_water(bool reset);
void set_backface_culling(bool enable);      // the 8-wide search steps over closest hits on the back of one-sided triangles (default on, like the device)
bool backface_culling();  // diagnostic: deepest traverse_wide8 stack since the last reset

// --- textures / materials ---------------------------------------------------------------------
float4 sample_texture(const HiprSceneDesc& scene, int texture_ID, float2 uv);
int wide_stack_high_water(bool reset);   // diagnostic: deepest traverse_wide stack since the last reset
float material_coverage(const HiprSceneDesc& scene, const HiprMaterial& m, float2 uv);

// --- integrator -------------------------------------------------------------------------------
struct RenderSettings {
    bool use_bvh = true;          // false: brute force over all triangles (tiny scenes)
    bool use_wide = false;        // true: the compressed 4-wide BVH
    bool use_wide8 = false;       // true: the compressed 8-wide BVH with leaf records (what the HIP kernels walk in scenes with more than 64 BVH2 nodes)
};

struct RenderCounters {
    uint64_t camera_rays = 0, closest_rays = 0, shadow_rays = 0, shaded_hits = 0;
    uint64_t rejected_hits = 0;      // closest hits the hit program refused (back side of a one-sided surface, coverage below the drawn number): retraced, MonteCarlo.cu:159-164
    TraversalCounters closest, shadow;
};

// Radiance of one pixel-sample: the lambda of path_tracing_RPG (SimpleRGPs.cu:131-140).
float3 path_trace_pixel(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam,
                        const float4* sample_offsets, int x, int y, int width, int height, uint32_t accumulation,
                        const RenderSettings& settings, RenderCounters* counters);

// The hit programs for one traced ray and its given closest hit (stage-level checker of K3; record layout in integrator.cpp).
void shade_hit_for_test(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam, const float4* sample_offsets, const float* ray8,
                        const float* throughput_bounces4, const float* hit4, uint32_t last_triangle, uint32_t pixel_hash, uint32_t accumulation, float* out32);

// One pixel-sample of an AOV entry point (HIPR_ENTRY_DEPTH ... HIPR_ENTRY_PRIMITIVE_ID), ORS/SimpleRGPs.cu:227-340.
float3 aov_pixel(const HiprSceneDesc& scene, const HiprSceneState& state, const HiprCameraState& cam, const float4* sample_offsets, int x, int y,
                 int width, int height, uint32_t accumulation, int entry, const RenderSettings& settings);

} // namespace oracle
