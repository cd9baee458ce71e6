// launch.h -- host-side launch interface between hiprenderer.hip and shade.hip.
#pragma once

#include "kernels.h"

namespace hipr {

struct ShadeLaunch {
    uint32_t grid;
    hipStream_t stream;
    DeviceScene scene;
    HiprCameraState camera;
    int entry;                 // HIPR_ENTRY_*
    PathState in;
    const float4* hits;
    PathState out;
    ShadowQueue shadows;
    float4* radiance;
    const uint32_t* in_count;
    uint32_t* out_count;
    uint32_t* shadow_count;
    DeviceCounters* counters;
};

void launch_shade(int shading_models, const ShadeLaunch& args);

} // namespace hipr
