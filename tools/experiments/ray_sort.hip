// ray_sort.hip -- the coherence pass at the shade -> trace hand-off (opt-in, HIPR_COHERENCE_SORT=1): the rays one fused trace launch serves -- the
// closest-hit rays of bounce k and the shadow rays shade(k - 1) queued -- are listed in the order of a 16-bit key
//     shadow ray ? 1 : 0  |  Morton code of the origin's cell in a 16 x 16 x 16 grid over the scene's bounds (12 bits)  |  direction octant (3 bits)
// and the trace kernel takes them in that order (k_trace_wide8<..., SORTED = true>: queue entry = order[i], results stored where they always are, so
// nothing after the trace kernel sees the order and frames stay bit-identical). The sort is rocPRIM's stable radix sort of (key, index) pairs:
// within a bucket the rays keep the queue's pixel-major order. Nothing to match in the reference: OptiX schedules its rays itself
// (OptiXRenderer/Renderer.cpp:471-476 launches a frame of them); the contract is the oracle's frame and counters.
#define HIPR_SHADE_TU 1      // kernels.h: declarations and types only, its kernels live in hiprenderer.hip
#include "launch.h"

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace hipr {

namespace {

__device__ inline uint32_t spread4(uint32_t v) {     // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

__global__ __launch_bounds__(256) void k_ray_sort_keys(RaySortLaunch a) {
    const uint32_t n_closest = *a.closest_count, n = n_closest + *a.shadow_count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < a.capacity; i += gridDim.x * 256u) {
        uint32_t key = 0xFFFFu;       // entries past the rays: last, never taken
        if (i < n) {
            const bool shadow = i >= n_closest;
            const float4 o = shadow ? a.shadow_o[i - n_closest] : a.closest_o[i];
            const float4 d = shadow ? a.shadow_d[i - n_closest] : a.closest_d[i];
            const float cx = (o.x - a.grid_min[0]) * a.cells_per_unit[0], cy = (o.y - a.grid_min[1]) * a.cells_per_unit[1], cz = (o.z - a.grid_min[2]) * a.cells_per_unit[2];
            const uint32_t ix = uint32_t(fminf(fmaxf(cx, 0.0f), 15.0f)), iy = uint32_t(fminf(fmaxf(cy, 0.0f), 15.0f)), iz = uint32_t(fminf(fmaxf(cz, 0.0f), 15.0f));
            const uint32_t octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
            key = (shadow ? 0x8000u : 0u) | ((spread4(ix) | (spread4(iy) << 1) | (spread4(iz) << 2)) << 3) | octant;
            if (key == 0xFFFFu) key = 0xFFFEu;
        }
        a.keys[i] = uint16_t(key);
    }
}

} // namespace

size_t ray_sort_temp_bytes(uint32_t capacity) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint16_t*)nullptr, (uint16_t*)nullptr, rocprim::counting_iterator<uint32_t>(0u), (uint32_t*)nullptr, size_t(capacity), 0u, 16u);
    return bytes;
}

int launch_ray_sort(const RaySortLaunch& a) {
    if (a.capacity == 0) return 0;
    const uint32_t blocks = std::min<uint32_t>((a.capacity + 255u) / 256u, 256u * 16u);
    hipLaunchKernelGGL(k_ray_sort_keys, dim3(blocks), dim3(256), 0, a.stream, a);
    size_t bytes = a.temp_bytes;
    const hipError_t e = rocprim::radix_sort_pairs(a.temp, bytes, (const uint16_t*)a.keys, a.keys_sorted, rocprim::counting_iterator<uint32_t>(0u), a.order, size_t(a.capacity), 0u, 16u, a.stream);
    return e == hipSuccess ? 0 : int(e);
}

} // namespace hipr
