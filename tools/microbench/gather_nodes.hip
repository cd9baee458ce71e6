// Microbenchmark: how fast can a wave fetch one random 64 B BVH node per lane?
//   A  lane gather : every lane issues 4 x dwordx4 to its own node (what k_trace_persistent does)
//   B  quad gather : the 4 lanes of a quad fetch the 4 quarters of ONE node per instruction (4 instructions serve the
//                    quad's 4 nodes); 64 B contiguous per quad and instruction
//   C  quad gather + LDS transpose so that every lane ends up with its own node (the usable form of B)
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_nodes gather_nodes.hip ; run: ./gather_nodes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(128) void k_gather(const float4* nodes, uint32_t node_mask, int iterations, float* out) {
    __shared__ float4 s_tile[MODE == 2 ? 128 * 4 : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t seed = hash(blockIdx.x * 128u + threadIdx.x + 1u);
    float acc = 0.0f;
    for (int it = 0; it < iterations; ++it) {
        seed = hash(seed + it);
        const uint32_t node = seed & node_mask;
        if (MODE == 0) {
            const float4* p = nodes + 4 * size_t(node);
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            const uint32_t j = lane & 3u;
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t leader_node = __shfl(node, (lane & ~3u) + i);
                v[i] = nodes[4 * size_t(leader_node) + j];
            }
            acc += v[0].x + v[1].y + v[2].z + v[3].w;
        } else {
            const uint32_t j = lane & 3u;
            float4* tile = s_tile + wave * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t leader_node = __shfl(node, (lane & ~3u) + i);
                // node of lane (4q + i), quarter j -> row (4q + i), padded so that the reads below spread over the banks
                tile[((lane & ~3u) + i) * 4 + (j ^ ((lane >> 2) & 3u))] = nodes[4 * size_t(leader_node) + j];
            }
            const uint32_t sw = (lane >> 2) & 3u;
            const float4 a = tile[lane * 4 + (0 ^ sw)], b = tile[lane * 4 + (1 ^ sw)], c = tile[lane * 4 + (2 ^ sw)], d = tile[lane * 4 + (3 ^ sw)];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    out[blockIdx.x * 128u + threadIdx.x] = acc;
}

template <int MODE>
double run(const float4* nodes, uint32_t mask, float* out, int blocks, int iterations) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(128), 0, 0, nodes, mask, iterations, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(128), 0, 0, nodes, mask, iterations, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5.0;
}

int main() {
    const int blocks = 256 * 10, iterations = 256;   // 10 blocks of 2 waves per CU = 5 waves per SIMD, like the trace kernel
    float* out;
    CHECK(hipMalloc(&out, size_t(blocks) * 128 * 4));
    for (uint32_t log_nodes : {12u, 14u, 17u, 20u, 23u}) {   // 256 KB, 1 MB, 8 MB, 64 MB, 512 MB of nodes
        const size_t count = size_t(1) << log_nodes;
        float4* nodes;
        CHECK(hipMalloc(&nodes, count * 64));
        CHECK(hipMemset(nodes, 0, count * 64));
        const double fetches = double(blocks) * 128 * iterations;
        const double a = run<0>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double b = run<1>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double c = run<2>(nodes, uint32_t(count - 1), out, blocks, iterations);
        printf("nodes %8zu (%7.1f MB): lane gather %7.3f ms = %6.1f Gnodes/s (%5.0f GB/s) | quad gather %7.3f ms = %6.1f Gnodes/s | quad + LDS transpose %7.3f ms = %6.1f Gnodes/s\n",
               count, count * 64 / 1e6, a, fetches / a / 1e6, fetches * 64 / a / 1e6, b, fetches / b / 1e6, c, fetches / c / 1e6);
        CHECK(hipFree(nodes));
    }
    return 0;
}
