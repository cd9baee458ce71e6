// Microbenchmark: how fast can a wave fetch one random 64 B BVH node per lane?
//   A  lane gather : every lane issues 4 x dwordx4 to its own node (what k_trace_persistent does)
//   B  quad gather : the 4 lanes of a quad fetch the 4 quarters of ONE node per instruction (4 instructions serve the
//                    quad's 4 nodes); 64 B contiguous per quad and instruction
//   C  quad gather + LDS transpose so that every lane ends up with its own node (the usable form of B)
//   D  quad gather + a 4 x 4 transpose inside the quad through DPP (two butterfly stages of v_cndmask_b32 with a quad_perm source: 32 VALU, no LDS)
//   F  quad gather straight into LDS (global_load_lds_dwordx4: no staging registers, no ds_write), planes padded by 16 B so that the b128 reads of a lane's
//      own node are conflict free; 38 of 64 lanes
//   E  like D, but only `ACTIVE` of the 64 lanes want a node (the trace kernel averages 38 working lanes per node iteration): lane gather with
//      the idle lanes masked against quad gather with the loads of idle quad-mates masked
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
    constexpr int PLANE = 65;   // uint4 per plane: 64 lanes + 16 B of padding
    __shared__ uint4 s_dma[MODE == 6 ? 2 * 4 * PLANE : 1];
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
        } else if (MODE == 6) {
            const bool want = (hash(seed ^ 0x9E3779B9u) & 63u) < 38u;
            const uint32_t j = lane & 3u;
            uint4* tile = s_dma + wave * 4 * PLANE;
            const uint32_t packed = (node << 1) | (want ? 1u : 0u);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t mate = i == 0 ? __builtin_amdgcn_mov_dpp(packed, 0x00, 0xF, 0xF, false) : i == 1 ? __builtin_amdgcn_mov_dpp(packed, 0x55, 0xF, 0xF, false)
                                    : i == 2 ? __builtin_amdgcn_mov_dpp(packed, 0xAA, 0xF, 0xF, false) : __builtin_amdgcn_mov_dpp(packed, 0xFF, 0xF, 0xF, false);
                if (mate & 1u)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const uint4*>(nodes) + 4 * size_t(mate >> 1) + j),
                                                     (__attribute__((address_space(3))) void*)(tile + i * PLANE), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the wave's DMA writes have landed
            if (want) {
                const uint4* mine = tile + (lane & 3u) * PLANE + (lane & ~3u);
                uint32_t x = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const uint4 v = mine[k]; x += v.x ^ (v.y + (v.z ^ v.w)); }
                acc += __uint_as_float(x);
            }
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 3 || MODE == 4 || MODE == 5) {
            const bool want = MODE == 3 || ((hash(seed ^ 0x9E3779B9u) & 63u) < 38u);   // E: 38 of 64 lanes on average, scattered
            const uint32_t j = lane & 3u;
            if (MODE == 4) {   // masked lane gather
                if (want) {
                    const uint4* p = reinterpret_cast<const uint4*>(nodes) + 4 * size_t(node);
                    const uint4 v[4] = {p[0], p[1], p[2], p[3]};
                    uint32_t x = 0; for (int k = 0; k < 4; ++k) x += v[k].x ^ (v[k].y + (v[k].z ^ v[k].w));
                    acc += __uint_as_float(x);
                }
            } else {
                uint4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // node index and request flag of quad-mate i, broadcast inside the quad (one DPP move each)
                    const uint32_t packed = (node << 1) | (want ? 1u : 0u);
                    const uint32_t mate = i == 0 ? __builtin_amdgcn_mov_dpp(packed, 0x00, 0xF, 0xF, false) : i == 1 ? __builtin_amdgcn_mov_dpp(packed, 0x55, 0xF, 0xF, false)
                                        : i == 2 ? __builtin_amdgcn_mov_dpp(packed, 0xAA, 0xF, 0xF, false) : __builtin_amdgcn_mov_dpp(packed, 0xFF, 0xF, 0xF, false);
                    v[i] = make_uint4(0, 0, 0, 0);
                    if (mate & 1u) v[i] = reinterpret_cast<const uint4*>(nodes)[4 * size_t(mate >> 1) + j];
                }
                // lane j holds quarter j of the nodes of mates 0..3; two butterfly stages leave quarter k of the lane's OWN node in register k
                const bool odd = (lane & 1u) != 0, upper = (lane & 2u) != 0;
                uint4 s[4];
#define XCHG1(x) uint32_t(__builtin_amdgcn_mov_dpp(int(x), 0xB1, 0xF, 0xF, false))   /* quad_perm [1, 0, 3, 2] */
#define XCHG2(x) uint32_t(__builtin_amdgcn_mov_dpp(int(x), 0x4E, 0xF, 0xF, false))   /* quad_perm [2, 3, 0, 1] */
#define STAGE(out0, out1, in0, in1, sel, X) \
                out0.x = sel ? X(in1.x) : in0.x; out0.y = sel ? X(in1.y) : in0.y; out0.z = sel ? X(in1.z) : in0.z; out0.w = sel ? X(in1.w) : in0.w; \
                out1.x = sel ? in1.x : X(in0.x); out1.y = sel ? in1.y : X(in0.y); out1.z = sel ? in1.z : X(in0.z); out1.w = sel ? in1.w : X(in0.w);
                STAGE(s[0], s[1], v[0], v[1], odd, XCHG1)
                STAGE(s[2], s[3], v[2], v[3], odd, XCHG1)
                STAGE(v[0], v[2], s[0], s[2], upper, XCHG2)
                STAGE(v[1], v[3], s[1], s[3], upper, XCHG2)
#undef STAGE
#undef XCHG1
#undef XCHG2
                if (want) { uint32_t x = 0; for (int k = 0; k < 4; ++k) x += v[k].x ^ (v[k].y + (v[k].z ^ v[k].w)); acc += __uint_as_float(x); }   // every dword is used: the transpose cannot be pruned
            }
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
    const int blocks = 256 * 12, iterations = 256;   // 12 blocks of 2 waves per CU = 6 waves per SIMD, like the trace kernel
    float* out;
    CHECK(hipMalloc(&out, size_t(blocks) * 128 * 4));
    for (uint32_t log_nodes : {7u, 8u, 9u, 12u, 14u, 16u, 17u, 20u, 23u}) {   // 8 / 16 / 32 KB (inside the 32 KB L1), 256 KB, 1 MB, 4 MB, 8 MB, 64 MB, 512 MB of nodes
        const size_t count = size_t(1) << log_nodes;
        float4* nodes;
        CHECK(hipMalloc(&nodes, count * 64));
        CHECK(hipMemset(nodes, 0, count * 64));
        const double fetches = double(blocks) * 128 * iterations;
        const double a = run<0>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double b = run<1>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double c = run<2>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double d = run<3>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double e_lane = run<4>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double e_quad = run<5>(nodes, uint32_t(count - 1), out, blocks, iterations);
        const double f_dma = run<6>(nodes, uint32_t(count - 1), out, blocks, iterations);
        printf("nodes %8zu (%7.1f MB): lane gather %7.3f ms = %6.1f Gnodes/s (%5.0f GB/s) | quad gather %7.3f ms = %6.1f Gnodes/s | quad + LDS transpose %7.3f ms = %6.1f Gnodes/s\n",
               count, count * 64 / 1e6, a, fetches / a / 1e6, fetches * 64 / a / 1e6, b, fetches / b / 1e6, c, fetches / c / 1e6);
        printf("                             quad + DPP transpose %7.3f ms = %6.1f Gnodes/s | 38 of 64 lanes: lane gather %7.3f ms = %6.1f Gnodes/s, quad + DPP %7.3f ms = %6.1f Gnodes/s\n",
               d, fetches / d / 1e6, e_lane, fetches * (38.0 / 64.0) / e_lane / 1e6, e_quad, fetches * (38.0 / 64.0) / e_quad / 1e6);
        printf("                             38 of 64 lanes, quad gather through LDS-DMA %7.3f ms = %6.1f Gnodes/s\n", f_dma, fetches * (38.0 / 64.0) / f_dma / 1e6);
        CHECK(hipFree(nodes));
    }
    return 0;
}
