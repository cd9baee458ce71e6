// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the path tracer's kernels.
//
// MI355X_MICROARCH.md "HBM": FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read of 16 B per lane; "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute". The trace kernels do not
// stream: a lane gathers a 64 B wide-BVH node (4 x dwordx4) or a 48 B triangle (3 x dwordx4) at a data-dependent address, spills stack entries
// as one dword per lane, and reads / writes its queues as 16 B per lane. Every kernel below moves a byte count known by construction:
//
//   k_cal_stream_read16      16 B per lane, coalesced, 2 GiB, each byte once                                  (the guide's case: expect 0.5)
//   k_cal_gather_node64      one 64 B record per lane at a random, unique index of a 2 GiB table (4 x dwordx4) -- each record exactly once
//   k_cal_gather_tri48       one 48 B record per lane at a random, unique index of a 1.5 GiB table (3 x dwordx4)
//   k_cal_gather_16_of_64    only the first 16 B of a random unique 64 B record (what a lane costs when it uses a quarter of a line)
//   k_cal_gather_node64_l2   64 B records of a 2 MiB table, 64 rounds (the table lives in every XCD's L2: what the counters see of L2 hits)
//   k_cal_gather_node64_mall 64 B records of a 48 MiB table, 16 rounds (beyond L2, inside the 256 MiB Infinity Cache: are its hits counted?)
//   k_cal_stream_write16     16 B per lane stores, coalesced, 2 GiB
//   k_cal_spill_dword        a [depth][lane] array of one dword per lane and depth, 256 B contiguous per wave and depth, written then read
//                            back by the same lane (the scratch-backed traversal stack), 1 GiB each way
//
// Run plain for the timings (prints one JSON line per kernel), then under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and the raw request
// counters; tools/fetch_calibration_summary.py divides the known bytes by what the counters report.
// Build: hipcc -O3 --offload-arch=gfx950 -o fetch_calibration fetch_calibration.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// bijection of [0, 2^bits): odd multiplier + xorshift, both invertible modulo 2^bits
__device__ inline uint32_t permute(uint32_t i, uint32_t bits) {
    const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    i = (i * 0x9E3779B1u) & mask;
    i ^= i >> (bits / 2 + 1);
    i = (i * 0x85EBCA6Bu) & mask;
    i ^= i >> (bits / 2 + 1);
    return i & mask;
}

__global__ __launch_bounds__(256) void k_cal_stream_read16(const float4* in, size_t n, float* out) {
    float acc = 0.0f;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) { const float4 v = in[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_cal_stream_write16(float4* outp, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) outp[i] = make_float4(float(i), 1.0f, 2.0f, 3.0f);
}

template <int QUADS, int USED>   // record of QUADS float4, of which the first USED are loaded
__global__ __launch_bounds__(256) void k_cal_gather(const float4* table, uint32_t bits, uint32_t rounds, float* out) {
    float acc = 0.0f;
    const uint32_t n = 1u << bits;
    for (uint32_t r = 0; r < rounds; ++r)
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            const uint32_t j = permute(i ^ (r * 0x632BE5ABu & (n - 1u)), bits);
            const float4* p = table + size_t(QUADS) * j;
#pragma unroll
            for (int q = 0; q < USED; ++q) { const float4 v = p[q]; acc += v.x + v.w; }
        }
    if (acc == 12345.678f) out[0] = acc;
}

// The scratch-backed stack: entry k of lane l of wave w lives at [w][k][l] (one dword), i.e. a wave's entry k is 256 contiguous bytes.
__global__ __launch_bounds__(256) void k_cal_spill_dword(uint32_t* spill, uint32_t depth, float* out) {
    const size_t wave = (size_t(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* mine = spill + wave * depth * 64u + lane;
    for (uint32_t k = 0; k < depth; ++k) mine[size_t(k) * 64u] = k * 2654435761u + lane;
    __threadfence_block();
    uint32_t acc = 0;
    for (uint32_t k = depth; k-- > 0;) acc += mine[size_t(k) * 64u];
    if (acc == 0x12345678u) out[0] = float(acc);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    void start() { CHECK(hipEventRecord(a, 0)); }
    float stop() { CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b)); float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

static void report(const char* kernel, const char* pattern, double read_bytes, double write_bytes, float ms) {
    printf("{\"kernel\": \"%s\", \"pattern\": \"%s\", \"read_bytes\": %.0f, \"write_bytes\": %.0f, \"ms\": %.4f, \"GB_per_s\": %.1f}\n", kernel, pattern, read_bytes, write_bytes, ms,
           (read_bytes + write_bytes) / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const size_t big = size_t(2) << 30;
    float4* table = nullptr;
    float* out = nullptr;
    CHECK(hipMalloc(&table, big));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(table, 0, big));
    CHECK(hipDeviceSynchronize());
    Timer t;
    const int grid = 256 * 16;

    for (int rep = 0; rep < 2; ++rep) {   // the second round is the one to read (same byte counts; the first warms the code objects)
        t.start();
        hipLaunchKernelGGL(k_cal_stream_read16, dim3(grid), dim3(256), 0, 0, table, big / 16, out);
        report("k_cal_stream_read16", "16 B per lane coalesced stream, 2 GiB once", double(big), 0, t.stop());

        t.start();
        hipLaunchKernelGGL((k_cal_gather<4, 4>), dim3(grid), dim3(256), 0, 0, table, 25u, 1u, out);
        report("k_cal_gather<4, 4>", "64 B record per lane, random unique index, 2 GiB table, each record once", double(big), 0, t.stop());

        t.start();
        hipLaunchKernelGGL((k_cal_gather<3, 3>), dim3(grid), dim3(256), 0, 0, table, 25u, 1u, out);
        report("k_cal_gather<3, 3>", "48 B record per lane, random unique index, 1.5 GiB table, each record once", 48.0 * double(1u << 25), 0, t.stop());

        t.start();
        hipLaunchKernelGGL((k_cal_gather<4, 1>), dim3(grid), dim3(256), 0, 0, table, 25u, 1u, out);
        report("k_cal_gather<4, 1>", "first 16 B of a random unique 64 B record, 2 GiB table", 16.0 * double(1u << 25), 0, t.stop());

        t.start();
        hipLaunchKernelGGL((k_cal_gather<4, 4>), dim3(grid), dim3(256), 0, 0, table, 15u, 1024u, out);
        report("k_cal_gather<4, 4> L2", "64 B records of a 2 MiB table, 1024 rounds (L2 resident)", 64.0 * double(1u << 15) * 1024.0, 0, t.stop());

        t.start();
        hipLaunchKernelGGL((k_cal_gather<3, 3>), dim3(grid), dim3(256), 0, 0, table, 20u, 32u, out);
        report("k_cal_gather<3, 3> MALL", "48 B records of a 48 MiB table, 32 rounds (beyond L2, inside the Infinity Cache)", 48.0 * double(1u << 20) * 32.0, 0, t.stop());

        t.start();
        hipLaunchKernelGGL(k_cal_stream_write16, dim3(grid), dim3(256), 0, 0, table, big / 16);
        report("k_cal_stream_write16", "16 B per lane coalesced stores, 2 GiB", 0, double(big), t.stop());

        // 1 GiB of spill: 65536 waves x 64 entries x 256 B
        t.start();
        hipLaunchKernelGGL(k_cal_spill_dword, dim3(65536 / 4), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(table), 64u, out);
        report("k_cal_spill_dword", "one dword per lane and stack entry, 256 B per wave and entry, 1 GiB written then read back by its lane", double(big) / 2, double(big) / 2, t.stop());
    }
    CHECK(hipFree(table));
    CHECK(hipFree(out));
    return 0;
}
