// issue_rates.hip -- how many wave64 instructions per clock and SIMD does gfx950 issue of the kinds a compressed-node test is made of?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_rates tools/microbench/issue_rates.hip && /tmp/issue_rates
// Every thread runs `iterations` rounds of 8 independent chains of ONE instruction (inline asm, so the compiler neither fuses nor vectorises), on a grid
// that fills every SIMD with 8 waves. Reported: instructions per second device-wide and cycles per wave instruction and SIMD at the measured clock of the
// v_fma_f32 row (taken as the reference).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(INSTR)                                                                                                        \
    for (int it = 0; it < iterations; ++it) {                                                                                \
        asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                                 \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));     \
    }

#define I_FMA32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PKFMA16(k) "v_pk_fma_f16 %" #k ", %" #k ", %8, %9\n"
#define I_PKMIN16(k) "v_pk_min_f16 %" #k ", %" #k ", %8\n"
#define I_PKMAX16(k) "v_pk_max_f16 %" #k ", %" #k ", %8\n"
#define I_PKMUL16(k) "v_pk_mul_f16 %" #k ", %" #k ", %8\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MIN32(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define I_CVTPKFP8(k) "v_cvt_f32_fp8 %" #k ", %" #k "\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"

#define KERNEL(NAME, INSTR)                                                                          \
    __global__ void NAME(uint32_t* out, const uint32_t* in, int iterations) {                         \
        uint32_t a0 = in[threadIdx.x & 7], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const uint32_t m = in[8 + (threadIdx.x & 1)], c = in[10 + (threadIdx.x & 1)];                 \
        CHAIN8(INSTR)                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }

KERNEL(k_fma32, I_FMA32)
KERNEL(k_pkfma16, I_PKFMA16)
KERNEL(k_pkmin16, I_PKMIN16)
KERNEL(k_pkmax16, I_PKMAX16)
KERNEL(k_pkmul16, I_PKMUL16)
KERNEL(k_perm, I_PERM)
KERNEL(k_cvtub, I_CVTUB)
KERNEL(k_max3, I_MAX3)
KERNEL(k_min32, I_MIN32)
KERNEL(k_and, I_AND)

template <typename K>
double run(K kernel, uint32_t* out, const uint32_t* in, int iterations) {
    const int blocks = 256 * 8, threads = 256;      // 8 waves per SIMD
    hipEvent_t begin, end;
    hipEventCreate(&begin); hipEventCreate(&end);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, in, iterations);
    hipEventRecord(begin);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, in, iterations);
    hipEventRecord(end);
    hipEventSynchronize(end);
    float ms = 0;
    hipEventElapsedTime(&ms, begin, end);
    return double(blocks) * (threads / 64) * iterations * 8.0 / (ms * 1e-3);     // wave instructions per second
}

int main() {
    uint32_t *out, *in;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t));
    hipMalloc(&in, 16 * sizeof(uint32_t));
    const uint32_t host[16] = {0x3C003C00u, 0x3C013C01u, 0x3C023C02u, 0x3C033C03u, 0x3C043C04u, 0x3C053C05u, 0x3C063C06u, 0x3C073C07u,
                               0x3BFF3BFFu, 0x3C003C01u, 0x00010001u, 0x00020002u, 0, 0, 0, 0};
    hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice);
    const int iterations = 8192;
    const double reference = run(k_fma32, out, in, iterations);
    struct Row { const char* name; double rate; } rows[] = {
        {"v_fma_f32", reference}, {"v_pk_fma_f16", run(k_pkfma16, out, in, iterations)}, {"v_pk_min_f16", run(k_pkmin16, out, in, iterations)},
        {"v_pk_max_f16", run(k_pkmax16, out, in, iterations)}, {"v_pk_mul_f16", run(k_pkmul16, out, in, iterations)}, {"v_perm_b32", run(k_perm, out, in, iterations)},
        {"v_cvt_f32_ubyte1", run(k_cvtub, out, in, iterations)}, {"v_max3_f32", run(k_max3, out, in, iterations)}, {"v_min_f32", run(k_min32, out, in, iterations)},
        {"v_and_b32", run(k_and, out, in, iterations)}};
    for (const Row& r : rows) printf("%-18s %8.2f G wave-instructions/s   %.2f x the time of a v_fma_f32\n", r.name, r.rate * 1e-9, reference / r.rate);
    return 0;
}
