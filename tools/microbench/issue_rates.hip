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
#define I_FMAMIX(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9 op_sel_hi:[1,0,0]\n"
#define I_FMAMIXHI(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define I_MAX3I(k) "v_max3_i32 %" #k ", %" #k ", %8, %9\n"
#define I_MAXI(k) "v_max_i32 %" #k ", %" #k ", %8\n"
#define I_MINU(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define I_MAXF(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_MULF(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_ADDF(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_SUBU(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 1, %8\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMPCND(k) "v_cmp_le_f32 vcc, %" #k ", %8\n v_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_CMPCNDI(k) "v_cmp_le_i32 vcc, %" #k ", %8\n v_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_ORSDWA(k) "v_or_b32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"

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
KERNEL(k_fmamix, I_FMAMIX)
KERNEL(k_fmamixhi, I_FMAMIXHI)
KERNEL(k_max3i, I_MAX3I)
KERNEL(k_maxi, I_MAXI)
KERNEL(k_minu, I_MINU)
KERNEL(k_maxf, I_MAXF)
KERNEL(k_mulf, I_MULF)
KERNEL(k_addf, I_ADDF)
KERNEL(k_subu, I_SUBU)
KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_bfe, I_BFE)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_cmpcnd, I_CMPCND)
KERNEL(k_cmpcndi, I_CMPCNDI)
KERNEL(k_orsdwa, I_ORSDWA)
KERNEL(k_med3, I_MED3)
KERNEL(k_mov, I_MOV)

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
        {"v_and_b32", run(k_and, out, in, iterations)}, {"v_fma_mix_f32 (f16 lo)", run(k_fmamix, out, in, iterations)}, {"v_fma_mix_f32 (f16 hi)", run(k_fmamixhi, out, in, iterations)},
        {"v_max3_i32", run(k_max3i, out, in, iterations)}, {"v_max_i32", run(k_maxi, out, in, iterations)}, {"v_min_u32", run(k_minu, out, in, iterations)},
        {"v_max_f32", run(k_maxf, out, in, iterations)}, {"v_mul_f32", run(k_mulf, out, in, iterations)}, {"v_add_f32", run(k_addf, out, in, iterations)},
        {"v_sub_u32", run(k_subu, out, in, iterations)}, {"v_lshl_or_b32", run(k_lshlor, out, in, iterations)}, {"v_bfe_u32", run(k_bfe, out, in, iterations)},
        {"v_cndmask_b32 (vcc)", run(k_cndmask, out, in, iterations)}, {"v_cmp_le_f32 + v_cndmask (2 instr)", run(k_cmpcnd, out, in, iterations)},
        {"v_cmp_le_i32 + v_cndmask (2 instr)", run(k_cmpcndi, out, in, iterations)}, {"v_or_b32_sdwa BYTE_1", run(k_orsdwa, out, in, iterations)},
        {"v_med3_f32", run(k_med3, out, in, iterations)}, {"v_mov_b32", run(k_mov, out, in, iterations)}};
    for (const Row& r : rows) printf("%-36s %8.2f G rounds/s (one per listed row entry)   %.2f x the time of a v_fma_f32\n", r.name, r.rate * 1e-9, reference / r.rate);
    return 0;
}
