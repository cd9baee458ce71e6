// packed_fp32.hip -- does v_pk_fma_f32 deliver two FMAs per lane and issue slot on this part, for VGPR and for SGPR operands?
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o /tmp/packed_fp32 tools/microbench/packed_fp32.hip && /tmp/packed_fp32
// Each thread runs `iterations` rounds of 8 independent FMA chains, scalar (8 v_fma_f32 per round) or packed (8 v_pk_fma_f32 per
// round = 16 FMAs). Reported: FMA/s per variant. Equal rates mean a packed instruction occupies two issue slots.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <bool UNIFORM_OPERAND>
__global__ void k_scalar(float* out, const float* coefficients, int iterations) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float m = UNIFORM_OPERAND ? coefficients[0] : coefficients[threadIdx.x & 1], c = UNIFORM_OPERAND ? coefficients[1] : coefficients[2 + (threadIdx.x & 1)];
    for (int it = 0; it < iterations; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool UNIFORM_OPERAND>
__global__ void k_packed(float* out, const float* coefficients, int iterations) {
    float2v a[8];
    for (int i = 0; i < 8; ++i) a[i] = float2v{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    const int o = UNIFORM_OPERAND ? 0 : (threadIdx.x & 1);
    const float2v m = {coefficients[o], coefficients[o + 1]}, c = {coefficients[o + 2], coefficients[o + 3]};
    for (int it = 0; it < iterations; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
double run(K kernel, float* out, const float* coefficients, int iterations, int fma_per_round) {
    const int blocks = 256 * 16, threads = 256;
    hipEvent_t begin, end;
    hipEventCreate(&begin); hipEventCreate(&end);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, coefficients, iterations);
    hipEventRecord(begin);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, coefficients, iterations);
    hipEventRecord(end);
    hipEventSynchronize(end);
    float ms = 0;
    hipEventElapsedTime(&ms, begin, end);
    return double(blocks) * threads * iterations * fma_per_round / (ms * 1e-3);
}

int main() {
    float *out, *coefficients;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    hipMalloc(&coefficients, 8 * sizeof(float));
    const float host[8] = {0.999f, 1.001f, 0.5f, 0.25f, 0.75f, 0.125f, 1.0f, 2.0f};
    hipMemcpy(coefficients, host, sizeof(host), hipMemcpyHostToDevice);
    const int iterations = 4096;
    printf("v_fma_f32,    VGPR operands: %.2f TFMA/s\n", run(k_scalar<false>, out, coefficients, iterations, 8) * 1e-12);
    printf("v_fma_f32,    SGPR operands: %.2f TFMA/s\n", run(k_scalar<true>, out, coefficients, iterations, 8) * 1e-12);
    printf("v_pk_fma_f32, VGPR operands: %.2f TFMA/s\n", run(k_packed<false>, out, coefficients, iterations, 16) * 1e-12);
    printf("v_pk_fma_f32, SGPR operands: %.2f TFMA/s\n", run(k_packed<true>, out, coefficients, iterations, 16) * 1e-12);
    return 0;
}
