// camera_effects.hip -- exposure, bloom, vignette, tonemapping and film grain on the path tracer's half4 frame: the kernels and
// the C-ABI of include/hipr_camera_effects_c.h. Written for gfx950; replaces DX11Renderer::CameraEffects
// (extensions/DX11Renderer/DX11Renderer/CameraEffects.{h,cpp} and Shaders/CameraEffects/*.hlsl), whose stages and arithmetic
// each kernel cites. All of it is HBM-bound image work: one pass over the 8 B/pixel frame per stage.
//
//   k_exposure_histogram        64-bin log-luminance histogram: LDS histograms replicated 16 x per wave, one histogram per block to global memory
//   k_exposure_from_histogram   one wave: sums the blocks' histograms, prefix sum, percentile clamp, weighted luminance, eye adaptation
//   k_log_luminance_partials / k_log_average_finish   two-level sum of log2 luminance -> log average or key-value exposure
//   k_exposure_from_bias        fixed exposure with eye adaptation
//   k_bloom_horizontal / k_bloom_vertical (+ _tiled)   separable Gaussian through bilinearly placed taps, half4 intermediates; the tiled
//                               forms stage a block's texel neighbourhood in LDS
//   k_kawase_extract / _downsample / _upsample   the dual Kawase bloom the reference tests next to the Gaussian one
//   k_tonemap<MODE>             exposure * (clamped pixel + bloom), vignette, operator, film grain -> RGBA16F / RGBA32F / RGBA8 sRGB
#include "../../include/hipr_camera_effects_c.h"

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace hipr_camera_effects {

constexpr int BINS = HIPR_EXPOSURE_HISTOGRAM_BINS;
constexpr int REDUCE_BLOCK = 256;
constexpr int MAX_PARTIALS = 1024;
constexpr int HISTOGRAM_REPLICAS = 16;      // per wave: lanes that meet in one bin spread over 16 counters

struct DeviceFrame {
    const uint2* pixels;    // half4 as two dwords
    uint32_t pitch, rows;
    int32_t x, y, width, height;
};

struct float3_ { float x, y, z; };

__device__ __forceinline__ float3_ unpack_rgb(uint2 p) {
    const __half2 rg = *reinterpret_cast<const __half2*>(&p.x), ba = *reinterpret_cast<const __half2*>(&p.y);
    return {__low2float(rg), __high2float(rg), __low2float(ba)};
}
__device__ __forceinline__ uint2 pack_rgba(float r, float g, float b, float a) {
    const __half2 rg = __floats2half2_rn(r, g), ba = __floats2half2_rn(b, a);
    uint2 p;
    p.x = *reinterpret_cast<const uint32_t*>(&rg); p.y = *reinterpret_cast<const uint32_t*>(&ba);
    return p;
}
__device__ __forceinline__ float luminance(float3_ c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }      // Utils.hlsl:98
__device__ __forceinline__ float log_luminance_of(uint2 pixel) { return log2f(fmaxf(luminance(unpack_rgb(pixel)), 0.0001f)); }

// CameraEffects/Utils.hlsl:42-47. Eye adaptation switched off arrives as infinite speeds (CameraEffects.cpp:432-436).
__device__ float eye_adaptation(float current_exposure, float target_exposure, float brightness, float darkness, float delta_time) {
    const float delta_exposure = target_exposure - current_exposure;
    const float adaption_speed = delta_exposure > 0.0f ? brightness : darkness;
    const float factor = 1.0f - exp2f(-delta_time * adaption_speed);
    return current_exposure + delta_exposure * factor;
}

struct ExposureConstants {
    float min_log_luminance, max_log_luminance, min_percentage, max_percentage, log_luminance_bias;
    float eye_adaptation_brightness, eye_adaptation_darkness, delta_time;
};

// ---- exposure histogram (ReduceExposureHistogram.hlsl:27-70) ---------------------------------------------------------------------
// Blocks stride over the viewport's pixels in row-major order (coalesced 8 B loads). Each wave owns 16 copies of the 64 bins in
// LDS, lane l counts into copy l % 16: a flat region of the image, where all 64 lanes hit one bin, costs 4 serialised LDS
// atomics instead of 64. Every block writes its own 64 bins; whoever consumes the histogram adds the blocks up (a few hundred
// coalesced loads), which costs less than tens of thousands of global atomics meeting in 64 addresses, and needs no clearing.
__global__ __launch_bounds__(REDUCE_BLOCK) void k_exposure_histogram(DeviceFrame frame, float min_log_luminance, float max_log_luminance, uint32_t* __restrict__ block_histograms) {
    __shared__ uint32_t s_bins[(REDUCE_BLOCK / 64) * HISTOGRAM_REPLICAS * BINS];
    for (uint32_t i = threadIdx.x; i < (REDUCE_BLOCK / 64) * HISTOGRAM_REPLICAS * BINS; i += REDUCE_BLOCK) s_bins[i] = 0u;
    __syncthreads();

    uint32_t* mine = s_bins + ((threadIdx.x / 64) * HISTOGRAM_REPLICAS + (threadIdx.x % HISTOGRAM_REPLICAS)) * BINS;
    const uint32_t pixel_count = uint32_t(frame.width) * uint32_t(frame.height);
    const float recip_range = max_log_luminance - min_log_luminance;
    for (uint32_t i = blockIdx.x * REDUCE_BLOCK + threadIdx.x; i < pixel_count; i += gridDim.x * REDUCE_BLOCK) {
        const uint32_t x = i % uint32_t(frame.width), y = i / uint32_t(frame.width);
        const uint2 pixel = frame.pixels[(x + uint32_t(frame.x)) + size_t(y + uint32_t(frame.y)) * frame.pitch];
        const float normalized_index = (log_luminance_of(pixel) - min_log_luminance) / recip_range;
        const int bin_index = min(max(int(normalized_index * BINS + 0.5f), 0), BINS - 1);
        atomicAdd(&mine[bin_index], 1u);
    }
    __syncthreads();

    if (threadIdx.x < BINS) {
        uint32_t total = 0;
        for (int copy = 0; copy < (REDUCE_BLOCK / 64) * HISTOGRAM_REPLICAS; ++copy) total += s_bins[copy * BINS + threadIdx.x];
        block_histograms[blockIdx.x * BINS + threadIdx.x] = total;
    }
}

// Adds up the blocks' histograms with all 16 waves of a 1024-thread block (wave w takes blocks w, w + 16, ...): a single wave
// walking hundreds of rows one load after the other would take longer than the histogram pass itself. Result in s_total[0..63].
constexpr int SUM_BLOCK = 1024;
__device__ __forceinline__ void sum_block_histograms(const uint32_t* __restrict__ block_histograms, uint32_t block_count, uint32_t* s_partial, uint32_t* s_total) {
    const uint32_t wave = threadIdx.x / 64, bin = threadIdx.x % 64;
    uint32_t total = 0;
#pragma unroll 4
    for (uint32_t b = wave; b < block_count; b += SUM_BLOCK / 64) total += block_histograms[b * BINS + bin];
    s_partial[wave * BINS + bin] = total;
    __syncthreads();
    if (threadIdx.x < BINS) {
        uint32_t sum = 0;
        for (int w = 0; w < SUM_BLOCK / 64; ++w) sum += s_partial[w * BINS + threadIdx.x];
        s_total[threadIdx.x] = sum;
    }
    __syncthreads();
}

__global__ __launch_bounds__(SUM_BLOCK) void k_sum_histograms(const uint32_t* __restrict__ block_histograms, uint32_t block_count, uint32_t* __restrict__ histogram) {
    __shared__ uint32_t s_partial[(SUM_BLOCK / 64) * BINS], s_total[BINS];
    sum_block_histograms(block_histograms, block_count, s_partial, s_total);
    if (threadIdx.x < BINS) histogram[threadIdx.x] = s_total[threadIdx.x];
}

// ---- exposure from the histogram (ReduceExposureHistogram.hlsl:82-154): one wave, lane = bin ----------------------------------------
__global__ __launch_bounds__(SUM_BLOCK) void k_exposure_from_histogram(const uint32_t* __restrict__ block_histograms, uint32_t block_count, ExposureConstants c, float* __restrict__ linear_exposure) {
    __shared__ uint32_t s_partial[(SUM_BLOCK / 64) * BINS], s_total[BINS];
    sum_block_histograms(block_histograms, block_count, s_partial, s_total);
    if (threadIdx.x >= BINS) return;        // the rest is one wave's work, lane = bin
    const int bin = threadIdx.x;
    const uint32_t bin_total = s_total[bin];

    // Exclusive prefix sum of the bin counts. The values are whole numbers below 2^24, so any summation order is exact in f32.
    const float count = float(bin_total);
    float inclusive = count;
    for (int offset = 1; offset < BINS; offset <<= 1) {
        const float below = __shfl_up(inclusive, offset);
        if (bin >= offset) inclusive += below;
    }
    const float total = __shfl(inclusive, BINS - 1);
    const float max_pixel_count = total * c.max_percentage;
    const float min_pixel_count = total * c.min_percentage;
    // Clamp the prefix sum to the percentile window: counts above the upper bound and below the lower bound drop out.
    const float clamped = fmaxf(0.0f, fminf(inclusive - count, max_pixel_count) - min_pixel_count);
    const float next_clamped = __shfl_down(clamped, 1);
    const float bin_count = (bin == BINS - 1 ? max_pixel_count - min_pixel_count : next_clamped) - clamped;
    const float normalized_index = (bin + 0.5f) / BINS;
    const float bin_log_luminance = c.min_log_luminance + normalized_index * (c.max_log_luminance - c.min_log_luminance);
    float weighted = exp2f(bin_log_luminance) * bin_count;
    for (int offset = BINS >> 1; offset > 0; offset >>= 1) weighted += __shfl_down(weighted, offset);   // the shader's tree: lane t takes t + offset

    if (bin == 0) {
        const float average_luminance = weighted / (max_pixel_count - min_pixel_count);
        const float target = exp2f(c.log_luminance_bias) / average_luminance;
        linear_exposure[0] = eye_adaptation(linear_exposure[0], target, c.eye_adaptation_brightness, c.eye_adaptation_darkness, c.delta_time);
    }
}

// ---- log average luminance (ReduceLogAverageLuminance.hlsl:23-106) ---------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* s_wave_sums) {
    for (int offset = 32; offset > 0; offset >>= 1) v += __shfl_down(v, offset);
    if ((threadIdx.x & 63) == 0) s_wave_sums[threadIdx.x / 64] = v;
    __syncthreads();
    float total = 0.0f;
    if (threadIdx.x == 0)
        for (int w = 0; w < REDUCE_BLOCK / 64; ++w) total += s_wave_sums[w];
    return total;     // valid in thread 0
}

__global__ __launch_bounds__(REDUCE_BLOCK) void k_log_luminance_partials(DeviceFrame frame, float* __restrict__ partials) {
    __shared__ float s_wave_sums[REDUCE_BLOCK / 64];
    const uint32_t pixel_count = uint32_t(frame.width) * uint32_t(frame.height);
    float sum = 0.0f;
    for (uint32_t i = blockIdx.x * REDUCE_BLOCK + threadIdx.x; i < pixel_count; i += gridDim.x * REDUCE_BLOCK) {
        const uint32_t x = i % uint32_t(frame.width), y = i / uint32_t(frame.width);
        sum += log_luminance_of(frame.pixels[(x + uint32_t(frame.x)) + size_t(y + uint32_t(frame.y)) * frame.pitch]);
    }
    const float total = block_sum(sum, s_wave_sums);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// mode 0: out = 2^mean (compute_log_average). mode 1: the geometric-mean key value exposure with eye adaptation (compute_linear_exposure).
__global__ __launch_bounds__(REDUCE_BLOCK) void k_log_average_finish(const float* __restrict__ partials, uint32_t partial_count, float pixel_count, int mode, ExposureConstants c,
                                                                     float* __restrict__ out) {
    __shared__ float s_wave_sums[REDUCE_BLOCK / 64];
    float sum = 0.0f;
    for (uint32_t i = threadIdx.x; i < partial_count; i += REDUCE_BLOCK) sum += partials[i];
    const float total = block_sum(sum, s_wave_sums);
    if (threadIdx.x != 0) return;
    float average_log_luminance = total / pixel_count;
    if (mode == 0) { out[0] = exp2f(average_log_luminance); return; }
    average_log_luminance = fminf(fmaxf(average_log_luminance, c.min_log_luminance), c.max_log_luminance);
    const float log_average_luminance = exp2f(average_log_luminance);
    const float key_value = 1.03f - (2.0f / (2 + log10f(log_average_luminance + 1)));      // MJP's geometric mean exposure, ReduceLogAverageLuminance.hlsl:59-62
    const float target = key_value / log_average_luminance * exp2f(c.log_luminance_bias);
    out[0] = eye_adaptation(out[0], target, c.eye_adaptation_brightness, c.eye_adaptation_darkness, c.delta_time);
}

__global__ void k_exposure_from_bias(ExposureConstants c, float* __restrict__ linear_exposure) {     // Tonemapping.hlsl:18-21
    linear_exposure[0] = eye_adaptation(linear_exposure[0], exp2f(c.log_luminance_bias), c.eye_adaptation_brightness, c.eye_adaptation_darkness, c.delta_time);
}

// ---- Gaussian bloom (Bloom.hlsl:24-67, CameraEffects.cpp:39-112) -------------------------------------------------------------------
// taps: (offset in texels, weight) as half2, as the reference uploads them. A sample at a fractional texel position is the
// linear blend of its two neighbours along the filtered axis; the other axis sits on texel centres.
__device__ __forceinline__ float3_ fetch_clamped(const uint2* __restrict__ pixels, uint32_t pitch, int x, int y, int max_x, int max_y) {
    x = min(max(x, 0), max_x); y = min(max(y, 0), max_y);
    return unpack_rgb(pixels[uint32_t(x) + size_t(y) * pitch]);
}
__device__ __forceinline__ float3_ lerp3(float3_ a, float3_ b, float t) { return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)}; }

// High intensity pass and horizontal filter: reads the whole frame (also beside the viewport), writes the viewport-sized intermediate.
__global__ __launch_bounds__(256) void k_bloom_horizontal(DeviceFrame frame, const __half2* __restrict__ taps, int sample_count, float threshold, uint2* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= frame.width || y >= frame.height) return;
    const int row = y + frame.y, max_x = int(frame.pitch) - 1, max_y = int(frame.rows) - 1;
    const float centre = float(x + frame.x);
    float3_ sum = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < sample_count; ++s) {
        const float offset = __low2float(taps[s]), weight = __high2float(taps[s]);
        const float lower_position = centre - offset, upper_position = centre + offset;
        const float lower_floor = floorf(lower_position), upper_floor = floorf(upper_position);
        const float3_ lower = lerp3(fetch_clamped(frame.pixels, frame.pitch, int(lower_floor), row, max_x, max_y),
                                    fetch_clamped(frame.pixels, frame.pitch, int(lower_floor) + 1, row, max_x, max_y), lower_position - lower_floor);
        const float3_ upper = lerp3(fetch_clamped(frame.pixels, frame.pitch, int(upper_floor), row, max_x, max_y),
                                    fetch_clamped(frame.pixels, frame.pitch, int(upper_floor) + 1, row, max_x, max_y), upper_position - upper_floor);
        sum.x += (fmaxf(lower.x - threshold, 0.0f) + fmaxf(upper.x - threshold, 0.0f)) * weight;
        sum.y += (fmaxf(lower.y - threshold, 0.0f) + fmaxf(upper.y - threshold, 0.0f)) * weight;
        sum.z += (fmaxf(lower.z - threshold, 0.0f) + fmaxf(upper.z - threshold, 0.0f)) * weight;
    }
    out[uint32_t(x) + size_t(y) * uint32_t(frame.width)] = pack_rgba(sum.x, sum.y, sum.z, 1.0f);
}

// Vertical filter over the viewport-sized intermediate, clamped at its edges.
__global__ __launch_bounds__(256) void k_bloom_vertical(const uint2* __restrict__ in, int width, int height, const __half2* __restrict__ taps, int sample_count, uint2* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const float centre = float(y);
    float3_ sum = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < sample_count; ++s) {
        const float offset = __low2float(taps[s]), weight = __high2float(taps[s]);
        const float upper_position = centre + offset, lower_position = centre - offset;
        const float upper_floor = floorf(upper_position), lower_floor = floorf(lower_position);
        const float3_ upper = lerp3(fetch_clamped(in, uint32_t(width), x, int(upper_floor), width - 1, height - 1),
                                    fetch_clamped(in, uint32_t(width), x, int(upper_floor) + 1, width - 1, height - 1), upper_position - upper_floor);
        const float3_ lower = lerp3(fetch_clamped(in, uint32_t(width), x, int(lower_floor), width - 1, height - 1),
                                    fetch_clamped(in, uint32_t(width), x, int(lower_floor) + 1, width - 1, height - 1), lower_position - lower_floor);
        sum.x += (upper.x + lower.x) * weight;
        sum.y += (upper.y + lower.y) * weight;
        sum.z += (upper.z + lower.z) * weight;
    }
    out[uint32_t(x) + size_t(y) * uint32_t(width)] = pack_rgba(sum.x, sum.y, sum.z, 1.0f);
}

// ---- Dual Kawase bloom (Bloom.hlsl:69-119, CameraEffects.cpp:140-232): extract, `half_passes` levels down, the same levels up -------------
struct float4_ { float x, y, z, w; };
__device__ __forceinline__ float4_ unpack_rgba(uint2 p) {
    const __half2 rg = *reinterpret_cast<const __half2*>(&p.x), ba = *reinterpret_cast<const __half2*>(&p.y);
    return {__low2float(rg), __high2float(rg), __low2float(ba), __high2float(ba)};
}
// SampleLevel(bilinear_sampler, uv, 0) on a width x height half4 image: clamped addressing, exact fractional weights.
__device__ __forceinline__ float4_ sample_bilinear(const uint2* __restrict__ pixels, int width, int height, float u, float v) {
    const float px = u * float(width) - 0.5f, py = v * float(height) - 0.5f;
    const float fx = floorf(px), fy = floorf(py);
    const float tx = px - fx, ty = py - fy;
    const int x0 = min(max(int(fx), 0), width - 1), x1 = min(max(int(fx) + 1, 0), width - 1);
    const int y0 = min(max(int(fy), 0), height - 1), y1 = min(max(int(fy) + 1, 0), height - 1);
    const float4_ a = unpack_rgba(pixels[x0 + size_t(y0) * width]), b = unpack_rgba(pixels[x1 + size_t(y0) * width]);
    const float4_ c = unpack_rgba(pixels[x0 + size_t(y1) * width]), d = unpack_rgba(pixels[x1 + size_t(y1) * width]);
    auto mix = [](float p, float q, float t) { return p + t * (q - p); };
    return {mix(mix(a.x, b.x, tx), mix(c.x, d.x, tx), ty), mix(mix(a.y, b.y, tx), mix(c.y, d.y, tx), ty), mix(mix(a.z, b.z, tx), mix(c.z, d.z, tx), ty),
            mix(mix(a.w, b.w, tx), mix(c.w, d.w, tx), ty)};
}
__device__ __forceinline__ void accumulate(float4_& sum, float4_ v, float weight) { sum.x += v.x * weight; sum.y += v.y * weight; sum.z += v.z * weight; sum.w += v.w * weight; }

__global__ __launch_bounds__(256) void k_kawase_extract(DeviceFrame frame, float threshold, uint2* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= frame.width || y >= frame.height) return;
    const float4_ pixel = unpack_rgba(frame.pixels[uint32_t(x + frame.x) + size_t(y + frame.y) * frame.pitch]);
    out[uint32_t(x) + size_t(y) * uint32_t(frame.width)] = pack_rgba(fmaxf(0.0f, pixel.x - threshold), fmaxf(0.0f, pixel.y - threshold), fmaxf(0.0f, pixel.z - threshold), pixel.w);
}

__global__ __launch_bounds__(256) void k_kawase_downsample(const uint2* __restrict__ in, int in_width, int in_height, uint2* __restrict__ out, int width, int height) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const float inverse_width = 1.0f / float(width), inverse_height = 1.0f / float(height);
    const float half_x = 0.5f * inverse_width, half_y = 0.5f * inverse_height;
    const float u = float(x) * inverse_width + half_x, v = float(y) * inverse_height + half_y;
    float4_ sum = {0.0f, 0.0f, 0.0f, 0.0f};
    accumulate(sum, sample_bilinear(in, in_width, in_height, u, v), 4.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u + half_x, v + half_y), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u + half_x, v - half_y), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u - half_x, v + half_y), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u - half_x, v - half_y), 1.0f);
    out[uint32_t(x) + size_t(y) * uint32_t(width)] = pack_rgba(sum.x / 8.0f, sum.y / 8.0f, sum.z / 8.0f, sum.w / 8.0f);
}

__global__ __launch_bounds__(256) void k_kawase_upsample(const uint2* __restrict__ in, int in_width, int in_height, uint2* __restrict__ out, int width, int height) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const float inverse_width = 1.0f / float(width), inverse_height = 1.0f / float(height);
    const float half_x = 0.5f * inverse_width, half_y = 0.5f * inverse_height;
    const float u = float(x) * inverse_width + half_x, v = float(y) * inverse_height + half_y;
    float4_ sum = {0.0f, 0.0f, 0.0f, 0.0f};
    accumulate(sum, sample_bilinear(in, in_width, in_height, u - half_x * 2.0f, v), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u - half_x, v + half_y), 2.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u, v + half_y * 2.0f), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u + half_x, v + half_y), 2.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u + half_x * 2.0f, v), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u + half_x, v - half_y), 2.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u, v - half_y * 2.0f), 1.0f);
    accumulate(sum, sample_bilinear(in, in_width, in_height, u - half_x, v - half_y), 2.0f);
    out[uint32_t(x) + size_t(y) * uint32_t(width)] = pack_rgba(sum.x / 12.0f, sum.y / 12.0f, sum.z / 12.0f, sum.w / 12.0f);
}

// The same two passes with the texels a block needs staged in LDS first. A tap reads two neighbouring texels on either side of
// the pixel, so a pixel touches the 4 * sample_count texels around it: from global memory that is 108 loads per pixel and pass
// at 1080p (support 54), all L1 / L2 hits but each an instruction; staged, a block of 256 pixels loads every texel of its
// neighbourhood once (2.7 x / 4.4 x its own pixels) and the taps read LDS. Same operations in the same order: bit-identical output.
// reach = 2 * sample_count texels on either side of a pixel (the outermost tap's far neighbour).
__global__ __launch_bounds__(256) void k_bloom_horizontal_tiled(DeviceFrame frame, const __half2* __restrict__ taps, int sample_count, float threshold, uint2* __restrict__ out) {
    extern __shared__ uint2 s_texels[];      // [4 rows][64 + 2 * reach]
    const int reach = 2 * sample_count, tile_width = 64 + 2 * reach;
    const int tile_x = blockIdx.x * 64 + frame.x - reach, tile_y = blockIdx.y * 4 + frame.y;      // frame coordinates of the tile's first staged texel
    const int max_x = int(frame.pitch) - 1, max_y = int(frame.rows) - 1;
    for (int i = threadIdx.x; i < 4 * tile_width; i += 256) {
        const int r = i / tile_width, c = i - r * tile_width;
        const int px = min(max(tile_x + c, 0), max_x), py = min(max(tile_y + r, 0), max_y);
        s_texels[i] = frame.pixels[uint32_t(px) + size_t(py) * frame.pitch];
    }
    __syncthreads();

    const int lane_x = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane_x, y = blockIdx.y * 4 + r;
    if (x >= frame.width || y >= frame.height) return;
    const uint2* row = s_texels + r * tile_width;
    const float centre = float(x + frame.x);
    float3_ sum = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < sample_count; ++s) {
        const float offset = __low2float(taps[s]), weight = __high2float(taps[s]);
        const float lower_position = centre - offset, upper_position = centre + offset;
        const float lower_floor = floorf(lower_position), upper_floor = floorf(upper_position);
        const int lower_index = int(lower_floor) - tile_x, upper_index = int(upper_floor) - tile_x;
        const float3_ lower = lerp3(unpack_rgb(row[lower_index]), unpack_rgb(row[lower_index + 1]), lower_position - lower_floor);
        const float3_ upper = lerp3(unpack_rgb(row[upper_index]), unpack_rgb(row[upper_index + 1]), upper_position - upper_floor);
        sum.x += (fmaxf(lower.x - threshold, 0.0f) + fmaxf(upper.x - threshold, 0.0f)) * weight;
        sum.y += (fmaxf(lower.y - threshold, 0.0f) + fmaxf(upper.y - threshold, 0.0f)) * weight;
        sum.z += (fmaxf(lower.z - threshold, 0.0f) + fmaxf(upper.z - threshold, 0.0f)) * weight;
    }
    out[uint32_t(x) + size_t(y) * uint32_t(frame.width)] = pack_rgba(sum.x, sum.y, sum.z, 1.0f);
}

// Tiles of 32 columns x 32 rows: lane % 32 is the column (consecutive 8 B LDS words, coalesced stores), thread / 32 one of eight
// groups of four rows.
__global__ __launch_bounds__(256) void k_bloom_vertical_tiled(const uint2* __restrict__ in, int width, int height, const __half2* __restrict__ taps, int sample_count, uint2* __restrict__ out) {
    extern __shared__ uint2 s_texels[];      // [32 + 2 * reach rows][32 columns]
    const int reach = 2 * sample_count, tile_rows = 32 + 2 * reach;
    const int tile_x = blockIdx.x * 32, tile_y = blockIdx.y * 32 - reach;
    for (int i = threadIdx.x; i < tile_rows * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        const int px = min(tile_x + c, width - 1), py = min(max(tile_y + r, 0), height - 1);
        s_texels[i] = in[uint32_t(px) + size_t(py) * uint32_t(width)];
    }
    __syncthreads();

    const int column = threadIdx.x & 31, x = tile_x + column;
    if (x >= width) return;
    for (int k = 0; k < 4; ++k) {
        const int y = blockIdx.y * 32 + (threadIdx.x >> 5) * 4 + k;
        if (y >= height) return;
        const float centre = float(y);
        float3_ sum = {0.0f, 0.0f, 0.0f};
        for (int s = 0; s < sample_count; ++s) {
            const float offset = __low2float(taps[s]), weight = __high2float(taps[s]);
            const float upper_position = centre + offset, lower_position = centre - offset;
            const float upper_floor = floorf(upper_position), lower_floor = floorf(lower_position);
            const int upper_index = (int(upper_floor) - tile_y) * 32 + column, lower_index = (int(lower_floor) - tile_y) * 32 + column;
            const float3_ upper = lerp3(unpack_rgb(s_texels[upper_index]), unpack_rgb(s_texels[upper_index + 32]), upper_position - upper_floor);
            const float3_ lower = lerp3(unpack_rgb(s_texels[lower_index]), unpack_rgb(s_texels[lower_index + 32]), lower_position - lower_floor);
            sum.x += (upper.x + lower.x) * weight;
            sum.y += (upper.y + lower.y) * weight;
            sum.z += (upper.z + lower.z) * weight;
        }
        out[uint32_t(x) + size_t(y) * uint32_t(width)] = pack_rgba(sum.x, sum.y, sum.z, 1.0f);
    }
}

// ---- tonemapping (Tonemapping.hlsl:38-227) -----------------------------------------------------------------------------------------
struct TonemapConstants {
    float bloom_threshold, vignette_strength, film_grain_strength, delta_time;
    float black_clip, toe, slope, shoulder, white_clip;
    float toe_scale, shoulder_scale, toe_match, shoulder_match;     // the curve's frame-uniform parameters (Tonemapping.hlsl:79-94), evaluated once on the host
    float sRGB_to_AP1[9];     // XYZ_to_AP1 * D65_to_D60 * sRGB_to_XYZ, multiplied on the host in f32 like the shader compiler's constant folding
};

__device__ __forceinline__ float saturate(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

__device__ float3_ unreal4(float3_ color, const TonemapConstants& c) {     // Tonemapping.hlsl:71-109
    const float* m = c.sRGB_to_AP1;
    float3_ working = {fmaxf(m[0] * color.x + m[1] * color.y + m[2] * color.z, 0.0f), fmaxf(m[3] * color.x + m[4] * color.y + m[5] * color.z, 0.0f),
                       fmaxf(m[6] * color.x + m[7] * color.y + m[8] * color.z, 0.0f)};
    const float pre_luminance = working.x * 0.2722287168f + working.y * 0.6740817658f + working.z * 0.0536895174f;
    working = lerp3({pre_luminance, pre_luminance, pre_luminance}, working, 0.96f);

    const float toe_scale = c.toe_scale, shoulder_scale = c.shoulder_scale, toe_match = c.toe_match, shoulder_match = c.shoulder_match;

    float tone[3];
    const float channels[3] = {working.x, working.y, working.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float log_color = log10f(channels[i]);
        const float toe_color = (-c.black_clip) + (2 * toe_scale) / (1 + expf((-2 * c.slope / toe_scale) * (log_color - toe_match)));
        const float shoulder_color = (1 + c.white_clip) - (2 * shoulder_scale) / (1 + expf((2 * c.slope / shoulder_scale) * (log_color - shoulder_match)));
        float t = saturate((log_color - toe_match) / (shoulder_match - toe_match));
        t = shoulder_match < toe_match ? 1.0f - t : t;
        t = (3.0f - t * 2.0f) * t * t;
        tone[i] = toe_color + t * (shoulder_color - toe_color);
    }
    const float post_luminance = tone[0] * 0.2722287168f + tone[1] * 0.6740817658f + tone[2] * 0.0536895174f;
    const float3_ tone_color = lerp3({post_luminance, post_luminance, post_luminance}, {tone[0], tone[1], tone[2]}, 0.93f);
    const float r = fmaxf(tone_color.x, 0.0f), g = fmaxf(tone_color.y, 0.0f), b = fmaxf(tone_color.z, 0.0f);
    return {1.70479095f * r + -0.621689737f * g + -0.0832421705f * b, -0.130263522f * r + 1.14082849f * g + -0.0105496496f * b,
            -0.0240088310f * r + -0.128999621f * g + 1.15324795f * b};
}

__device__ float3_ agx(float3_ color) {     // Tonemapping.hlsl:111-142
    float c[3] = {0.842479062253094f * color.x + 0.0784335999999992f * color.y + 0.0792237451477643f * color.z,
                  0.0423282422610123f * color.x + 0.878468636469772f * color.y + 0.0791661274605434f * color.z,
                  0.0423756549057051f * color.x + 0.0784336f * color.y + 0.879142973793104f * color.z};
    const float min_exposure_value = -12.47393f, max_exposure_value = 4.026069f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float v = saturate((log2f(c[i]) - min_exposure_value) / (max_exposure_value - min_exposure_value));
        c[i] = -0.00232f + v * (0.1191f + v * (0.4298f + v * (-6.868f + v * (31.96f + v * (-40.14f + v * 15.5f)))));
    }
    const float r = 1.19687900512017f * c[0] + -0.0980208811401368f * c[1] + -0.0990297440797205f * c[2];
    const float g = -0.0528968517574562f * c[0] + 1.15190312990417f * c[1] + -0.0989611768448433f * c[2];
    const float b = -0.0529716355144438f * c[0] + -0.0980434501171241f * c[1] + 1.15107367264116f * c[2];
    return {powf(fabsf(r), 2.2f), powf(fabsf(g), 2.2f), powf(fabsf(b), 2.2f)};
}

__device__ float3_ khronos_neutral(float3_ c) {     // Tonemapping.hlsl:144-160
    const float start_compression = 0.8f - 0.04f, desaturation = 0.15f;
    const float x = fminf(c.x, fminf(c.y, c.z));
    const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    c = {c.x - offset, c.y - offset, c.z - offset};
    const float peak = fmaxf(c.x, fmaxf(c.y, c.z));
    if (peak < start_compression) return c;
    const float d = 1.0f - start_compression;
    const float new_peak = 1.0f - d * d / (peak + d - start_compression);
    const float scale = new_peak / peak;
    c = {c.x * scale, c.y * scale, c.z * scale};
    const float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    return lerp3(c, {new_peak, new_peak, new_peak}, g);
}

__device__ __forceinline__ float linear_to_sRGB(float v) { return v < 0.0031308f ? v * 12.92f : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f; }

// postprocess_pixel, Tonemapping.hlsl:205-227. One thread per viewport pixel, rows of 64 pixels per wave.
template <int MODE>
__global__ __launch_bounds__(256) void k_tonemap(DeviceFrame frame, const uint2* __restrict__ bloom, const float* __restrict__ linear_exposure_buffer, TonemapConstants c, void* __restrict__ target,
                                                 int target_format, uint32_t target_pitch, int32_t target_x, int32_t target_y) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= frame.width || y >= frame.height) return;

    const float linear_exposure = linear_exposure_buffer[0];
    const float3_ pixel = unpack_rgb(frame.pixels[uint32_t(x + frame.x) + size_t(y + frame.y) * frame.pitch]);
    float3_ color = {fminf(pixel.x, c.bloom_threshold), fminf(pixel.y, c.bloom_threshold), fminf(pixel.z, c.bloom_threshold)};
    if (bloom) {
        const float3_ bloom_color = unpack_rgb(bloom[uint32_t(x) + size_t(y) * uint32_t(frame.width)]);
        color = {color.x + bloom_color.x, color.y + bloom_color.y, color.z + bloom_color.z};
    }
    color = {color.x * linear_exposure, color.y * linear_exposure, color.z * linear_exposure};

    // Vignette (Tonemapping.hlsl:166-170), with the shader's uv: pixel index over the viewport size, no half-pixel offset.
    const float u = float(x) / float(frame.width), v = float(y) / float(frame.height);
    {
        const float cx = u - 0.5f, cy = v - 0.5f;
        const float t = saturate((sqrtf(cx * cx + cy * cy) * 1.5f * c.vignette_strength - 0.1f) / (0.9f - 0.1f));
        const float tint = 1.0f - t * t * (3.0f - 2.0f * t);
        color = {color.x * tint, color.y * tint, color.z * tint};
    }

    if (MODE == HIPR_TONEMAPPING_FILMIC) color = unreal4(color, c);
    else if (MODE == HIPR_TONEMAPPING_AGX) color = agx(color);
    else if (MODE == HIPR_TONEMAPPING_KHRONOS_NEUTRAL) color = khronos_neutral(color);

    {   // Film grain (Tonemapping.hlsl:176-180)
        const float s = sinf((u + c.delta_time) * 12.9898f + (v + c.delta_time) * 78.233f) * 43758.5453f;
        const float grain = c.film_grain_strength * ((s - floorf(s)) - 0.5f);
        color = {color.x + grain, color.y + grain, color.z + grain};
    }

    const size_t index = uint32_t(x + target_x) + size_t(y + target_y) * target_pitch;
    if (target_format == HIPR_TARGET_RGBA16F) static_cast<uint2*>(target)[index] = pack_rgba(color.x, color.y, color.z, 1.0f);
    else if (target_format == HIPR_TARGET_RGBA32F) static_cast<float4*>(target)[index] = make_float4(color.x, color.y, color.z, 1.0f);
    else {
        const uint32_t r = uint32_t(saturate(linear_to_sRGB(saturate(color.x))) * 255.0f + 0.5f), g = uint32_t(saturate(linear_to_sRGB(saturate(color.y))) * 255.0f + 0.5f),
                       b = uint32_t(saturate(linear_to_sRGB(saturate(color.z))) * 255.0f + 0.5f);
        static_cast<uint32_t*>(target)[index] = r | (g << 8) | (b << 16) | (255u << 24);
    }
}

} // namespace hipr_camera_effects

// =====================================================================================================================================
// Host side: the C-ABI object
// =====================================================================================================================================
using namespace hipr_camera_effects;

struct HiprCameraEffects {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    std::string last_error;

    uint32_t* histogram = nullptr;          // 64 bins
    uint32_t* block_histograms = nullptr;   // MAX_PARTIALS x 64 bins, one row per block of the histogram kernel
    unsigned histogram_blocks = 0;
    float* linear_exposure = nullptr;       // carried from frame to frame
    float* partials = nullptr;              // MAX_PARTIALS
    float* scratch_scalar = nullptr;

    __half2* taps = nullptr;                // Gaussian taps, refilled when the standard deviation changes (CameraEffects.cpp:51-76)
    int taps_capacity = 0;
    float taps_std_dev = INFINITY;

    uint2* ping = nullptr; uint2* pong = nullptr;   // viewport-sized half4 intermediates, grow only (CameraEffects.cpp:78-98)
    uint2* kawase_chain = nullptr;                   // the levels of the dual Kawase filter back to back, grow only (CameraEffects.cpp:150-199)
    size_t kawase_capacity = 0;
    size_t intermediate_pixels = 0;

    bool direct_bloom = false;              // HIPR_BLOOM_DIRECT=1: the kernels without LDS staging (comparison runs)
    bool instrument = false;
    hipEvent_t event_begin = nullptr, event_end = nullptr;
    HiprCameraEffectsTimes times = {};
    unsigned compute_units = 256;
};

namespace {

int fail(HiprCameraEffects* fx, int status, const std::string& message) {
    if (fx) fx->last_error = message;
    return status;
}
int fail_hip(HiprCameraEffects* fx, hipError_t error, const char* what) { return fail(fx, HIPR_ERROR_HIP, std::string(what) + ": " + hipGetErrorString(error)); }

#define FX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip(fx, e_, #call); } while (0)

bool valid_frame(const HiprFrameView* frame) {
    return frame && frame->pixels && frame->pitch > 0 && frame->rows > 0 && frame->viewport.width > 0 && frame->viewport.height > 0 && frame->viewport.x >= 0 && frame->viewport.y >= 0 &&
           uint32_t(frame->viewport.x + frame->viewport.width) <= frame->pitch && uint32_t(frame->viewport.y + frame->viewport.height) <= frame->rows;
}
DeviceFrame device_frame(const HiprFrameView& f) { return {static_cast<const uint2*>(f.pixels), f.pitch, f.rows, f.viewport.x, f.viewport.y, f.viewport.width, f.viewport.height}; }

ExposureConstants exposure_constants(const HiprCameraEffectsSettings& s, float delta_time) {
    ExposureConstants c;
    c.min_log_luminance = s.min_log_luminance; c.max_log_luminance = s.max_log_luminance;
    c.min_percentage = s.min_histogram_percentage; c.max_percentage = s.max_histogram_percentage;
    c.log_luminance_bias = s.log_luminance_bias;
    c.eye_adaptation_brightness = s.eye_adaptation_enabled ? s.eye_adaptation_brightness : INFINITY;     // CameraEffects.cpp:432-436
    c.eye_adaptation_darkness = s.eye_adaptation_enabled ? s.eye_adaptation_darkness : INFINITY;
    c.delta_time = delta_time;
    return c;
}

unsigned reduce_blocks(const HiprCameraEffects* fx, const HiprFrameView& frame) {
    const uint64_t pixels = uint64_t(frame.viewport.width) * uint64_t(frame.viewport.height);
    const uint64_t wanted = (pixels + REDUCE_BLOCK * 4 - 1) / (REDUCE_BLOCK * 4);       // at least four pixels per thread before another block pays for itself
    return unsigned(std::min<uint64_t>(std::max<uint64_t>(wanted, 1), std::min<uint64_t>(MAX_PARTIALS, fx->compute_units * 2ull)));
}

struct StageTimer {
    HiprCameraEffects* fx; float* ms; uint32_t* launches;
    StageTimer(HiprCameraEffects* fx, float* ms, uint32_t* launches) : fx(fx), ms(ms), launches(launches) {
        if (fx->instrument) (void)hipEventRecord(fx->event_begin, fx->stream);
    }
    ~StageTimer() {
        if (!fx->instrument) return;
        (void)hipEventRecord(fx->event_end, fx->stream);
        (void)hipEventSynchronize(fx->event_end);
        float elapsed = 0.0f;
        if (hipEventElapsedTime(&elapsed, fx->event_begin, fx->event_end) == hipSuccess) { *ms += elapsed; ++*launches; }
    }
};

int enqueue_histogram(HiprCameraEffects* fx, const HiprCameraEffectsSettings& s, const HiprFrameView& frame) {
    fx->histogram_blocks = reduce_blocks(fx, frame);
    hipLaunchKernelGGL(k_exposure_histogram, dim3(fx->histogram_blocks), dim3(REDUCE_BLOCK), 0, fx->stream, device_frame(frame), s.min_log_luminance, s.max_log_luminance, fx->block_histograms);
    FX_HIP(hipGetLastError());
    return HIPR_OK;
}

int enqueue_log_average(HiprCameraEffects* fx, const HiprFrameView& frame, int mode, const ExposureConstants& c, float* out) {
    const unsigned blocks = reduce_blocks(fx, frame);
    hipLaunchKernelGGL(k_log_luminance_partials, dim3(blocks), dim3(REDUCE_BLOCK), 0, fx->stream, device_frame(frame), fx->partials);
    hipLaunchKernelGGL(k_log_average_finish, dim3(1), dim3(REDUCE_BLOCK), 0, fx->stream, fx->partials, blocks, float(frame.viewport.width) * float(frame.viewport.height), mode, c, out);
    FX_HIP(hipGetLastError());
    return HIPR_OK;
}

// Bifrost/Math/Utils.h:286-312 fill_bilinear_gaussian_samples, stored as half2 like the reference's R16G16_FLOAT buffer.
int prepare_taps(HiprCameraEffects* fx, int support) {
    const int needed = (support + 1) / 2;
    if (needed > fx->taps_capacity) {
        int capacity = 64;
        while (capacity < needed) capacity <<= 1;
        FX_HIP(hipStreamSynchronize(fx->stream));
        if (fx->taps) (void)hipFree(fx->taps);
        fx->taps = nullptr;
        FX_HIP(hipMalloc(&fx->taps, sizeof(__half2) * capacity));
        fx->taps_capacity = capacity;
        fx->taps_std_dev = INFINITY;
    }
    const float std_dev = support * 0.25f;
    if (fx->taps_std_dev == std_dev) return HIPR_OK;

    const int count = fx->taps_capacity;
    std::vector<float> offsets(count), weights(count);
    const float double_variance = 2.0f * std_dev * std_dev;
    float total_weight = 0.0f;
    for (int s = 0; s < count; ++s) {
        const int t1 = s * 2, t2 = t1 + 1;
        float w1 = std::exp(-(t1 * t1) / double_variance);
        if (s == 0) w1 *= 0.5f;
        const float w2 = std::exp(-(t2 * t2) / double_variance);
        const float weight = w1 + w2;
        float offset = (t1 * w1 + t2 * w2) / weight;
        if (std::isnan(offset)) offset = float(t1);
        offsets[s] = offset; weights[s] = weight;
        total_weight += weight;
    }
    total_weight *= 2;      // the table holds one half of the symmetric bell
    std::vector<__half2> packed(count);
    for (int s = 0; s < count; ++s) packed[s] = __floats2half2_rn(offsets[s], weights[s] / total_weight);
    FX_HIP(hipStreamSynchronize(fx->stream));       // the previous table may still be read
    FX_HIP(hipMemcpy(fx->taps, packed.data(), sizeof(__half2) * count, hipMemcpyHostToDevice));
    fx->taps_std_dev = std_dev;
    return HIPR_OK;
}

int ensure_intermediates(HiprCameraEffects* fx, size_t pixels) {
    if (pixels <= fx->intermediate_pixels) return HIPR_OK;
    FX_HIP(hipStreamSynchronize(fx->stream));
    if (fx->ping) (void)hipFree(fx->ping);
    if (fx->pong) (void)hipFree(fx->pong);
    fx->ping = fx->pong = nullptr; fx->intermediate_pixels = 0;
    FX_HIP(hipMalloc(&fx->ping, pixels * sizeof(uint2)));
    FX_HIP(hipMalloc(&fx->pong, pixels * sizeof(uint2)));
    fx->intermediate_pixels = pixels;
    return HIPR_OK;
}

// out == nullptr: the result stays in fx->ping.
int enqueue_bloom(HiprCameraEffects* fx, float threshold, int support, const HiprFrameView& frame, uint2* out) {
    int status = prepare_taps(fx, support);
    if (status != HIPR_OK) return status;
    const size_t pixels = size_t(frame.viewport.width) * size_t(frame.viewport.height);
    status = ensure_intermediates(fx, pixels);
    if (status != HIPR_OK) return status;
    const dim3 grid((frame.viewport.width + 63) / 64, (frame.viewport.height + 3) / 4), block(256);
    const int sample_count = support / 2;
    // LDS the tiled kernels stage (their 64 KB default limit decides; wider filters fall back to the direct kernels)
    const size_t horizontal_lds = size_t(4) * (64 + 4 * sample_count) * sizeof(uint2), vertical_lds = size_t(32 + 4 * sample_count) * 32 * sizeof(uint2);
    const bool tiled = sample_count > 0 && !fx->direct_bloom && horizontal_lds <= 64 * 1024 && vertical_lds <= 64 * 1024;
    {
        StageTimer timer(fx, &fx->times.bloom_horizontal_ms, &fx->times.bloom_horizontal_launches);
        if (tiled) hipLaunchKernelGGL(k_bloom_horizontal_tiled, grid, block, horizontal_lds, fx->stream, device_frame(frame), fx->taps, sample_count, threshold, fx->pong);
        else hipLaunchKernelGGL(k_bloom_horizontal, grid, block, 0, fx->stream, device_frame(frame), fx->taps, sample_count, threshold, fx->pong);
    }
    {
        StageTimer timer(fx, &fx->times.bloom_vertical_ms, &fx->times.bloom_vertical_launches);
        const dim3 tile_grid((frame.viewport.width + 31) / 32, (frame.viewport.height + 31) / 32);
        if (tiled) hipLaunchKernelGGL(k_bloom_vertical_tiled, tile_grid, block, vertical_lds, fx->stream, fx->pong, frame.viewport.width, frame.viewport.height, fx->taps, sample_count, out ? out : fx->ping);
        else hipLaunchKernelGGL(k_bloom_vertical, grid, block, 0, fx->stream, fx->pong, frame.viewport.width, frame.viewport.height, fx->taps, sample_count, out ? out : fx->ping);
    }
    FX_HIP(hipGetLastError());
    return HIPR_OK;
}

int enqueue_dual_kawase(HiprCameraEffects* fx, float threshold, uint32_t half_passes, const HiprFrameView& frame, uint2* out) {
    const int width = frame.viewport.width, height = frame.viewport.height;
    uint32_t level_count = 1;      // levels until both dimensions have collapsed (CameraEffects.cpp:166-168)
    while ((width >> level_count) > 0 || (height >> level_count) > 0) ++level_count;
    half_passes = std::min(half_passes, level_count - 1);
    std::vector<size_t> offset(half_passes + 2, 0);
    auto level_width = [&](uint32_t m) { return std::max(1, width >> m); };
    auto level_height = [&](uint32_t m) { return std::max(1, height >> m); };
    for (uint32_t m = 0; m <= half_passes; ++m) offset[m + 1] = offset[m] + size_t(level_width(m)) * level_height(m);
    if (offset[half_passes + 1] > fx->kawase_capacity) {
        FX_HIP(hipStreamSynchronize(fx->stream));
        if (fx->kawase_chain) (void)hipFree(fx->kawase_chain);
        fx->kawase_chain = nullptr; fx->kawase_capacity = 0;
        FX_HIP(hipMalloc(&fx->kawase_chain, offset[half_passes + 1] * sizeof(uint2)));
        fx->kawase_capacity = offset[half_passes + 1];
    }
    auto level = [&](uint32_t m) { return m == 0 && half_passes == 0 ? out : fx->kawase_chain + offset[m]; };
    auto grid_of = [](int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); };
    const dim3 block(256);
    hipLaunchKernelGGL(k_kawase_extract, grid_of(width, height), block, 0, fx->stream, device_frame(frame), threshold, level(0));
    for (uint32_t p = 0; p < half_passes; ++p)
        hipLaunchKernelGGL(k_kawase_downsample, grid_of(level_width(p + 1), level_height(p + 1)), block, 0, fx->stream, level(p), level_width(p), level_height(p), level(p + 1), level_width(p + 1), level_height(p + 1));
    for (uint32_t p = half_passes; p > 0; --p)
        hipLaunchKernelGGL(k_kawase_upsample, grid_of(level_width(p - 1), level_height(p - 1)), block, 0, fx->stream, level(p), level_width(p), level_height(p), p == 1 ? out : level(p - 1), level_width(p - 1), level_height(p - 1));
    FX_HIP(hipGetLastError());
    return HIPR_OK;
}

void multiply3x3(const float a[9], const float b[9], float out[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

} // namespace

extern "C" {

int hipr_camera_effects_create(int device_index, HiprCameraEffects** out) {
    if (!out) return HIPR_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_index < 0 || device_index >= device_count) return HIPR_ERROR_NO_DEVICE;
    HiprCameraEffects* fx = new HiprCameraEffects();
    fx->device = device_index;
    auto cleanup = [&](int status) { hipr_camera_effects_destroy(fx); return status; };
    if (hipSetDevice(device_index) != hipSuccess) return cleanup(HIPR_ERROR_NO_DEVICE);
    hipDeviceProp_t properties;
    if (hipGetDeviceProperties(&properties, device_index) == hipSuccess && properties.multiProcessorCount > 0) fx->compute_units = unsigned(properties.multiProcessorCount);
    if (hipStreamCreateWithFlags(&fx->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(HIPR_ERROR_HIP);
    if (const char* direct = getenv("HIPR_BLOOM_DIRECT")) fx->direct_bloom = atoi(direct) != 0;
    if (hipMalloc(&fx->histogram, BINS * sizeof(uint32_t)) != hipSuccess || hipMalloc(&fx->linear_exposure, sizeof(float)) != hipSuccess ||
        hipMalloc(&fx->partials, MAX_PARTIALS * sizeof(float)) != hipSuccess || hipMalloc(&fx->block_histograms, MAX_PARTIALS * BINS * sizeof(uint32_t)) != hipSuccess || hipMalloc(&fx->scratch_scalar, sizeof(float)) != hipSuccess)
        return cleanup(HIPR_ERROR_OUT_OF_MEMORY);
    if (hipMemset(fx->linear_exposure, 0, sizeof(float)) != hipSuccess) return cleanup(HIPR_ERROR_HIP);
    if (hipEventCreate(&fx->event_begin) != hipSuccess || hipEventCreate(&fx->event_end) != hipSuccess) return cleanup(HIPR_ERROR_HIP);
    *out = fx;
    return HIPR_OK;
}

void hipr_camera_effects_destroy(HiprCameraEffects* fx) {
    if (!fx) return;
    (void)hipSetDevice(fx->device);
    if (fx->stream) (void)hipStreamSynchronize(fx->stream);
    for (void* p : {(void*)fx->histogram, (void*)fx->block_histograms, (void*)fx->linear_exposure, (void*)fx->partials, (void*)fx->scratch_scalar, (void*)fx->taps, (void*)fx->ping, (void*)fx->pong, (void*)fx->kawase_chain})
        if (p) (void)hipFree(p);
    if (fx->event_begin) (void)hipEventDestroy(fx->event_begin);
    if (fx->event_end) (void)hipEventDestroy(fx->event_end);
    if (fx->stream && fx->owns_stream) (void)hipStreamDestroy(fx->stream);
    delete fx;
}

const char* hipr_camera_effects_last_error(const HiprCameraEffects* fx) { return fx ? fx->last_error.c_str() : "null camera effects object"; }

int hipr_camera_effects_set_stream(HiprCameraEffects* fx, void* hip_stream) {
    if (!fx) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_set_stream: null object");      // a null stream is the device's default stream (what a host framework's "current stream" usually is)
    FX_HIP(hipStreamSynchronize(fx->stream));
    if (fx->owns_stream) (void)hipStreamDestroy(fx->stream);
    fx->stream = static_cast<hipStream_t>(hip_stream);
    fx->owns_stream = false;
    return HIPR_OK;
}

int hipr_camera_effects_synchronize(HiprCameraEffects* fx) {
    if (!fx) return HIPR_ERROR_INVALID_ARGUMENT;
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_get_linear_exposure(HiprCameraEffects* fx, float* out_host) {
    if (!fx || !out_host) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_get_linear_exposure: null argument");
    FX_HIP(hipMemcpyAsync(out_host, fx->linear_exposure, sizeof(float), hipMemcpyDeviceToHost, fx->stream));
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_set_linear_exposure(HiprCameraEffects* fx, float linear_exposure) {
    if (!fx) return HIPR_ERROR_INVALID_ARGUMENT;
    FX_HIP(hipMemcpyAsync(fx->linear_exposure, &linear_exposure, sizeof(float), hipMemcpyHostToDevice, fx->stream));
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_reduce_histogram(HiprCameraEffects* fx, const HiprCameraEffectsSettings* settings, const HiprFrameView* frame, uint32_t* out_histogram_host) {
    if (!fx || !settings || !out_histogram_host || !valid_frame(frame)) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_reduce_histogram: null argument or viewport outside the frame");
    FX_HIP(hipSetDevice(fx->device));
    const int status = enqueue_histogram(fx, *settings, *frame);
    if (status != HIPR_OK) return status;
    hipLaunchKernelGGL(k_sum_histograms, dim3(1), dim3(SUM_BLOCK), 0, fx->stream, fx->block_histograms, fx->histogram_blocks, fx->histogram);
    FX_HIP(hipGetLastError());
    FX_HIP(hipMemcpyAsync(out_histogram_host, fx->histogram, BINS * sizeof(uint32_t), hipMemcpyDeviceToHost, fx->stream));
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_exposure_from_histogram(HiprCameraEffects* fx, const HiprCameraEffectsSettings* settings, float delta_time, const uint32_t* histogram_host, float* io_linear_exposure_host) {
    if (!fx || !settings || !histogram_host || !io_linear_exposure_host) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_exposure_from_histogram: null argument");
    FX_HIP(hipSetDevice(fx->device));
    FX_HIP(hipMemcpyAsync(fx->histogram, histogram_host, BINS * sizeof(uint32_t), hipMemcpyHostToDevice, fx->stream));
    FX_HIP(hipMemcpyAsync(fx->scratch_scalar, io_linear_exposure_host, sizeof(float), hipMemcpyHostToDevice, fx->stream));
    hipLaunchKernelGGL(k_exposure_from_histogram, dim3(1), dim3(SUM_BLOCK), 0, fx->stream, fx->histogram, 1u, exposure_constants(*settings, delta_time), fx->scratch_scalar);
    FX_HIP(hipGetLastError());
    FX_HIP(hipMemcpyAsync(io_linear_exposure_host, fx->scratch_scalar, sizeof(float), hipMemcpyDeviceToHost, fx->stream));
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_log_average(HiprCameraEffects* fx, const HiprFrameView* frame, float* out_log_average_host) {
    if (!fx || !out_log_average_host || !valid_frame(frame)) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_log_average: null argument or viewport outside the frame");
    FX_HIP(hipSetDevice(fx->device));
    const int status = enqueue_log_average(fx, *frame, 0, ExposureConstants{}, fx->scratch_scalar);
    if (status != HIPR_OK) return status;
    FX_HIP(hipMemcpyAsync(out_log_average_host, fx->scratch_scalar, sizeof(float), hipMemcpyDeviceToHost, fx->stream));
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_exposure_from_log_average(HiprCameraEffects* fx, const HiprCameraEffectsSettings* settings, float delta_time, const HiprFrameView* frame, float* io_linear_exposure_host) {
    if (!fx || !settings || !io_linear_exposure_host || !valid_frame(frame))
        return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_exposure_from_log_average: null argument or viewport outside the frame");
    FX_HIP(hipSetDevice(fx->device));
    FX_HIP(hipMemcpyAsync(fx->scratch_scalar, io_linear_exposure_host, sizeof(float), hipMemcpyHostToDevice, fx->stream));
    const int status = enqueue_log_average(fx, *frame, 1, exposure_constants(*settings, delta_time), fx->scratch_scalar);
    if (status != HIPR_OK) return status;
    FX_HIP(hipMemcpyAsync(io_linear_exposure_host, fx->scratch_scalar, sizeof(float), hipMemcpyDeviceToHost, fx->stream));
    FX_HIP(hipStreamSynchronize(fx->stream));
    return HIPR_OK;
}

int hipr_camera_effects_bloom(HiprCameraEffects* fx, float threshold, int32_t support, const HiprFrameView* frame, void* out_half4_device) {
    if (!fx || !out_half4_device || !valid_frame(frame) || support < 0) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_bloom: null argument, negative support or viewport outside the frame");
    FX_HIP(hipSetDevice(fx->device));
    return enqueue_bloom(fx, threshold, support, *frame, static_cast<uint2*>(out_half4_device));
}

int hipr_camera_effects_dual_kawase_bloom(HiprCameraEffects* fx, float threshold, uint32_t half_passes, const HiprFrameView* frame, void* out_half4_device) {
    if (!fx || !out_half4_device || !valid_frame(frame)) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_dual_kawase_bloom: null argument or viewport outside the frame");
    FX_HIP(hipSetDevice(fx->device));
    return enqueue_dual_kawase(fx, threshold, half_passes, *frame, static_cast<uint2*>(out_half4_device));
}

int hipr_camera_effects_process(HiprCameraEffects* fx, const HiprCameraEffectsSettings* settings, float delta_time, const HiprFrameView* frame, void* target, int target_format,
                                uint32_t target_pitch, uint32_t target_rows, int32_t target_viewport_x, int32_t target_viewport_y) {
    if (!fx || !settings || !target || !valid_frame(frame)) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_process: null argument or viewport outside the frame");
    if (target_format != HIPR_TARGET_RGBA16F && target_format != HIPR_TARGET_RGBA32F && target_format != HIPR_TARGET_RGBA8_SRGB)
        return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_process: unknown target format");
    if (target_viewport_x < 0 || target_viewport_y < 0 || uint32_t(target_viewport_x + frame->viewport.width) > target_pitch || uint32_t(target_viewport_y + frame->viewport.height) > target_rows)
        return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_process: the viewport does not fit the target");
    const HiprCameraEffectsSettings& s = *settings;
    if (s.exposure_mode < HIPR_EXPOSURE_FIXED || s.exposure_mode > HIPR_EXPOSURE_HISTOGRAM || s.tonemapping_mode < HIPR_TONEMAPPING_LINEAR || s.tonemapping_mode > HIPR_TONEMAPPING_KHRONOS_NEUTRAL)
        return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_process: unknown exposure or tonemapping mode");
    FX_HIP(hipSetDevice(fx->device));

    const ExposureConstants exposure = exposure_constants(s, delta_time);
    {   // Determine exposure (CameraEffects.cpp:456-469).
        StageTimer timer(fx, &fx->times.exposure_ms, &fx->times.exposure_launches);
        if (s.exposure_mode == HIPR_EXPOSURE_HISTOGRAM) {
            const int status = enqueue_histogram(fx, s, *frame);
            if (status != HIPR_OK) return status;
            hipLaunchKernelGGL(k_exposure_from_histogram, dim3(1), dim3(SUM_BLOCK), 0, fx->stream, fx->block_histograms, fx->histogram_blocks, exposure, fx->linear_exposure);
        } else if (s.exposure_mode == HIPR_EXPOSURE_LOG_AVERAGE) {
            const int status = enqueue_log_average(fx, *frame, 1, exposure, fx->linear_exposure);
            if (status != HIPR_OK) return status;
        } else
            hipLaunchKernelGGL(k_exposure_from_bias, dim3(1), dim3(1), 0, fx->stream, exposure, fx->linear_exposure);
    }

    // Bloom filter (CameraEffects.cpp:471-476).
    const uint2* bloom = nullptr;
    if (s.bloom_threshold < INFINITY) {
        const int support = int(s.bloom_support * frame->viewport.height);
        const int status = enqueue_bloom(fx, s.bloom_threshold, support, *frame, nullptr);
        if (status != HIPR_OK) return status;
        bloom = fx->ping;
    }

    TonemapConstants constants;
    constants.bloom_threshold = s.bloom_threshold; constants.vignette_strength = s.vignette; constants.film_grain_strength = s.film_grain; constants.delta_time = delta_time;
    constants.black_clip = s.tonemapping_black_clip; constants.toe = s.tonemapping_toe; constants.slope = s.tonemapping_slope; constants.shoulder = s.tonemapping_shoulder;
    constants.white_clip = s.tonemapping_white_clip;
    {   // Tonemapping.hlsl:79-94
        constants.toe_scale = 1.0f + constants.black_clip - constants.toe;
        constants.shoulder_scale = 1.0f + constants.white_clip - constants.shoulder;
        const float in_match = 0.18f, out_match = 0.18f;
        if (constants.toe > 0.8f)
            constants.toe_match = (1.0f - constants.toe - out_match) / constants.slope + std::log10(in_match);
        else {
            const float bt = (out_match + constants.black_clip) / constants.toe_scale - 1.0f;
            constants.toe_match = std::log10(in_match) - 0.5f * std::log((1.0f + bt) / (1.0f - bt)) * (constants.toe_scale / constants.slope);
        }
        const float straight_match = (1.0f - constants.toe) / constants.slope - constants.toe_match;
        constants.shoulder_match = constants.shoulder / constants.slope - straight_match;
    }
    {
        const float D65_to_D60[9] = {1.01303f, 0.00610531f, -0.014971f, 0.00769823f, 0.998165f, -0.00503203f, -0.00284131f, 0.00468516f, 0.924507f};
        const float sRGB_to_XYZ[9] = {0.4124564f, 0.3575761f, 0.1804375f, 0.2126729f, 0.7151522f, 0.0721750f, 0.0193339f, 0.1191920f, 0.9503041f};
        const float XYZ_to_AP1[9] = {1.6410233797f, -0.3248032942f, -0.2364246952f, -0.6636628587f, 1.6153315917f, 0.0167563477f, 0.0117218943f, -0.0082844420f, 0.9883948585f};
        float adapted[9];
        multiply3x3(D65_to_D60, sRGB_to_XYZ, adapted);
        multiply3x3(XYZ_to_AP1, adapted, constants.sRGB_to_AP1);
    }

    const DeviceFrame device = device_frame(*frame);
    const dim3 grid((frame->viewport.width + 63) / 64, (frame->viewport.height + 3) / 4), block(256);
    {
        StageTimer timer(fx, &fx->times.tonemap_ms, &fx->times.tonemap_launches);
        switch (s.tonemapping_mode) {
        case HIPR_TONEMAPPING_FILMIC:
            hipLaunchKernelGGL(k_tonemap<HIPR_TONEMAPPING_FILMIC>, grid, block, 0, fx->stream, device, bloom, fx->linear_exposure, constants, target, target_format, target_pitch, target_viewport_x, target_viewport_y);
            break;
        case HIPR_TONEMAPPING_AGX:
            hipLaunchKernelGGL(k_tonemap<HIPR_TONEMAPPING_AGX>, grid, block, 0, fx->stream, device, bloom, fx->linear_exposure, constants, target, target_format, target_pitch, target_viewport_x, target_viewport_y);
            break;
        case HIPR_TONEMAPPING_KHRONOS_NEUTRAL:
            hipLaunchKernelGGL(k_tonemap<HIPR_TONEMAPPING_KHRONOS_NEUTRAL>, grid, block, 0, fx->stream, device, bloom, fx->linear_exposure, constants, target, target_format, target_pitch, target_viewport_x, target_viewport_y);
            break;
        default:
            hipLaunchKernelGGL(k_tonemap<HIPR_TONEMAPPING_LINEAR>, grid, block, 0, fx->stream, device, bloom, fx->linear_exposure, constants, target, target_format, target_pitch, target_viewport_x, target_viewport_y);
            break;
        }
    }
    FX_HIP(hipGetLastError());
    return HIPR_OK;
}

int hipr_camera_effects_set_instrumentation(HiprCameraEffects* fx, int time_stages) {
    if (!fx) return HIPR_ERROR_INVALID_ARGUMENT;
    fx->instrument = time_stages != 0;
    return HIPR_OK;
}

int hipr_camera_effects_reset_timers(HiprCameraEffects* fx) {
    if (!fx) return HIPR_ERROR_INVALID_ARGUMENT;
    fx->times = HiprCameraEffectsTimes{};
    return HIPR_OK;
}

int hipr_camera_effects_get_times(HiprCameraEffects* fx, HiprCameraEffectsTimes* out) {
    if (!fx || !out) return fail(fx, HIPR_ERROR_INVALID_ARGUMENT, "hipr_camera_effects_get_times: null argument");
    *out = fx->times;
    return HIPR_OK;
}

} // extern "C"
