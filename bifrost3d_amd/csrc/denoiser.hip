// denoiser.hip -- the filter stage of the denoising backend: kernels and the C-ABI of include/hipr_denoiser_c.h. Written for gfx950.
// Stands where the reference's AIDenoisedBackend runs NVIDIA's closed DLDenoiser stage (extensions/OptiXRenderer/OptiXRenderer/
// IBackend.cpp:26-31, 48-80); the filter is the open edge-avoiding a-trous wavelet transform the header describes.
//
//   k_denoiser_prepare<Source>   noisy / albedo' -> float4 (demodulated rgb, log2(1 + luminance)); albedo -> float4
//   k_denoiser_pass              one a-trous pass: 25 taps at stride 2^i, B3 spline x albedo edge stop x log-luminance edge stop
//   k_denoiser_finish            re-modulate by albedo' -> the filtered float4 image the object keeps
//   k_denoiser_output            filtered / noisy / albedo -> half4 with the caller's pitch (AIDenoiser::copy_to_output, ORS/SimpleRGPs.cu:203-219)
// All of it is image work on 16 B / pixel planes: one pass reads 25 x 2 float4 per pixel, nearly all from L1 / L2 (neighbouring
// pixels share 20 of their 25 taps at step 1, and a 1080p plane is 33 MB against 32 MB of L2 + 256 MB of MALL); the roofline that
// bounds it is HBM: 2 planes read + 1 written per pass = 48 B / pixel / pass.
#include "../../include/hipr_denoiser_c.h"

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

namespace hipr_denoiser {

constexpr int TILE_X = 32, TILE_Y = 8;   // one wave covers 32 x 2 pixels: two full 512 B rows of a float4 plane per load

__device__ __forceinline__ float luminance(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
__device__ __forceinline__ float4 unpack_half4(uint2 p) {
    const __half2 lo = *reinterpret_cast<const __half2*>(&p.x), hi = *reinterpret_cast<const __half2*>(&p.y);
    const float2 a = __half22float2(lo), b = __half22float2(hi);
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ uint2 pack_half4(float4 v) {
    const __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
    uint2 p;
    p.x = *reinterpret_cast<const uint32_t*>(&lo); p.y = *reinterpret_cast<const uint32_t*>(&hi);
    return p;
}

struct HalfSource {
    const uint2* pixels; uint32_t pitch;
    __device__ float4 load(uint32_t x, uint32_t y) const { return unpack_half4(pixels[size_t(y) * pitch + x]); }
};
struct FloatSource {
    const float4* pixels; uint32_t pitch;
    __device__ float4 load(uint32_t x, uint32_t y) const { return pixels[size_t(y) * pitch + x]; }
};

// albedo': the albedo where it can be divided by, 1 where there is none (emitters, misses, black surfaces)
__device__ __forceinline__ float demodulator(float albedo, float floor) { return albedo > floor ? albedo : 1.0f; }

template <typename Source>
__global__ __launch_bounds__(TILE_X * TILE_Y) void k_denoiser_prepare(Source noisy, Source albedo, uint32_t width, uint32_t height, float albedo_floor, float4* __restrict__ color,
                                                                       float4* __restrict__ feature) {
    const uint32_t x = blockIdx.x * TILE_X + threadIdx.x, y = blockIdx.y * TILE_Y + threadIdx.y;
    if (x >= width || y >= height) return;
    const float4 n = noisy.load(x, y), a = albedo.load(x, y);
    const float r = n.x / demodulator(a.x, albedo_floor), g = n.y / demodulator(a.y, albedo_floor), b = n.z / demodulator(a.z, albedo_floor);
    color[size_t(y) * width + x] = make_float4(r, g, b, log2f(1.0f + luminance(r, g, b)));
    feature[size_t(y) * width + x] = make_float4(a.x, a.y, a.z, 0.0f);
}

__global__ __launch_bounds__(TILE_X * TILE_Y) void k_denoiser_pass(const float4* __restrict__ color, const float4* __restrict__ feature, uint32_t width, uint32_t height, int step,
                                                                    float inverse_sigma_albedo_squared, float inverse_sigma_luminance, float4* __restrict__ out) {
    const uint32_t x = blockIdx.x * TILE_X + threadIdx.x, y = blockIdx.y * TILE_Y + threadIdx.y;
    if (x >= width || y >= height) return;
    const float4 cp = color[size_t(y) * width + x], ap = feature[size_t(y) * width + x];
    const float spline[3] = {0.375f, 0.25f, 0.0625f};
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
        const int qy = int(y) + dy * step;
        if (qy < 0 || qy >= int(height)) continue;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int qx = int(x) + dx * step;
            if (qx < 0 || qx >= int(width)) continue;
            const float4 cq = color[size_t(qy) * width + qx], aq = feature[size_t(qy) * width + qx];
            const float dr = ap.x - aq.x, dg = ap.y - aq.y, db = ap.z - aq.z;
            const float albedo_distance = dr * dr + dg * dg + db * db;
            const float w = spline[dx < 0 ? -dx : dx] * spline[dy < 0 ? -dy : dy] * expf(-albedo_distance * inverse_sigma_albedo_squared) *
                            expf(-fabsf(cp.w - cq.w) * inverse_sigma_luminance);
            sr += w * cq.x; sg += w * cq.y; sb += w * cq.z; sw += w;
        }
    }
    const float inverse = 1.0f / sw;   // the centre tap alone weighs 9 / 64
    const float r = sr * inverse, g = sg * inverse, b = sb * inverse;
    out[size_t(y) * width + x] = make_float4(r, g, b, log2f(1.0f + luminance(r, g, b)));
}

__global__ __launch_bounds__(TILE_X * TILE_Y) void k_denoiser_finish(const float4* __restrict__ color, const float4* __restrict__ feature, uint32_t width, uint32_t height, float albedo_floor,
                                                                      float4* __restrict__ filtered) {
    const uint32_t x = blockIdx.x * TILE_X + threadIdx.x, y = blockIdx.y * TILE_Y + threadIdx.y;
    if (x >= width || y >= height) return;
    const float4 c = color[size_t(y) * width + x], a = feature[size_t(y) * width + x];
    filtered[size_t(y) * width + x] = make_float4(c.x * demodulator(a.x, albedo_floor), c.y * demodulator(a.y, albedo_floor), c.z * demodulator(a.z, albedo_floor), 1.0f);
}

__global__ __launch_bounds__(TILE_X * TILE_Y) void k_denoiser_output(const float4* __restrict__ filtered, HalfSource noisy, HalfSource albedo, uint32_t width, uint32_t height, int show,
                                                                      uint2* __restrict__ out, uint32_t out_pitch) {
    const uint32_t x = blockIdx.x * TILE_X + threadIdx.x, y = blockIdx.y * TILE_Y + threadIdx.y;
    if (x >= width || y >= height) return;
    float4 pixel;
    if (show == HIPR_DENOISER_SHOW_NOISE) pixel = noisy.load(x, y);
    else if (show == HIPR_DENOISER_SHOW_ALBEDO) pixel = albedo.load(x, y);
    else pixel = filtered[size_t(y) * width + x];
    pixel.w = 1.0f;
    out[size_t(y) * out_pitch + x] = pack_half4(pixel);
}

} // namespace hipr_denoiser

using namespace hipr_denoiser;

struct HiprDenoiser {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    std::string last_error;
    float4 *ping = nullptr, *pong = nullptr, *feature = nullptr, *filtered = nullptr;   // width x height planes, grow only
    size_t capacity_pixels = 0;
    uint32_t filtered_width = 0, filtered_height = 0;                                   // frame size the filtered image is valid for
};

namespace {

int fail(HiprDenoiser* d, int status, const std::string& message) {
    if (d) d->last_error = message;
    return status;
}
#define DN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(d, HIPR_ERROR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

bool valid_settings(const HiprDenoiserSettings* s) {
    return s && s->iterations >= 1 && s->iterations <= 12 && s->sigma_albedo > 0.0f && s->sigma_luminance > 0.0f && s->albedo_floor >= 0.0f;
}

int reserve(HiprDenoiser* d, uint32_t width, uint32_t height) {
    const size_t pixels = size_t(width) * height;
    if (pixels <= d->capacity_pixels) return HIPR_OK;
    DN_HIP(hipStreamSynchronize(d->stream));
    for (float4** plane : {&d->ping, &d->pong, &d->feature, &d->filtered}) {
        if (*plane) (void)hipFree(*plane);
        *plane = nullptr;
    }
    d->capacity_pixels = 0; d->filtered_width = d->filtered_height = 0;
    for (float4** plane : {&d->ping, &d->pong, &d->feature, &d->filtered})
        if (hipMalloc(reinterpret_cast<void**>(plane), pixels * sizeof(float4)) != hipSuccess) return fail(d, HIPR_ERROR_OUT_OF_MEMORY, "hipr_denoiser: cannot allocate the working planes");
    d->capacity_pixels = pixels;
    return HIPR_OK;
}

dim3 grid_of(uint32_t width, uint32_t height) { return dim3((width + TILE_X - 1) / TILE_X, (height + TILE_Y - 1) / TILE_Y); }

// prepare has filled ping / feature: run the passes and leave the result in d->filtered
int enqueue_filter(HiprDenoiser* d, const HiprDenoiserSettings& s, uint32_t width, uint32_t height) {
    const dim3 grid = grid_of(width, height), block(TILE_X, TILE_Y);
    float4 *in = d->ping, *out = d->pong;
    for (uint32_t i = 0; i < s.iterations; ++i) {
        const float sigma_luminance = s.sigma_luminance / float(1u << i);
        hipLaunchKernelGGL(k_denoiser_pass, grid, block, 0, d->stream, in, d->feature, width, height, int(1u << i), 1.0f / (s.sigma_albedo * s.sigma_albedo), 1.0f / sigma_luminance, out);
        std::swap(in, out);
    }
    hipLaunchKernelGGL(k_denoiser_finish, grid, block, 0, d->stream, in, d->feature, width, height, s.albedo_floor, d->filtered);
    DN_HIP(hipGetLastError());
    d->filtered_width = width; d->filtered_height = height;
    return HIPR_OK;
}

} // namespace

extern "C" {

int hipr_denoiser_create(int device_index, HiprDenoiser** out) {
    if (!out) return HIPR_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_index < 0 || device_index >= device_count) return HIPR_ERROR_NO_DEVICE;
    if (hipSetDevice(device_index) != hipSuccess) return HIPR_ERROR_NO_DEVICE;
    HiprDenoiser* d = new HiprDenoiser();
    d->device = device_index;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { delete d; return HIPR_ERROR_HIP; }
    *out = d;
    return HIPR_OK;
}

void hipr_denoiser_destroy(HiprDenoiser* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    for (float4* plane : {d->ping, d->pong, d->feature, d->filtered})
        if (plane) (void)hipFree(plane);
    if (d->stream && d->owns_stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

const char* hipr_denoiser_last_error(const HiprDenoiser* d) { return d ? d->last_error.c_str() : "null denoiser object"; }

int hipr_denoiser_set_stream(HiprDenoiser* d, void* hip_stream) {
    if (!d || !hip_stream) return fail(d, HIPR_ERROR_INVALID_ARGUMENT, "hipr_denoiser_set_stream: null argument");
    DN_HIP(hipStreamSynchronize(d->stream));
    if (d->owns_stream) (void)hipStreamDestroy(d->stream);
    d->stream = static_cast<hipStream_t>(hip_stream);
    d->owns_stream = false;
    return HIPR_OK;
}

int hipr_denoiser_synchronize(HiprDenoiser* d) {
    if (!d) return HIPR_ERROR_INVALID_ARGUMENT;
    DN_HIP(hipStreamSynchronize(d->stream));
    return HIPR_OK;
}

int hipr_denoiser_default_settings(HiprDenoiserSettings* out) {
    if (!out) return HIPR_ERROR_INVALID_ARGUMENT;
    *out = {5u, 0.1f, 1.0f, 0.001f};
    return HIPR_OK;
}

int hipr_denoiser_process(HiprDenoiser* d, const HiprDenoiserSettings* settings, const void* noisy_half4, uint32_t noisy_pitch, const void* albedo_half4, uint32_t albedo_pitch,
                          uint32_t width, uint32_t height, int update_filtered, int show, void* out_half4, uint32_t out_pitch) {
    if (!d) return HIPR_ERROR_INVALID_ARGUMENT;
    if (!valid_settings(settings)) return fail(d, HIPR_ERROR_INVALID_ARGUMENT, "hipr_denoiser_process: settings out of range (1..12 iterations, positive sigmas, albedo_floor >= 0)");
    if (!noisy_half4 || !albedo_half4 || !out_half4 || width == 0 || height == 0 || noisy_pitch < width || albedo_pitch < width || out_pitch < width)
        return fail(d, HIPR_ERROR_INVALID_ARGUMENT, "hipr_denoiser_process: null frame, empty frame or a pitch below the width");
    if (show < HIPR_DENOISER_SHOW_FILTERED || show > HIPR_DENOISER_SHOW_ALBEDO) return fail(d, HIPR_ERROR_INVALID_ARGUMENT, "hipr_denoiser_process: unknown `show` value");
    DN_HIP(hipSetDevice(d->device));
    if (int status = reserve(d, width, height)) return status;
    const HalfSource noisy = {static_cast<const uint2*>(noisy_half4), noisy_pitch}, albedo = {static_cast<const uint2*>(albedo_half4), albedo_pitch};
    const dim3 grid = grid_of(width, height), block(TILE_X, TILE_Y);
    if (update_filtered || d->filtered_width != width || d->filtered_height != height) {
        hipLaunchKernelGGL(k_denoiser_prepare<HalfSource>, grid, block, 0, d->stream, noisy, albedo, width, height, settings->albedo_floor, d->ping, d->feature);
        if (int status = enqueue_filter(d, *settings, width, height)) return status;
    }
    hipLaunchKernelGGL(k_denoiser_output, grid, block, 0, d->stream, d->filtered, noisy, albedo, width, height, show, static_cast<uint2*>(out_half4), out_pitch);
    DN_HIP(hipGetLastError());
    return HIPR_OK;
}

int hipr_denoiser_filter_host(HiprDenoiser* d, const HiprDenoiserSettings* settings, const float* noisy_rgba, const float* albedo_rgba, uint32_t width, uint32_t height, float* out_rgba) {
    if (!d) return HIPR_ERROR_INVALID_ARGUMENT;
    if (!valid_settings(settings)) return fail(d, HIPR_ERROR_INVALID_ARGUMENT, "hipr_denoiser_filter_host: settings out of range (1..12 iterations, positive sigmas, albedo_floor >= 0)");
    if (!noisy_rgba || !albedo_rgba || !out_rgba || width == 0 || height == 0) return fail(d, HIPR_ERROR_INVALID_ARGUMENT, "hipr_denoiser_filter_host: null or empty frame");
    DN_HIP(hipSetDevice(d->device));
    if (int status = reserve(d, width, height)) return status;
    const size_t bytes = size_t(width) * height * sizeof(float4);
    // the inputs are staged in the two planes the filter does not read first: pong (noisy) and filtered (albedo)
    DN_HIP(hipMemcpyAsync(d->pong, noisy_rgba, bytes, hipMemcpyHostToDevice, d->stream));
    DN_HIP(hipMemcpyAsync(d->filtered, albedo_rgba, bytes, hipMemcpyHostToDevice, d->stream));
    const FloatSource noisy = {d->pong, width}, albedo = {d->filtered, width};
    hipLaunchKernelGGL(k_denoiser_prepare<FloatSource>, grid_of(width, height), dim3(TILE_X, TILE_Y), 0, d->stream, noisy, albedo, width, height, settings->albedo_floor, d->ping, d->feature);
    if (int status = enqueue_filter(d, *settings, width, height)) return status;
    DN_HIP(hipMemcpyAsync(out_rgba, d->filtered, bytes, hipMemcpyDeviceToHost, d->stream));
    DN_HIP(hipStreamSynchronize(d->stream));
    d->filtered_width = d->filtered_height = 0;   // a test frame, not the backend's
    return HIPR_OK;
}

} // extern "C"
