// oracle/camera_effects.cpp -- CPU restatement of the reference's camera effects (exposure, bloom, tonemapping).
//
// TEST INFRASTRUCTURE ONLY: the checker the GPU stages of bifrost3d_amd/csrc/camera_effects.hip are compared against.
// Nothing in the product links or calls this file.
//
// Follows extensions/DX11Renderer/DX11Renderer/Shaders/CameraEffects/{Utils,ReduceExposureHistogram,ReduceLogAverageLuminance,
// Bloom,Tonemapping}.hlsl, CameraEffects.cpp:39-112,412-507 and Bifrost/Math/{CameraEffects.h,Utils.h:275-312}; each function
// cites its lines. Pinned by what the reference's own tests expect of these stages (tests/test_camera_effects_cpu.py):
// ExposureHistogramTest.h (bin placement, compute_average_luminance_without_outlier), LogAverageLuminanceTest.h (double
// precision log average, geometric-mean exposure), BloomTest.h (energy conservation, thresholding), Math/UtilsTest.h:69-123
// (tap weights sum to one half, taps filter like a Gaussian). The DX11 shaders themselves cannot run here; where the
// hardware sampler's fixed-point bilinear weights would enter, the mathematically intended weights are used.
#include "../include/hipr_camera_effects_c.h"

#include <immintrin.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline float half_to_float(uint16_t h) { return _cvtsh_ss(h); }
inline uint16_t float_to_half(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT); }

struct float3 { float x, y, z; };
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 max0(float3 a) { return {std::fmax(a.x, 0.0f), std::fmax(a.y, 0.0f), std::fmax(a.z, 0.0f)}; }
inline float3 lerp3(float3 a, float3 b, float t) { return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)}; }
inline float lerp1(float a, float b, float t) { return a + t * (b - a); }
inline float saturate(float v) { return std::fmin(std::fmax(v, 0.0f), 1.0f); }

// Utils.hlsl:98
inline float luminance(float3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }

inline float3 load(const HiprFrameView& frame, int x, int y) {     // clamped addressing
    x = std::min(std::max(x, 0), int(frame.pitch) - 1);
    y = std::min(std::max(y, 0), int(frame.rows) - 1);
    const uint16_t* p = static_cast<const uint16_t*>(frame.pixels) + 4 * (size_t(x) + size_t(y) * frame.pitch);
    return {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2])};
}

// CameraEffects/Utils.hlsl:42-47
float eye_adaptation(const HiprCameraEffectsSettings& s, float delta_time, float current_exposure, float target_exposure) {
    const float brightness = s.eye_adaptation_enabled ? s.eye_adaptation_brightness : INFINITY;     // CameraEffects.cpp:432-436
    const float darkness = s.eye_adaptation_enabled ? s.eye_adaptation_darkness : INFINITY;
    const float delta_exposure = target_exposure - current_exposure;
    const float adaption_speed = delta_exposure > 0.0f ? brightness : darkness;
    const float factor = 1.0f - std::exp2(-delta_time * adaption_speed);
    return current_exposure + delta_exposure * factor;
}

// ---- tonemapping operators (Tonemapping.hlsl:38-160, Bifrost/Math/CameraEffects.h:135-283) ------------------------------------

struct Matrix3 { float m[3][3]; };
inline float3 mul(const Matrix3& a, float3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z, a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
Matrix3 mul(const Matrix3& a, const Matrix3& b) {
    Matrix3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
const Matrix3 D65_to_D60 = {{{1.01303f, 0.00610531f, -0.014971f}, {0.00769823f, 0.998165f, -0.00503203f}, {-0.00284131f, 0.00468516f, 0.924507f}}};
const Matrix3 sRGB_to_XYZ = {{{0.4124564f, 0.3575761f, 0.1804375f}, {0.2126729f, 0.7151522f, 0.0721750f}, {0.0193339f, 0.1191920f, 0.9503041f}}};
const Matrix3 XYZ_to_AP1 = {{{1.6410233797f, -0.3248032942f, -0.2364246952f}, {-0.6636628587f, 1.6153315917f, 0.0167563477f}, {0.0117218943f, -0.0082844420f, 0.9883948585f}}};
const Matrix3 sRGB_to_AP1 = mul(XYZ_to_AP1, mul(D65_to_D60, sRGB_to_XYZ));
const Matrix3 AP1_to_sRGB = {{{1.70479095f, -0.621689737f, -0.0832421705f}, {-0.130263522f, 1.14082849f, -0.0105496496f}, {-0.0240088310f, -0.128999621f, 1.15324795f}}};
const float3 AP1_RGB2Y = {0.2722287168f, 0.6740817658f, 0.0536895174f};

float3 unreal4(float3 color, float black_clip, float toe, float slope, float shoulder, float white_clip) {     // Tonemapping.hlsl:71-109
    float3 working_color = max0(mul(sRGB_to_AP1, color));
    const float pre_luminance = working_color.x * AP1_RGB2Y.x + working_color.y * AP1_RGB2Y.y + working_color.z * AP1_RGB2Y.z;
    working_color = lerp3({pre_luminance, pre_luminance, pre_luminance}, working_color, 0.96f);      // pre desaturate

    const float toe_scale = 1.0f + black_clip - toe;
    const float shoulder_scale = 1.0f + white_clip - shoulder;
    const float in_match = 0.18f, out_match = 0.18f;
    float toe_match;
    if (toe > 0.8f)
        toe_match = (1.0f - toe - out_match) / slope + std::log10(in_match);
    else {
        const float bt = (out_match + black_clip) / toe_scale - 1.0f;
        toe_match = std::log10(in_match) - 0.5f * std::log((1.0f + bt) / (1.0f - bt)) * (toe_scale / slope);
    }
    const float straight_match = (1.0f - toe) / slope - toe_match;
    const float shoulder_match = shoulder / slope - straight_match;

    auto channel = [&](float c) {
        const float log_color = std::log10(c);
        const float toe_color = (-black_clip) + (2 * toe_scale) / (1 + std::exp((-2 * slope / toe_scale) * (log_color - toe_match)));
        const float shoulder_color = (1 + white_clip) - (2 * shoulder_scale) / (1 + std::exp((2 * slope / shoulder_scale) * (log_color - shoulder_match)));
        float t = saturate((log_color - toe_match) / (shoulder_match - toe_match));
        t = shoulder_match < toe_match ? 1.0f - t : t;
        t = (3.0f - t * 2.0f) * t * t;
        return lerp1(toe_color, shoulder_color, t);
    };
    float3 tone_color = {channel(working_color.x), channel(working_color.y), channel(working_color.z)};
    const float post_luminance = tone_color.x * AP1_RGB2Y.x + tone_color.y * AP1_RGB2Y.y + tone_color.z * AP1_RGB2Y.z;
    tone_color = lerp3({post_luminance, post_luminance, post_luminance}, tone_color, 0.93f);           // post desaturate
    return mul(AP1_to_sRGB, max0(tone_color));
}

float3 agx(float3 linear_color) {      // Tonemapping.hlsl:111-142
    const Matrix3 linear_to_agx = {{{0.842479062253094f, 0.0784335999999992f, 0.0792237451477643f}, {0.0423282422610123f, 0.878468636469772f, 0.0791661274605434f},
                                    {0.0423756549057051f, 0.0784336f, 0.879142973793104f}}};
    float3 c = mul(linear_to_agx, linear_color);
    const float min_exposure_value = -12.47393f, max_exposure_value = 4.026069f;
    auto encode = [&](float v) {
        v = (std::log2(v) - min_exposure_value) / (max_exposure_value - min_exposure_value);
        v = saturate(v);
        return -0.00232f + v * (0.1191f + v * (0.4298f + v * (-6.868f + v * (31.96f + v * (-40.14f + v * 15.5f)))));   // sigmoid approximation
    };
    c = {encode(c.x), encode(c.y), encode(c.z)};
    const Matrix3 agx_to_tonemapped = {{{1.19687900512017f, -0.0980208811401368f, -0.0990297440797205f}, {-0.0528968517574562f, 1.15190312990417f, -0.0989611768448433f},
                                        {-0.0529716355144438f, -0.0980434501171241f, 1.15107367264116f}}};
    c = mul(agx_to_tonemapped, c);
    return {std::pow(std::fabs(c.x), 2.2f), std::pow(std::fabs(c.y), 2.2f), std::pow(std::fabs(c.z), 2.2f)};
}

float3 khronos_neutral(float3 c) {     // Tonemapping.hlsl:144-160
    const float start_compression = 0.8f - 0.04f, desaturation = 0.15f;
    const float x = std::fmin(c.x, std::fmin(c.y, c.z));
    const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    c = {c.x - offset, c.y - offset, c.z - offset};
    const float peak = std::fmax(c.x, std::fmax(c.y, c.z));
    if (peak < start_compression) return c;
    const float d = 1.0f - start_compression;
    const float new_peak = 1.0f - d * d / (peak + d - start_compression);
    c = c * (new_peak / peak);
    const float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    return lerp3(c, {new_peak, new_peak, new_peak}, g);
}

float3 tonemap(const HiprCameraEffectsSettings& s, float3 color) {
    switch (s.tonemapping_mode) {
    case HIPR_TONEMAPPING_FILMIC: return unreal4(color, s.tonemapping_black_clip, s.tonemapping_toe, s.tonemapping_slope, s.tonemapping_shoulder, s.tonemapping_white_clip);
    case HIPR_TONEMAPPING_AGX: return agx(color);
    case HIPR_TONEMAPPING_KHRONOS_NEUTRAL: return khronos_neutral(color);
    default: return color;
    }
}

inline float smoothstep(float lo, float hi, float v) { const float t = saturate((v - lo) / (hi - lo)); return t * t * (3.0f - 2.0f * t); }
float simple_vignette_tint(float u, float v, float scale) {     // Tonemapping.hlsl:166-170
    const float cx = u - 0.5f, cy = v - 0.5f;
    return 1.0f - smoothstep(0.1f, 0.9f, std::sqrt(cx * cx + cy * cy) * 1.5f * scale);
}
float film_grain(float u, float v, float delta_time, float scale) {     // Tonemapping.hlsl:176-180
    const float gu = u + delta_time, gv = v + delta_time;
    const float s = std::sin(gu * 12.9898f + gv * 78.233f) * 43758.5453f;
    return scale * ((s - std::floor(s)) - 0.5f);
}

// Bifrost/Math/Utils.h:286-312 fill_bilinear_gaussian_samples
void fill_taps(float std_dev, int count, float* offsets, float* weights) {
    const float double_variance = 2.0f * std_dev * std_dev;
    float total_weight = 0.0f;
    for (int s = 0; s < count; ++s) {
        const int t1 = s * 2;
        float w1 = std::exp(-(t1 * t1) / double_variance);
        if (s == 0) w1 *= 0.5f;
        const int t2 = t1 + 1;
        const float w2 = std::exp(-(t2 * t2) / double_variance);
        const float weight = w1 + w2;
        float offset = (t1 * w1 + t2 * w2) / weight;
        if (std::isnan(offset)) offset = float(t1);
        offsets[s] = offset; weights[s] = weight;
        total_weight += weight;
    }
    total_weight *= 2;
    for (int s = 0; s < count; ++s) weights[s] /= total_weight;
}

inline int next_power_of_two(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// GaussianBloom::filter (CameraEffects.cpp:39-112) + Bloom.hlsl:24-67. out: viewport-sized float3 rows (values as the half4 target stores them).
void bloom(float threshold, int support, const HiprFrameView& frame, std::vector<float3>& out) {
    const int width = frame.viewport.width, height = frame.viewport.height;
    const int sample_count = support / 2;
    // The tap table is filled to its capacity (64 entries, or the next power of two that holds ceil(support / 2)) and normalised
    // over all of it; the filters read the first support / 2 taps, stored as half2 (CameraEffects.cpp:51-76).
    const int capacity = std::max(64, next_power_of_two((support + 1) / 2));
    std::vector<float> offsets(capacity), weights(capacity);
    fill_taps(support * 0.25f, capacity, offsets.data(), weights.data());
    for (int s = 0; s < capacity; ++s) { offsets[s] = half_to_float(float_to_half(offsets[s])); weights[s] = half_to_float(float_to_half(weights[s])); }

    auto as_stored = [](float3 v) { return float3{half_to_float(float_to_half(v.x)), half_to_float(float_to_half(v.y)), half_to_float(float_to_half(v.z))}; };

    // Horizontal: threshold after the bilinear fetch, taps along x in texel units, rows exact. Reads the frame beyond the viewport.
    std::vector<float3> pong(size_t(width) * height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            float3 sum = {0, 0, 0};
            const float centre = float(x + frame.viewport.x);
            for (int s = 0; s < sample_count; ++s) {
                auto fetch = [&](float position) {      // texel-space coordinate of the sample along x; texel i covers [i - 0.5, i + 0.5]
                    const float lower = std::floor(position);
                    const float t = position - lower;
                    return lerp3(load(frame, int(lower), y + frame.viewport.y), load(frame, int(lower) + 1, y + frame.viewport.y), t);
                };
                const float3 lower_sample = fetch(centre - offsets[s]), upper_sample = fetch(centre + offsets[s]);
                const float3 high = max0(lower_sample - float3{threshold, threshold, threshold}) + max0(upper_sample - float3{threshold, threshold, threshold});
                sum = sum + high * weights[s];
            }
            pong[x + size_t(y) * width] = as_stored(sum);
        }

    // Vertical over the viewport-sized intermediate, clamped at its edges.
    out.assign(size_t(width) * height, {0, 0, 0});
    auto pong_at = [&](int x, int y) { return pong[x + size_t(std::min(std::max(y, 0), height - 1)) * width]; };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            float3 sum = {0, 0, 0};
            for (int s = 0; s < sample_count; ++s) {
                auto fetch = [&](float position) {
                    const float lower = std::floor(position);
                    const float t = position - lower;
                    return lerp3(pong_at(x, int(lower)), pong_at(x, int(lower) + 1), t);
                };
                sum = sum + (fetch(float(y) + offsets[s]) + fetch(float(y) - offsets[s])) * weights[s];
            }
            out[x + size_t(y) * width] = as_stored(sum);
        }
}

// ---- Dual Kawase bloom: DualKawaseBloom::filter (CameraEffects.cpp:140-232) over extract_high_intensity / dual_kawase_downsample /
// dual_kawase_upsample (Bloom.hlsl:69-119). Levels are half4 images (rounded to half when stored), level m is max(1, w >> m) x max(1, h >> m)
// of the viewport; SampleLevel(bilinear_sampler) = clamped addressing, exact fractional weights.
struct float4v { float x, y, z, w; };
struct HalfImage {
    int width = 0, height = 0;
    std::vector<uint16_t> texels;      // 4 per pixel
    void resize(int w, int h) { width = w; height = h; texels.assign(size_t(w) * h * 4, 0); }
    float4v at(int x, int y) const {
        x = std::min(std::max(x, 0), width - 1); y = std::min(std::max(y, 0), height - 1);
        const uint16_t* p = &texels[4 * (size_t(x) + size_t(y) * width)];
        return {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2]), half_to_float(p[3])};
    }
    void store(int x, int y, float4v v) {
        uint16_t* p = &texels[4 * (size_t(x) + size_t(y) * width)];
        p[0] = float_to_half(v.x); p[1] = float_to_half(v.y); p[2] = float_to_half(v.z); p[3] = float_to_half(v.w);
    }
    float4v sample(float u, float v) const {
        const float px = u * float(width) - 0.5f, py = v * float(height) - 0.5f;
        const float fx = std::floor(px), fy = std::floor(py);
        const float tx = px - fx, ty = py - fy;
        const float4v a = at(int(fx), int(fy)), b = at(int(fx) + 1, int(fy)), c = at(int(fx), int(fy) + 1), d = at(int(fx) + 1, int(fy) + 1);
        return {lerp1(lerp1(a.x, b.x, tx), lerp1(c.x, d.x, tx), ty), lerp1(lerp1(a.y, b.y, tx), lerp1(c.y, d.y, tx), ty), lerp1(lerp1(a.z, b.z, tx), lerp1(c.z, d.z, tx), ty),
                lerp1(lerp1(a.w, b.w, tx), lerp1(c.w, d.w, tx), ty)};
    }
};
inline void accumulate(float4v& sum, float4v v, float weight) { sum.x += v.x * weight; sum.y += v.y * weight; sum.z += v.z * weight; sum.w += v.w * weight; }

void dual_kawase_bloom(float threshold, unsigned half_passes, const HiprFrameView& frame, HalfImage& result) {
    const int width = frame.viewport.width, height = frame.viewport.height;
    unsigned level_count = 1;
    while ((width >> level_count) > 0 || (height >> level_count) > 0) ++level_count;
    half_passes = std::min(half_passes, level_count - 1);
    std::vector<HalfImage> level(half_passes + 1);
    for (unsigned m = 0; m <= half_passes; ++m) level[m].resize(std::max(1, width >> m), std::max(1, height >> m));
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const uint16_t* p = static_cast<const uint16_t*>(frame.pixels) + 4 * (size_t(x + frame.viewport.x) + size_t(y + frame.viewport.y) * frame.pitch);
            level[0].store(x, y, {std::fmax(0.0f, half_to_float(p[0]) - threshold), std::fmax(0.0f, half_to_float(p[1]) - threshold), std::fmax(0.0f, half_to_float(p[2]) - threshold), half_to_float(p[3])});
        }
    for (unsigned p = 0; p < half_passes; ++p) {
        const HalfImage& in = level[p];
        HalfImage& out = level[p + 1];
        const float inverse_width = 1.0f / float(out.width), inverse_height = 1.0f / float(out.height), half_x = 0.5f * inverse_width, half_y = 0.5f * inverse_height;
        for (int y = 0; y < out.height; ++y)
            for (int x = 0; x < out.width; ++x) {
                const float u = float(x) * inverse_width + half_x, v = float(y) * inverse_height + half_y;
                float4v sum = {0, 0, 0, 0};
                accumulate(sum, in.sample(u, v), 4.0f);
                accumulate(sum, in.sample(u + half_x, v + half_y), 1.0f);
                accumulate(sum, in.sample(u + half_x, v - half_y), 1.0f);
                accumulate(sum, in.sample(u - half_x, v + half_y), 1.0f);
                accumulate(sum, in.sample(u - half_x, v - half_y), 1.0f);
                out.store(x, y, {sum.x / 8.0f, sum.y / 8.0f, sum.z / 8.0f, sum.w / 8.0f});
            }
    }
    for (unsigned p = half_passes; p > 0; --p) {
        const HalfImage& in = level[p];
        HalfImage& out = level[p - 1];
        const float inverse_width = 1.0f / float(out.width), inverse_height = 1.0f / float(out.height), half_x = 0.5f * inverse_width, half_y = 0.5f * inverse_height;
        for (int y = 0; y < out.height; ++y)
            for (int x = 0; x < out.width; ++x) {
                const float u = float(x) * inverse_width + half_x, v = float(y) * inverse_height + half_y;
                float4v sum = {0, 0, 0, 0};
                accumulate(sum, in.sample(u - half_x * 2.0f, v), 1.0f);
                accumulate(sum, in.sample(u - half_x, v + half_y), 2.0f);
                accumulate(sum, in.sample(u, v + half_y * 2.0f), 1.0f);
                accumulate(sum, in.sample(u + half_x, v + half_y), 2.0f);
                accumulate(sum, in.sample(u + half_x * 2.0f, v), 1.0f);
                accumulate(sum, in.sample(u + half_x, v - half_y), 2.0f);
                accumulate(sum, in.sample(u, v - half_y * 2.0f), 1.0f);
                accumulate(sum, in.sample(u - half_x, v - half_y), 2.0f);
                out.store(x, y, {sum.x / 12.0f, sum.y / 12.0f, sum.z / 12.0f, sum.w / 12.0f});
            }
    }
    result = std::move(level[0]);
}

void histogram(const HiprCameraEffectsSettings& s, const HiprFrameView& frame, uint32_t* bins) {     // ReduceExposureHistogram.hlsl:27-70
    std::fill(bins, bins + HIPR_EXPOSURE_HISTOGRAM_BINS, 0u);
    for (int y = 0; y < frame.viewport.height; ++y)
        for (int x = 0; x < frame.viewport.width; ++x) {
            const float3 pixel = load(frame, x + frame.viewport.x, y + frame.viewport.y);
            const float log_luminance = std::log2(std::fmax(luminance(pixel), 0.0001f));
            const float normalized_index = (log_luminance - s.min_log_luminance) / (s.max_log_luminance - s.min_log_luminance);
            const int bin_index = std::min(std::max(int(normalized_index * HIPR_EXPOSURE_HISTOGRAM_BINS + 0.5f), 0), HIPR_EXPOSURE_HISTOGRAM_BINS - 1);
            ++bins[bin_index];
        }
}

float exposure_from_histogram(const HiprCameraEffectsSettings& s, float delta_time, const uint32_t* bins, float current_exposure) {     // ReduceExposureHistogram.hlsl:82-154
    const int N = HIPR_EXPOSURE_HISTOGRAM_BINS;
    float prefix[N + 1];
    float running = 0.0f;
    for (int i = 0; i < N; ++i) { prefix[i] = running; running += float(bins[i]); }     // exclusive prefix sum; integer valued, exact in f32
    prefix[N] = running;
    const float max_pixel_count = prefix[N] * s.max_histogram_percentage;
    const float min_pixel_count = prefix[N] * s.min_histogram_percentage;
    for (int i = 0; i < N; ++i) prefix[i] = std::fmax(0.0f, std::fmin(prefix[i], max_pixel_count) - min_pixel_count);
    prefix[N] = max_pixel_count - min_pixel_count;
    float weighted[N];
    for (int i = 0; i < N; ++i) {
        const float bin_count = prefix[i + 1] - prefix[i];
        const float normalized_index = (i + 0.5f) / N;
        weighted[i] = std::exp2(lerp1(s.min_log_luminance, s.max_log_luminance, normalized_index)) * bin_count;
    }
    for (int offset = N >> 1; offset > 0; offset >>= 1)      // the shader's tree reduction, same pairing
        for (int t = 0; t < offset; ++t) weighted[t] += weighted[t + offset];
    const float average_luminance = weighted[0] / (max_pixel_count - min_pixel_count);
    const float linear_exposure = std::exp2(s.log_luminance_bias) / average_luminance;
    return eye_adaptation(s, delta_time, current_exposure, linear_exposure);
}

double mean_log_luminance(const HiprFrameView& frame) {     // ReduceLogAverageLuminance.hlsl:23-50, summed in double as LogAverageLuminanceTest.h:57-62 does
    double sum = 0.0;
    for (int y = 0; y < frame.viewport.height; ++y)
        for (int x = 0; x < frame.viewport.width; ++x)
            sum += std::log2(std::fmax(luminance(load(frame, x + frame.viewport.x, y + frame.viewport.y)), 0.0001f));
    return sum / (double(frame.viewport.width) * frame.viewport.height);
}

float exposure_from_log_average(const HiprCameraEffectsSettings& s, float delta_time, float average_log_luminance, float current_exposure) {   // ReduceLogAverageLuminance.hlsl:55-106
    average_log_luminance = std::fmin(std::fmax(average_log_luminance, s.min_log_luminance), s.max_log_luminance);
    const float log_average_luminance = std::exp2(average_log_luminance);
    const float key_value = 1.03f - (2.0f / (2 + std::log10(log_average_luminance + 1)));
    const float linear_exposure = key_value / log_average_luminance * std::exp2(s.log_luminance_bias);
    return eye_adaptation(s, delta_time, current_exposure, linear_exposure);
}

} // namespace

extern "C" {

void oracle_ce_histogram(const HiprCameraEffectsSettings* settings, const HiprFrameView* frame, uint32_t* out_bins) { histogram(*settings, *frame, out_bins); }
float oracle_ce_exposure_from_histogram(const HiprCameraEffectsSettings* settings, float delta_time, const uint32_t* bins, float current_exposure) {
    return exposure_from_histogram(*settings, delta_time, bins, current_exposure);
}
double oracle_ce_log_average(const HiprFrameView* frame) { return std::exp2(mean_log_luminance(*frame)); }
float oracle_ce_exposure_from_log_average(const HiprCameraEffectsSettings* settings, float delta_time, const HiprFrameView* frame, float current_exposure) {
    return exposure_from_log_average(*settings, delta_time, float(mean_log_luminance(*frame)), current_exposure);
}
void oracle_ce_gaussian_taps(float std_dev, int count, float* out_offsets, float* out_weights) { fill_taps(std_dev, count, out_offsets, out_weights); }
void oracle_ce_bloom(float threshold, int support, const HiprFrameView* frame, float* out_rgb) {
    std::vector<float3> result;
    bloom(threshold, support, *frame, result);
    std::memcpy(out_rgb, result.data(), result.size() * sizeof(float3));
}
// out: viewport-sized half4 pixels (4 x uint16 each), as the device writes them
void oracle_ce_dual_kawase_bloom(float threshold, unsigned half_passes, const HiprFrameView* frame, uint16_t* out_half4) {
    HalfImage result;
    dual_kawase_bloom(threshold, half_passes, *frame, result);
    std::memcpy(out_half4, result.texels.data(), result.texels.size() * sizeof(uint16_t));
}
void oracle_ce_tonemap(const HiprCameraEffectsSettings* settings, const float* rgb_in, int count, float* rgb_out) {
    for (int i = 0; i < count; ++i) {
        const float3 c = tonemap(*settings, {rgb_in[3 * i], rgb_in[3 * i + 1], rgb_in[3 * i + 2]});
        rgb_out[3 * i] = c.x; rgb_out[3 * i + 1] = c.y; rgb_out[3 * i + 2] = c.z;
    }
}
float oracle_ce_vignette(float u, float v, float scale) { return simple_vignette_tint(u, v, scale); }
float oracle_ce_film_grain(float u, float v, float delta_time, float scale) { return film_grain(u, v, delta_time, scale); }

// CameraEffects::process (CameraEffects.cpp:412-507) + Tonemapping.hlsl:205-227 postprocess_pixel. out: viewport-sized RGBA float rows.
// *io_linear_exposure carries the exposure from frame to frame.
void oracle_ce_process(const HiprCameraEffectsSettings* settings, float delta_time, const HiprFrameView* frame, float* io_linear_exposure, float* out_rgba) {
    const HiprCameraEffectsSettings& s = *settings;
    if (s.exposure_mode == HIPR_EXPOSURE_HISTOGRAM) {
        uint32_t bins[HIPR_EXPOSURE_HISTOGRAM_BINS];
        histogram(s, *frame, bins);
        *io_linear_exposure = exposure_from_histogram(s, delta_time, bins, *io_linear_exposure);
    } else if (s.exposure_mode == HIPR_EXPOSURE_LOG_AVERAGE)
        *io_linear_exposure = exposure_from_log_average(s, delta_time, float(mean_log_luminance(*frame)), *io_linear_exposure);
    else
        *io_linear_exposure = eye_adaptation(s, delta_time, *io_linear_exposure, std::exp2(s.log_luminance_bias));     // Tonemapping.hlsl:18-21

    const int width = frame->viewport.width, height = frame->viewport.height;
    std::vector<float3> bloom_pixels;
    const bool has_bloom = s.bloom_threshold < INFINITY;
    if (has_bloom) bloom(s.bloom_threshold, int(s.bloom_support * height), *frame, bloom_pixels);

    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float3 pixel = load(*frame, x + frame->viewport.x, y + frame->viewport.y);
            const float3 low_intensity_color = {std::fmin(pixel.x, s.bloom_threshold), std::fmin(pixel.y, s.bloom_threshold), std::fmin(pixel.z, s.bloom_threshold)};
            const float3 bloom_color = has_bloom ? bloom_pixels[x + size_t(y) * width] : float3{0, 0, 0};     // an unbound texture reads as zero
            float3 color = (low_intensity_color + bloom_color) * *io_linear_exposure;
            const float u = float(x) / float(width), v = float(y) / float(height);
            color = color * simple_vignette_tint(u, v, s.vignette);
            color = tonemap(s, color);
            const float grain = film_grain(u, v, delta_time, s.film_grain);
            float* o = out_rgba + 4 * (x + size_t(y) * width);
            o[0] = color.x + grain; o[1] = color.y + grain; o[2] = color.z + grain; o[3] = 1.0f;
        }
}

} // extern "C"
