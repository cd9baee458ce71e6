/* hipr_camera_effects_c.h -- C-ABI of the step after the path: exposure, bloom, vignette, tonemapping, film grain.
 *
 * Replaces DX11Renderer::CameraEffects, the post-process the reference's compositor runs on the path tracer's half4 frame
 * before presenting it (extensions/DX11Renderer/DX11Renderer/CameraEffects.h:196-239, CameraEffects.cpp:412-507 process();
 * shaders under Shaders/CameraEffects/; settings Bifrost/Math/CameraEffects.h:33-118). SURVEY.md 8(f)4.
 *
 * Same conventions as hiprenderer_c.h: plain pointers and sizes, int status (HIPR_OK or a negative HiprStatus), no
 * exceptions across the boundary. Frame and target pointers are DEVICE pointers (the frame is what hipr_render_pass /
 * hipr_present_flipped produce); histograms and scalars cross as HOST memory. Everything runs on the object's stream.
 *
 * The stage entry points exist because the reference tests the stages one by one
 * (tests/DX11RendererTests/{ExposureHistogram,LogAverageLuminance,Bloom}Test.h); process() is the product path.
 */
#ifndef HIPR_CAMERA_EFFECTS_C_H
#define HIPR_CAMERA_EFFECTS_C_H

#include "hiprenderer_c.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { HIPR_EXPOSURE_FIXED = 0, HIPR_EXPOSURE_LOG_AVERAGE = 1, HIPR_EXPOSURE_HISTOGRAM = 2 };                                         /* ExposureMode, BF/Math/CameraEffects.h:19 */
enum { HIPR_TONEMAPPING_LINEAR = 0, HIPR_TONEMAPPING_FILMIC = 1, HIPR_TONEMAPPING_AGX = 2, HIPR_TONEMAPPING_KHRONOS_NEUTRAL = 3 };    /* TonemappingMode, :18 */
enum { HIPR_EXPOSURE_HISTOGRAM_BINS = 64 };                                                                                            /* ExposureHistogram::bin_count, CameraEffects.h:149 */
/* What process() writes. RGBA16F / RGBA32F hold the value the reference's pixel shader returns (linear, alpha 1);
 * RGBA8_SRGB is that value as an sRGB render target stores it (clamped, sRGB encoded, 8 bit). */
enum { HIPR_TARGET_RGBA16F = 0, HIPR_TARGET_RGBA32F = 1, HIPR_TARGET_RGBA8_SRGB = 2 };

/* Bifrost::Math::CameraEffects::Settings (BF/Math/CameraEffects.h:33-63), flattened. bloom_support is relative to the
 * viewport height; bloom_threshold = INFINITY switches bloom off; eye adaptation off means the target exposure is taken at once. */
typedef struct HiprCameraEffectsSettings {
    int32_t exposure_mode;
    float min_log_luminance, max_log_luminance;
    float min_histogram_percentage, max_histogram_percentage;
    float log_luminance_bias;
    int32_t eye_adaptation_enabled;
    float eye_adaptation_brightness, eye_adaptation_darkness;
    float bloom_threshold, bloom_support;
    float vignette;
    int32_t tonemapping_mode;
    float tonemapping_black_clip, tonemapping_toe, tonemapping_slope, tonemapping_shoulder, tonemapping_white_clip;
    float film_grain;
} HiprCameraEffectsSettings;

typedef struct HiprRect { int32_t x, y, width, height; } HiprRect;

/* A frame in device memory: half4 pixels, `pitch` pixels per row, `rows` rows; the part to process is `viewport`. */
typedef struct HiprFrameView {
    const void* pixels;
    uint32_t pitch, rows;
    HiprRect viewport;
} HiprFrameView;

/* Time spent per stage since the last reset, measured with HIP events on the object's stream when instrumentation is on. */
typedef struct HiprCameraEffectsTimes {
    float exposure_ms, bloom_horizontal_ms, bloom_vertical_ms, tonemap_ms;
    uint32_t exposure_launches, bloom_horizontal_launches, bloom_vertical_launches, tonemap_launches;
} HiprCameraEffectsTimes;

typedef struct HiprCameraEffects HiprCameraEffects;

int hipr_camera_effects_create(int device_index, HiprCameraEffects** out);      /* CameraEffects::CameraEffects, CameraEffects.cpp:361-398; linear exposure starts at 0 */
void hipr_camera_effects_destroy(HiprCameraEffects* effects);
const char* hipr_camera_effects_last_error(const HiprCameraEffects* effects);
int hipr_camera_effects_set_stream(HiprCameraEffects* effects, void* hip_stream);      /* the stream every stage runs on from now on (caller-owned; null = the default stream) */
int hipr_camera_effects_synchronize(HiprCameraEffects* effects);

/* CameraEffects::process (CameraEffects.cpp:412-507): exposure by the settings' mode with eye adaptation over `delta_time`
 * seconds -> Gaussian bloom of what exceeds the threshold -> per pixel: exposure * (min(pixel, threshold) + bloom), vignette,
 * tonemapping operator, film grain -> target. The target viewport has the frame viewport's size. */
int hipr_camera_effects_process(HiprCameraEffects* effects, const HiprCameraEffectsSettings* settings, float delta_time, const HiprFrameView* frame,
                                void* target, int target_format, uint32_t target_pitch, uint32_t target_rows, int32_t target_viewport_x, int32_t target_viewport_y);

/* The linear exposure carried from frame to frame (m_linear_exposure, CameraEffects.h:222-223). */
int hipr_camera_effects_get_linear_exposure(HiprCameraEffects* effects, float* out_host);
int hipr_camera_effects_set_linear_exposure(HiprCameraEffects* effects, float linear_exposure);

/* ---- stages ------------------------------------------------------------------------------------------------------------- */
/* ExposureHistogram::reduce_histogram (CameraEffects.cpp:303-323, ReduceExposureHistogram.hlsl:27-70): 64 bins of
 * log2(max(luminance, 1e-4)) between the settings' min and max log luminance. Uses min / max_log_luminance only. */
int hipr_camera_effects_reduce_histogram(HiprCameraEffects* effects, const HiprCameraEffectsSettings* settings, const HiprFrameView* frame, uint32_t* out_histogram_host);
/* ReduceExposureHistogram.hlsl:82-154 compute_linear_exposure on a caller's histogram: the average luminance of the pixels
 * between the min and max percentage, exposure = 2^bias / average, eye adaptation from *io_linear_exposure_host. */
int hipr_camera_effects_exposure_from_histogram(HiprCameraEffects* effects, const HiprCameraEffectsSettings* settings, float delta_time,
                                                const uint32_t* histogram_host, float* io_linear_exposure_host);
/* LogAverageLuminance::compute_log_average (CameraEffects.cpp:262-266, ReduceLogAverageLuminance.hlsl:23-88): 2^mean(log2(max(luminance, 1e-4))). */
int hipr_camera_effects_log_average(HiprCameraEffects* effects, const HiprFrameView* frame, float* out_log_average_host);
/* LogAverageLuminance::compute_linear_exposure (ReduceLogAverageLuminance.hlsl:55-106): the geometric-mean key value exposure. */
int hipr_camera_effects_exposure_from_log_average(HiprCameraEffects* effects, const HiprCameraEffectsSettings* settings, float delta_time,
                                                  const HiprFrameView* frame, float* io_linear_exposure_host);
/* GaussianBloom::filter (CameraEffects.cpp:39-112, Bloom.hlsl:24-67): max(0, pixel - threshold) blurred by a separable Gaussian
 * of `support` pixels (standard deviation support / 4) through bilinearly placed taps. out: viewport width x height half4, tightly packed, device. */
int hipr_camera_effects_bloom(HiprCameraEffects* effects, float threshold, int32_t support, const HiprFrameView* frame, void* out_half4_device);
/* DualKawaseBloom::filter (CameraEffects.cpp:140-232, Bloom.hlsl:69-119): what exceeds the threshold (alpha kept), `half_passes` times down a chain
 * of half-sized half4 images (5 bilinear taps each) and back up (8 taps each); half_passes 0 only extracts. The reference tests this filter
 * (tests/DX11RendererTests/BloomTest.h:218-245) and its CameraEffects::process uses the Gaussian one. out_half4_device: viewport-sized, tightly packed.
 * Level m is max(1, width >> m) x max(1, height >> m) of the VIEWPORT (the reference sizes its chain by a grow-only buffer and leaves levels that
 * collapse to zero rows unwritten); bilinear weights are exact fractions, addressing is clamped. */
int hipr_camera_effects_dual_kawase_bloom(HiprCameraEffects* effects, float threshold, uint32_t half_passes, const HiprFrameView* frame, void* out_half4_device);

int hipr_camera_effects_set_instrumentation(HiprCameraEffects* effects, int time_stages);
int hipr_camera_effects_reset_timers(HiprCameraEffects* effects);
int hipr_camera_effects_get_times(HiprCameraEffects* effects, HiprCameraEffectsTimes* out);

#ifdef __cplusplus
}
#endif

#endif /* HIPR_CAMERA_EFFECTS_C_H */
