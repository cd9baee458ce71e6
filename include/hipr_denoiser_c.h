/* hipr_denoiser_c.h -- C-ABI of the denoising stage of the reference's AIDenoisedBackend.
 *
 * The reference's `Backend::AIDenoisedPathTracing` (extensions/OptiXRenderer/OptiXRenderer/IBackend.cpp:19-80) runs a command
 * list per frame: the path tracing launch that also accumulates an albedo feature image (ORS/SimpleRGPs.cu:149-201), NVIDIA's
 * closed "DLDenoiser" post-processing stage on {noisy radiance, albedo} (IBackend.cpp:26-31), and a launch that writes the
 * filtered image -- or, on request, the noisy or the albedo image -- as half4 (SimpleRGPs.cu:203-219). The DL stage cannot be
 * rebuilt (weights and code are not public); what stands in for it here is an open filter on the same two
 * inputs: an edge-avoiding a-trous wavelet transform (Dammertz, Sewtz, Hanika, Lensch, "Edge-Avoiding A-Trous Wavelet
 * Transform for fast Global Illumination Filtering", HPG 2010) on the radiance demodulated by the albedo.
 *   c0 = noisy / albedo'                          (albedo' = albedo where a component is above albedo_floor, else 1: emitters, misses)
 *   pass i = 0 .. iterations-1, step 2^i:         c_{i+1}(p) = sum_q w c_i(q) / sum_q w over the 5 x 5 taps q = p + 2^i (dx, dy) inside the frame,
 *       w = h(dx) h(dy)                           B3 spline, h = {1/16, 1/4, 3/8, 1/4, 1/16}
 *         * exp(-|albedo(p) - albedo(q)|^2 / sigma_albedo^2)
 *         * exp(-|l_i(p) - l_i(q)| / (sigma_luminance 2^-i)),   l_i = log2(1 + luminance(c_i))    (HDR: differences of logarithms)
 *   out = c_iterations * albedo'
 * There is no reference output to be equal to (parity unpinned, by construction); the parity tests compare this implementation with
 * its CPU restatement (oracle/denoiser.cpp) and check what a denoiser must do (constants stay, noise drops, albedo edges stay).
 *
 * Conventions as in hiprenderer_c.h: plain pointers and sizes, int status (HIPR_OK or a negative HiprStatus). Frames are DEVICE
 * pointers to half4 pixels with a row pitch in pixels, as hipr_render_pass / hipr_accumulate_samples write them.
 */
#ifndef HIPR_DENOISER_C_H
#define HIPR_DENOISER_C_H

#include "hiprenderer_c.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HiprDenoiser HiprDenoiser;

typedef struct HiprDenoiserSettings {
    uint32_t iterations;      /* a-trous passes; 5 gives a 125 x 125 pixel footprint */
    float sigma_albedo;       /* edge stop on the albedo feature */
    float sigma_luminance;    /* edge stop on log2(1 + luminance) of the demodulated radiance, halved every pass */
    float albedo_floor;       /* albedo components at or below this are not demodulated */
} HiprDenoiserSettings;

/* What hipr_denoiser_process writes: AIDenoiserFlag::VisualizeNoise / VisualizeAlbedo (OR/PublicTypes.h:50-57, SimpleRGPs.cu:206-216). */
enum { HIPR_DENOISER_SHOW_FILTERED = 0, HIPR_DENOISER_SHOW_NOISE = 1, HIPR_DENOISER_SHOW_ALBEDO = 2 };

int hipr_denoiser_create(int device_index, HiprDenoiser** out_denoiser);
void hipr_denoiser_destroy(HiprDenoiser* denoiser);
const char* hipr_denoiser_last_error(const HiprDenoiser* denoiser);
int hipr_denoiser_set_stream(HiprDenoiser* denoiser, void* hip_stream);
int hipr_denoiser_synchronize(HiprDenoiser* denoiser);
int hipr_denoiser_default_settings(HiprDenoiserSettings* out_settings);   /* 5 passes, sigma_albedo 0.1, sigma_luminance 1.0, albedo_floor 0.001 */

/* One frame of the backend's command list (IBackend.cpp:48-80). With update_filtered != 0 (the "presenting" list: every frame, or
 * the power-of-two and every 32nd frame under AIDenoiserFlag::LogarithmicFeedback) the filter runs on {noisy, albedo} and its result
 * replaces the filtered image the object keeps; otherwise the filtered image of the last update is reused, as the reference's
 * denoised_pixels_buffer is. Then `show` selects what goes to out_half4: the filtered, the noisy or the albedo image. The first call
 * for a frame size always filters. Runs on the object's stream; returns once the work is queued. */
int hipr_denoiser_process(HiprDenoiser* denoiser, const HiprDenoiserSettings* settings, const void* noisy_half4, uint32_t noisy_pitch,
                          const void* albedo_half4, uint32_t albedo_pitch, uint32_t width, uint32_t height, int update_filtered, int show,
                          void* out_half4, uint32_t out_pitch);

/* The filter alone on float4 frames in HOST memory (row-major, no padding): the parity entry point the tests compare with oracle/denoiser.cpp. */
int hipr_denoiser_filter_host(HiprDenoiser* denoiser, const HiprDenoiserSettings* settings, const float* noisy_rgba, const float* albedo_rgba,
                              uint32_t width, uint32_t height, float* out_rgba);

#ifdef __cplusplus
}
#endif
#endif
