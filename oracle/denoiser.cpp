// oracle/denoiser.cpp -- CPU restatement of the denoising filter (TEST INFRASTRUCTURE: only tests/ may use it).
//
// PARITY UNPINNED, by construction: the stage this stands in for is NVIDIA's closed DLDenoiser (extensions/OptiXRenderer/OptiXRenderer/
// IBackend.cpp:26-31); no output of it exists or can be produced here. What is restated is the open filter of
// include/hipr_denoiser_c.h (edge-avoiding a-trous wavelet transform, Dammertz et al. 2010, on albedo-demodulated radiance), so that
// the HIP kernels of csrc/denoiser.hip have an independent implementation to be compared with, written from the header's formulas
// in plain loops: same taps, same order of the sums (row by row, left to right), f32 throughout.
#include "../include/hipr_denoiser_c.h"

#include <cmath>
#include <cstddef>
#include <vector>

namespace {

struct Pixel { float r, g, b, l; };

inline float luminance(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
inline float demodulator(float albedo, float floor) { return albedo > floor ? albedo : 1.0f; }

} // namespace

extern "C" {

// noisy, albedo, out: width x height float4, row-major. Returns 0, or -1 for settings the product rejects too.
int oracle_denoise(const float* noisy, const float* albedo, int width, int height, const HiprDenoiserSettings* s, float* out) {
    if (!noisy || !albedo || !out || !s || width <= 0 || height <= 0 || s->iterations < 1 || s->iterations > 12 || !(s->sigma_albedo > 0.0f) || !(s->sigma_luminance > 0.0f) || s->albedo_floor < 0.0f)
        return -1;
    const size_t n = size_t(width) * size_t(height);
    std::vector<Pixel> a(n), b(n);
    for (size_t i = 0; i < n; ++i) {
        const float r = noisy[4 * i] / demodulator(albedo[4 * i], s->albedo_floor), g = noisy[4 * i + 1] / demodulator(albedo[4 * i + 1], s->albedo_floor),
                    bl = noisy[4 * i + 2] / demodulator(albedo[4 * i + 2], s->albedo_floor);
        a[i] = {r, g, bl, log2f(1.0f + luminance(r, g, bl))};
    }
    const float spline[3] = {0.375f, 0.25f, 0.0625f};
    const float inverse_sigma_albedo_squared = 1.0f / (s->sigma_albedo * s->sigma_albedo);
    std::vector<Pixel>*in = &a, *result = &b;
    for (unsigned pass = 0; pass < s->iterations; ++pass) {
        const int step = 1 << pass;
        const float inverse_sigma_luminance = 1.0f / (s->sigma_luminance / float(1u << pass));
#pragma omp parallel for schedule(static)
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {
                const size_t p = size_t(y) * width + x;
                const Pixel cp = (*in)[p];
                float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
                for (int dy = -2; dy <= 2; ++dy) {
                    const int qy = y + dy * step;
                    if (qy < 0 || qy >= height) continue;
                    for (int dx = -2; dx <= 2; ++dx) {
                        const int qx = x + dx * step;
                        if (qx < 0 || qx >= width) continue;
                        const size_t q = size_t(qy) * width + qx;
                        const Pixel cq = (*in)[q];
                        const float dr = albedo[4 * p] - albedo[4 * q], dg = albedo[4 * p + 1] - albedo[4 * q + 1], db = albedo[4 * p + 2] - albedo[4 * q + 2];
                        const float albedo_distance = dr * dr + dg * dg + db * db;
                        const float w = spline[dx < 0 ? -dx : dx] * spline[dy < 0 ? -dy : dy] * expf(-albedo_distance * inverse_sigma_albedo_squared) *
                                        expf(-fabsf(cp.l - cq.l) * inverse_sigma_luminance);
                        sr += w * cq.r; sg += w * cq.g; sb += w * cq.b; sw += w;
                    }
                }
                const float inverse = 1.0f / sw;
                const float r = sr * inverse, g = sg * inverse, bl = sb * inverse;
                (*result)[p] = {r, g, bl, log2f(1.0f + luminance(r, g, bl))};
            }
        std::swap(in, result);
    }
    for (size_t i = 0; i < n; ++i) {
        const Pixel c = (*in)[i];
        out[4 * i] = c.r * demodulator(albedo[4 * i], s->albedo_floor);
        out[4 * i + 1] = c.g * demodulator(albedo[4 * i + 1], s->albedo_floor);
        out[4 * i + 2] = c.b * demodulator(albedo[4 * i + 2], s->albedo_floor);
        out[4 * i + 3] = 1.0f;
    }
    return 0;
}

} // extern "C"
