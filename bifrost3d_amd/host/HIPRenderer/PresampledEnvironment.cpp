// PresampledEnvironment.cpp -- see PresampledEnvironment.h.
#include "PresampledEnvironment.h"

#include "../RNG.h"

#include <cmath>

using namespace Bifrost;

namespace HIPRenderer {

static unsigned int next_power_of_two(unsigned int v) { unsigned int p = 1; while (p < v) p *= 2; return p; }

PresampledEnvironment presample_environment(const Assets::InfiniteAreaLight& light, unsigned int sample_count) {
    PresampledEnvironment result;
    const bool disable_importance_sampling = light.image_integral() < 0.00001f || sample_count == 0;
    if (disable_importance_sampling) {
        result.per_pixel_PDF.assign(1, 0.0f);
        result.samples.assign(1, HiprLightSample{{0, 0, 0}, 0.0f, {0, 1, 0}, 0.0f});   // LightSample::none()
        return result;
    }

    result.pdf_width = light.get_PDF_width();
    result.pdf_height = light.get_PDF_height();
    result.per_pixel_PDF.resize(size_t(result.pdf_width) * result.pdf_height);
    Assets::InfiniteAreaLightUtils::reconstruct_solid_angle_PDF_sans_sin_theta(light, result.per_pixel_PDF.data());

    sample_count = std::max(2u, next_power_of_two(sample_count));
    const int exponent = int(std::log2(float(sample_count)));
    std::vector<Math::Vector2f> points(sample_count);
    Math::RNG::fill_progressive_multijittered_bluenoise_samples(points.data(), points.data() + sample_count);
    result.samples.resize(sample_count);
    for (unsigned int i = 0; i < sample_count; ++i) {
        // The renderer picks sample int(u * count) with a stratified u: neighbouring u should give neighbouring light samples, so
        // the progressive (hence scattered) points are visited in bit-reversed order, which groups them by quadrant.
        const unsigned int adjusted = Math::RNG::reverse_bits(i) >> (32 - exponent);
        const Assets::LightSample s = light.sample(points[adjusted]);
        result.samples[i] = {{s.radiance.r, s.radiance.g, s.radiance.b}, s.PDF, {s.direction_to_light.x, s.direction_to_light.y, s.direction_to_light.z}, s.distance};
    }
    return result;
}

} // namespace HIPRenderer
