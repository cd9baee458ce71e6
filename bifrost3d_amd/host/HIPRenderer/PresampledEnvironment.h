// PresampledEnvironment.h -- turns an InfiniteAreaLight into the device data of HiprEnvironment: the per-texel PDF image and
// the presampled light samples. Follows OptiXRenderer/PresampledEnvironmentMap.cpp:19-101.
#pragma once

#include "../InfiniteAreaLight.h"
#include "../../../include/hiprenderer_c.h"

#include <vector>

namespace HIPRenderer {

struct PresampledEnvironment {
    uint32_t pdf_width = 1, pdf_height = 1;
    std::vector<float> per_pixel_PDF;       // a single 0 when importance sampling is disabled
    std::vector<HiprLightSample> samples;   // a single invalid sample when importance sampling is disabled
};

// sample_count is rounded up to a power of two (at least 2). A dark image (integral < 1e-5) or sample_count == 0 disables
// importance sampling: one PDF texel of 0 and one sample with a zero PDF, so the shaders need no extra branch.
PresampledEnvironment presample_environment(const Bifrost::Assets::InfiniteAreaLight& light, unsigned int sample_count = 8192);

} // namespace HIPRenderer
