// CameraEffects.h -- Bifrost::Math::CameraEffects settings (Bifrost/Math/CameraEffects.h:18-118): exposure, bloom, vignette,
// tonemapping and film grain parameters of a camera, with the reference's presets. The effects themselves run on the device
// (csrc/camera_effects.hip, driven by HIPRenderer::CameraEffects in HIPRenderer/Compositor.h).
#pragma once

#include <cmath>

namespace Bifrost {
namespace Math {
namespace CameraEffects {

enum class TonemappingMode { Linear, Filmic, AgX, KhronosNeutral, Count };
enum class ExposureMode { Fixed, LogAverage, Histogram, Count };

struct TonemappingSettings {
    float black_clip, toe, slope, shoulder, white_clip;

    static TonemappingSettings ACES() { return {0.0f, 0.53f, 0.91f, 0.23f, 0.035f}; }
    static TonemappingSettings uncharted2() { return {0.0f, 0.55f, 0.63f, 0.47f, 0.01f}; }
    static TonemappingSettings HP() { return {0.0f, 0.63f, 0.65f, 0.45f, 0.0f}; }
    static TonemappingSettings legacy() { return {0.0f, 0.3f, 0.98f, 0.22f, 0.025f}; }
};

struct Settings final {
    struct {
        ExposureMode mode;
        float min_log_luminance, max_log_luminance;
        float min_histogram_percentage, max_histogram_percentage;
        float log_lumiance_bias;
        bool eye_adaptation_enabled;
        float eye_adaptation_brightness, eye_adaptation_darkness;
    } exposure;

    struct {
        float threshold;
        float support;      // normalised: relative to the height of the image
        float std_dev(int height) { return (support * height) * 0.25f; }
        float variance(int height) { return std_dev(height) * std_dev(height); }
    } bloom;

    float vignette;

    struct {
        TonemappingMode mode;
        TonemappingSettings settings;
    } tonemapping;

    float film_grain;

    static Settings preset() {
        Settings res = {};
        res.exposure.mode = ExposureMode::Histogram;
        res.exposure.min_log_luminance = -4;
        res.exposure.max_log_luminance = 4;
        res.exposure.min_histogram_percentage = 0.7f;
        res.exposure.max_histogram_percentage = 0.95f;
        res.exposure.log_lumiance_bias = 0;
        res.exposure.eye_adaptation_enabled = true;
        res.exposure.eye_adaptation_brightness = 3.0f;
        res.exposure.eye_adaptation_darkness = 1.0f;
        res.bloom.threshold = INFINITY;
        res.bloom.support = 0.05f;
        res.vignette = 0.63f;
        res.tonemapping.mode = TonemappingMode::Filmic;
        res.tonemapping.settings = TonemappingSettings::ACES();
        res.film_grain = 1 / 255.0f;
        return res;
    }

    // Linear colours without exposure, vignetting or other screen space effects.
    static Settings linear() {
        Settings res = {};
        res.exposure.mode = ExposureMode::Fixed;
        res.exposure.min_log_luminance = -4;
        res.exposure.max_log_luminance = 4;
        res.exposure.min_histogram_percentage = 0.7f;
        res.exposure.max_histogram_percentage = 0.95f;
        res.exposure.log_lumiance_bias = 0;
        res.exposure.eye_adaptation_enabled = false;
        res.exposure.eye_adaptation_brightness = INFINITY;
        res.exposure.eye_adaptation_darkness = INFINITY;
        res.bloom.threshold = INFINITY;
        res.bloom.support = 0.00f;
        res.vignette = 0.0f;
        res.tonemapping.mode = TonemappingMode::Linear;
        res.tonemapping.settings = TonemappingSettings::ACES();
        res.film_grain = 0.0f;
        return res;
    }
};

} // namespace CameraEffects
} // namespace Math
} // namespace Bifrost
