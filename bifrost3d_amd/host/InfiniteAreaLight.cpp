// InfiniteAreaLight.cpp -- see InfiniteAreaLight.h.
#include "InfiniteAreaLight.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace Bifrost::Math;

namespace Bifrost {
namespace Assets {

static float sRGB_to_linear(float v) { return v < 0.04045f ? v * 0.0773993808f : std::pow(v * 0.9478672986f + 0.0521327014f, 2.4f); }   // Color.h:356-361

RGBA get_pixel(ImageID image_ID, unsigned int x, unsigned int y) {
    const size_t index = x + size_t(y) * Images::get_width(image_ID);
    const unsigned char* bytes = static_cast<const unsigned char*>(Images::get_pixels(image_ID));
    const float* floats = static_cast<const float*>(Images::get_pixels(image_ID));
    const float unorm8 = 1.0f / 255.0f;
    RGBA c = {1, 0, 0, 1};
    switch (Images::get_pixel_format(image_ID)) {
    case PixelFormat::Alpha8: c = {1.0f, 1.0f, 1.0f, bytes[index] * unorm8}; break;
    case PixelFormat::Intensity8: { const float i = bytes[index] * unorm8; c = {i, i, i, 1.0f}; break; }
    case PixelFormat::RGB24: c = {bytes[3 * index] * unorm8, bytes[3 * index + 1] * unorm8, bytes[3 * index + 2] * unorm8, 1.0f}; break;
    case PixelFormat::RGBA32: c = {bytes[4 * index] * unorm8, bytes[4 * index + 1] * unorm8, bytes[4 * index + 2] * unorm8, bytes[4 * index + 3] * unorm8}; break;
    case PixelFormat::Intensity_Float: c = {floats[index], floats[index], floats[index], 1.0f}; break;
    case PixelFormat::RGB_Float: c = {floats[3 * index], floats[3 * index + 1], floats[3 * index + 2], 1.0f}; break;
    case PixelFormat::RGBA_Float: c = {floats[4 * index], floats[4 * index + 1], floats[4 * index + 2], floats[4 * index + 3]}; break;
    default: break;
    }
    if (Images::is_sRGB(image_ID)) { c.r = sRGB_to_linear(c.r); c.g = sRGB_to_linear(c.g); c.b = sRGB_to_linear(c.b); }
    return c;
}

static RGBA lerp(RGBA a, RGBA b, float t) { return {a.r + (b.r - a.r) * t, a.g + (b.g - a.g) * t, a.b + (b.b - a.b) * t, a.a + (b.a - a.a) * t}; }

RGBA sample2D(TextureID texture_ID, Vector2f texcoord) {
    const ImageID image_ID = Textures::get_image_ID(texture_ID);
    const bool clamp_u = Textures::get_wrapmode_U(texture_ID) == WrapMode::Clamp, clamp_v = Textures::get_wrapmode_V(texture_ID) == WrapMode::Clamp;
    auto wrap = [](float t, bool clamp) {
        if (clamp) return std::fmin(std::fmax(t, 0.0f), nearly_one);
        t -= float(int(t));
        return t < -0.0f ? t + 1.0f : t;
    };
    texcoord = {wrap(texcoord.x, clamp_u), wrap(texcoord.y, clamp_v)};

    const int width = int(Images::get_width(image_ID)), height = int(Images::get_height(image_ID));
    if (Textures::get_minification_filter(texture_ID) == MinificationFilter::None)
        return get_pixel(image_ID, unsigned(texcoord.x * float(width)), unsigned(texcoord.y * float(height)));

    const float px = texcoord.x * float(width) - 0.5f, py = texcoord.y * float(height) - 0.5f;
    const int x0 = int(px), y0 = int(py);
    float u_lerp = px - float(x0), v_lerp = py - float(y0);
    if (u_lerp < 0.0f) u_lerp += 1.0f;
    if (v_lerp < 0.0f) v_lerp += 1.0f;
    auto lookup = [&](int x, int y) {
        x = clamp_u ? std::min(std::max(x, 0), width - 1) : (x + width) % width;
        y = clamp_v ? std::min(std::max(y, 0), height - 1) : (y + height) % height;
        return get_pixel(image_ID, unsigned(x), unsigned(y));
    };
    const RGBA lower = lerp(lookup(x0, y0), lookup(x0 + 1, y0), u_lerp), upper = lerp(lookup(x0, y0 + 1), lookup(x0 + 1, y0 + 1), u_lerp);
    return lerp(lower, upper, v_lerp);
}

// Per texel importance: (r + g + b) * sin(theta) (PBRT's account for the shrinking texels near the poles). Images lower than
// MINIMUM_PDF_HEIGHT are resampled to that height; filtered textures get their importance blurred with the 3x3 tent the
// bilinear footprint implies, so that a black texel next to a bright one keeps a non-zero PDF (InfiniteAreaLight.cpp:15-132).
static Distribution2D<float> compute_distribution(TextureID latlong) {
    const ImageID image = Textures::get_image_ID(latlong);
    const int width = int(Images::get_width(image));
    const int height = int(std::max(Images::get_height(image), InfiniteAreaLight::MINIMUM_PDF_HEIGHT));
    const bool resample_height = height != int(Images::get_height(image));
    const bool filter_pixels = Textures::get_magnification_filter(latlong) == MagnificationFilter::Linear || resample_height;

    std::vector<float> importance(size_t(width) * height);
    for (int y = 0; y < height; ++y) {
        const float sin_theta = std::sin(PI<float>() * (y + 0.5f) / float(height));
        for (int x = 0; x < width; ++x) {
            const RGBA pixel = resample_height ? sample2D(latlong, {(x + 0.5f) / width, (y + 0.5f) / height}) : get_pixel(image, unsigned(x), unsigned(y));
            importance[x + size_t(y) * width] = (pixel.r + pixel.g + pixel.b) * sin_theta;
        }
    }
    if (!filter_pixels) return Distribution2D<float>(importance.data(), width, height);

    std::vector<float> blurred(importance.size());
    for (int y = 0; y < height; ++y) {
        const int below = std::max(0, y - 1), above = std::min(height - 1, y + 1);   // clamp vertically
        for (int x = 0; x < width; ++x) {
            const int left = x - 1 < 0 ? width - 1 : x - 1, right = x + 1 == width ? 0 : x + 1;   // repeat horizontally
            auto at = [&](int px, int py) { return importance[px + size_t(py) * width]; };
            float sum = 0.0f;   // same accumulation order as the reference: left column, right column, middle column, centre last
            sum += at(left, below); sum += at(left, y) * 2.0f; sum += at(left, above);
            sum += at(right, below); sum += at(right, y) * 2.0f; sum += at(right, above);
            sum += at(x, below) * 2.0f; sum += at(x, above) * 2.0f;
            sum += at(x, y) * 20.0f;
            blurred[x + size_t(y) * width] = sum / 32.0f;
        }
    }
    return Distribution2D<float>(blurred.data(), width, height);
}

InfiniteAreaLight::InfiniteAreaLight(TextureID latlong) : m_latlong(latlong), m_distribution(compute_distribution(latlong)) {}

LightSample InfiniteAreaLight::sample(Vector2f random_sample) const {
    const auto cdf_sample = m_distribution.sample_continuous(random_sample);
    const Vector2f uv = {cdf_sample.x, cdf_sample.y};
    LightSample sample;
    sample.direction_to_light = latlong_texcoord_to_direction(uv);
    sample.distance = 1e30f;
    sample.radiance = evaluate(uv);
    const float sin_theta = std::fabs(std::sqrt(1.0f - sample.direction_to_light.y * sample.direction_to_light.y));
    const float pdf = float(cdf_sample.PDF) / (2.0f * PI<float>() * PI<float>() * sin_theta);
    sample.PDF = sin_theta == 0.0f ? 0.0f : pdf;
    return sample;
}

float InfiniteAreaLight::PDF(Vector3f direction_to_light) const {
    const float sin_theta = std::fabs(std::sqrt(1.0f - direction_to_light.y * direction_to_light.y));
    Vector2f uv = direction_to_latlong_texcoord(direction_to_light);
    uv.y = std::fmin(uv.y, nearly_one);
    const float pdf = float(m_distribution.PDF_continuous(uv)) / (2.0f * PI<float>() * PI<float>() * sin_theta);
    return sin_theta == 0.0f ? 0.0f : pdf;
}

namespace InfiniteAreaLightUtils {

void reconstruct_solid_angle_PDF_sans_sin_theta(const InfiniteAreaLight& light, float* per_pixel_PDF) {
    const int width = int(light.get_PDF_width()), height = int(light.get_PDF_height());
    const float PDF_scale = float(width * height) * (1.0f / (2.0f * PI<float>() * PI<float>()));
    for (int y = 0; y < height; ++y) {
        const float marginal_PDF = light.get_image_marginal_CDF()[y + 1] - light.get_image_marginal_CDF()[y];
        for (int x = 0; x < width; ++x) {
            const float* conditional = light.get_image_conditional_CDF() + x + size_t(y) * (width + 1);
            per_pixel_PDF[x + size_t(y) * width] = marginal_PDF * (conditional[1] - conditional[0]) * PDF_scale;
        }
    }
}

} // namespace InfiniteAreaLightUtils
} // namespace Assets
} // namespace Bifrost
