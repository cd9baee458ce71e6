// InfiniteAreaLight.h -- importance sampled latitude-longitude environment light.
// Mirror of core/Bifrost/Bifrost/Assets/InfiniteAreaLight.h:25-125 and InfiniteAreaLight.cpp:15-160, plus the texture lookup
// it relies on (Assets::sample2D, core/Bifrost/Bifrost/Assets/Texture.cpp:114-176, and Images::get_pixel, Image.cpp:221-286).
#pragma once

#include "Bifrost.h"
#include "Distribution2D.h"

namespace Bifrost {
namespace Assets {

// Linear RGBA of a pixel: 8-bit formats as UNorm8, single channel formats replicated (Alpha8: white with that alpha), sRGB decoded.
Math::RGBA get_pixel(ImageID image_ID, unsigned int x, unsigned int y);

// Wrap modes and filtering of the texture applied on the host. Like the reference, the MINIFICATION filter selects between
// nearest and bilinear.
Math::RGBA sample2D(TextureID texture_ID, Math::Vector2f texcoord);

struct LightSample {
    Math::RGB radiance;
    float PDF;
    Math::Vector3f direction_to_light;
    float distance;
};

class InfiniteAreaLight {
public:
    static constexpr unsigned int MINIMUM_PDF_HEIGHT = 128;      // constexpr: an inline variable, std::max binds a reference to it

    explicit InfiniteAreaLight(TextureID latlong);

    TextureID get_texture_ID() const { return m_latlong; }
    unsigned int get_width() const { return Images::get_width(Textures::get_image_ID(m_latlong)); }
    unsigned int get_height() const { return Images::get_height(Textures::get_image_ID(m_latlong)); }
    const float* get_image_marginal_CDF() const { return m_distribution.get_marginal_CDF(); }
    const float* get_image_conditional_CDF() const { return m_distribution.get_conditional_CDF(); }
    unsigned int get_PDF_width() const { return unsigned(m_distribution.get_width()); }
    unsigned int get_PDF_height() const { return unsigned(m_distribution.get_height()); }
    float image_integral() const { return m_distribution.get_integral(); }

    Math::RGB evaluate(Math::Vector2f uv) const { const Math::RGBA c = sample2D(m_latlong, uv); return Math::RGB(c.r, c.g, c.b); }
    Math::RGB evaluate(Math::Vector3f direction_to_light) const {
        Math::Vector2f uv = Math::direction_to_latlong_texcoord(direction_to_light);
        uv.y = std::fmin(uv.y, Math::nearly_one);
        return evaluate(uv);
    }

    // Importance samples the image; PDF with respect to solid angle (0 at the poles).
    LightSample sample(Math::Vector2f random_sample) const;
    float PDF(Math::Vector3f direction_to_light) const;

private:
    TextureID m_latlong;
    Math::Distribution2D<float> m_distribution;
};

namespace InfiniteAreaLightUtils {
// Per PDF texel: the solid angle PDF times sin(theta) (the renderer divides by sin(theta) per direction). width * height floats.
void reconstruct_solid_angle_PDF_sans_sin_theta(const InfiniteAreaLight& light, float* per_pixel_PDF);
}

} // namespace Assets
} // namespace Bifrost
