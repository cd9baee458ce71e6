// oracle/ref/reference_api.cpp -- C entry points onto the REAL reference code, for validating the restatements.
//
// TEST INFRASTRUCTURE ONLY. This file is this repository's own; everything it calls is compiled from the reference's sources
// where they lie under /root/reference (oracle/Makefile, target _ref): Bifrost/Assets/Shading/*.cpp (the rho and alpha tables
// and their lookups), Bifrost/Math/{CameraEffects,Utils,OctahedralNormal,Color}.h, Bifrost/Scene/Camera.cpp,
// Bifrost/Assets/{InfiniteAreaLight,Image,Texture}.cpp with Math/Distribution2D.h. No reference source is copied, patched or
// stood in for: the files build as they are with clang's -fms-extensions -fdelayed-template-parsing and the standard headers
// the MSVC dialect gets implicitly (-include cmath ...). What does not build that way (Math/RNG.cpp, Assets/Mesh*.cpp,
// apps/SmallPT, and everything that includes OptiX headers; extensions/StbImageLoader -- the reference's image loader with its vendored stb_image.h -- does build) stays out, and the oracle keeps being pinned by test vectors there. (Math/Distributions.h builds but is
// not the path's code: the renderer samples with its own differently parameterised OptiXRenderer/Distributions.h.)
#include <Bifrost/Assets/Image.h>
#include <Bifrost/Assets/InfiniteAreaLight.h>
#include <Bifrost/Assets/Shading/Fittings.h>
#include <Bifrost/Assets/Texture.h>
#include <Bifrost/Math/CameraEffects.h>
#include <Bifrost/Math/Color.h>
#include <Bifrost/Math/OctahedralNormal.h>
#include <Bifrost/Math/RNG.h>
#include <Bifrost/Math/Utils.h>
#include <Bifrost/Scene/Camera.h>
#include <Bifrost/Scene/SceneRoot.h>
#include <StbImageLoader/StbImageLoader.h>

#include <cstring>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace {
void allocate_managers() {
    static bool done = false;
    if (done) return;
    Images::allocate(8u); Textures::allocate(8u); SceneNodes::allocate(8u); SceneRoots::allocate(2u); Cameras::allocate(2u);
    done = true;
}
} // namespace

extern "C" {

// ---- Assets/Shading: tables and lookups ----------------------------------------------------------------------------------------
// which: 0 GGX_with_fresnel, 1 GGX, 2 dielectric into light medium, 3 dielectric into dense medium (float2 entries), 4 alphas
int ref_table(int which, const float** data, int* float_count) {
    using namespace Shading;
    switch (which) {
    case 0: *data = Rho::GGX_with_fresnel; *float_count = Rho::GGX_with_fresnel_angle_sample_count * Rho::GGX_with_fresnel_roughness_sample_count; return 0;
    case 1: *data = Rho::GGX; *float_count = Rho::GGX_angle_sample_count * Rho::GGX_roughness_sample_count; return 0;
    case 2: *data = &Rho::dielectric_GGX_into_light_medium[0].x; break;
    case 3: *data = &Rho::dielectric_GGX_into_dense_medium[0].x; break;
    case 4: *data = Estimate_GGX_bounded_VNDF_alpha::alphas; *float_count = Estimate_GGX_bounded_VNDF_alpha::wo_dot_normal_sample_count * Estimate_GGX_bounded_VNDF_alpha::max_PDF_sample_count; return 0;
    default: return -1;
    }
    *float_count = 2 * Rho::dielectric_GGX_angle_sample_count * Rho::dielectric_GGX_roughness_sample_count * Rho::dielectric_GGX_ior_i_over_o_sample_count;
    return 0;
}
void ref_dielectric_ior_ranges(float* out4) {
    out4[0] = Shading::Rho::dielectric_GGX_minimum_IOR_into_light_medium; out4[1] = Shading::Rho::dielectric_GGX_maximum_IOR_into_light_medium;
    out4[2] = Shading::Rho::dielectric_GGX_minimum_IOR_into_dense_medium; out4[3] = Shading::Rho::dielectric_GGX_maximum_IOR_into_dense_medium;
}
float ref_sample_GGX(float wo_dot_normal, float roughness) { return Shading::Rho::sample_GGX(wo_dot_normal, roughness); }
float ref_sample_GGX_with_fresnel(float wo_dot_normal, float roughness) { return Shading::Rho::sample_GGX_with_fresnel(wo_dot_normal, roughness); }
void ref_sample_dielectric_GGX(float wo_dot_normal, float roughness, float ior_i_over_o, float* out2) {
    const Shading::Rho::DielectricRho r = Shading::Rho::sample_dielectric_GGX(wo_dot_normal, roughness, ior_i_over_o);
    out2[0] = r.total_rho; out2[1] = r.reflected_rho;
}
float ref_estimate_alpha(float wo_dot_normal, float max_PDF) { return Shading::Estimate_GGX_bounded_VNDF_alpha::estimate_alpha(wo_dot_normal, max_PDF); }
float ref_encode_PDF(float pdf) { return Shading::Estimate_GGX_bounded_VNDF_alpha::encode_PDF(pdf); }

// ---- Math/CameraEffects.h, Math/Utils.h, Math/Color.h ---------------------------------------------------------------------------
// mode: 1 filmic(settings5 = black_clip, toe, slope, shoulder, white_clip), 2 agx, 3 khronos neutral, 4 reinhard(settings5[0] = white level squared)
void ref_tonemap(int mode, const float* settings5, const float* rgb_in, int count, float* rgb_out) {
    for (int i = 0; i < count; ++i) {
        const RGB in = RGB(rgb_in[3 * i], rgb_in[3 * i + 1], rgb_in[3 * i + 2]);
        RGB out = in;
        if (mode == 1) out = CameraEffects::filmic(in, settings5[2], settings5[1], settings5[3], settings5[0], settings5[4]);
        else if (mode == 2) out = CameraEffects::agx(in);
        else if (mode == 3) out = CameraEffects::khronos_neutral_tone_mapping(in);
        else if (mode == 4) out = CameraEffects::reinhard(in, settings5[0]);
        rgb_out[3 * i] = out.r; rgb_out[3 * i + 1] = out.g; rgb_out[3 * i + 2] = out.b;
    }
}
void ref_gaussian_taps(float std_dev, int count, float* offsets, float* weights) {
    Tap* taps = new Tap[count];
    fill_bilinear_gaussian_samples(std_dev, taps, taps + count);
    for (int i = 0; i < count; ++i) { offsets[i] = taps[i].offset; weights[i] = taps[i].weight; }
    delete[] taps;
}
float ref_sRGB_to_linear(float v) { return sRGB_to_linear(v); }
float ref_linear_to_sRGB(float v) { return linear_to_sRGB(v); }

// ---- Math/OctahedralNormal.h -----------------------------------------------------------------------------------------------------
void ref_octahedral_encode_precise(const float* normals_n3, int n, short* out_n2) {
    for (int i = 0; i < n; ++i) {
        const OctahedralNormal e = OctahedralNormal::encode_precise(Vector3f(normals_n3[3 * i], normals_n3[3 * i + 1], normals_n3[3 * i + 2]));
        out_n2[2 * i] = e.encoding.x; out_n2[2 * i + 1] = e.encoding.y;
    }
}
void ref_octahedral_decode(const short* encoded_n2, int n, float* out_n3) {
    for (int i = 0; i < n; ++i) {
        OctahedralNormal e; e.encoding = Vector2s(encoded_n2[2 * i], encoded_n2[2 * i + 1]);
        const Vector3f d = e.decode();
        out_n3[3 * i] = d.x; out_n3[3 * i + 1] = d.y; out_n3[3 * i + 2] = d.z;
    }
}

// ---- Math/RNG.h (the header's inline functions; RNG.cpp does not build outside MSVC) --------------------------------------
unsigned int ref_reverse_bits(unsigned int n) { return RNG::reverse_bits(n); }
unsigned int ref_jenkins_hash(unsigned int n) { return RNG::jenkins_hash(n); }
void ref_sample02(unsigned int n, float* out2) { const Vector2f s = RNG::sample02(n); out2[0] = s.x; out2[1] = s.y; }
float ref_power_heuristic(float pdf1, float pdf2) { return RNG::power_heuristic(pdf1, pdf2); }

// ---- Scene/Camera.cpp -------------------------------------------------------------------------------------------------------------
void ref_perspective_projection(float near_distance, float far_distance, float field_of_view, float aspect_ratio, float* projection16, float* inverse16) {
    Matrix4x4f p, ip;
    CameraUtils::compute_perspective_projection(near_distance, far_distance, field_of_view, aspect_ratio, p, ip);
    std::memcpy(projection16, p.begin(), 64); std::memcpy(inverse16, ip.begin(), 64);
}
void ref_orthographic_projection(float width, float height, float depth, float* projection16, float* inverse16) {
    Matrix4x4f p, ip;
    CameraUtils::compute_orthographic_projection(width, height, depth, p, ip);
    std::memcpy(projection16, p.begin(), 64); std::memcpy(inverse16, ip.begin(), 64);
}
// The rays of a perspective camera at `position` with rotation quaternion (x, y, z, w) through n viewport points: out_n6 = origin, direction.
void ref_rays_from_viewport_points(const float* position3, const float* rotation4, float near_distance, float far_distance, float field_of_view, float aspect_ratio,
                                   const float* viewport_points_n2, int n, float* out_n6) {
    allocate_managers();
    Matrix4x4f p, ip;
    CameraUtils::compute_perspective_projection(near_distance, far_distance, field_of_view, aspect_ratio, p, ip);
    SceneRoot scene = SceneRoot("Root", RGB::white());
    CameraID camera_ID = Cameras::create("Camera", scene.get_ID(), p, ip);
    Cameras::set_transform(camera_ID, Transform(Vector3f(position3[0], position3[1], position3[2]), Quaternionf(rotation4[0], rotation4[1], rotation4[2], rotation4[3])));
    for (int i = 0; i < n; ++i) {
        const Ray ray = CameraUtils::ray_from_viewport_point(camera_ID, Vector2f(viewport_points_n2[2 * i], viewport_points_n2[2 * i + 1]));
        out_n6[6 * i] = ray.origin.x; out_n6[6 * i + 1] = ray.origin.y; out_n6[6 * i + 2] = ray.origin.z;
        out_n6[6 * i + 3] = ray.direction.x; out_n6[6 * i + 4] = ray.direction.y; out_n6[6 * i + 5] = ray.direction.z;
    }
    Cameras::destroy(camera_ID);
    SceneRoots::destroy(scene.get_ID());
}

// ---- Assets/Texture.cpp sample2D over Images::get_pixel (Image.cpp) --------------------------------------------------------------------
// format / filters / wrap modes by enum value (the host mirror keeps the reference's); out_n4 = linear RGBA per texcoord.
void ref_sample2D(int format, int is_sRGB, int width, int height, const void* pixels, int byte_count, int magnification, int minification, int wrap_U, int wrap_V,
                  const float* uv_n2, int n, float* out_n4) {
    allocate_managers();
    Image image = Image::create2D("texture", PixelFormat(format), is_sRGB != 0, Vector2ui(width, height));
    std::memcpy(image.get_pixels(), pixels, size_t(byte_count));
    Texture texture = Texture::create2D(image, MagnificationFilter(magnification), MinificationFilter(minification), WrapMode(wrap_U), WrapMode(wrap_V));
    for (int i = 0; i < n; ++i) {
        const RGBA c = sample2D(texture, Vector2f(uv_n2[2 * i], uv_n2[2 * i + 1]));
        out_n4[4 * i] = c.r; out_n4[4 * i + 1] = c.g; out_n4[4 * i + 2] = c.b; out_n4[4 * i + 3] = c.a;
    }
    Textures::destroy(texture.get_ID());
    Images::destroy(image.get_ID());
}

// ---- Assets/InfiniteAreaLight.cpp over Image / Texture / Distribution2D -------------------------------------------------------------
// An RGBA float latitude-longitude image -> samples (radiance[3], PDF, direction[3], distance) for n random pairs, PDF(direction) of
// those sample directions, the PDF image's size and (when capacity allows) the per pixel solid angle PDF sans sin(theta).
int ref_infinite_area_light(int width, int height, const float* rgba, const float* u_n2, int n, float* out_samples_n8, float* out_pdf_n, int* out_pdf_size2,
                            float* out_per_pixel_PDF, int per_pixel_capacity) {
    allocate_managers();
    Image image = Image::create2D("environment", PixelFormat::RGBA_Float, false, Vector2ui(width, height));
    std::memcpy(image.get_pixels(), rgba, size_t(width) * height * 16);
    Texture texture = Texture::create2D(image, MagnificationFilter::Linear, MinificationFilter::Linear, WrapMode::Repeat, WrapMode::Clamp);
    int status = 0;
    {
        InfiniteAreaLight light(texture);
        for (int i = 0; i < n; ++i) {
            const LightSample s = light.sample(Vector2f(u_n2[2 * i], u_n2[2 * i + 1]));
            float* o = out_samples_n8 + 8 * i;
            o[0] = s.radiance.r; o[1] = s.radiance.g; o[2] = s.radiance.b; o[3] = s.PDF;
            o[4] = s.direction_to_light.x; o[5] = s.direction_to_light.y; o[6] = s.direction_to_light.z; o[7] = s.distance;
            out_pdf_n[i] = light.PDF(s.direction_to_light);
        }
        out_pdf_size2[0] = int(light.get_PDF_width()); out_pdf_size2[1] = int(light.get_PDF_height());
        if (out_per_pixel_PDF && per_pixel_capacity >= out_pdf_size2[0] * out_pdf_size2[1])
            InfiniteAreaLightUtils::reconstruct_solid_angle_PDF_sans_sin_theta(light, out_per_pixel_PDF);
        else if (out_per_pixel_PDF)
            status = 1;
    }
    Textures::destroy(texture.get_ID());
    Images::destroy(image.get_ID());
    return status;
}

// The reference's image loader (extensions/StbImageLoader/StbImageLoader/StbImageLoader.cpp:99-113, stb_image 2.29 underneath): the file's pixels as
// the reference's Image holds them (bottom row first). Returns the byte count, 0 when the file does not load; info4 = width, height, channels, float flag.
size_t ref_image_load(const char* path, int* info4, void* out, size_t capacity) {
    allocate_managers();
    Image image = StbImageLoader::load(path);
    if (!image.exists()) return 0;
    const PixelFormat format = image.get_pixel_format();
    const bool floats = format == PixelFormat::Intensity_Float || format == PixelFormat::RGB_Float || format == PixelFormat::RGBA_Float;
    const int channels = channel_count(format);
    const size_t bytes = size_t(image.get_width()) * image.get_height() * channels * (floats ? 4 : 1);
    info4[0] = int(image.get_width()); info4[1] = int(image.get_height()); info4[2] = channels; info4[3] = floats;
    if (out && capacity >= bytes) std::memcpy(out, image.get_pixels(), bytes);
    Images::destroy(image.get_ID());
    return bytes;
}

} // extern "C"
