// EnvironmentTest.cpp -- host-side environment light math against the reference's own test cases:
//   Distribution2D      tests/BifrostTests/Math/Distribution2DTest.h:21-150
//   InfiniteAreaLight   tests/BifrostTests/Assets/InfiniteAreaLightTest.h:34-130
// EXPECT_FLOAT_EQ of googletest is "within 4 ulp"; here it is a relative 1e-6.
#include "MiniTest.h"

#include "../../bifrost3d_amd/host/Distribution2D.h"
#include "../../bifrost3d_amd/host/InfiniteAreaLight.h"
#include "../../bifrost3d_amd/host/RNG.h"

#include <algorithm>
#include <cstring>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;

namespace {

struct EnvironmentFixture {
    void SetUp() { deallocate_all(); }
    void TearDown() { deallocate_all(); }
    bool usable() const { return true; }
};

#define EXPECT_FLOAT_NEAR_REL(expected, actual) EXPECT_FLOAT_EQ_EPS(expected, actual, 1e-6 * std::fmax(1.0, std::fabs(double(expected))))

float previous_float(float v) { return std::nextafter(v, -1e30f); }

TextureID intensity_texture(const char* name, unsigned width, unsigned height, const unsigned char* pixels, MagnificationFilter mag, MinificationFilter min) {
    const ImageID image = Images::create2D(name, PixelFormat::Intensity8, false, width, height, pixels, size_t(width) * height);
    return Textures::create2D(image, mag, min, WrapMode::Repeat, WrapMode::Clamp);
}

} // namespace

CPU_TEST_F(EnvironmentFixture, distribution2D_single_value_and_constant_function) {
    const double one = 1.0;
    const Distribution2D<float> single(&one, 1, 1);
    for (Vector2f r : {Vector2f{0.0f, 0.0f}, Vector2f{nearly_one, 0.0f}, Vector2f{0.0f, nearly_one}, Vector2f{nearly_one, nearly_one}, Vector2f{0.5f, 0.5f}}) {
        EXPECT_EQ(0, single.sample_discrete(r).x);
        EXPECT_EQ(0, single.sample_discrete(r).y);
        EXPECT_FLOAT_NEAR_REL(1.0f, single.sample_discrete(r).PDF);
    }

    const float f[] = {3, 3, 3, 3};
    const Distribution2D<float> constant(f, 2, 2);
    for (int y = 0; y < 5; ++y)
        for (int x = 0; x < 5; ++x) {
            const Vector2f r = {x / 5.0f, y / 5.0f};
            EXPECT_FLOAT_NEAR_REL(3.0f, constant.evaluate(r));
            EXPECT_FLOAT_NEAR_REL(r.x, constant.sample_continuous(r).x);
            EXPECT_FLOAT_NEAR_REL(r.y, constant.sample_continuous(r).y);
            EXPECT_FLOAT_NEAR_REL(1.0f, constant.sample_continuous(r).PDF);
        }
}

CPU_TEST_F(EnvironmentFixture, distribution2D_non_constant_function) {
    const float f[] = {0, 5, 0, 3, 2, 1, 1, 4};
    const Distribution2D<float> distribution(f, 4, 2);
    const float integral = distribution.get_integral();
    for (int y = 0; y < 2; ++y)
        for (int x = 0; x < 4; ++x) EXPECT_FLOAT_NEAR_REL(f[x + 4 * y], distribution.evaluate(x, y));

    auto s = distribution.sample_continuous({0.0f, 0.0f});
    EXPECT_FLOAT_NEAR_REL(0.25f, s.x); EXPECT_FLOAT_NEAR_REL(0.0f, s.y); EXPECT_FLOAT_NEAR_REL(f[1] / integral, s.PDF);
    s = distribution.sample_continuous({0.3125f, 0.25f});
    EXPECT_FLOAT_NEAR_REL(0.375f, s.x); EXPECT_FLOAT_NEAR_REL(0.25f, s.y); EXPECT_FLOAT_NEAR_REL(f[1] / integral, s.PDF);
    s = distribution.sample_continuous({nearly_one, previous_float(0.5f)});
    EXPECT_FLOAT_NEAR_REL(1.0f, s.x); EXPECT_FLOAT_NEAR_REL(0.5f, s.y); EXPECT_FLOAT_NEAR_REL(f[3] / integral, s.PDF);

    for (unsigned i = 0; i < 32; ++i) {   // consistent_PDF
        const auto continuous = distribution.sample_continuous(RNG::sample02(i));
        EXPECT_FLOAT_NEAR_REL(continuous.PDF, distribution.PDF_continuous({continuous.x, continuous.y}));
        const auto discrete = distribution.sample_discrete(RNG::sample02(i));
        EXPECT_FLOAT_NEAR_REL(discrete.PDF, distribution.PDF_discrete(discrete.x, discrete.y));
    }

    // reconstruct_continuous_function / reconstruct_discrete_function: importance sampled estimates of every cell
    const int iterations = 8192;
    float continuous_estimate[8] = {}, discrete_estimate[8] = {};
    for (int i = 0; i < iterations; ++i) {
        const auto c = distribution.sample_continuous(RNG::sample02(i, 0u, 0u));
        continuous_estimate[int(c.x * 4) + int(c.y * 2) * 4] += distribution.evaluate(Vector2f{c.x, c.y}) / c.PDF * 8;
        const auto d = distribution.sample_discrete(RNG::sample02(i, 0u, 0u));
        discrete_estimate[d.x + d.y * 4] += distribution.evaluate(d.x, d.y) / d.PDF;
    }
    for (int e = 0; e < 8; ++e) {
        EXPECT_FLOAT_EQ_EPS(f[e], continuous_estimate[e] / iterations, 1e-5f * std::fmax(1.0f, f[e]));
        EXPECT_FLOAT_EQ_EPS(f[e], discrete_estimate[e] / iterations, 1e-5f * std::fmax(1.0f, f[e]));
    }
}

CPU_TEST_F(EnvironmentFixture, infinite_area_light_consistent_PDF_and_evaluate) {
    const unsigned char f[] = {0, 5, 0, 3, 1, 2, 1, 4, 3, 7, 5, 1, 9, 4, 1, 1};
    const InfiniteAreaLight light(intensity_texture("Noisy", 4, 4, f, MagnificationFilter::Linear, MinificationFilter::Linear));
    for (unsigned i = 0; i < 32; ++i) {
        const LightSample sample = light.sample(RNG::sample02(i));
        EXPECT_FLOAT_NEAR_REL(sample.PDF, light.PDF(sample.direction_to_light));
        const RGB evaluated = light.evaluate(sample.direction_to_light);
        EXPECT_FLOAT_EQ_EPS(sample.radiance.r, evaluated.r, 0.000001f);
        EXPECT_FLOAT_EQ_EPS(sample.radiance.g, evaluated.g, 0.000001f);
        EXPECT_FLOAT_EQ_EPS(sample.radiance.b, evaluated.b, 0.000001f);
    }
}

CPU_TEST_F(EnvironmentFixture, infinite_area_light_diffuse_integrates_to_white) {
    std::vector<unsigned char> white(256, 255);
    const InfiniteAreaLight light(intensity_texture("White", 1, 256, white.data(), MagnificationFilter::Linear, MinificationFilter::Linear));
    const int sample_count = 8192;
    for (int up_axis = 1; up_axis <= 2; ++up_axis) {   // a diffuse surface with y, then z as its normal
        std::vector<double> radiance(sample_count);
        for (int i = 0; i < sample_count; ++i) {
            const LightSample sample = light.sample(RNG::sample02(i));
            const float cos_theta = std::fmax(0.0f, up_axis == 1 ? sample.direction_to_light.y : sample.direction_to_light.z);
            radiance[i] = sample.PDF != 0.0f ? sample.radiance.r / PI<float>() * cos_theta / sample.PDF : 0.0f;
        }
        std::sort(radiance.begin(), radiance.end());
        double sum = 0.0;
        for (double r : radiance) sum += r;
        EXPECT_TRUE(0.9999 < sum / sample_count && sum / sample_count < 1.0001);
    }
}

CPU_TEST_F(EnvironmentFixture, infinite_area_light_PDF_resampling) {
    const unsigned minimum = InfiniteAreaLight::MINIMUM_PDF_HEIGHT;
    const unsigned char small_pixels[] = {0, 1};
    std::vector<unsigned char> large_pixels(minimum);
    for (unsigned p = 0; p < minimum; ++p) large_pixels[p] = p < minimum / 2 ? 0 : 1;
    const InfiniteAreaLight small_light(intensity_texture("Small", 1, 2, small_pixels, MagnificationFilter::None, MinificationFilter::None));
    const InfiniteAreaLight large_light(intensity_texture("Large", 1, minimum, large_pixels.data(), MagnificationFilter::None, MinificationFilter::None));
    EXPECT_EQ(1u, small_light.get_PDF_width());
    EXPECT_EQ(minimum, small_light.get_PDF_height());   // resampled
    EXPECT_EQ(1u, large_light.get_PDF_width());
    EXPECT_EQ(minimum, large_light.get_PDF_height());   // one to one
    for (unsigned i = 1; i < 32; ++i) {   // not the first sample: it is next to the pole, where the PDF filtering deviates
        const LightSample a = small_light.sample(RNG::sample02(i)), b = large_light.sample(RNG::sample02(i));
        EXPECT_TRUE(dot(a.direction_to_light, b.direction_to_light) > std::cos(0.01f));
        EXPECT_FLOAT_EQ_EPS(a.PDF, b.PDF, 0.02f);
    }
}

CPU_TEST_F(EnvironmentFixture, per_pixel_PDF_reconstructs_the_solid_angle_PDF) {
    // reconstruct_solid_angle_PDF_sans_sin_theta / sin(theta) is what light.PDF returns for the same direction
    const unsigned char f[] = {0, 5, 0, 3, 1, 2, 1, 4, 3, 7, 5, 1, 9, 4, 1, 1};
    const InfiniteAreaLight light(intensity_texture("Noisy", 4, 4, f, MagnificationFilter::Linear, MinificationFilter::Linear));
    std::vector<float> per_pixel(size_t(light.get_PDF_width()) * light.get_PDF_height());
    InfiniteAreaLightUtils::reconstruct_solid_angle_PDF_sans_sin_theta(light, per_pixel.data());
    for (unsigned i = 0; i < 64; ++i) {
        const LightSample sample = light.sample(RNG::sample02(i));
        Vector2f uv = direction_to_latlong_texcoord(sample.direction_to_light);
        uv.y = std::fmin(uv.y, nearly_one);
        const unsigned x = unsigned(uv.x * light.get_PDF_width()) % light.get_PDF_width(), y = unsigned(uv.y * light.get_PDF_height());
        const float sin_theta = std::sqrt(1.0f - sample.direction_to_light.y * sample.direction_to_light.y);
        EXPECT_FLOAT_EQ_EPS(light.PDF(sample.direction_to_light), per_pixel[x + y * light.get_PDF_width()] / sin_theta, 1e-4f * std::fmax(1.0f, sample.PDF));
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Through the renderer: SceneRoot::set_environment_map -> handle_updates -> presampled environment -> miss evaluation.
// ------------------------------------------------------------------------------------------------------------------------
#include "../../bifrost3d_amd/host/HIPRenderer/Renderer.h"

#include <filesystem>

GPU_TEST_F(EnvironmentFixture, renderer_shows_the_environment_map_behind_an_empty_scene) {
    using namespace Bifrost::Scene;
    std::error_code error;
    const std::filesystem::path data = std::filesystem::read_symlink("/proc/self/exe", error).parent_path() / ".." / ".." / "bifrost3d_amd" / "data";
    HIPRenderer::Renderer* renderer = HIPRenderer::Renderer::initialize(0, data);
    EXPECT_TRUE(renderer != nullptr);
    if (!renderer) return;

    // lower hemisphere green, upper hemisphere blue, nearest filtering: a perspective camera looking along +z sees both halves
    const unsigned width = 8, height = 4;
    std::vector<float> pixels(width * height * 4);
    for (unsigned y = 0; y < height; ++y)
        for (unsigned x = 0; x < width; ++x) {
            float* p = pixels.data() + 4 * (x + y * width);
            p[0] = 0.0f; p[1] = y < height / 2 ? 0.5f : 0.0f; p[2] = y < height / 2 ? 0.0f : 2.0f; p[3] = 1.0f;
        }
    const ImageID image = Images::create2D("two tone sky", PixelFormat::RGBA_Float, false, width, height, pixels.data(), pixels.size() * 4);
    const TextureID environment = Textures::create2D(image, MagnificationFilter::None, MinificationFilter::None, WrapMode::Repeat, WrapMode::Clamp);

    SceneRoot scene = SceneRoot("Sky", RGB(1.0f, 1.0f, 0.5f));   // the tint scales the map
    scene.set_environment_map(environment);
    const Vector2i frame_size(32, 18);
    Matrix4x4f projection, inverse_projection;
    CameraUtils::compute_perspective_projection(0.1f, 100.0f, PI<float>() / 3.0f, float(frame_size.x) / frame_size.y, projection, inverse_projection);
    const CameraID camera_ID = Cameras::create("Camera", scene.get_ID(), projection, inverse_projection);
    Cameras::set_renderer_ID(camera_ID, renderer->get_renderer_ID());

    renderer->handle_updates();
    EXPECT_EQ(1u, renderer->render(camera_ID, nullptr, 0, frame_size));
    std::vector<double> accumulation;
    EXPECT_TRUE(renderer->read_accumulation(accumulation));
    // latlong v = (asin(y) + pi/2) / pi: rays pointing down (image rows at the bottom) land in the first rows of the map (green),
    // rays pointing up in the last rows (blue); tint (1, 1, 0.5).
    const double* bottom = accumulation.data() + 4 * (frame_size.x / 2 + 1 * frame_size.x);
    const double* top = accumulation.data() + 4 * (frame_size.x / 2 + (frame_size.y - 2) * frame_size.x);
    EXPECT_FLOAT_EQ_EPS(0.5, bottom[1], 1e-6); EXPECT_FLOAT_EQ_EPS(0.0, bottom[2], 1e-6);
    EXPECT_FLOAT_EQ_EPS(0.0, top[1], 1e-6); EXPECT_FLOAT_EQ_EPS(1.0, top[2], 1e-6);

    // removing nothing but changing the tint restarts the accumulation and rescales the map
    scene.set_environment_tint(RGB(2.0f, 2.0f, 2.0f));
    renderer->handle_updates();
    EXPECT_EQ(1u, renderer->render(camera_ID, nullptr, 0, frame_size));
    EXPECT_TRUE(renderer->read_accumulation(accumulation));
    EXPECT_FLOAT_EQ_EPS(4.0, accumulation[4 * (frame_size.x / 2 + (frame_size.y - 2) * frame_size.x) + 2], 1e-6);
    delete renderer;
}
