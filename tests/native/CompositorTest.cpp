// CompositorTest.cpp -- the headless compositor: per camera, renderer -> camera effects -> the window's RGBA8 back buffer
// (mirror of the render loop of DX11Renderer::Compositor, Compositor.cpp:255-325), and the settings plumbing around it.
#include "MiniTest.h"

#include "../../bifrost3d_amd/host/HIPRenderer/Compositor.h"

#include <cmath>

using namespace Bifrost;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace {

std::filesystem::path data_directory() {
    std::error_code error;
    return std::filesystem::read_symlink("/proc/self/exe", error).parent_path() / ".." / ".." / "bifrost3d_amd" / "data";
}

struct CompositorFixture {
    void SetUp() { deallocate_all(); }
    void TearDown() { deallocate_all(); }
    bool usable() const { return true; }

    // An orthographic camera over an empty scene: every pixel is the environment tint (RendererTest.h:147-153).
    CameraID create_ortho_camera(Vector2i size, RGB environment_tint) { return create_ortho_camera(size, SceneRoot("Test", environment_tint)); }
    CameraID create_ortho_camera(Vector2i size, SceneRoot scene) {
        Matrix4x4f orthographic_matrix, inverse_orthographic_matrix;
        CameraUtils::compute_orthographic_projection(float(size.x), float(size.y), 1000.0f, orthographic_matrix, inverse_orthographic_matrix);
        return Cameras::create("Test", scene.get_ID(), orthographic_matrix, inverse_orthographic_matrix);
    }
};

int to_sRGB8(float linear) {
    linear = std::fmin(std::fmax(linear, 0.0f), 1.0f);
    const float encoded = linear < 0.0031308f ? linear * 12.92f : 1.055f * std::pow(linear, 1.0f / 2.4f) - 0.055f;
    return int(encoded * 255.0f + 0.5f);
}

} // namespace

CPU_TEST_F(CompositorFixture, cameras_carry_the_effects_preset) {
    const CameraID camera_ID = create_ortho_camera(Vector2i(4, 4), RGB(1, 1, 1));
    const Math::CameraEffects::Settings preset = Cameras::get_effects_settings(camera_ID);       // Camera.cpp:157
    EXPECT_TRUE(preset.exposure.mode == Math::CameraEffects::ExposureMode::Histogram);
    EXPECT_TRUE(preset.tonemapping.mode == Math::CameraEffects::TonemappingMode::Filmic);
    EXPECT_EQ(0.63f, preset.vignette);
    EXPECT_TRUE(std::isinf(preset.bloom.threshold));
    EXPECT_FLOAT_EQ_EPS(13.5f, Math::CameraEffects::Settings(preset).bloom.std_dev(1080), 1e-4f);     // 5 % of the height / 4

    const HiprCameraEffectsSettings c = HIPRenderer::to_c_settings(preset);
    EXPECT_EQ(int(HIPR_EXPOSURE_HISTOGRAM), c.exposure_mode); EXPECT_EQ(int(HIPR_TONEMAPPING_FILMIC), c.tonemapping_mode);
    EXPECT_EQ(-4.0f, c.min_log_luminance); EXPECT_EQ(4.0f, c.max_log_luminance);
    EXPECT_EQ(0.7f, c.min_histogram_percentage); EXPECT_EQ(0.95f, c.max_histogram_percentage);
    EXPECT_EQ(1, c.eye_adaptation_enabled); EXPECT_EQ(3.0f, c.eye_adaptation_brightness); EXPECT_EQ(1.0f, c.eye_adaptation_darkness);
    EXPECT_EQ(0.53f, c.tonemapping_toe); EXPECT_EQ(0.91f, c.tonemapping_slope); EXPECT_EQ(0.23f, c.tonemapping_shoulder); EXPECT_EQ(0.035f, c.tonemapping_white_clip);
    EXPECT_FLOAT_EQ_EPS(1.0f / 255.0f, c.film_grain, 1e-9f);

    Cameras::set_effects_settings(camera_ID, Math::CameraEffects::Settings::linear());
    const HiprCameraEffectsSettings linear = HIPRenderer::to_c_settings(Cameras::get_effects_settings(camera_ID));
    EXPECT_EQ(int(HIPR_EXPOSURE_FIXED), linear.exposure_mode); EXPECT_EQ(int(HIPR_TONEMAPPING_LINEAR), linear.tonemapping_mode);
    EXPECT_EQ(0, linear.eye_adaptation_enabled); EXPECT_EQ(0.0f, linear.vignette); EXPECT_EQ(0.0f, linear.film_grain);

    int x, y, width, height;
    Cameras::set_viewport(camera_ID, 0.5f, 0.0f, 0.5f, 1.0f);
    Cameras::get_window_viewport(camera_ID, Vector2i(64, 36), x, y, width, height);
    EXPECT_EQ(32, x); EXPECT_EQ(0, y); EXPECT_EQ(32, width); EXPECT_EQ(36, height);
}

GPU_TEST_F(CompositorFixture, composites_two_cameras_into_their_viewports) {
    const Vector2i window_size(64, 36);
    HIPRenderer::HeadlessCompositor* compositor = HIPRenderer::HeadlessCompositor::initialize(0, data_directory(), window_size);
    EXPECT_TRUE(compositor != nullptr);
    if (!compositor) return;
    const Core::RendererID renderer_ID = compositor->add_renderer(HIPRenderer::HeadlessAdaptor::initialize);
    EXPECT_TRUE(renderer_ID != Core::RendererID::invalid_UID());
    if (renderer_ID == Core::RendererID::invalid_UID()) { delete compositor; return; }

    // Two cameras on one scene (a renderer holds one scene, like the reference's). Left half: the linear settings show the
    // scene's colour as it is. Right half: the preset without eye adaptation exposes it, vignettes the corners and adds grain.
    const RGB left_tint(0.1f, 0.5f, 0.9f);
    SceneRoot scene = SceneRoot("Test", left_tint);
    const CameraID left = create_ortho_camera(Vector2i(32, 36), scene), right = create_ortho_camera(Vector2i(32, 36), scene);
    Cameras::set_renderer_ID(left, renderer_ID); Cameras::set_renderer_ID(right, renderer_ID);
    Cameras::set_viewport(left, 0.0f, 0.0f, 0.5f, 1.0f); Cameras::set_viewport(right, 0.5f, 0.0f, 0.5f, 1.0f);
    Cameras::set_effects_settings(left, Math::CameraEffects::Settings::linear());
    Math::CameraEffects::Settings preset = Math::CameraEffects::Settings::preset();
    preset.exposure.eye_adaptation_enabled = false;
    Cameras::set_effects_settings(right, preset);

    EXPECT_EQ(2u, compositor->render(1.0f / 60.0f));
    EXPECT_EQ(1u, compositor->get_iteration_count(left));
    reset_all_change_notifications();       // what the application does at the end of a tick (apps/SimpleViewer/main.cpp:298-308)
    EXPECT_EQ(2u, compositor->render(1.0f / 60.0f));
    EXPECT_EQ(2u, compositor->get_iteration_count(right));

    std::vector<unsigned char> pixels;
    EXPECT_TRUE(compositor->read_back_buffer(pixels));
    EXPECT_EQ(size_t(64 * 36 * 4), pixels.size());
    if (pixels.size() == size_t(64 * 36 * 4)) {
        bool left_is_the_tint = true, alpha_is_opaque = true;
        for (int y = 0; y < 36; ++y)
            for (int x = 0; x < 32; ++x) {
                const unsigned char* p = pixels.data() + 4 * (x + 64 * y);
                left_is_the_tint = left_is_the_tint && std::abs(p[0] - to_sRGB8(left_tint.r)) <= 1 && std::abs(p[1] - to_sRGB8(left_tint.g)) <= 1 && std::abs(p[2] - to_sRGB8(left_tint.b)) <= 1;
                alpha_is_opaque = alpha_is_opaque && p[3] == 255;
            }
        if (!left_is_the_tint) fprintf(stderr, "left pixel (3, 5): %d %d %d, expected %d %d %d\n", pixels[4 * (3 + 64 * 5)], pixels[4 * (3 + 64 * 5) + 1], pixels[4 * (3 + 64 * 5) + 2],
                                       to_sRGB8(left_tint.r), to_sRGB8(left_tint.g), to_sRGB8(left_tint.b));
        EXPECT_TRUE(left_is_the_tint);
        EXPECT_TRUE(alpha_is_opaque);
        // Right half: the histogram exposure brings the luminance 0.44 to about 1 (within a bin), so the colour is brighter
        // than on the left, channel order kept by the filmic curve; the vignette darkens the corner.
        const unsigned char* centre = pixels.data() + 4 * (48 + 64 * 18);
        const unsigned char* corner = pixels.data() + 4 * (63 + 64 * 35);
        EXPECT_TRUE(centre[0] > to_sRGB8(left_tint.r) + 10 && centre[1] > to_sRGB8(left_tint.g) + 10);
        EXPECT_TRUE(centre[0] < centre[1] && centre[1] < centre[2]);
        EXPECT_TRUE(corner[1] + 20 < centre[1]);
    }
    delete compositor;
}

GPU_TEST_F(CompositorFixture, overlapping_cameras_composite_in_z_order_over_a_cleared_back_buffer) {
    // DX11Renderer/Compositor.cpp:268 iterates Cameras::get_z_sorted_IDs(): the camera with the larger z-index is drawn later and ends on top,
    // whatever the creation order; what no viewport covers is the cleared back buffer, not last frame's pixels.
    const Vector2i window_size(64, 36);
    HIPRenderer::HeadlessCompositor* compositor = HIPRenderer::HeadlessCompositor::initialize(0, data_directory(), window_size);
    EXPECT_TRUE(compositor != nullptr);
    if (!compositor) return;
    const Core::RendererID renderer_ID = compositor->add_renderer(HIPRenderer::HeadlessAdaptor::initialize);
    if (renderer_ID == Core::RendererID::invalid_UID()) { EXPECT_TRUE(false); delete compositor; return; }
    const RGB tint(0.2f, 0.4f, 0.8f);
    SceneRoot scene = SceneRoot("Test", tint);
    // created first but drawn last: an inset that applies the preset (brighter than the linear main view)
    const CameraID inset = create_ortho_camera(Vector2i(16, 18), scene), main_view = create_ortho_camera(Vector2i(32, 36), scene);
    Cameras::set_renderer_ID(inset, renderer_ID); Cameras::set_renderer_ID(main_view, renderer_ID);
    Cameras::set_viewport(main_view, 0.0f, 0.0f, 0.5f, 1.0f);
    Cameras::set_viewport(inset, 0.25f, 0.5f, 0.25f, 0.5f);
    Cameras::set_z_index(main_view, 0); Cameras::set_z_index(inset, 5);
    Cameras::set_effects_settings(main_view, Math::CameraEffects::Settings::linear());
    Math::CameraEffects::Settings preset = Math::CameraEffects::Settings::preset();
    preset.exposure.eye_adaptation_enabled = false; preset.vignette = 0.0f; preset.film_grain = 0.0f;
    Cameras::set_effects_settings(inset, preset);
    const std::vector<CameraID> order = Cameras::get_z_sorted_IDs();
    EXPECT_EQ(size_t(2), order.size());
    if (order.size() == 2) { EXPECT_TRUE(order[0] == main_view); EXPECT_TRUE(order[1] == inset); }

    EXPECT_EQ(2u, compositor->render(1.0f / 60.0f));
    std::vector<unsigned char> pixels;
    EXPECT_TRUE(compositor->read_back_buffer(pixels));
    if (pixels.size() == size_t(64 * 36 * 4)) {
        const unsigned char* main_pixel = pixels.data() + 4 * (4 + 64 * 4);          // main view only
        const unsigned char* inset_pixel = pixels.data() + 4 * (24 + 64 * 27);       // inside both viewports: the inset is on top
        const unsigned char* uncovered = pixels.data() + 4 * (50 + 64 * 10);         // right half: no camera
        EXPECT_TRUE(std::abs(main_pixel[1] - to_sRGB8(tint.g)) <= 1);
        EXPECT_TRUE(inset_pixel[1] > to_sRGB8(tint.g) + 10);
        EXPECT_EQ(0, int(uncovered[0])); EXPECT_EQ(0, int(uncovered[1])); EXPECT_EQ(0, int(uncovered[2]));
    }
    // swap the order: the main view now covers the inset
    Cameras::set_z_index(inset, -1);
    EXPECT_EQ(2u, compositor->render(1.0f / 60.0f));
    EXPECT_TRUE(compositor->read_back_buffer(pixels));
    if (pixels.size() == size_t(64 * 36 * 4)) EXPECT_TRUE(std::abs(pixels[4 * (24 + 64 * 27) + 1] - to_sRGB8(tint.g)) <= 1);
    delete compositor;
}
