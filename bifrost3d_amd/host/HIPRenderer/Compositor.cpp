// HIPRenderer/Compositor.cpp -- see Compositor.h.
#include "Compositor.h"

#include <cstdio>
#include <map>

using namespace Bifrost;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace HIPRenderer {

// CameraEffects::process, CameraEffects.cpp:421-449: the settings as the shaders' constants see them.
HiprCameraEffectsSettings to_c_settings(const Math::CameraEffects::Settings& s) {
    HiprCameraEffectsSettings c;
    c.exposure_mode = int(s.exposure.mode);
    c.min_log_luminance = s.exposure.min_log_luminance; c.max_log_luminance = s.exposure.max_log_luminance;
    c.min_histogram_percentage = s.exposure.min_histogram_percentage; c.max_histogram_percentage = s.exposure.max_histogram_percentage;
    c.log_luminance_bias = s.exposure.log_lumiance_bias;
    c.eye_adaptation_enabled = s.exposure.eye_adaptation_enabled ? 1 : 0;
    c.eye_adaptation_brightness = s.exposure.eye_adaptation_brightness; c.eye_adaptation_darkness = s.exposure.eye_adaptation_darkness;
    c.bloom_threshold = s.bloom.threshold; c.bloom_support = s.bloom.support;
    c.vignette = s.vignette;
    c.tonemapping_mode = int(s.tonemapping.mode);
    c.tonemapping_black_clip = s.tonemapping.settings.black_clip; c.tonemapping_toe = s.tonemapping.settings.toe; c.tonemapping_slope = s.tonemapping.settings.slope;
    c.tonemapping_shoulder = s.tonemapping.settings.shoulder; c.tonemapping_white_clip = s.tonemapping.settings.white_clip;
    c.film_grain = s.film_grain;
    return c;
}

CameraEffects::CameraEffects(int device_ID) {
    if (hipr_camera_effects_create(device_ID, &m_effects) != HIPR_OK) m_effects = nullptr;
}
CameraEffects::~CameraEffects() { hipr_camera_effects_destroy(m_effects); }

bool CameraEffects::process(const Math::CameraEffects::Settings& settings, float delta_time, const void* frame_pixels, unsigned frame_pitch, unsigned frame_rows, Recti frame_viewport,
                            void* backbuffer, unsigned backbuffer_pitch, unsigned backbuffer_rows, Recti backbuffer_viewport) {
    if (!m_effects) return false;
    const HiprCameraEffectsSettings c_settings = to_c_settings(settings);
    const HiprFrameView frame = {frame_pixels, frame_pitch, frame_rows, {frame_viewport.x, frame_viewport.y, frame_viewport.width, frame_viewport.height}};
    return hipr_camera_effects_process(m_effects, &c_settings, delta_time, &frame, backbuffer, HIPR_TARGET_RGBA8_SRGB, backbuffer_pitch, backbuffer_rows, backbuffer_viewport.x,
                                       backbuffer_viewport.y) == HIPR_OK &&
           hipr_camera_effects_synchronize(m_effects) == HIPR_OK;
}

float CameraEffects::get_linear_exposure() {
    float exposure = 0.0f;
    if (m_effects) hipr_camera_effects_get_linear_exposure(m_effects, &exposure);
    return exposure;
}
const char* CameraEffects::last_error() const { return hipr_camera_effects_last_error(m_effects); }

struct HeadlessCompositor::Implementation {
    int device_ID = 0;
    std::filesystem::path data_directory;
    Vector2i window_size = Vector2i(0, 0);
    HiprContext* context = nullptr;     // owns the back buffer allocation
    void* backbuffer = nullptr;         // RGBA8, window_size.x * window_size.y
    std::unique_ptr<CameraEffects> camera_effects;
    std::vector<std::unique_ptr<IRenderer>> renderers = std::vector<std::unique_ptr<IRenderer>>(1);      // slot 0 is the invalid renderer
    std::map<unsigned, unsigned> iteration_counts;

    ~Implementation() {
        renderers.clear();
        camera_effects.reset();
        if (context) { hipr_device_free(context, backbuffer); hipr_destroy(context); }
    }
};

HeadlessCompositor* HeadlessCompositor::initialize(int device_ID, const std::filesystem::path& data_directory, Vector2i window_size) {
    if (window_size.x <= 0 || window_size.y <= 0) return nullptr;
    HeadlessCompositor* compositor = new HeadlessCompositor();
    Implementation* impl = compositor->m_impl = new Implementation();
    impl->device_ID = device_ID; impl->data_directory = data_directory; impl->window_size = window_size;
    impl->camera_effects.reset(new CameraEffects(device_ID));
    const bool ready = impl->camera_effects->is_valid() && hipr_create(device_ID, &impl->context) == HIPR_OK &&
                       hipr_device_malloc(impl->context, uint64_t(window_size.x) * window_size.y * 4u, &impl->backbuffer) == HIPR_OK;
    if (ready) return compositor;
    delete compositor;
    return nullptr;
}

HeadlessCompositor::~HeadlessCompositor() { delete m_impl; }

Core::RendererID HeadlessCompositor::add_renderer(RendererCreator renderer_creator) {
    IRenderer* renderer = renderer_creator(m_impl->device_ID, m_impl->data_directory);
    if (renderer == nullptr) return Core::RendererID::invalid_UID();
    const unsigned index = renderer->get_ID().get_index();
    if (m_impl->renderers.size() <= index) m_impl->renderers.resize(index + 1);
    m_impl->renderers[index].reset(renderer);
    return renderer->get_ID();
}

IRenderer* HeadlessCompositor::get_renderer(Core::RendererID renderer_ID) {
    return renderer_ID.get_index() < m_impl->renderers.size() ? m_impl->renderers[renderer_ID.get_index()].get() : nullptr;
}

unsigned HeadlessCompositor::render(float delta_time) {
    for (auto& renderer : m_impl->renderers)
        if (renderer) renderer->handle_updates();

    // The back buffer starts black every frame (the swap chain's target is cleared before the cameras draw): regions no viewport covers must not
    // show last frame's pixels. Cameras composite in ascending z-index, so an overlay camera lands on top (DX11Renderer/Compositor.cpp:268).
    // The clear runs on the compositor context's stream, the camera effects write the same buffer on a stream of their own: the clear is
    // waited for here, so no viewport composited below can be overwritten by it.
    if (hipr_device_memset(m_impl->context, m_impl->backbuffer, 0, uint64_t(m_impl->window_size.x) * m_impl->window_size.y * 4) != HIPR_OK ||
        hipr_synchronize(m_impl->context) != HIPR_OK) {
        fprintf(stderr, "HIPRenderer compositor: clearing the back buffer failed: %s\n", hipr_last_error());
        return 0;
    }
    unsigned composited = 0;
    for (CameraID camera_ID : Cameras::get_z_sorted_IDs()) {
        Recti viewport;
        Cameras::get_window_viewport(camera_ID, m_impl->window_size, viewport.x, viewport.y, viewport.width, viewport.height);
        const Vector2i frame_size = Vector2i(viewport.width, viewport.height);
        if (frame_size.x <= 0 || frame_size.y <= 0) continue;      // no content, e.g. a minimised window

        IRenderer* renderer = get_renderer(Cameras::get_renderer_ID(camera_ID));
        if (!renderer) continue;
        const RenderedFrame frame = renderer->render(camera_ID, frame_size);
        if (!frame.frame_pixels) continue;
        m_impl->iteration_counts[camera_ID.get_index()] = frame.iteration_count;

        // Post process the image with the camera effects.
        const unsigned frame_rows = unsigned(frame.frame_viewport.y + frame.frame_viewport.height);
        if (!m_impl->camera_effects->process(Cameras::get_effects_settings(camera_ID), delta_time, frame.frame_pixels, frame.frame_pitch, frame_rows, frame.frame_viewport,
                                             m_impl->backbuffer, unsigned(m_impl->window_size.x), unsigned(m_impl->window_size.y), viewport)) {
            fprintf(stderr, "HIPRenderer compositor: camera effects failed: %s\n", m_impl->camera_effects->last_error());
            continue;
        }
        ++composited;
    }
    return composited;
}

Vector2i HeadlessCompositor::get_window_size() const { return m_impl->window_size; }

bool HeadlessCompositor::read_back_buffer(std::vector<unsigned char>& out) const {
    out.resize(size_t(m_impl->window_size.x) * m_impl->window_size.y * 4);
    return hipr_copy_to_host(m_impl->context, out.data(), m_impl->backbuffer, out.size()) == HIPR_OK;
}

unsigned HeadlessCompositor::get_iteration_count(CameraID camera_ID) const {
    const auto found = m_impl->iteration_counts.find(camera_ID.get_index());
    return found == m_impl->iteration_counts.end() ? 0u : found->second;
}

} // namespace HIPRenderer
