// HIPRenderer/Compositor.h -- the headless half of the reference's compositor: for every camera, render through the camera's
// renderer, then post-process the frame with the camera's effects settings into the window's back buffer.
//
// Mirrors the render loop of DX11Renderer::Compositor (extensions/DX11Renderer/DX11Renderer/Compositor.cpp:255-325) and its
// CameraEffects member (CameraEffects.h:196-239). There is no window or swap chain here: the back buffer is an RGBA8 sRGB
// device buffer of the window's size (row 0 = top) that a presenter would blit and that read_back_buffer() downloads; the
// time between frames is handed in instead of read from a clock. GUI renderers and v-sync stay out (out of scope).
#pragma once

#include "Adaptor.h"

#include "../../../include/hipr_camera_effects_c.h"

#include <memory>
#include <vector>

namespace HIPRenderer {

HiprCameraEffectsSettings to_c_settings(const Bifrost::Math::CameraEffects::Settings& settings);

// RAII owner of a HiprCameraEffects object; process() is DX11Renderer::CameraEffects::process with device pointers for the views.
class CameraEffects {
public:
    explicit CameraEffects(int device_ID);
    ~CameraEffects();
    CameraEffects(const CameraEffects&) = delete;
    CameraEffects& operator=(const CameraEffects&) = delete;

    bool is_valid() const { return m_effects != nullptr; }
    HiprCameraEffects* get() { return m_effects; }

    // Blocking. `backbuffer`: RGBA8 sRGB device buffer, `backbuffer_pitch` pixels per row. False on failure (see last_error()).
    bool process(const Bifrost::Math::CameraEffects::Settings& settings, float delta_time, const void* frame_pixels, unsigned frame_pitch, unsigned frame_rows, Recti frame_viewport,
                 void* backbuffer, unsigned backbuffer_pitch, unsigned backbuffer_rows, Recti backbuffer_viewport);
    float get_linear_exposure();
    const char* last_error() const;

private:
    HiprCameraEffects* m_effects = nullptr;
};

class HeadlessCompositor {
public:
    // Returns nullptr when the device has no camera effects (no GPU). The window size is the back buffer size.
    static HeadlessCompositor* initialize(int device_ID, const std::filesystem::path& data_directory, Bifrost::Math::Vector2i window_size);
    ~HeadlessCompositor();

    // Compositor::add_renderer (Compositor.cpp:170-186): the creator is handed the device; invalid_UID when it fails.
    Bifrost::Core::RendererID add_renderer(RendererCreator renderer_creator);
    IRenderer* get_renderer(Bifrost::Core::RendererID renderer_ID);

    // One pass of Compositor::render: handle_updates on all renderers, then every camera. `delta_time` drives eye adaptation and the film grain.
    // Returns the number of cameras composited.
    unsigned render(float delta_time);

    Bifrost::Math::Vector2i get_window_size() const;
    // The whole window as tightly packed RGBA8 (sRGB encoded), row 0 = top.
    bool read_back_buffer(std::vector<unsigned char>& out_rgba8) const;
    // The iteration count the camera's renderer reported for the last frame (0 when it has not been rendered).
    unsigned get_iteration_count(Bifrost::Scene::CameraID camera_ID) const;

private:
    HeadlessCompositor() = default;
    struct Implementation;
    Implementation* m_impl = nullptr;
};

} // namespace HIPRenderer
