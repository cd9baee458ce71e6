// HIPRenderer/Adaptor.cpp -- see Adaptor.h. Behaviour follows DX11OptiXAdaptor/Adaptor.cpp:
//   render                 :141-225  (grow the render target to the largest size seen, render, flip-blit, viewport)
//   resize_render_target   :227-247  (capacity only grows)
#include "Adaptor.h"

#include "../../../include/hiprenderer_c.h"

#include <algorithm>
#include <cstdio>

using namespace Bifrost;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace HIPRenderer {

struct HeadlessAdaptor::Implementation {
    Renderer* renderer = nullptr;
    HiprContext* context = nullptr;   // owns the render target and back buffer allocations and the blit stream

    struct { void* pixels = nullptr; int width = 0, height = 0, capacity = 0; } render_target;
    struct { void* pixels = nullptr; int width = 0, height = 0; } backbuffer;

    ~Implementation() {
        if (context) {
            hipr_device_free(context, render_target.pixels);
            hipr_device_free(context, backbuffer.pixels);
            hipr_destroy(context);
        }
        delete renderer;
    }

    bool resize_render_target(int width, int height) {
        if (render_target.capacity < width * height) {
            hipr_device_free(context, render_target.pixels);
            render_target.pixels = nullptr;
            render_target.capacity = width * height;
            if (hipr_device_malloc(context, uint64_t(render_target.capacity) * 8u, &render_target.pixels) != HIPR_OK) { render_target.capacity = 0; return false; }
        }
        render_target.width = width;
        render_target.height = height;
        return true;
    }

    RenderedFrame render(CameraID camera_ID, Vector2i frame_size) {
        RenderedFrame failed = {nullptr, 0, {0, 0, 0, 0}, 0};
        if (frame_size.x <= 0 || frame_size.y <= 0) return failed;
        if (render_target.width < frame_size.x || render_target.height < frame_size.y) {
            const int buffer_width = std::max(render_target.width, frame_size.x), buffer_height = std::max(render_target.height, frame_size.y);
            hipr_device_free(context, backbuffer.pixels);
            backbuffer.pixels = nullptr;
            if (hipr_device_malloc(context, uint64_t(buffer_width) * buffer_height * 8u, &backbuffer.pixels) != HIPR_OK) return failed;
            backbuffer.width = buffer_width;
            backbuffer.height = buffer_height;
            if (!resize_render_target(buffer_width, buffer_height)) return failed;
        }
        const unsigned int iteration_count = renderer->render(camera_ID, render_target.pixels, render_target.width, frame_size);
        // Renderer::render is blocking, so the blit on this context's stream sees the finished pixels.
        if (hipr_present_flipped(context, render_target.pixels, render_target.width, frame_size.x, frame_size.y, backbuffer.pixels, backbuffer.width) != HIPR_OK ||
            hipr_synchronize(context) != HIPR_OK) {
            fprintf(stderr, "HIPRenderer adaptor: %s\n", hipr_last_error());
            return failed;
        }
        return {backbuffer.pixels, (unsigned int)backbuffer.width, {0, 0, frame_size.x, frame_size.y}, iteration_count};
    }
};

IRenderer* HeadlessAdaptor::initialize(int device_ID, const std::filesystem::path& data_directory) {
    HeadlessAdaptor* adaptor = new HeadlessAdaptor(device_ID, data_directory);
    if (adaptor->m_impl->renderer && adaptor->m_impl->context) return adaptor;
    delete adaptor;
    return nullptr;
}

HeadlessAdaptor::HeadlessAdaptor(int device_ID, const std::filesystem::path& data_directory) : m_impl(new Implementation()) {
    m_impl->renderer = Renderer::initialize(device_ID, data_directory);
    if (m_impl->renderer && hipr_create(device_ID, &m_impl->context) != HIPR_OK) m_impl->context = nullptr;
}

HeadlessAdaptor::~HeadlessAdaptor() { delete m_impl; }

Renderer* HeadlessAdaptor::get_renderer() { return m_impl->renderer; }
const Renderer* HeadlessAdaptor::get_renderer() const { return m_impl->renderer; }
Core::RendererID HeadlessAdaptor::get_ID() const { return m_impl->renderer->get_renderer_ID(); }
void HeadlessAdaptor::handle_updates() { m_impl->renderer->handle_updates(); }
RenderedFrame HeadlessAdaptor::render(CameraID camera_ID, Vector2i frame_size) { return m_impl->render(camera_ID, frame_size); }

std::vector<Screenshot> HeadlessAdaptor::request_auxiliary_buffers(CameraID camera_ID, Cameras::ScreenshotContent content_requested, Vector2i frame_size) {
    return m_impl->renderer->request_auxiliary_buffers(camera_ID, content_requested, frame_size);
}

bool HeadlessAdaptor::read_back_buffer(const RenderedFrame& frame, std::vector<unsigned short>& out) const {
    if (!frame.frame_pixels) return false;
    const int w = frame.frame_viewport.width, h = frame.frame_viewport.height;
    std::vector<unsigned short> whole(size_t(frame.frame_pitch) * h * 4);
    if (hipr_copy_to_host(m_impl->context, whole.data(), frame.frame_pixels, whole.size() * 2) != HIPR_OK) return false;
    out.resize(size_t(w) * h * 4);
    for (int y = 0; y < h; ++y) std::copy_n(whole.begin() + size_t(y) * frame.frame_pitch * 4, size_t(w) * 4, out.begin() + size_t(y) * w * 4);
    return true;
}

} // namespace HIPRenderer
