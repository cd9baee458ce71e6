// HIPRenderer/Adaptor.h -- the compositor-facing adaptor around HIPRenderer::Renderer.
//
// Mirrors DX11OptiXAdaptor::Adaptor (extensions/DX11OptiXAdapter/DX11OptiXAdaptor/Adaptor.h:28-51), which
// implements the compositor's renderer interface (extensions/DX11Renderer/DX11Renderer/Compositor.h:50-68):
// same four virtuals, same creator signature shape, same ownership (the adaptor owns the renderer and the
// render target). Linux has no D3D11: the "back buffer" is a half4 device buffer with row 0 at the TOP
// (what the adaptor's blit produces, Adaptor.cpp:96-100) instead of an ID3D11ShaderResourceView.
#pragma once

#include "Renderer.h"

namespace HIPRenderer {

struct Recti { int x, y, width, height; };

// RenderedFrame of Compositor.h:50-54 with the SRV replaced by the device pointer + pitch of the back buffer.
struct RenderedFrame {
    const void* frame_pixels;          // half4 (R16G16B16A16_FLOAT) in device memory, row 0 = top
    unsigned int frame_pitch;          // pixels per row of the back buffer (>= frame_viewport.width)
    Recti frame_viewport;
    unsigned int iteration_count;
};

class IRenderer {   // Compositor.h:56-66
public:
    virtual ~IRenderer() {}
    virtual Bifrost::Core::RendererID get_ID() const = 0;
    virtual void handle_updates() = 0;
    virtual RenderedFrame render(Bifrost::Scene::CameraID camera_ID, Bifrost::Math::Vector2i frame_size) = 0;
    virtual std::vector<Bifrost::Scene::Screenshot> request_auxiliary_buffers(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::Cameras::ScreenshotContent content_requested,
                                                                              Bifrost::Math::Vector2i frame_size) = 0;
};

// typedef IRenderer*(*RendererCreator)(ODevice1&, const std::filesystem::path& data_directory) of Compositor.h:68,
// with the D3D device replaced by the HIP device index.
typedef IRenderer* (*RendererCreator)(int device_ID, const std::filesystem::path& data_directory);

class HeadlessAdaptor final : public IRenderer {
public:
    // Returns nullptr when the renderer cannot be created (the compositor then keeps slot 0, Compositor.cpp:178-180).
    static IRenderer* initialize(int device_ID, const std::filesystem::path& data_directory);
    ~HeadlessAdaptor();

    Renderer* get_renderer();
    const Renderer* get_renderer() const;

    Bifrost::Core::RendererID get_ID() const override;
    void handle_updates() override;
    RenderedFrame render(Bifrost::Scene::CameraID camera_ID, Bifrost::Math::Vector2i frame_size) override;
    std::vector<Bifrost::Scene::Screenshot> request_auxiliary_buffers(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::Cameras::ScreenshotContent content_requested,
                                                                      Bifrost::Math::Vector2i frame_size) override;

    // Reads the viewport of the last rendered frame back (row 0 = top, tightly packed half4 bit patterns).
    bool read_back_buffer(const RenderedFrame& frame, std::vector<unsigned short>& out_rgba16f) const;

private:
    HeadlessAdaptor(int device_ID, const std::filesystem::path& data_directory);
    HeadlessAdaptor(HeadlessAdaptor&) = delete;
    HeadlessAdaptor& operator=(HeadlessAdaptor&) = delete;

    struct Implementation;
    Implementation* m_impl;
};

} // namespace HIPRenderer
