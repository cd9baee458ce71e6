// HIPRenderer/Renderer.h -- the renderer plugin class, a drop-in for OptiXRenderer::Renderer.
//
// Same public method set, argument meaning and error behaviour as the reference class
// (extensions/OptiXRenderer/OptiXRenderer/Renderer.h:40-86, PublicTypes.h:20-58). Two signature changes,
// both forced by the removal of OptiX:
//   * render() takes a device pointer to half4 pixels plus an explicit row pitch instead of an optix::Buffer
//     (the adaptor's buffer may be wider than the frame, Adaptor.cpp:141-153);
//   * get_context() is gone.
// The class owns no GPU code: it flattens the Bifrost scene (SceneBuilder) and drives the C-ABI of
// include/hiprenderer_c.h.
#pragma once

#include "../Bifrost.h"

#include <filesystem>
#include <vector>

namespace HIPRenderer {

enum class Backend {   // PublicTypes.h:20-30
    None,
    PathTracing,
    AIDenoisedPathTracing,
    DepthVisualization,
    AlbedoVisualization,
    TintVisualization,
    RoughnessVisualization,
    ShadingNormalVisualization,
    PrimitiveIdVisualization,
};

struct PathRegularizationSettings {   // PublicTypes.h:40-45
    float PDF_scale;
    float scale_decay;
    float PDF_scale_at_accumulation(int accumulation) { return PDF_scale * (1.0f + scale_decay * accumulation); }
};

enum class AIDenoiserFlag : unsigned char { None = 0, LogarithmicFeedback = 1 << 0, VisualizeNoise = 1 << 1, VisualizeAlbedo = 1 << 2, Default = LogarithmicFeedback };
typedef Bifrost::Core::Bitmask<AIDenoiserFlag> AIDenoiserFlags;

class Renderer final {
public:
    // Returns nullptr when no usable device exists or initialisation fails (OR/Renderer.cpp:1365-1378).
    // data_directory must contain HIPRenderer/shading_tables.bin (the reference loads its PTX from <data>/OptiXRenderer/ptx).
    static Renderer* initialize(int device_ID, const std::filesystem::path& data_directory);
    // Not in the reference (single device, OR/Renderer.cpp:289-291): the same renderer tracing on several GPUs of the node. Pixel tiles of
    // 8 x 8 are dealt round-robin over the devices, scene and tables are replicated, every device keeps the accumulation of its tiles;
    // each render() gathers the finished half4 tiles on device_IDs[0] (RCCL over xGMI) and assembles the frame there, so the buffer handed
    // to render() lives on device_IDs[0]. Frames are bit-identical to the single-device ones.
    static Renderer* initialize(const std::vector<int>& device_IDs, const std::filesystem::path& data_directory);
    ~Renderer();

    Bifrost::Core::RendererID get_renderer_ID() const { return m_renderer_ID; }

    Backend get_backend(Bifrost::Scene::CameraID camera_ID) const;
    void set_backend(Bifrost::Scene::CameraID camera_ID, Backend backend);

    unsigned int get_max_bounce_count(Bifrost::Scene::CameraID camera_ID) const;
    void set_max_bounce_count(Bifrost::Scene::CameraID camera_ID, unsigned int bounce_count);

    unsigned int get_max_accumulation_count(Bifrost::Scene::CameraID camera_ID) const;
    void set_max_accumulation_count(Bifrost::Scene::CameraID camera_ID, unsigned int accumulation_count);

    int get_next_event_sample_count(Bifrost::Scene::SceneRootID scene_root_ID) const;
    void set_next_event_sample_count(Bifrost::Scene::SceneRootID scene_root_ID, int sample_count);

    PathRegularizationSettings get_path_regularization_settings() const;
    void set_path_regularization_settings(PathRegularizationSettings settings);

    // Not in the reference: how many accumulations render() may trace together (default 32; 1 = one launch per accumulation like the
    // reference). Every render() still returns exactly one more accumulation and the same pixels for any setting; see Renderer.cpp.
    unsigned int get_max_batch_size() const;
    void set_max_batch_size(unsigned int accumulations);

    // Not in the reference (a build flag there: --use_fast_math, extensions/OptiXRenderer/CMakeLists.txt:82-83): the arithmetic of the shade stage. Fast (default):
    // hardware-approximate division / sqrt / sin / cos / pow like the reference's PTX. Exact: IEEE operations and specified transcendentals -- every frame equals
    // the CPU restatement bit for bit (include/hiprenderer_c.h hipr_set_arithmetic). Changing it restarts every camera's accumulation.
    enum class Arithmetic { Fast = 0, Exact = 1 };
    Arithmetic get_arithmetic() const;
    void set_arithmetic(Arithmetic arithmetic);

    AIDenoiserFlags get_AI_denoiser_flags() const;
    void set_AI_denoiser_flags(AIDenoiserFlags flags);

    // Pulls the change sets of the Bifrost managers; must run after the mutating callbacks of a tick and before the
    // application resets the change notifications (apps/SimpleViewer/main.cpp:298-308).
    void handle_updates();

    // One accumulation more in the frame (traced ahead in batches, see set_max_batch_size). `half4_device_buffer`: R16G16B16A16_FLOAT pixels in device memory, row 0 = bottom,
    // `buffer_pitch` pixels per row (>= frame_size.x). Returns the iteration count like the reference.
    unsigned int render(Bifrost::Scene::CameraID camera_ID, void* half4_device_buffer, unsigned int buffer_pitch, Bifrost::Math::Vector2i frame_size);

    std::vector<Bifrost::Scene::Screenshot> request_auxiliary_buffers(Bifrost::Scene::CameraID camera_ID, Bifrost::Scene::Cameras::ScreenshotContent content_requested,
                                                                      Bifrost::Math::Vector2i frame_size);

    // Test / tooling access: the accumulation buffer as RGBA f64 (row-major, row 0 = bottom).
    bool read_accumulation(std::vector<double>& out_rgba) const;

private:
    Renderer(const std::vector<int>& device_IDs, const std::filesystem::path& data_directory);
    Renderer(Renderer&) = delete;
    Renderer& operator=(Renderer&) = delete;

    Bifrost::Core::RendererID m_renderer_ID;
    struct Implementation;
    Implementation* m_impl;
};

class SceneBuilder;
// Flattens the current state of the Bifrost managers (meshes, models, materials, textures, lights) into `scene` and
// finalizes it (world-space triangles + BVH2). What handle_updates() triggers whenever geometry, materials or lights changed.
void flatten_bifrost_scene(SceneBuilder& scene);

} // namespace HIPRenderer
