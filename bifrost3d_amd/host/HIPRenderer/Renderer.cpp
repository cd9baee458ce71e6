// HIPRenderer/Renderer.cpp -- host side of the MI355X path tracer: scene mirroring + launch.
//
// Follows the behaviour of extensions/OptiXRenderer/OptiXRenderer/Renderer.cpp ("OR/Renderer.cpp"):
//   handle_updates           OR/Renderer.cpp:578-1205  (pull-based scene sync, accumulation reset rules)
//   prepare_camera_state     OR/Renderer.cpp:1207-1248
//   render                   OR/Renderer.cpp:1250-1265
//   request_auxiliary_buffers OR/Renderer.cpp:1267-1358
//   settings accessors       OR/Renderer.cpp:1389-1461
// Where the reference edits an OptiX scene graph incrementally, this host rebuilds the flat HiprSceneDesc
// (world-space triangles + BVH2) whenever geometry, materials or lights changed -- scenes are static between
// edits, and the rebuild is off the render path.
#include "Renderer.h"

#include "PresampledEnvironment.h"
#include "../SceneBuilder.h"
#include "../../../include/hiprenderer_c.h"
#include "../../../include/hipr_denoiser_c.h"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace HIPRenderer {

static const int MAX_RNG_SAMPLE_OFFSETS = 256;   // OR/Renderer.cpp:46

static int entry_of(Backend backend) {
    switch (backend) {
    case Backend::PathTracing: return HIPR_ENTRY_PATH_TRACING;
    case Backend::DepthVisualization: return HIPR_ENTRY_DEPTH;
    case Backend::AlbedoVisualization: return HIPR_ENTRY_ALBEDO;
    case Backend::TintVisualization: return HIPR_ENTRY_TINT;
    case Backend::RoughnessVisualization: return HIPR_ENTRY_ROUGHNESS;
    case Backend::ShadingNormalVisualization: return HIPR_ENTRY_SHADING_NORMAL;
    case Backend::PrimitiveIdVisualization: return HIPR_ENTRY_PRIMITIVE_ID;
    default: return -1;
    }
}

// ---- scene flattening: Bifrost managers -> SceneBuilder -> HiprSceneDesc --------------------------------------------
static HiprMaterial upload_material(MaterialID material_ID) {   // OR/Renderer.cpp:754-812
    HiprMaterial m = {};
    if (material_ID == MaterialID::invalid_UID()) return m;
    Assets::Material host = material_ID;
    m.flags = uint16_t(host.get_flags().raw());
    m.shading_model = uint16_t(host.get_shading_model());
    RGB tint = host.get_tint();
    m.tint[0] = tint.r; m.tint[1] = tint.g; m.tint[2] = tint.b;
    m.tint_roughness_texture_ID = host.has_tint_texture() ? int(host.get_tint_roughness_texture_ID().get_index()) : 0;
    m.roughness = host.get_roughness();
    m.roughness_texture_ID = (m.tint_roughness_texture_ID == 0 && host.has_roughness_texture()) ? int(host.get_tint_roughness_texture_ID().get_index()) : 0;
    m.specularity = host.get_specularity();
    m.metallic = host.get_metallic();
    m.metallic_texture_ID = int(host.get_metallic_texture_ID().get_index());
    m.coat = SceneBuilder::unorm16(host.get_coat());
    m.coat_roughness = SceneBuilder::unorm16(host.get_coat_roughness());
    m.coverage = host.is_cutout() ? host.get_cutout_threshold() : host.get_coverage();
    m.coverage_texture_ID = int(host.get_coverage_texture_ID().get_index());
    RGB e = host.get_emission();
    m.emission[0] = e.r; m.emission[1] = e.g; m.emission[2] = e.b;
    return m;
}

static ImageData convert_image(ImageID image_ID) {   // OR/Renderer.cpp:650-701: RGB24 is expanded to RGBA, alpha 255
    ImageData out;
    out.width = Images::get_width(image_ID);
    out.height = Images::get_height(image_ID);
    out.is_sRGB = Images::is_sRGB(image_ID);
    const size_t n = size_t(out.width) * out.height;
    const uint8_t* src = static_cast<const uint8_t*>(Images::get_pixels(image_ID));
    switch (Images::get_pixel_format(image_ID)) {
    case PixelFormat::Alpha8: case PixelFormat::Intensity8: out.format = HIPR_TEXEL_R8; out.pixels.assign(src, src + n); break;
    case PixelFormat::RGB24:
        out.format = HIPR_TEXEL_RGBA8;
        out.pixels.resize(4 * n);
        for (size_t i = 0; i < n; ++i) { out.pixels[4 * i] = src[3 * i]; out.pixels[4 * i + 1] = src[3 * i + 1]; out.pixels[4 * i + 2] = src[3 * i + 2]; out.pixels[4 * i + 3] = 255; }
        break;
    case PixelFormat::RGBA32: out.format = HIPR_TEXEL_RGBA8; out.pixels.assign(src, src + 4 * n); break;
    case PixelFormat::Intensity_Float: out.format = HIPR_TEXEL_R32F; out.pixels.assign(src, src + 4 * n); break;
    case PixelFormat::RGB_Float: {
        out.format = HIPR_TEXEL_RGBA32F;
        out.pixels.resize(16 * n);
        const float* s = reinterpret_cast<const float*>(src);
        float* d = reinterpret_cast<float*>(out.pixels.data());
        for (size_t i = 0; i < n; ++i) { d[4 * i] = s[3 * i]; d[4 * i + 1] = s[3 * i + 1]; d[4 * i + 2] = s[3 * i + 2]; d[4 * i + 3] = 1.0f; }
        break;
    }
    case PixelFormat::RGBA_Float: out.format = HIPR_TEXEL_RGBA32F; out.pixels.assign(src, src + 16 * n); break;
    default: out.format = HIPR_TEXEL_R8; out.width = out.height = 1; out.pixels.assign(1, 255); break;
    }
    return out;
}

// Dense light array in ID order (light_creation, OR/Renderer.cpp:855-902).
static std::vector<HiprLight> collect_lights() {
    std::vector<HiprLight> lights;
    for (LightSourceID light_ID : LightSources::get_iterable()) {
        Transform t = SceneNodes::get_global_transform(LightSources::get_node_ID(light_ID));
        RGB power = LightSources::get_power(light_ID);
        switch (LightSources::get_type(light_ID)) {
        case LightSources::Type::Sphere: lights.push_back(SceneBuilder::sphere_light(t.translation, power, LightSources::get_radius(light_ID))); break;
        case LightSources::Type::Spot:
            lights.push_back(SceneBuilder::spot_light(t.translation, t.rotation.forward(), power, LightSources::get_radius(light_ID), LightSources::get_cos_angle(light_ID)));
            break;
        case LightSources::Type::Directional: lights.push_back(SceneBuilder::directional_light(t.rotation.forward(), power)); break;
        }
    }
    return lights;
}

void flatten_bifrost_scene(SceneBuilder& sb) {

    // Textures and materials keep their Bifrost indices (slot 0 = invalid), like the reference's per-ID arrays.
    for (unsigned int t = 1; t < Textures::capacity(); ++t) {
        TextureID id(t);
        ImageData image;
        bool alive = Textures::get_image_ID(id) != ImageID::invalid_UID();
        if (alive) image = convert_image(Textures::get_image_ID(id));
        else { image.width = image.height = 1; image.format = HIPR_TEXEL_R8; image.pixels.assign(1, 255); }
        sb.add_texture(image, Textures::get_wrapmode_U(id) == WrapMode::Repeat, Textures::get_wrapmode_V(id) == WrapMode::Repeat,
                       Textures::get_magnification_filter(id) == MagnificationFilter::Linear, Textures::get_minification_filter(id) != MinificationFilter::None);
    }
    std::vector<bool> material_alive(Materials::capacity(), false);
    for (MaterialID id : Materials::get_iterable()) material_alive[id] = true;
    for (unsigned int m = 1; m < Materials::capacity(); ++m) sb.add_material(material_alive[m] ? upload_material(MaterialID(m)) : HiprMaterial{});

    // Meshes are added once and shared by the models that reference them (load_mesh, OR/Renderer.cpp:92-136).
    std::map<unsigned int, uint32_t> mesh_index;
    for (MeshModelID model_ID : MeshModels::get_iterable()) {
        MeshModel model = model_ID;
        Mesh mesh = model.get_mesh();
        auto it = mesh_index.find(mesh.get_ID());
        if (it == mesh_index.end()) {
            MeshData data;
            const unsigned int vertex_count = mesh.get_vertex_count(), primitive_count = mesh.get_primitive_count();
            data.positions.assign(mesh.get_positions(), mesh.get_positions() + vertex_count);
            if (mesh.get_normals()) data.normals.assign(mesh.get_normals(), mesh.get_normals() + vertex_count);
            if (mesh.get_texcoords()) data.texcoords.assign(mesh.get_texcoords(), mesh.get_texcoords() + vertex_count);
            if (mesh.get_tint_and_roughness()) {
                data.tints.resize(vertex_count);
                std::memcpy(data.tints.data(), mesh.get_tint_and_roughness(), vertex_count * 4);
            }
            if (mesh.get_emission()) data.emission.assign(mesh.get_emission(), mesh.get_emission() + vertex_count);
            data.primitives.assign(mesh.get_primitives(), mesh.get_primitives() + primitive_count);
            it = mesh_index.emplace(mesh.get_ID(), sb.add_mesh(std::move(data))).first;
        }
        // create_model: InstanceID = (MeshModel << 30) | model index, material_index = MaterialID index (OR/Renderer.cpp:138-159)
        sb.add_model(it->second, model.get_material().get_ID().get_index(), model.get_scene_node().get_global_transform(), model_ID.get_index());
    }

    for (const HiprLight& light : collect_lights()) sb.add_light(light);
    // Environment map of the scene (OR/Renderer.cpp:1136-1160): only four channel images, importance sampled through presampled lights.
    for (SceneRootID scene_ID : SceneRoots::get_iterable()) {
        const TextureID environment_map = SceneRoots::get_environment_map(scene_ID);
        if (environment_map == TextureID::invalid_UID()) continue;
        const ImageID image = Textures::get_image_ID(environment_map);
        if (channel_count(Images::get_pixel_format(image)) != 4) {
            printf("HIPRenderer only supports environments with 4 channels. '%s' has %u.\n", Images::get_name(image).c_str(), unsigned(channel_count(Images::get_pixel_format(image))));
            continue;
        }
        const InfiniteAreaLight light(environment_map);
        PresampledEnvironment presampled = presample_environment(light);
        sb.set_environment(environment_map.get_index(), presampled.pdf_width, presampled.pdf_height, std::move(presampled.per_pixel_PDF), std::move(presampled.samples));
        break;   // one scene root is rendered
    }
    sb.finalize();
}


struct Renderer::Implementation {
    int device_ID = -1;                 // the device frames are delivered on (= device_IDs[0])
    std::vector<int> device_IDs;        // every device the renderer traces on; tiles are dealt round-robin over them (hipr_group_*)
    Core::RendererID owning_renderer_ID;
    std::vector<float> tables[5];

    struct CameraState {
        HiprGroup* context = nullptr;     // per camera: one C-ABI context (accumulation buffer + queues) on every device of the renderer, as a group
        Vector2i frame_size = {0, 0};
        bool initialized = false;
        unsigned int accumulations = 0, max_accumulation_count = UINT_MAX, max_bounce_count = 4;   // OR/Renderer.cpp:211-221
        Matrix4x4f inverse_view_projection_matrix = Matrix4x4f::identity();
        Backend backend = Backend::None;
        bool scene_uploaded = false;
        bool geometry_update_pending = false;   // the uploaded scene's instances / lights moved: hipr_update_scene_geometry, not a new upload
        // Progressive batching (render()): samples [batch_used, batch_size) of the pass traced from accumulation batch_first are waiting
        // in the context's per-sample radiance buffer; they stay valid while nothing that shaped them changes.
        unsigned int batch_first = 0, batch_size = 0, batch_used = 0;
        HiprCameraState batch_camera = {};
        HiprSceneState batch_scene_state = {};
        int batch_entry = -1;
        void drop_batch() { batch_size = batch_used = 0; }
        // Backend::AIDenoisedPathTracing (OR/IBackend.cpp:19-80): the filter object and the two half4 frames it reads, on the device frames are delivered on
        HiprDenoiser* denoiser = nullptr;
        void *noisy_frame = nullptr, *albedo_frame = nullptr;
        size_t feature_frame_pixels = 0;
    };
    unsigned int max_batch_size = 64;   // accumulations traced together at most (64: 2 % more rays per second than 32 on the 1080p atrium, 128 gains nothing more: profiles/r04_ab_spp_per_pass.txt) (set_max_batch_size; 1 = the reference's one launch per accumulation)
    std::vector<CameraState> per_camera_state = std::vector<CameraState>(1);
    AIDenoiserFlags AI_denoiser_flags = AIDenoiserFlag::Default;
    PathRegularizationSettings path_regularization = {0.5f, 0.0f};   // OR/Renderer.cpp:482-483
    int arithmetic = HIPR_ARITHMETIC_FAST;      // set_arithmetic: applied to every member context of every camera's group
    HiprSceneState scene_state = {{0, 0, 0}, 3};                     // next_event_sample_count = 3, OR/Renderer.cpp:479
    std::unique_ptr<SceneBuilder> scene;

    bool is_valid() const { return device_ID >= 0; }

    ~Implementation() {
        for (CameraState& c : per_camera_state) {
            if (c.denoiser) hipr_denoiser_destroy(c.denoiser);
            if (c.context) {
                HiprContext* root = hipr_group_context(c.context, 0);
                if (c.noisy_frame) hipr_device_free(root, c.noisy_frame);
                if (c.albedo_frame) hipr_device_free(root, c.albedo_frame);
                hipr_group_destroy(c.context);
            }
        }
    }

    bool conditional_per_camera_state_resize(unsigned int camera_ID) {
        if (per_camera_state.size() <= camera_ID) { per_camera_state.resize(std::max<size_t>(Cameras::capacity(), camera_ID + 1)); return true; }
        return false;
    }

    bool load_tables(const std::filesystem::path& data_directory) {
        std::ifstream f(data_directory / "HIPRenderer" / "shading_tables.bin", std::ios::binary);
        if (!f) return false;
        char magic[8];
        uint32_t counts[5];
        f.read(magic, 8);
        f.read(reinterpret_cast<char*>(counts), sizeof(counts));
        if (!f || std::memcmp(magic, "HIPRTBL1", 8) != 0) return false;
        const uint32_t expected[5] = {1024, 1024, 8192, 8192, 1024};
        for (int i = 0; i < 5; ++i) {
            if (counts[i] != expected[i]) return false;
            tables[i].resize(counts[i]);
            f.read(reinterpret_cast<char*>(tables[i].data()), counts[i] * sizeof(float));
        }
        return bool(f);
    }

    HiprGroup* create_context() {
        HiprGroup* group = nullptr;
        if (hipr_group_create(device_IDs.data(), uint32_t(device_IDs.size()), &group) != HIPR_OK) return nullptr;
        HiprTables t = {tables[0].data(), tables[1].data(), tables[2].data(), tables[3].data(), tables[4].data()};
        if (hipr_group_upload_tables(group, &t) != HIPR_OK) { hipr_group_destroy(group); return nullptr; }
        apply_arithmetic(group);
        return group;
    }

    void apply_arithmetic(HiprGroup* group) {
        for (uint32_t m = 0; m < hipr_group_size(group); ++m) hipr_set_arithmetic(hipr_group_context(group, m), arithmetic);
    }

    void rebuild_scene() {
        scene.reset(new SceneBuilder());
        flatten_bifrost_scene(*scene);
        for (CameraState& c : per_camera_state) c.scene_uploaded = false;
    }

    bool upload_scene_to(CameraState& c) {
        if (!scene) rebuild_scene();
        if (hipr_group_upload_scene(c.context, &scene->desc()) != HIPR_OK) return false;
        c.scene_uploaded = true;
        c.geometry_update_pending = false;
        c.drop_batch();
        return true;
    }

    // ---- handle_updates -----------------------------------------------------------------------------------------------
    void handle_updates() {
        bool should_reset_accumulations = false, scene_dirty = false;

        for (CameraID cam_ID : Cameras::get_changed_cameras()) {   // OR/Renderer.cpp:581-619
            auto changes = Cameras::get_changes(cam_ID);
            if (changes.contains(Cameras::Change::Destroyed)) {
                if (cam_ID < per_camera_state.size()) {
                    if (per_camera_state[cam_ID].context) hipr_group_destroy(per_camera_state[cam_ID].context);
                    per_camera_state[cam_ID] = CameraState();
                }
                continue;
            }
            bool camera_initialized = per_camera_state.size() > cam_ID && per_camera_state[cam_ID].initialized;
            bool uses_this_renderer = owning_renderer_ID == Cameras::get_renderer_ID(cam_ID);
            bool create = uses_this_renderer && changes.is_set(Cameras::Change::Created);
            bool switch_to = uses_this_renderer && changes.is_set(Cameras::Change::Renderer);
            if (!camera_initialized && (create || switch_to)) {
                conditional_per_camera_state_resize(cam_ID);
                CameraState& state = per_camera_state[cam_ID];
                state.initialized = true;
                state.accumulations = 0;
                state.frame_size = {0, 0};
                if (state.backend == Backend::None) state.backend = Backend::PathTracing;   // preserve a backend set before handle_updates
            }
        }

        for (MeshID mesh_ID : Meshes::get_changed_meshes())
            if (Meshes::get_changes(mesh_ID).any_set(Meshes::Change::Created, Meshes::Change::Destroyed)) scene_dirty = true;
        if (!Images::get_changed_images().is_empty() || !Textures::get_changed_textures().is_empty()) scene_dirty = true;

        for (MaterialID material_ID : Materials::get_changed_materials()) {   // OR/Renderer.cpp:753-850
            auto changes = Materials::get_changes(material_ID);
            if (changes.any_set(Materials::Change::Created, Materials::Change::Updated, Materials::Change::ShadingModel)) { scene_dirty = true; should_reset_accumulations = true; }
        }
        // Lights (:852-1008): created or destroyed lights change the light count (a new scene description); updated ones are replaced in place.
        bool lights_moved = false;
        for (LightSourceID light_ID : LightSources::get_changed_lights()) {
            if (LightSources::get_changes(light_ID).any_set(LightSources::Change::Created, LightSources::Change::Destroyed)) scene_dirty = true;
            else lights_moved = true;
            should_reset_accumulations = true;
        }

        // Moved nodes (:1010-1041; only nodes that carry renderables or lights matter). The reference updates the node's transform and
        // marks the root acceleration structure dirty, which OptiX REFITS (:472); here the scene's world-space triangles of the moved
        // models are recomputed and the BVH is refitted (SceneBuilder::update_model_transforms), the rest of the scene stays as uploaded.
        std::vector<std::pair<uint32_t, Transform>> moved_models;
        for (SceneNodeID node_ID : SceneNodes::get_changed_nodes()) {
            if (!SceneNodes::get_changes(node_ID).contains(SceneNodes::Change::Transform)) continue;
            for (MeshModelID m : MeshModels::get_iterable())
                if (MeshModels::get_scene_node_ID(m) == node_ID) { moved_models.emplace_back(m.get_index(), SceneNodes::get_global_transform(node_ID)); should_reset_accumulations = true; }
            for (LightSourceID l : LightSources::get_iterable())
                if (LightSources::get_node_ID(l) == node_ID) { lights_moved = true; should_reset_accumulations = true; }
        }
        for (MeshModelID model_ID : MeshModels::get_changed_models())   // :1043-1110
            if (MeshModels::get_changes(model_ID).any_set(MeshModels::Change::Created, MeshModels::Change::Destroyed, MeshModels::Change::Material)) {
                scene_dirty = true;
                should_reset_accumulations = true;
            }

        for (SceneRootID scene_ID : SceneRoots::get_changed_scenes()) {   // :1112-1200
            auto changes = SceneRoots::get_changes(scene_ID);
            if (changes.contains(SceneRoots::Change::Destroyed)) {
                scene_state.environment_tint[0] = scene_state.environment_tint[1] = scene_state.environment_tint[2] = 0.0f;
                should_reset_accumulations = true;
                continue;
            }
            if (changes.any_set(SceneRoots::Change::EnvironmentMap, SceneRoots::Change::Created) && SceneRoots::get_environment_map(scene_ID) != TextureID::invalid_UID()) {
                scene_dirty = true;
                should_reset_accumulations = true;
            }
            if (changes.any_set(SceneRoots::Change::EnvironmentTint, SceneRoots::Change::Created)) {
                RGB tint = SceneRoots::get_environment_tint(scene_ID);
                scene_state.environment_tint[0] = tint.r; scene_state.environment_tint[1] = tint.g; scene_state.environment_tint[2] = tint.b;
                should_reset_accumulations = true;
            }
        }

        if (!scene_dirty && scene && (!moved_models.empty() || lights_moved)) {
            // transform-only tick: refit in place; a tree the motion has stretched too far is rebuilt by the scene builder itself
            if (lights_moved && !scene->replace_lights(collect_lights())) scene_dirty = true;      // the light count changed after all
            else {
                const bool topology_kept = moved_models.empty() || scene->update_model_transforms(moved_models);
                for (CameraState& c : per_camera_state) {
                    if (topology_kept) c.geometry_update_pending = c.scene_uploaded;
                    else c.scene_uploaded = false;
                    c.drop_batch();
                }
            }
        }
        if (scene_dirty) scene.reset();   // rebuilt lazily by the next render
        if (scene_dirty)
            for (CameraState& c : per_camera_state) c.scene_uploaded = false;
        if (should_reset_accumulations)
            for (CameraState& c : per_camera_state) c.accumulations = 0;
    }

    // ---- render -------------------------------------------------------------------------------------------------------
    bool prepare_camera_state(CameraID camera_ID, Vector2i frame_size, HiprCameraState& out) {   // OR/Renderer.cpp:1207-1248
        CameraState& state = per_camera_state[camera_ID];
        if (!state.context) {
            state.context = create_context();
            if (!state.context) return false;
        }
        if (!state.scene_uploaded && !upload_scene_to(state)) return false;
        if (state.geometry_update_pending) {
            if (!scene || hipr_group_update_scene_geometry(state.context, &scene->desc()) != HIPR_OK) { if (!upload_scene_to(state)) return false; }
            state.geometry_update_pending = false;
            state.drop_batch();
        }
        if (frame_size.x != state.frame_size.x || frame_size.y != state.frame_size.y) {
            if (hipr_group_set_frame(state.context, uint32_t(frame_size.x), uint32_t(frame_size.y), 1) != HIPR_OK) return false;
            state.frame_size = frame_size;
            state.accumulations = 0;
            state.drop_batch();
        }
        Matrix4x4f inverse_projection = Cameras::get_inverse_projection_matrix(camera_ID);
        Matrix4x4f inverse_view_projection = Cameras::get_inverse_view_projection_matrix(camera_ID);
        if (state.inverse_view_projection_matrix != inverse_view_projection) state.accumulations = 0;
        state.inverse_view_projection_matrix = inverse_view_projection;

        out = {};
        std::memcpy(out.inverse_projection_matrix, inverse_projection.begin(), sizeof(out.inverse_projection_matrix));
        std::memcpy(out.inverse_view_projection_matrix, inverse_view_projection.begin(), sizeof(out.inverse_view_projection_matrix));
        Matrix3x3f rotation = to_matrix3x3(Cameras::get_inverse_view_transform(camera_ID).rotation);   // holds view -> world (OR/Renderer.cpp:1238-1239)
        std::memcpy(out.view_to_world_rotation, rotation.begin(), sizeof(out.view_to_world_rotation));
        out.accumulations = state.accumulations;
        out.max_bounce_count = state.max_bounce_count;
        out.path_regularization_PDF_scale = path_regularization.PDF_scale;       // PDF_scale_at_accumulation (OR/Renderer.cpp:1244) is evaluated per path on the device
        out.path_regularization_scale_decay = path_regularization.scale_decay;
        return true;
    }

    // The reference traces ONE accumulation per render() (a blocking context->launch, OR/Renderer.cpp:1250-1265). Here the tracing is
    // batched without changing what a call returns: when the samples traced earlier are used up, the next `batch` accumulations are
    // traced in one wavefront pass (hipr_trace_pass) and every render() folds exactly one of them into the running mean and
    // writes the frame (hipr_accumulate_samples) -- bit for bit the image one launch per accumulation gives, since the per-sample
    // radiance and the f64 fold are the same. The batch grows with the accumulation count (1, 1, 1, 1, 2, 3, 4, 6, ... up to
    // max_batch_size), so a camera that keeps moving still gets one accumulation per call, and at most a third of the work done since
    // the last reset is ever thrown away by the next one.
    unsigned int next_batch_size(const CameraState& state) const {
        unsigned int batch = std::max(1u, std::min(max_batch_size, state.accumulations / 2));
        const unsigned int remaining = state.max_accumulation_count - state.accumulations;
        return std::min(batch, std::max(1u, remaining));
    }

    // AIDenoisedBackend::render (OR/IBackend.cpp:62-80) with the reference's command list spelled out: the path tracing launch, whose ray
    // generation program also accumulates the albedo feature image (ORS/SimpleRGPs.cu:149-201) -- here a second, one-segment pass of the
    // HIPR_ENTRY_DENOISER_ALBEDO entry into the context's second running mean --, the filter (the presenting list only: every frame, or the
    // power-of-two and every 32nd frame under LogarithmicFeedback), and copy_to_output with its two debug views. One accumulation per
    // call, not batched: the albedo pass between two folds would overwrite the samples a batch keeps in the path slots.
    unsigned int render_denoised(CameraState& state, const HiprCameraState& camera, void* buffer, unsigned int pitch, Vector2i frame_size) {
        HiprContext* root = hipr_group_context(state.context, 0);
        const size_t pixels = size_t(frame_size.x) * size_t(frame_size.y);
        if (!state.denoiser && hipr_denoiser_create(device_ID, &state.denoiser) != HIPR_OK) { printf("HIPRenderer: cannot create the denoiser.\n"); return state.accumulations; }
        if (pixels > state.feature_frame_pixels) {
            if (state.noisy_frame) hipr_device_free(root, state.noisy_frame);
            if (state.albedo_frame) hipr_device_free(root, state.albedo_frame);
            state.noisy_frame = state.albedo_frame = nullptr; state.feature_frame_pixels = 0;
            if (hipr_device_malloc(root, pixels * 8, &state.noisy_frame) != HIPR_OK || hipr_device_malloc(root, pixels * 8, &state.albedo_frame) != HIPR_OK) return state.accumulations;
            state.feature_frame_pixels = pixels;
        }
        state.drop_batch();
        const uint32_t width = uint32_t(frame_size.x), height = uint32_t(frame_size.y);
        hipr_group_set_scene_state(state.context, &scene_state);
        hipr_group_set_entry_point(state.context, HIPR_ENTRY_PATH_TRACING);
        if (hipr_group_set_samples_per_pass(state.context, 1) != HIPR_OK || hipr_group_trace_pass(state.context, &camera) != HIPR_OK ||
            hipr_group_accumulate_samples(state.context, 0, 1, state.accumulations, state.noisy_frame, width, 1) != HIPR_OK) return state.accumulations;
        hipr_group_set_entry_point(state.context, HIPR_ENTRY_DENOISER_ALBEDO);
        bool albedo_ok = hipr_group_use_scratch_accumulation(state.context, 2) == HIPR_OK && hipr_group_trace_pass(state.context, &camera) == HIPR_OK &&
                         hipr_group_accumulate_samples(state.context, 0, 1, state.accumulations, state.albedo_frame, width, 1) == HIPR_OK;
        hipr_group_use_scratch_accumulation(state.context, 0);
        hipr_group_set_entry_point(state.context, HIPR_ENTRY_PATH_TRACING);
        if (!albedo_ok) return state.accumulations;

        const unsigned int frame_number = state.accumulations + 1;
        const bool presenting = (frame_number & (frame_number - 1)) == 0 || frame_number % 32 == 0 || !AI_denoiser_flags.is_set(AIDenoiserFlag::LogarithmicFeedback);
        const int show = AI_denoiser_flags.is_set(AIDenoiserFlag::VisualizeNoise) ? HIPR_DENOISER_SHOW_NOISE
                       : AI_denoiser_flags.is_set(AIDenoiserFlag::VisualizeAlbedo) ? HIPR_DENOISER_SHOW_ALBEDO : HIPR_DENOISER_SHOW_FILTERED;
        HiprDenoiserSettings settings;
        hipr_denoiser_default_settings(&settings);
        if (hipr_denoiser_process(state.denoiser, &settings, state.noisy_frame, width, state.albedo_frame, width, width, height, presenting ? 1 : 0, show, buffer, pitch) != HIPR_OK ||
            hipr_denoiser_synchronize(state.denoiser) != HIPR_OK) {
            printf("HIPRenderer: denoising failed: %s\n", hipr_denoiser_last_error(state.denoiser));
            return state.accumulations;
        }
        return ++state.accumulations;
    }

    unsigned int render(CameraID camera_ID, void* buffer, unsigned int pitch, Vector2i frame_size) {   // OR/Renderer.cpp:1250-1265
        conditional_per_camera_state_resize(camera_ID);
        HiprCameraState camera;
        if (!prepare_camera_state(camera_ID, frame_size, camera)) return 0;
        CameraState& state = per_camera_state[camera_ID];
        if (state.accumulations >= state.max_accumulation_count) return state.accumulations;
        if (state.backend == Backend::AIDenoisedPathTracing) return render_denoised(state, camera, buffer, pitch, frame_size);
        int entry = entry_of(state.backend);
        entry = entry < 0 ? HIPR_ENTRY_PATH_TRACING : entry;

        // Is the next accumulation already traced, with the settings in force now? (Accumulation resets change `accumulations`;
        // bounce count, regularisation, next event sample count, environment tint and backend take effect immediately as in the reference.)
        HiprCameraState comparable = camera;
        comparable.accumulations = state.batch_camera.accumulations;
        const bool batch_valid = state.batch_used < state.batch_size && state.batch_first + state.batch_used == state.accumulations && state.batch_entry == entry &&
                                 std::memcmp(&comparable, &state.batch_camera, sizeof(HiprCameraState)) == 0 &&
                                 std::memcmp(&scene_state, &state.batch_scene_state, sizeof(HiprSceneState)) == 0;
        if (!batch_valid) {
            const unsigned int batch = next_batch_size(state);
            hipr_group_set_scene_state(state.context, &scene_state);
            hipr_group_set_entry_point(state.context, entry);
            if (hipr_group_set_samples_per_pass(state.context, batch) != HIPR_OK || hipr_group_trace_pass(state.context, &camera) != HIPR_OK) { state.drop_batch(); return state.accumulations; }
            state.batch_first = state.accumulations; state.batch_size = batch; state.batch_used = 0;
            state.batch_camera = camera; state.batch_scene_state = scene_state; state.batch_entry = entry;
        }
        if (hipr_group_accumulate_samples(state.context, state.batch_used, 1, state.accumulations, buffer, pitch, 1) != HIPR_OK) return state.accumulations;   // launch is blocking in the reference
        ++state.batch_used;
        ++state.accumulations;
        return state.accumulations;
    }
};

// ------------------------------------------------------------------------------------------------------------------------
// Renderer
// ------------------------------------------------------------------------------------------------------------------------
Renderer* Renderer::initialize(int device_ID, const std::filesystem::path& data_directory) { return initialize(std::vector<int>{device_ID}, data_directory); }

Renderer* Renderer::initialize(const std::vector<int>& device_IDs, const std::filesystem::path& data_directory) {
    if (device_IDs.empty()) return nullptr;
    Renderer* r = new Renderer(device_IDs, data_directory);
    if (r->m_impl->is_valid()) return r;
    delete r;
    return nullptr;
}

Renderer::Renderer(const std::vector<int>& device_IDs, const std::filesystem::path& data_directory)
    : m_renderer_ID(Core::Renderers::create("HIPRenderer")), m_impl(new Implementation()) {
    m_impl->owning_renderer_ID = m_renderer_ID;
    if (hipr_device_count() == 0) { fprintf(stderr, "HIPRenderer: no HIP device available.\n"); return; }   // OR/Renderer.cpp:280-281
    if (!m_impl->load_tables(data_directory)) { fprintf(stderr, "HIPRenderer failed to initialize: cannot read %s/HIPRenderer/shading_tables.bin\n", data_directory.c_str()); return; }
    m_impl->device_ID = device_IDs[0];
    m_impl->device_IDs = device_IDs;
    HiprGroup* probe = m_impl->create_context();   // fail here, like the OptiX context creation would
    if (!probe) { fprintf(stderr, "HIPRenderer failed to initialize:\n%s\n", hipr_last_error()); m_impl->device_ID = -1; return; }
    const std::string gather = hipr_group_gather_description(probe);
    hipr_group_destroy(probe);
    if (device_IDs.size() == 1) printf("HIPRenderer using HIP device %d.\n", device_IDs[0]);
    else {
        printf("HIPRenderer using %d HIP devices (", int(device_IDs.size()));
        for (size_t i = 0; i < device_IDs.size(); ++i) printf(i ? ", %d" : "%d", device_IDs[i]);
        printf("): 8x8 pixel tiles dealt round-robin, frames assembled on device %d by %s.\n", device_IDs[0], gather.c_str());
    }
}

Renderer::~Renderer() {
    Core::Renderers::destroy(m_renderer_ID);
    delete m_impl;
}

int Renderer::get_next_event_sample_count(SceneRootID) const { return m_impl->scene_state.next_event_sample_count; }
void Renderer::set_next_event_sample_count(SceneRootID, int sample_count) {
    m_impl->scene_state.next_event_sample_count = std::min(sample_count, MAX_RNG_SAMPLE_OFFSETS);   // OR/Renderer.cpp:1390-1392
}

unsigned int Renderer::get_max_bounce_count(CameraID camera_ID) const { m_impl->conditional_per_camera_state_resize(camera_ID); return m_impl->per_camera_state[camera_ID].max_bounce_count; }
void Renderer::set_max_bounce_count(CameraID camera_ID, unsigned int bounce_count) { m_impl->conditional_per_camera_state_resize(camera_ID); m_impl->per_camera_state[camera_ID].max_bounce_count = bounce_count; }

unsigned int Renderer::get_max_accumulation_count(CameraID camera_ID) const { m_impl->conditional_per_camera_state_resize(camera_ID); return m_impl->per_camera_state[camera_ID].max_accumulation_count; }
void Renderer::set_max_accumulation_count(CameraID camera_ID, unsigned int count) { m_impl->conditional_per_camera_state_resize(camera_ID); m_impl->per_camera_state[camera_ID].max_accumulation_count = count; }

Backend Renderer::get_backend(CameraID camera_ID) const { m_impl->conditional_per_camera_state_resize(camera_ID); return m_impl->per_camera_state[camera_ID].backend; }

void Renderer::set_backend(CameraID camera_ID, Backend backend) {   // OR/Renderer.cpp:1417-1455
    if (backend == Backend::None) return;
    m_impl->conditional_per_camera_state_resize(camera_ID);
    auto& state = m_impl->per_camera_state[camera_ID];
    if (backend != Backend::AIDenoisedPathTracing && entry_of(backend) < 0) {
        printf("HIPRenderer: Backend %u not supported.\n", unsigned(backend));
        backend = Backend::AlbedoVisualization;
    }
    state.backend = backend;
    state.accumulations = 0u;
}

PathRegularizationSettings Renderer::get_path_regularization_settings() const { return m_impl->path_regularization; }
void Renderer::set_path_regularization_settings(PathRegularizationSettings settings) { m_impl->path_regularization = settings; }
unsigned int Renderer::get_max_batch_size() const { return m_impl->max_batch_size; }
void Renderer::set_max_batch_size(unsigned int accumulations) { m_impl->max_batch_size = std::max(1u, std::min(accumulations, 256u)); }
Renderer::Arithmetic Renderer::get_arithmetic() const { return m_impl->arithmetic == HIPR_ARITHMETIC_EXACT ? Arithmetic::Exact : Arithmetic::Fast; }
void Renderer::set_arithmetic(Arithmetic arithmetic) {
    const int mode = arithmetic == Arithmetic::Exact ? HIPR_ARITHMETIC_EXACT : HIPR_ARITHMETIC_FAST;
    if (mode == m_impl->arithmetic) return;
    m_impl->arithmetic = mode;
    for (auto& state : m_impl->per_camera_state) {      // the samples traced ahead and the running mean belong to the other estimator
        if (state.context) m_impl->apply_arithmetic(state.context);
        state.drop_batch();
        state.accumulations = 0;
    }
}
AIDenoiserFlags Renderer::get_AI_denoiser_flags() const { return m_impl->AI_denoiser_flags; }
void Renderer::set_AI_denoiser_flags(AIDenoiserFlags flags) { m_impl->AI_denoiser_flags = flags; }

void Renderer::handle_updates() { m_impl->handle_updates(); }

unsigned int Renderer::render(CameraID camera_ID, void* half4_device_buffer, unsigned int buffer_pitch, Vector2i frame_size) {
    return m_impl->render(camera_ID, half4_device_buffer, buffer_pitch, frame_size);
}

bool Renderer::read_accumulation(std::vector<double>& out_rgba) const {
    for (auto& state : m_impl->per_camera_state)
        if (state.context && state.frame_size.x > 0) {
            out_rgba.resize(size_t(state.frame_size.x) * state.frame_size.y * 4);
            return hipr_group_read_accumulation(state.context, out_rgba.data(), out_rgba.size() / 4) == HIPR_OK;
        }
    return false;
}

static float round_through_half(float v) {
    // The reference reads AOV screenshots back from the half4 output buffer (OR/Renderer.cpp:1317-1329):
    // round to binary16 (nearest even) and back. Values here are colours in [0, 1] or small positives.
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    const uint32_t sign = bits & 0x80000000u;
    bits &= 0x7FFFFFFFu;
    if (bits >= 0x47800000u) bits = bits > 0x7F800000u ? 0x7FC00000u : 0x7F800000u;   // >= 65536 (or inf / nan) -> inf / nan
    else if (bits < 0x38800000u) {                                                       // half subnormal: quantum 2^-24
        float a; std::memcpy(&a, &bits, 4);
        a = std::nearbyintf(a * 16777216.0f) * (1.0f / 16777216.0f);
        std::memcpy(&bits, &a, 4);
    } else {
        const uint32_t lsb = (bits >> 13) & 1u;
        bits = (bits + 0x0FFFu + lsb) & ~0x1FFFu;
        if (bits >= 0x47800000u) bits = 0x7F800000u;
    }
    bits |= sign;
    float out;
    std::memcpy(&out, &bits, 4);
    return out;
}

std::vector<Screenshot> Renderer::request_auxiliary_buffers(CameraID camera_ID, Cameras::ScreenshotContent content_requested, Vector2i frame_size) {
    typedef Screenshot::Content Content;
    std::vector<Screenshot> screenshots;
    const unsigned int supported = unsigned(Content::Depth) | unsigned(Content::Albedo) | unsigned(Content::Tint) | unsigned(Content::Roughness);
    if ((content_requested.raw() & supported) == 0) return screenshots;

    m_impl->conditional_per_camera_state_resize(camera_ID);
    HiprCameraState camera;
    if (!m_impl->prepare_camera_state(camera_ID, frame_size, camera)) return screenshots;
    auto& state = m_impl->per_camera_state[camera_ID];
    const unsigned int accumulation_count = std::max(1u, state.accumulations);
    const int pixel_count = frame_size.x * frame_size.y;
    hipr_group_set_scene_state(state.context, &m_impl->scene_state);

    state.drop_batch();   // the auxiliary passes reuse the context's per-sample radiance buffer
    if (hipr_group_set_samples_per_pass(state.context, 1) != HIPR_OK) return screenshots;
    auto render_auxiliary_feature = [&](int entry, std::vector<double>& accumulation) -> bool {
        if (hipr_group_use_scratch_accumulation(state.context, 1) != HIPR_OK) return false;
        hipr_group_set_entry_point(state.context, entry);
        bool ok = true;
        for (camera.accumulations = 0; ok && camera.accumulations < accumulation_count; ++camera.accumulations)
            ok = hipr_group_trace_pass(state.context, &camera) == HIPR_OK && hipr_group_accumulate_samples(state.context, 0, 1, camera.accumulations, nullptr, 0, 1) == HIPR_OK;
        accumulation.resize(size_t(pixel_count) * 4);
        ok = ok && hipr_group_read_accumulation(state.context, accumulation.data(), pixel_count) == HIPR_OK;
        hipr_group_use_scratch_accumulation(state.context, 0);
        return ok;
    };
    auto unorm8 = [](float v) { v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); return (unsigned char)(v * 255.0f + 0.5f); };

    std::vector<double> acc;
    if (content_requested.is_set(Content::Depth) && render_auxiliary_feature(HIPR_ENTRY_DEPTH, acc)) {
        float* pixels = new float[pixel_count];
        for (int i = 0; i < pixel_count; ++i) pixels[i] = float(acc[4 * i]);
        Screenshot s; s.width = frame_size.x; s.height = frame_size.y; s.content = Content::Depth; s.format = PixelFormat::Intensity_Float; s.pixels = pixels;
        screenshots.push_back(s);
    }
    auto rgb24_screenshot = [&](Content content, int entry) {
        if (!content_requested.is_set(content) || !render_auxiliary_feature(entry, acc)) return;
        unsigned char* pixels = new unsigned char[3 * pixel_count];
        for (int i = 0; i < pixel_count; ++i)
            for (int c = 0; c < 3; ++c) pixels[3 * i + c] = unorm8(round_through_half(float(acc[4 * i + c])));
        Screenshot s; s.width = frame_size.x; s.height = frame_size.y; s.content = content; s.format = PixelFormat::RGB24; s.pixels = pixels;
        screenshots.push_back(s);
    };
    rgb24_screenshot(Content::Albedo, HIPR_ENTRY_ALBEDO);
    rgb24_screenshot(Content::Tint, HIPR_ENTRY_TINT);
    if (content_requested.is_set(Content::Roughness) && render_auxiliary_feature(HIPR_ENTRY_ROUGHNESS, acc)) {
        unsigned char* pixels = new unsigned char[pixel_count];
        for (int i = 0; i < pixel_count; ++i) pixels[i] = unorm8(round_through_half(float(acc[4 * i])));
        Screenshot s; s.width = frame_size.x; s.height = frame_size.y; s.content = Content::Roughness; s.format = PixelFormat::Intensity8; s.pixels = pixels;
        screenshots.push_back(s);
    }
    int entry = entry_of(state.backend);
    hipr_group_set_entry_point(state.context, entry < 0 ? HIPR_ENTRY_PATH_TRACING : entry);
    return screenshots;
}

} // namespace HIPRenderer
