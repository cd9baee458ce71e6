// host/SceneBuilder.h -- flattens meshes / models / materials / lights into the HiprSceneDesc the
// kernels consume. This is the host work OptiXRenderer::Renderer::handle_updates() does against the
// OptiX scene graph (OptiXRenderer/Renderer.cpp:92-182 load_mesh / create_model / transformable_model,
// :754-812 upload_material, :855-902 light_creation), re-targeted at flat SoA arrays + a BVH2.
#pragma once

#include "BvhBuilder.h"
#include "Math.h"

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace HIPRenderer {

using namespace Bifrost::Math;

struct MeshData {
    std::string name;
    std::vector<Vector3f> positions;
    std::vector<Vector3f> normals;      // empty or one per vertex
    std::vector<Vector2f> texcoords;    // empty or one per vertex
    std::vector<uint32_t> tints;        // uchar4 (tint rgb, roughness), empty or one per vertex
    std::vector<Vector3f> emission;     // empty or one per vertex
    std::vector<Vector3ui> primitives;
};

struct ImageData {
    uint32_t width = 0, height = 0;
    uint8_t format = HIPR_TEXEL_RGBA8;  // HIPR_TEXEL_*
    bool is_sRGB = false;
    std::vector<uint8_t> pixels;
};

struct CameraDescription {
    Transform transform = Transform::identity();
    float field_of_view = 3.14159265358979323846f / 4.0f;   // SimpleViewer default, CameraHandlers.cpp:111
    float near_plane = 0.1f, far_plane = 100.0f;
    uint32_t max_bounce_count = 4;
    bool orthographic = false;                               // compute_orthographic_projection, Camera.cpp:268-286
    float ortho_width = 1.0f, ortho_height = 1.0f, ortho_depth = 1000.0f;
};

class SceneBuilder {
public:
    SceneBuilder();
    static constexpr bool takes_textures = true;        // Scenes::build_atrium's textured variant

    uint32_t add_mesh(MeshData mesh);                                   // returns mesh index
    uint32_t add_material(const HiprMaterial& material);                // returns material index (0 is the invalid material)
    uint32_t add_texture(const ImageData& image, bool repeat_u, bool repeat_v, bool linear_magnification, bool linear_minification);
    // model_index: the MeshModel UID index used for InstanceID; 0 = next sequential index. Returns the index used.
    uint32_t add_model(uint32_t mesh, uint32_t material, const Transform& transform, uint32_t model_index = 0);
    void add_light(const HiprLight& light);
    // The lat-long image is texture `texture_index` (add_texture); PDF texels and samples as PresampledEnvironment.h builds them. With two or
    // more samples the environment also joins the light list for next event estimation (OptiXRenderer/Renderer.cpp:1160-1196).
    void set_environment(uint32_t texture_index, uint32_t pdf_width, uint32_t pdf_height, std::vector<float> per_pixel_PDF, std::vector<HiprLightSample> samples);

    static HiprLight sphere_light(Vector3f position, RGB power, float radius);
    static HiprLight spot_light(Vector3f position, Vector3f direction, RGB power, float radius, float cos_angle);
    static HiprLight directional_light(Vector3f direction, RGB radiance);
    static HiprMaterial make_material(RGB tint, float roughness, float specularity, float metallic, uint16_t flags = 0, uint16_t shading_model = HIPR_SHADING_DEFAULT);
    static uint16_t unorm16(float v);

    void set_environment_tint(RGB tint) { m_state.environment_tint[0] = tint.r; m_state.environment_tint[1] = tint.g; m_state.environment_tint[2] = tint.b; }
    void force_shading_model(uint16_t shading_model);   // e.g. the diffuse-only configuration of BASELINE.json

    // Transforms every model to world space, builds the BVH and fills desc().
    void finalize(uint32_t bvh_max_depth = 62);

    // Transform-only update after finalize(): the models named by their model index (add_model's return value) get new object-to-world
    // transforms; their world-space triangles are recomputed in place (BVH leaf order kept) and the BVH is REFIT, not rebuilt -- what the
    // reference does when a node moves (root acceleration refit, OR/Renderer.cpp:472,1010-1041). When the refit tree's box area has grown
    // past `rebuild_threshold` times the built tree's (a badly stretched tree traverses slowly) the scene is rebuilt instead. Returns true when
    // the topology was kept (nodes, triangle order and counts unchanged: hipr_update_scene_geometry suffices), false after a rebuild.
    bool update_model_transforms(const std::vector<std::pair<uint32_t, Transform>>& model_transforms, double rebuild_threshold = 1.5);
    // The lights of a finalized scene replaced one for one (moved or re-coloured lights; the count must match).
    bool replace_lights(const std::vector<HiprLight>& lights);

    const HiprSceneDesc& desc() const { return m_desc; }
    const HiprSceneState& state() const { return m_state; }
    HiprSceneState& state() { return m_state; }
    CameraDescription camera;
    AABB bounds() const { return m_bounds; }
    size_t mesh_count() const { return m_meshes.size(); }
    size_t model_count() const { return m_instances.size(); }

private:
    struct MeshRecord { uint32_t index_offset, vertex_offset, primitive_count, vertex_count, flags; uint32_t mirrored_index_offset = 0xFFFFFFFFu; };
    // The index triples an instance of `mesh` under `object_to_world` uses: the mesh's own, or -- a transform that mirrors (negative determinant) -- a copy of
    // them with the second and third corner exchanged, so that the world-space corners wind the way the reference's transformed geometric normal points
    // (rtTransformNormal through the inverse transpose, ORS/MonteCarlo.cu:147) and every consumer of the triple order sees the same triangle.
    uint32_t index_offset_for(uint32_t mesh, const float* object_to_world);
    std::vector<MeshRecord> m_meshes;
    std::vector<uint32_t> m_indices;
    std::vector<HiprVertexGeometry> m_geometry;
    std::vector<float> m_texcoords, m_emissions;
    std::vector<uint32_t> m_tints;
    bool m_any_texcoords = false, m_any_tints = false, m_any_emission = false;
    std::vector<HiprMaterial> m_materials;
    std::vector<HiprLight> m_lights;
    std::vector<HiprTexture> m_textures;
    std::vector<uint8_t> m_texels;
    std::vector<HiprInstance> m_instances;
    std::vector<uint32_t> m_instance_mesh;
    std::vector<HiprTriangle> m_triangles;
    BvhBuildResult m_bvh;
    double m_built_bvh_area = 0.0;
    uint32_t m_bvh_max_depth_limit = 62;
    std::vector<float> m_environment_PDF;
    std::vector<HiprLightSample> m_environment_samples;
    HiprEnvironment m_environment = {};
    bool m_has_environment = false;
    HiprSceneDesc m_desc = {};
    HiprSceneState m_state = {};
    AABB m_bounds = AABB::invalid();
};

// Fills the matrices of HiprCameraState the way prepare_camera_state does (OptiXRenderer/Renderer.cpp:1207-1248)
// from a perspective camera (Bifrost/Scene/Camera.cpp:237-266 compute_perspective_projection).
HiprCameraState make_camera_state(const CameraDescription& camera, float aspect_ratio, uint32_t accumulations, float path_regularization_PDF_scale);

} // namespace HIPRenderer
