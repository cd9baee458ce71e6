// host/SceneBuilder.cpp -- see SceneBuilder.h.
#include "SceneBuilder.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace HIPRenderer {

SceneBuilder::SceneBuilder() {
    // Slot 0 holds the invalid material / "no texture", as in the reference where per-ID arrays are
    // sized capacity() and index 0 is the invalid sentinel (OptiXRenderer/Renderer.cpp:821, :528-562).
    HiprMaterial invalid = {};
    m_materials.push_back(invalid);
    HiprTexture none = {};
    m_textures.push_back(none);
    m_state.next_event_sample_count = 3;   // Renderer.cpp:479
}

uint16_t SceneBuilder::unorm16(float v) {
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return (uint16_t)(v * 65535.0f + 0.5f);
}

HiprMaterial SceneBuilder::make_material(RGB tint, float roughness, float specularity, float metallic, uint16_t flags, uint16_t shading_model) {
    HiprMaterial m = {};
    m.flags = flags;
    m.shading_model = shading_model;
    m.tint[0] = tint.r; m.tint[1] = tint.g; m.tint[2] = tint.b;
    m.roughness = roughness;
    m.specularity = specularity;
    m.metallic = metallic;
    m.coverage = 1.0f;
    return m;
}

HiprLight SceneBuilder::sphere_light(Vector3f position, RGB power, float radius) {
    HiprLight l = {};
    l.flags = HIPR_LIGHT_SPHERE;
    l.data[0] = power.r; l.data[1] = power.g; l.data[2] = power.b;
    l.data[3] = position.x; l.data[4] = position.y; l.data[5] = position.z;
    l.data[6] = radius;
    return l;
}

HiprLight SceneBuilder::spot_light(Vector3f position, Vector3f direction, RGB power, float radius, float cos_angle) {
    HiprLight l = sphere_light(position, power, radius);
    l.flags = HIPR_LIGHT_SPOT;
    l.data[7] = direction.x; l.data[8] = direction.y; l.data[9] = direction.z;
    l.data[10] = cos_angle;
    return l;
}

HiprLight SceneBuilder::directional_light(Vector3f direction, RGB radiance) {
    HiprLight l = {};
    l.flags = HIPR_LIGHT_DIRECTIONAL;
    l.data[0] = radiance.r; l.data[1] = radiance.g; l.data[2] = radiance.b;
    l.data[3] = direction.x; l.data[4] = direction.y; l.data[5] = direction.z;
    return l;
}

uint32_t SceneBuilder::add_mesh(MeshData mesh) {
    // load_mesh, OptiXRenderer/Renderer.cpp:92-136
    MeshRecord r;
    r.index_offset = uint32_t(m_indices.size() / 3);
    r.vertex_offset = uint32_t(m_geometry.size());
    r.primitive_count = uint32_t(mesh.primitives.size());
    r.vertex_count = uint32_t(mesh.positions.size());
    r.flags = 0;
    if (!mesh.normals.empty()) r.flags |= HIPR_MESH_NORMALS;
    if (!mesh.texcoords.empty()) r.flags |= HIPR_MESH_TEXCOORDS;
    if (!mesh.tints.empty()) r.flags |= HIPR_MESH_TINTS;
    if (!mesh.emission.empty()) r.flags |= HIPR_MESH_EMISSIVE;

    for (const Vector3ui& p : mesh.primitives) { m_indices.push_back(p.x); m_indices.push_back(p.y); m_indices.push_back(p.z); }
    for (uint32_t i = 0; i < r.vertex_count; ++i) {
        HiprVertexGeometry g = {};
        g.position[0] = mesh.positions[i].x; g.position[1] = mesh.positions[i].y; g.position[2] = mesh.positions[i].z;
        if (r.flags & HIPR_MESH_NORMALS) {
            OctahedralNormal e = OctahedralNormal::encode_precise(mesh.normals[i]);
            g.oct_normal[0] = e.encoding.x; g.oct_normal[1] = e.encoding.y;
        }
        m_geometry.push_back(g);
        Vector2f tc = (r.flags & HIPR_MESH_TEXCOORDS) ? mesh.texcoords[i] : Vector2f{0, 0};
        m_texcoords.push_back(tc.x); m_texcoords.push_back(tc.y);
        m_tints.push_back((r.flags & HIPR_MESH_TINTS) ? mesh.tints[i] : 0xFFFFFFFFu);
        Vector3f em = (r.flags & HIPR_MESH_EMISSIVE) ? mesh.emission[i] : Vector3f(0, 0, 0);
        m_emissions.push_back(em.x); m_emissions.push_back(em.y); m_emissions.push_back(em.z);
    }
    m_any_texcoords |= (r.flags & HIPR_MESH_TEXCOORDS) != 0;
    m_any_tints |= (r.flags & HIPR_MESH_TINTS) != 0;
    m_any_emission |= (r.flags & HIPR_MESH_EMISSIVE) != 0;
    m_meshes.push_back(r);
    return uint32_t(m_meshes.size() - 1);
}

uint32_t SceneBuilder::add_material(const HiprMaterial& material) {
    m_materials.push_back(material);
    return uint32_t(m_materials.size() - 1);
}

uint32_t SceneBuilder::add_texture(const ImageData& image, bool repeat_u, bool repeat_v, bool linear_mag, bool linear_min) {
    HiprTexture t = {};
    t.width = image.width; t.height = image.height;
    while (m_texels.size() % 16) m_texels.push_back(0);
    t.texel_offset = uint64_t(m_texels.size());
    t.format = image.format;
    t.wrap_u = repeat_u; t.wrap_v = repeat_v;
    t.filter = uint8_t((linear_mag ? 1 : 0) | (linear_min ? 2 : 0));
    t.is_sRGB = image.is_sRGB;
    m_texels.insert(m_texels.end(), image.pixels.begin(), image.pixels.end());
    m_textures.push_back(t);
    return uint32_t(m_textures.size() - 1);
}

static bool mirrors(const float* M) {
    const double det = double(M[0]) * (double(M[5]) * M[10] - double(M[6]) * M[9]) - double(M[1]) * (double(M[4]) * M[10] - double(M[6]) * M[8]) +
                       double(M[2]) * (double(M[4]) * M[9] - double(M[5]) * M[8]);
    return det < 0.0;
}

uint32_t SceneBuilder::index_offset_for(uint32_t mesh, const float* object_to_world) {
    MeshRecord& r = m_meshes[mesh];
    if (!mirrors(object_to_world)) return r.index_offset;
    if (r.mirrored_index_offset == 0xFFFFFFFFu) {
        r.mirrored_index_offset = uint32_t(m_indices.size() / 3);
        for (uint32_t p = 0; p < r.primitive_count; ++p) {
            const size_t at = 3 * size_t(r.index_offset + p);
            const uint32_t a = m_indices[at], b = m_indices[at + 1], c = m_indices[at + 2];
            m_indices.push_back(a); m_indices.push_back(c); m_indices.push_back(b);
        }
    }
    return r.mirrored_index_offset;
}

uint32_t SceneBuilder::add_model(uint32_t mesh, uint32_t material, const Transform& transform, uint32_t explicit_model_index) {
    // create_model + transformable_model, OptiXRenderer/Renderer.cpp:138-182
    const MeshRecord& r = m_meshes[mesh];
    HiprInstance inst = {};
    Matrix3x4f m = to_matrix3x4(transform);
    std::memcpy(inst.object_to_world, m.begin(), sizeof(inst.object_to_world));
    const uint32_t vertex_offset = r.vertex_offset, mesh_flags = r.flags;      // `r` may move: the mirrored copy grows m_indices only, but keep to values
    inst.index_offset = index_offset_for(mesh, inst.object_to_world);
    inst.vertex_offset = vertex_offset;
    const uint32_t model_index = explicit_model_index ? explicit_model_index : uint32_t(m_instances.size()) + 1;   // UID index, 0 is invalid
    inst.instance_id = int32_t((1u << 30) | model_index);                 // InstanceID::make(MeshModel, index)
    inst.material_index = int32_t(material);
    inst.mesh_flags = mesh_flags;
    m_instances.push_back(inst);
    m_instance_mesh.push_back(mesh);
    return model_index;
}

void SceneBuilder::add_light(const HiprLight& light) { m_lights.push_back(light); }

void SceneBuilder::set_environment(uint32_t texture_index, uint32_t pdf_width, uint32_t pdf_height, std::vector<float> per_pixel_PDF, std::vector<HiprLightSample> samples) {
    m_environment_PDF = std::move(per_pixel_PDF);
    m_environment_samples = std::move(samples);
    m_environment = {int32_t(texture_index), pdf_width, pdf_height, nullptr, nullptr, uint32_t(m_environment_samples.size())};
    m_has_environment = true;
    if (m_environment_samples.size() > 1) {   // next_event_estimation_possible (PresampledEnvironmentMap.h:64)
        HiprLight light = {};
        light.flags = HIPR_LIGHT_PRESAMPLED_ENVIRONMENT;
        m_lights.push_back(light);
    }
}

void SceneBuilder::force_shading_model(uint16_t shading_model) {
    for (size_t i = 1; i < m_materials.size(); ++i) m_materials[i].shading_model = shading_model;
}

static bool statically_opaque(const HiprMaterial& m) {
    // get_coverage (OptiXRenderer/Types.h:405-414) with no coverage texture: cutout -> (1 < threshold ? 0 : 1), else coverage.
    if (m.coverage_texture_ID) return false;
    if (m.flags & HIPR_MATERIAL_CUTOUT) return !(1.0f < m.coverage);
    return m.coverage >= 1.0f;
}

// A triangle of a material that is NOT statically opaque can still be: where its coverage texture covers it everywhere -- a finely tessellated cut-out surface
// (a fence, a lace banner) has many triangles that lie wholly on solid texels. Those get HIPR_TRIANGLE_OPAQUE too, and a shadow ray that hits one ends without
// the material -> texture -> texel lookups; get_coverage (OptiXRenderer/Types.h:405-414) would have returned 1 for every point of the triangle, so nothing changes
// for the reference's any-hit program, the oracle's or the kernels'. Decided conservatively from the texels the sampler can touch for ANY point of the triangle:
// the texture coordinates of its points lie in the bounding box of its corners' (the interpolation's rounding is covered by a texel of slack on every side), so
// every texel under that box, plus the bilinear neighbour, must pass. 8-bit linear textures only; boxes of more than 64 x 64 texels are left to the sampler.
static bool covered_everywhere(const HiprMaterial& m, const HiprTexture& t, const uint8_t* texels, const float (&uv)[3][2]) {
    if ((t.format != HIPR_TEXEL_R8 && t.format != HIPR_TEXEL_RGBA8) || t.is_sRGB || t.width == 0 || t.height == 0) return false;
    const bool cutout = (m.flags & HIPR_MATERIAL_CUTOUT) != 0;
    if (!cutout && !(m.coverage >= 1.0f)) return false;
    const bool linear = (t.filter & 1) != 0;
    const int size[2] = {int(t.width), int(t.height)};
    int first[2], last[2];
    for (int a = 0; a < 2; ++a) {
        const float lo = std::min(uv[0][a], std::min(uv[1][a], uv[2][a])) * float(size[a]), hi = std::max(uv[0][a], std::max(uv[1][a], uv[2][a])) * float(size[a]);
        if (!(std::fabs(lo) < 1048576.0f) || !(std::fabs(hi) < 1048576.0f)) return false;      // also NaN
        first[a] = int(std::floor(lo - (linear ? 0.5f : 0.0f))) - 1;
        last[a] = int(std::floor(hi - (linear ? 0.5f : 0.0f))) + (linear ? 1 : 0) + 1;
        if (last[a] - first[a] + 1 > 64) return false;
    }
    const int channels = t.format == HIPR_TEXEL_RGBA8 ? 4 : 1;
    const uint8_t* base = texels + t.texel_offset;
    auto wrap = [](int i, int n, bool repeat) { if (repeat) { i %= n; return i < 0 ? i + n : i; } return i < 0 ? 0 : (i >= n ? n - 1 : i); };
    for (int y = first[1]; y <= last[1]; ++y)
        for (int x = first[0]; x <= last[0]; ++x) {
            const uint8_t value = base[(size_t(wrap(y, size[1], t.wrap_v != 0)) * t.width + size_t(wrap(x, size[0], t.wrap_u != 0))) * size_t(channels)];      // the sampler's .x
            // cut-out: the sampled value must not fall below the threshold -- with a margin that a bilinear blend of passing texels cannot round through;
            // plain coverage: coverage * texture must be 1, i.e. every texel exactly 1
            if (cutout ? !(float(value) / 255.0f > m.coverage + 1e-5f) : value != 255) return false;
        }
    return true;
}

// backside_cull of the hit program (OptiXRenderer/Shading/MonteCarlo.cu:147-164): !hit_from_front && !thin_walled && !transmissive, thin_walled = cut-out or thin-walled.
static bool refuses_hits_from_behind(const HiprMaterial& m) {
    return !(m.flags & (HIPR_MATERIAL_CUTOUT | HIPR_MATERIAL_THIN_WALLED)) && m.shading_model != HIPR_SHADING_TRANSMISSIVE;
}

void SceneBuilder::finalize(uint32_t bvh_max_depth) {
    std::vector<HiprTriangle> world;
    m_bounds = AABB::invalid();
    for (uint32_t i = 0; i < m_instances.size(); ++i) {
        const HiprInstance& inst = m_instances[i];
        const MeshRecord& mesh = m_meshes[m_instance_mesh[i]];
        const float* M = inst.object_to_world;
        const HiprMaterial& material = m_materials[inst.material_index];
        const bool opaque = statically_opaque(material);
        const bool one_sided = refuses_hits_from_behind(m_materials[inst.material_index]);
        auto to_world = [&](uint32_t v, float* out) {
            const float* p = m_geometry[mesh.vertex_offset + v].position;
            for (int r = 0; r < 3; ++r) out[r] = M[4 * r] * p[0] + M[4 * r + 1] * p[1] + M[4 * r + 2] * p[2] + M[4 * r + 3];
            m_bounds.grow_to_contain(Vector3f(out[0], out[1], out[2]));
        };
        for (uint32_t p = 0; p < mesh.primitive_count; ++p) {
            const uint32_t* idx = &m_indices[3 * size_t(inst.index_offset + p)];      // the instance's triples: mirrored instances have their own
            HiprTriangle t = {};
            to_world(idx[0], t.v0); to_world(idx[1], t.v1); to_world(idx[2], t.v2);
            t.instance_index = i;
            t.primitive_index = p;
            bool triangle_opaque = opaque;
            if (!opaque && material.coverage_texture_ID > 0 && size_t(material.coverage_texture_ID) < m_textures.size()) {      // covered_everywhere: this triangle may still be
                float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                if (inst.mesh_flags & HIPR_MESH_TEXCOORDS)
                    for (int k = 0; k < 3; ++k) { uv[k][0] = m_texcoords[2 * size_t(mesh.vertex_offset + idx[k])]; uv[k][1] = m_texcoords[2 * size_t(mesh.vertex_offset + idx[k]) + 1]; }
                triangle_opaque = covered_everywhere(material, m_textures[size_t(material.coverage_texture_ID)], m_texels.data(), uv);
            }
            t.flags = (triangle_opaque ? HIPR_TRIANGLE_OPAQUE : 0) | (one_sided ? HIPR_TRIANGLE_ONE_SIDED : 0);
            world.push_back(t);
        }
    }

    m_bvh_max_depth_limit = bvh_max_depth;
    m_bvh = build_bvh(world, bvh_max_depth);
    m_built_bvh_area = bvh_child_area(m_bvh);
    m_triangles.resize(world.size());
    for (size_t k = 0; k < world.size(); ++k) m_triangles[k] = world[m_bvh.order[k]];

    HiprSceneDesc& d = m_desc;
    d = {};
    d.nodes = m_bvh.nodes.data(); d.node_count = uint32_t(m_bvh.nodes.size());
    d.triangles = m_triangles.data(); d.triangle_count = uint32_t(m_triangles.size());
    d.instances = m_instances.data(); d.instance_count = uint32_t(m_instances.size());
    d.indices = m_indices.data(); d.index_count = uint32_t(m_indices.size());
    d.geometry = m_geometry.data(); d.vertex_count = uint32_t(m_geometry.size());
    d.texcoords = m_any_texcoords ? m_texcoords.data() : nullptr;
    d.tints = m_any_tints ? m_tints.data() : nullptr;
    d.emissions = m_any_emission ? m_emissions.data() : nullptr;
    d.materials = m_materials.data(); d.material_count = uint32_t(m_materials.size());
    d.lights = m_lights.data(); d.light_count = uint32_t(m_lights.size());
    d.textures = m_textures.data(); d.texture_count = uint32_t(m_textures.size());
    d.texels = m_texels.data(); d.texel_bytes = uint64_t(m_texels.size());
    d.bvh_max_depth = m_bvh.max_depth;
    d.wide_nodes = m_bvh.wide_nodes.data(); d.wide_node_count = uint32_t(m_bvh.wide_nodes.size());
    d.wide_stack_entries = m_bvh.wide_stack_entries;
    d.wide8_slots = m_bvh.wide8.slots.data(); d.wide8_slot_count = uint32_t(m_bvh.wide8.slots.size()); d.wide8_height = m_bvh.wide8.height;
    for (int a = 0; a < 3; ++a) { d.wide8_grid_min[a] = m_bvh.wide8.grid_min[a]; d.wide8_grid_cell[a] = m_bvh.wide8.grid_cell[a]; }
    m_environment.per_pixel_PDF = m_environment_PDF.data();
    m_environment.samples = m_environment_samples.data();
    d.environment = m_has_environment ? &m_environment : nullptr;
}

bool SceneBuilder::update_model_transforms(const std::vector<std::pair<uint32_t, Transform>>& model_transforms, double rebuild_threshold) {
    std::vector<bool> moved(m_instances.size(), false);
    // Decided before anything is touched: does an update turn an instance inside out? Such an instance is drawn from its own index triples (two corners
    // exchanged, index_offset_for), so the triangle array is rebuilt from the instances -- with EVERY update of the batch applied first; the caller is told
    // "rebuilt" (false) and uploads a desc() that carries all the new poses.
    bool flips = false;
    for (const auto& update : model_transforms)
        for (size_t i = 0; i < m_instances.size(); ++i)
            if (uint32_t(m_instances[i].instance_id) == ((1u << 30) | update.first)) {
                Matrix3x4f m = to_matrix3x4(update.second);
                flips = flips || mirrors(m.begin()) != mirrors(m_instances[i].object_to_world);
                std::memcpy(m_instances[i].object_to_world, m.begin(), sizeof(m_instances[i].object_to_world));
                moved[i] = true;
            }
    if (flips) {
        for (size_t i = 0; i < m_instances.size(); ++i)
            if (moved[i]) m_instances[i].index_offset = index_offset_for(m_instance_mesh[i], m_instances[i].object_to_world);
        finalize(m_bvh_max_depth_limit);
        return false;
    }
    m_bounds = AABB::invalid();
    for (HiprTriangle& t : m_triangles) {
        if (moved[t.instance_index]) {
            const HiprInstance& inst = m_instances[t.instance_index];
            const MeshRecord& mesh = m_meshes[m_instance_mesh[t.instance_index]];
            const float* M = inst.object_to_world;
            const uint32_t* idx = &m_indices[3 * size_t(inst.index_offset + t.primitive_index)];
            float* corners[3] = {t.v0, t.v1, t.v2};
            for (int k = 0; k < 3; ++k) {
                const float* p = m_geometry[mesh.vertex_offset + idx[k]].position;
                for (int r = 0; r < 3; ++r) corners[k][r] = M[4 * r] * p[0] + M[4 * r + 1] * p[1] + M[4 * r + 2] * p[2] + M[4 * r + 3];
            }
        }
        m_bounds.grow_to_contain(Vector3f(t.v0[0], t.v0[1], t.v0[2]));
        m_bounds.grow_to_contain(Vector3f(t.v1[0], t.v1[1], t.v1[2]));
        m_bounds.grow_to_contain(Vector3f(t.v2[0], t.v2[1], t.v2[2]));
    }
    const double area = refit_bvh(m_bvh, m_triangles);
    for (int a = 0; a < 3; ++a) { m_desc.wide8_grid_min[a] = m_bvh.wide8.grid_min[a]; m_desc.wide8_grid_cell[a] = m_bvh.wide8.grid_cell[a]; }     // the refit moves the grid with the scene
    if (area < 0.0 || (m_built_bvh_area > 0.0 && area > rebuild_threshold * m_built_bvh_area)) {      // area < 0: a leaf record of the 8-wide tree could not be refitted
        finalize(m_bvh_max_depth_limit);      // the instances already carry the new transforms
        return false;
    }
    return true;
}

bool SceneBuilder::replace_lights(const std::vector<HiprLight>& lights) {
    const size_t environment_lights = m_has_environment && m_environment_samples.size() > 1 ? 1 : 0;
    if (lights.size() + environment_lights != m_lights.size()) return false;
    std::copy(lights.begin(), lights.end(), m_lights.begin());      // the presampled environment light, if any, stays last
    return true;
}

HiprCameraState make_camera_state(const CameraDescription& camera, float aspect_ratio, uint32_t accumulations, float path_regularization_PDF_scale) {
    Matrix4x4f inverse_projection = {};
    if (camera.orthographic) {
        // compute_orthographic_projection, Bifrost/Scene/Camera.cpp:268-286
        inverse_projection[0][0] = 0.5f * camera.ortho_width;
        inverse_projection[1][1] = 0.5f * camera.ortho_height;
        inverse_projection[2][2] = 0.5f * camera.ortho_depth;
        inverse_projection[2][3] = 0.5f * camera.ortho_depth;
        inverse_projection[3][3] = 1.0f;
    } else {
        // compute_perspective_projection, Bifrost/Scene/Camera.cpp:237-266
        const float n = camera.near_plane, fa = camera.far_plane;
        const float f = 1.0f / std::tan(camera.field_of_view * 0.5f);
        const float a = (fa + n) / (n - fa);
        const float b = (2.0f * fa * n) / (n - fa);
        Matrix4x4f projection = {};
        projection[0][0] = f / aspect_ratio;
        projection[1][1] = f;
        projection[2][2] = -a;
        projection[2][3] = b;
        projection[3][2] = 1.0f;
        inverse_projection[0][0] = 1.0f / projection[0][0];
        inverse_projection[1][1] = 1.0f / projection[1][1];
        inverse_projection[2][3] = 1.0f;
        inverse_projection[3][2] = 1.0f / projection[2][3];
        inverse_projection[3][3] = -projection[2][2] / projection[2][3];
    }

    // Cameras::get_inverse_view_projection_matrix = to_matrix4x4(inverse view transform) * inverse projection (Camera.h:112-114)
    Matrix4x4f inverse_view_projection = to_matrix4x4(camera.transform) * inverse_projection;
    Matrix3x3f rotation = to_matrix3x3(camera.transform.rotation);   // Renderer.cpp:1238-1239

    HiprCameraState s = {};
    std::memcpy(s.view_to_world_rotation, rotation.begin(), sizeof(s.view_to_world_rotation));
    std::memcpy(s.inverse_projection_matrix, inverse_projection.begin(), sizeof(s.inverse_projection_matrix));
    std::memcpy(s.inverse_view_projection_matrix, inverse_view_projection.begin(), sizeof(s.inverse_view_projection_matrix));
    s.accumulations = accumulations;
    s.max_bounce_count = camera.max_bounce_count;
    s.path_regularization_PDF_scale = path_regularization_PDF_scale;
    return s;
}

} // namespace HIPRenderer
