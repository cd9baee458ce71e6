// host/Scenes.cpp -- the built-in scenes of the headless driver.
//
//   create_cornell_box : apps/SimpleViewer/Scenes/CornellBox.h:23-122 (geometry, materials, light, camera),
//                        meshes from core/Bifrost/Bifrost/Assets/MeshCreation.cpp:30-156 (plane, box),
//                        camera / near / far rules of apps/SimpleViewer/main.cpp:395-429.
//   create_atrium      : deterministic procedural stand-in for the Sponza-class glTF configs the
//                        reference does not ship: colonnades, arches and
//                        draped cloth with 25 metal-rough materials, loader defaults of main.cpp:380-426.
//   create_quad_scene  : the ortho tint quad of tests/OptiXRendererTests/RendererTest.h:88-117.
#include "Scenes.h"
#include "AtriumScene.h"

#include <cmath>

namespace HIPRenderer {
namespace Scenes {

// ---------------------------------------------------------------------------------------------
// MeshCreation::plane / box (vertex order, winding and normals as the reference)
// ---------------------------------------------------------------------------------------------
MeshData plane(unsigned quads_per_edge, bool normals, bool texcoords) {
    MeshData m;
    m.name = "Plane";
    const unsigned size = quads_per_edge + 1;
    const float tc_normalizer = 1.0f / quads_per_edge;
    for (unsigned z = 0; z < size; ++z)
        for (unsigned x = 0; x < size; ++x) {
            Vector2f tc = {float(x) * tc_normalizer, float(z) * tc_normalizer};
            m.positions.push_back(Vector3f(tc.x - 0.5f, 0.0f, tc.y - 0.5f));
            if (normals) m.normals.push_back(Vector3f(0, 1, 0));
            if (texcoords) m.texcoords.push_back(tc);
        }
    for (unsigned z = 0; z < quads_per_edge; ++z)
        for (unsigned x = 0; x < quads_per_edge; ++x) {
            unsigned base = x + z * size;
            m.primitives.push_back({base, base + size, base + 1});
            m.primitives.push_back({base + 1, base + size, base + size + 1});
        }
    return m;
}

MeshData box(unsigned quads_per_edge, Vector3f size, bool tints, bool texcoords) {
    MeshData m;
    m.name = "Box";
    const unsigned verts_per_edge = quads_per_edge + 1;
    const Vector3f quad = size / float(quads_per_edge);
    const Vector3f half = 0.5f * size;
    const unsigned verts_per_side = verts_per_edge * verts_per_edge;
    auto side = [&](auto position_of) {
        for (unsigned i = 0; i < verts_per_edge; ++i)
            for (unsigned j = 0; j < verts_per_edge; ++j) m.positions.push_back(position_of(float(i), float(j)));
    };
    side([&](float i, float j) { return Vector3f(half.x - i * quad.x, half.y, j * quad.z - half.z); });    // top
    side([&](float i, float j) { return Vector3f(half.x - i * quad.x, -half.y, half.z - j * quad.z); });   // bottom
    side([&](float i, float j) { return Vector3f(-half.x, half.y - i * quad.y, j * quad.z - half.z); });   // left
    side([&](float i, float j) { return Vector3f(half.x, i * quad.y - half.y, j * quad.z - half.z); });    // right
    side([&](float i, float j) { return Vector3f(half.x - i * quad.x, half.y - j * quad.y, half.z); });    // front
    side([&](float i, float j) { return Vector3f(i * quad.x - half.x, half.y - j * quad.y, -half.z); });   // back
    const Vector3f side_normals[6] = {{0, 1, 0}, {0, -1, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 0, 1}, {0, 0, -1}};
    for (int s = 0; s < 6; ++s)
        for (unsigned v = 0; v < verts_per_side; ++v) m.normals.push_back(side_normals[s]);
    if (tints) m.tints.assign(m.positions.size(), 0xFFFFFFFFu);   // default_initialize_shading: UNorm8::one()
    if (texcoords)   // every side carries the same (i, j) / quads_per_edge grid (MeshCreation.cpp:126-137)
        for (int s = 0; s < 6; ++s)
            for (unsigned i = 0; i < verts_per_edge; ++i)
                for (unsigned j = 0; j < verts_per_edge; ++j) m.texcoords.push_back(Vector2f{float(i) * (1.0f / quads_per_edge), float(j) * (1.0f / quads_per_edge)});
    for (unsigned side_offset = 0; side_offset < 6 * verts_per_side; side_offset += verts_per_side)
        for (unsigned i = 0; i < quads_per_edge; ++i)
            for (unsigned j = 0; j < quads_per_edge; ++j) {
                m.primitives.push_back({j + i * verts_per_edge + side_offset, j + (i + 1) * verts_per_edge + side_offset, j + 1 + i * verts_per_edge + side_offset});
                m.primitives.push_back({j + 1 + i * verts_per_edge + side_offset, j + (i + 1) * verts_per_edge + side_offset,
                                        j + 1 + (i + 1) * verts_per_edge + side_offset});
            }
    return m;
}

// Scene bounds as SimpleViewer approximates them (main.cpp:396-405): bounding spheres of the mesh AABBs.
struct BoundsEstimator {
    AABB scene = AABB::invalid();
    void add(AABB mesh_bounds, const Transform& t) {
        Vector3f center = t * mesh_bounds.center();
        float radius = magnitude(mesh_bounds.size() * t.scale) * 0.5f;
        scene.grow_to_contain(AABB{center - Vector3f(radius), center + Vector3f(radius)});
    }
    float scene_size() const { return magnitude(scene.size()); }
};

void create_cornell_box(SceneBuilder& sb, unsigned wall_quads_per_edge) {
    const uint16_t thin = HIPR_MATERIAL_THIN_WALLED;
    const uint32_t white = sb.add_material(SceneBuilder::make_material(RGB(0.98f), 1.0f, 0.02f, 0.0f, thin));
    const uint32_t red = sb.add_material(SceneBuilder::make_material(RGB(0.98f, 0.02f, 0.02f), 1.0f, 0.02f, 0.0f, thin));
    const uint32_t green = sb.add_material(SceneBuilder::make_material(RGB(0.02f, 0.98f, 0.02f), 1.0f, 0.02f, 0.0f, thin));
    const uint32_t iron = sb.add_material(SceneBuilder::make_material(RGB(0.560f, 0.570f, 0.580f), 0.4f, 1.0f, 1.0f));
    const uint32_t copper = sb.add_material(SceneBuilder::make_material(RGB(0.955f, 0.637f, 0.538f), 0.02f, 1.0f, 1.0f));

    sb.camera.transform = Transform(Vector3f(0, 0.0f, -1.5f));
    sb.add_light(SceneBuilder::sphere_light(Vector3f(0.0f, 0.45f, 0.0f), RGB(2.0f), 0.05f));

    BoundsEstimator bounds;
    const AABB plane_bounds = {Vector3f(-0.5f, 0.0f, -0.5f), Vector3f(0.5f, 0.0f, 0.5f)};
    const AABB box_bounds = {Vector3f(-0.5f), Vector3f(0.5f)};
    const float PI_half = PI<float>() * 0.5f;

    const uint32_t plane_mesh = sb.add_mesh(plane(wall_quads_per_edge ? wall_quads_per_edge : 1u, true, false));   // MeshFlag::GeometryBuffers
    auto add_wall = [&](uint32_t material, Transform t) { sb.add_model(plane_mesh, material, t); bounds.add(plane_bounds, t); };
    add_wall(white, Transform(Vector3f(0.0f, -0.5f, 0.0f)));
    add_wall(white, Transform(Vector3f(0.0f, 0.5f, 0.0f), Quaternionf::from_angle_axis(PI<float>(), Vector3f::forward())));
    add_wall(white, Transform(Vector3f(0.0f, 0.0f, 0.5f), Quaternionf::from_angle_axis(-PI_half, Vector3f::right())));
    add_wall(red, Transform(Vector3f(-0.5f, 0.0f, 0.0f), Quaternionf::from_angle_axis(-PI_half, Vector3f::forward())));
    add_wall(green, Transform(Vector3f(0.5f, 0.0f, 0.0f), Quaternionf::from_angle_axis(PI_half, Vector3f::forward())));

    {   // small box: DefaultBuffers = positions + normals + tints
        const uint32_t mesh = sb.add_mesh(box(1, Vector3f::one(), true));
        Transform t(Vector3f(0.2f, -0.35f, -0.2f), Quaternionf::from_angle_axis(PI<float>() / 6.0f, Vector3f::up()), 0.3f);
        sb.add_model(mesh, iron, t);
        bounds.add(box_bounds, t);
    }
    {   // big box: y stretched after creation; the reference leaves the mesh bounds untouched
        MeshData big = box(1, Vector3f::one(), true);
        for (Vector3f& p : big.positions) p.y *= 2.0f;
        const uint32_t mesh = sb.add_mesh(std::move(big));
        Transform t(Vector3f(-0.2f, -0.2f, 0.2f), Quaternionf::from_angle_axis(-PI<float>() / 6.0f, Vector3f::up()), 0.3f);
        sb.add_model(mesh, copper, t);
        bounds.add(box_bounds, t);
    }

    const float scene_size = bounds.scene_size();
    sb.camera.near_plane = scene_size / 10000.0f;
    sb.camera.far_plane = scene_size * 3.0f;
    sb.camera.max_bounce_count = 32;   // SimpleViewer built-in scenes, main.cpp:353
    sb.set_environment_tint(RGB(0.68f, 0.92f, 1.0f));   // SimpleViewer default g_environment_color, main.cpp:58
}

void create_atrium(SceneBuilder& sb, unsigned target_triangles, unsigned seed, bool textured) { build_atrium(sb, target_triangles, seed, textured); }

void create_quad_scene(SceneBuilder& sb, unsigned width, unsigned height) {
    // create_ortho_camera_with_quad_scene, tests/OptiXRendererTests/RendererTest.h:67-117: a quad covering the
    // orthographic camera, vertex tints ramp red with x and green with y, thin walled diffuse white material.
    MeshData quad;
    const float w = float(width), h = float(height);
    quad.positions = {Vector3f(-0.5f * w, -0.5f * h, 1.0f), Vector3f(-0.5f * w, 0.5f * h, 1.0f), Vector3f(0.5f * w, -0.5f * h, 1.0f), Vector3f(0.5f * w, 0.5f * h, 1.0f)};
    quad.tints = {0x00FF0000u, 0x00FFFF00u, 0x00FF00FFu, 0x00FFFFFFu};   // (r, g, b = 1, roughness = 0), little endian uchar4
    quad.primitives = {{0, 1, 2}, {1, 2, 3}};
    HiprMaterial m = SceneBuilder::make_material(RGB(1.0f), 0.0f, 0.04f, 0.0f, HIPR_MATERIAL_THIN_WALLED, HIPR_SHADING_DIFFUSE);
    sb.add_model(sb.add_mesh(std::move(quad)), sb.add_material(m), Transform::identity());
    create_empty_ortho_scene(sb, width, height, RGB(1.0f));
}

void create_empty_ortho_scene(SceneBuilder& sb, unsigned width, unsigned height, RGB environment_tint) {
    // create_ortho_camera, RendererTest.h:67-80
    sb.set_environment_tint(environment_tint);
    sb.camera.transform = Transform::identity();
    sb.camera.orthographic = true;
    sb.camera.ortho_width = float(width);
    sb.camera.ortho_height = float(height);
    sb.camera.ortho_depth = 1000.0f;
    sb.camera.max_bounce_count = 4;
}

} // namespace Scenes
} // namespace HIPRenderer
