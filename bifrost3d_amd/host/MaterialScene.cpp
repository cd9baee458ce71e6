// MaterialScene.cpp -- see MaterialScene.h.
#include "MaterialScene.h"
#include "AtriumScene.h"

#include "glTFLoader/glTFLoader.h"

#include <cmath>
#include <cstdio>
#include <vector>

using namespace Bifrost;
using namespace Bifrost::Assets;
using namespace Bifrost::Math;
using namespace Bifrost::Scene;

namespace ViewerScenes {

namespace {

const RGB gold_tint = RGB(1.000f, 0.766f, 0.336f);      // BF/Assets/Material.h:67

inline float lerp(float a, float b, float t) { return a + (b - a) * t; }
inline RGB lerp(RGB a, RGB b, float t) { return RGB(lerp(a.r, b.r, t), lerp(a.g, b.g, t), lerp(a.b, b.b, t)); }

MeshModel attached_mesh_model(SceneNode node) {
    for (MeshModelID model_ID : MeshModels::get_iterable())
        if (MeshModels::get_scene_node_ID(model_ID) == node.get_ID()) return model_ID;
    return MeshModel();
}

template <typename F>
void apply_to_children_recursively(SceneNode node, F&& f) {
    for (SceneNodeID child : node.get_children()) { f(SceneNode(child)); apply_to_children_recursively(SceneNode(child), f); }
}

void apply_delta_transform(SceneNode node, Transform delta) { node.set_global_transform(node.get_global_transform() * delta); }   // BF/Scene/SceneNode.cpp:227-234

// Material.cpp:121-141: a new node per node of the hierarchy, sharing meshes and materials.
SceneNode shallow_clone(SceneNode node) {
    SceneNode cloned_node = SceneNode(node.get_name(), node.get_global_transform());
    MeshModel mesh_model = attached_mesh_model(node);
    if (mesh_model.get_ID() != MeshModelID::invalid_UID()) MeshModel(cloned_node, mesh_model.get_mesh(), mesh_model.get_material());
    for (SceneNodeID child : node.get_children()) {
        SceneNode cloned_child = shallow_clone(SceneNode(child));
        cloned_child.set_parent(cloned_node);
    }
    return cloned_node;
}

void replace_material(Material material, SceneNode parent_node, const std::string& child_scene_node_name) {     // Utils.cpp:64-72
    apply_to_children_recursively(parent_node, [&](SceneNode node) {
        if (node.get_name() != child_scene_node_name) return;
        MeshModel mesh_model = attached_mesh_model(node);
        if (mesh_model.get_ID() != MeshModelID::invalid_UID()) MeshModels::set_material_ID(mesh_model.get_ID(), material.get_ID());
    });
}

// BF/Assets/MeshCreation.cpp:328-392 revolved_sphere: a latitude-longitude sphere of radius 0.5, the pole rows collapsed to points
// and their degenerate triangles left out.
Mesh revolved_sphere(const std::string& name, unsigned longitude_quads, unsigned latitude_quads) {
    const unsigned latitude_size = latitude_quads + 1, longitude_size = longitude_quads + 1;
    const float radius = 0.5f;
    MeshFlags buffers = MeshFlag::Position; buffers |= MeshFlag::Normal; buffers |= MeshFlag::Texcoord;
    Mesh mesh = Mesh(name, (latitude_quads * longitude_quads - longitude_quads) * 2, latitude_size * longitude_size, buffers);
    for (unsigned y = 0; y < latitude_size; ++y)
        for (unsigned x = 0; x < longitude_size; ++x) {
            const unsigned v = y * longitude_size + x;
            const Vector2f tc = {float(x) * (1.0f / longitude_quads), float(y) * (1.0f / latitude_quads)};
            const float theta = tc.y * PI<float>(), phi = tc.x * 2.0f * PI<float>(), sin_theta = std::sin(theta);
            mesh.get_texcoords()[v] = tc;
            mesh.get_positions()[v] = Vector3f(-sin_theta * std::sin(phi), std::cos(theta), sin_theta * std::cos(phi)) * radius;
            mesh.get_normals()[v] = normalize(mesh.get_positions()[v]);
        }
    for (unsigned x = 0; x < longitude_size; ++x) {
        mesh.get_positions()[x] = Vector3f(0, radius, 0);
        mesh.get_positions()[(latitude_size - 1) * longitude_size + x] = Vector3f(0, -radius, 0);
    }
    Vector3ui* primitives = mesh.get_primitives();
    for (unsigned y = 0; y < latitude_quads; ++y)
        for (unsigned x = 0; x < longitude_quads; ++x) {
            const unsigned base = x + y * longitude_size;
            if (y != 0) *primitives++ = Vector3ui{base, base + 1, base + longitude_size};
            if (y != latitude_quads - 1) *primitives++ = Vector3ui{base + 1, base + longitude_size + 1, base + longitude_size};
        }
    mesh.set_bounds(AABB{Vector3f(-radius), Vector3f(radius)});
    return mesh;
}

// ---- the procedural shader ball ------------------------------------------------------------------------------------------------------------------
// A stand-in for Resources/Shaderball.gltf (the Mori knob; not in this repository and not on the GPU box) that costs a ray what the asset costs it.
// The asset, measured where it is (tests/test_loaders_cpu.py): 11 952 + 13 332 triangles in a 2 x 2 x 2 box holding FIVE to SIX times the area of the
// enclosing sphere -- a thick outer shell with round openings, a ball inside it that is itself several layers, a base of radius 1.1 -- with
// triangle edges between 0.0005 and 0.4 and aspect ratios of 20 at the 90th percentile. A ray visits 4.2 nodes and tests 4.8 triangles of it. The first
// stand-in (one sphere over a squashed sphere, regular grid) cost 2.8 and 1.6. This one is built from the same ingredients as the asset:
//   * layers: latitude-longitude patches of a sphere, outward or inward facing, with quads left out where a predicate says so (the openings);
//   * an irregular tessellation: the grid's rows and columns are unevenly spaced, its meridians are helices and every vertex is moved inside its cell
//     along the surface, so that triangles are long, slanted and of very different sizes while the shape stays the sphere's.
// The shape parameters below were tuned until the oracle's counters on this scene matched the asset's within 10 % (enforced in the build container by
// tests/test_loaders_cpu.py::test_material_scene_on_the_reference_shader_ball_against_the_stand_in).
struct Patches {
    std::vector<Vector3f> positions, normals;
    std::vector<Vector2f> texcoords;
    std::vector<Vector3ui> primitives;
};

inline float hash01(uint32_t a, uint32_t b, uint32_t seed) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return float(h >> 8) * (1.0f / 16777216.0f);
}

// A layer: the part of the ellipsoid centre + scale * direction(theta, phi) whose quads `keep(direction of the quad's middle)` accepts, theta in
// [theta0, theta1] of pi, the grid unevenly spaced (`unevenness` in [0, 1)) and jittered (`jitter` in cells). `inward`: the surface faces the centre.
template <typename Keep>
void add_layer(Patches& out, unsigned longitude_quads, unsigned latitude_quads, Vector3f centre, Vector3f scale, float theta0, float theta1, bool inward, float unevenness, float jitter,
               uint32_t seed, Keep&& keep) {
    // quads 1.56 times as wide (along a parallel) as they are high at equal counts: with the helical meridians below they become slanted slivers
    longitude_quads = unsigned(longitude_quads / 0.8f + 0.5f); latitude_quads = unsigned(latitude_quads * 0.8f + 0.5f);
    const float J_TWIST = 1.0f;      // the grid's meridians are helices: quads become slanted parallelograms whose boxes overlap their neighbours', as a sculpted mesh's do
    const unsigned latitude_size = latitude_quads + 1, longitude_size = longitude_quads + 1;
    const unsigned first_vertex = unsigned(out.positions.size());
    auto direction = [](float theta, float phi) { const float s = std::sin(theta); return Vector3f(-s * std::sin(phi), std::cos(theta), s * std::cos(phi)); };
    // uneven spacing: a smooth monotone warp of the unit interval, u + a sin(2 pi k u) / (2 pi k) has derivative 1 + a cos(.) > 0
    auto warp = [&](float u, float k) { return u + unevenness * std::sin(2.0f * PI<float>() * k * u) / (2.0f * PI<float>() * k); };
    for (unsigned y = 0; y < latitude_size; ++y)
        for (unsigned x = 0; x < longitude_size; ++x) {
            float u = float(x) / longitude_quads, v = float(y) / latitude_quads;
            const bool seam = x == 0 || x == longitude_quads, rim = y == 0 || y == latitude_quads;
            if (!seam) u += (hash01(x, y, seed) - 0.5f) * jitter / longitude_quads;
            if (!rim) v += (hash01(y, x, seed ^ 0x51ED27u) - 0.5f) * jitter / latitude_quads;
            u = warp(u, 3.0f); v = warp(v, 2.0f);
            const float theta = (theta0 + (theta1 - theta0) * v) * PI<float>(), phi = (u + J_TWIST * v) * 2.0f * PI<float>();
            const Vector3f d = direction(theta, phi);
            out.positions.push_back(centre + Vector3f(d.x * scale.x, d.y * scale.y, d.z * scale.z));
            const Vector3f n = normalize(Vector3f(d.x / scale.x, d.y / scale.y, d.z / scale.z));
            out.normals.push_back(inward ? -n : n);
            out.texcoords.push_back(Vector2f{u, theta0 + (theta1 - theta0) * v});
        }
    for (unsigned y = 0; y < latitude_quads; ++y)
        for (unsigned x = 0; x < longitude_quads; ++x) {
            const float theta = (theta0 + (theta1 - theta0) * (y + 0.5f) / latitude_quads) * PI<float>(), phi = ((x + 0.5f) / longitude_quads + J_TWIST * (y + 0.5f) / latitude_quads) * 2.0f * PI<float>();
            if (!keep(direction(theta, phi))) continue;
            const unsigned base = first_vertex + x + y * longitude_size;
            const bool north_pole = y == 0 && theta0 == 0.0f, south_pole = y == latitude_quads - 1 && theta1 == 1.0f;      // the collapsed rows: one triangle per quad
            const unsigned a = base, b = base + 1, c = base + longitude_size, d = base + longitude_size + 1;
            if (!north_pole) out.primitives.push_back(inward ? Vector3ui{a, c, b} : Vector3ui{a, b, c});
            if (!south_pole) out.primitives.push_back(inward ? Vector3ui{b, c, d} : Vector3ui{b, d, c});
        }
}

Mesh mesh_of(const std::string& name, const Patches& patches) {
    MeshFlags buffers = MeshFlag::Position; buffers |= MeshFlag::Normal; buffers |= MeshFlag::Texcoord;
    Mesh mesh = Mesh(name, unsigned(patches.primitives.size()), unsigned(patches.positions.size()), buffers);
    std::copy(patches.positions.begin(), patches.positions.end(), mesh.get_positions());
    std::copy(patches.normals.begin(), patches.normals.end(), mesh.get_normals());
    std::copy(patches.texcoords.begin(), patches.texcoords.end(), mesh.get_texcoords());
    std::copy(patches.primitives.begin(), patches.primitives.end(), mesh.get_primitives());
    mesh.compute_bounds();
    return mesh;
}

SceneNode create_procedural_shader_ball(Material outer_material, Material inner_material) {
    // tuned against the asset (see above): row / column spacing varies by 0.6, vertices move up to 0.3 cells, the grid's meridians make one turn from pole to pole
    const float J_UNEVEN = 0.6f, J_JITTER = 0.6f;
    SceneNode ball_node = SceneNode("ShaderBall");
    // Five round openings in the shell: four around the upper half, one on top (object space; the ball is scaled by two in the scene).
    const Vector3f opening_axes[5] = {normalize(Vector3f(1, 0.45f, 1)), normalize(Vector3f(-1, 0.45f, 1)), normalize(Vector3f(1, 0.45f, -1)), normalize(Vector3f(-1, 0.45f, -1)), Vector3f(0, 1, 0)};
    const float opening_cosine = std::cos(0.40f);
    auto shell = [&](Vector3f d) { for (const Vector3f& axis : opening_axes) if (dot(d, axis) > opening_cosine) return false; return true; };
    auto everywhere = [](Vector3f) { return true; };
    const Vector3f centre = Vector3f(0, 0.06f, 0);

    // The tested material ("Node5"): the shell's outside and inside faces and the base it stands on.
    Patches outer;
    add_layer(outer, 74, 46, centre, Vector3f(0.5f, 0.5f, 0.5f), 0.0f, 1.0f, false, J_UNEVEN, J_JITTER, 1u, shell);
    add_layer(outer, 64, 42, centre, Vector3f(0.44f, 0.44f, 0.44f), 0.0f, 1.0f, true, J_UNEVEN, J_JITTER, 2u, shell);
    add_layer(outer, 56, 20, Vector3f(0, -0.42f, 0), Vector3f(0.54f, 0.08f, 0.54f), 0.0f, 1.0f, false, J_UNEVEN, J_JITTER, 3u, everywhere);
    SceneNode outer_node = SceneNode("Node5");
    MeshModel(outer_node, mesh_of("ShaderBallOuter", outer), outer_material);
    outer_node.set_parent(ball_node);

    // The rubber inside ("Node2"): a ball of three layers, the outer two grooved so that the ones below show.
    Patches inner;
    auto grooved = [](Vector3f d) { const float phi = std::atan2(d.x, d.z); return std::fmod(std::fabs(phi) * (8.0f / PI<float>()), 2.0f) < 1.4f; };
    auto banded = [](Vector3f d) { return std::fmod((d.y + 1.0f) * 4.0f, 2.0f) < 1.5f; };
    add_layer(inner, 84, 42, centre, Vector3f(0.365f, 0.365f, 0.365f), 0.0f, 1.0f, false, J_UNEVEN, J_JITTER, 4u, grooved);
    add_layer(inner, 74, 42, centre, Vector3f(0.355f, 0.355f, 0.355f), 0.0f, 1.0f, false, J_UNEVEN, J_JITTER, 5u, banded);
    add_layer(inner, 64, 42, centre, Vector3f(0.34f, 0.34f, 0.34f), 0.0f, 1.0f, false, J_UNEVEN, J_JITTER, 6u, everywhere);
    SceneNode inner_node = SceneNode("Node2");
    MeshModel(inner_node, mesh_of("ShaderBallInner", inner), inner_material);
    inner_node.set_parent(ball_node);
    return ball_node;
}

} // namespace

SceneNode create_checkered_floor(float floor_size, float checker_size) {
    const unsigned size = 2;
    unsigned char tint_roughness_pixels[size * size * 4];
    for (unsigned y = 0; y < size; ++y)
        for (unsigned x = 0; x < size; ++x) {
            const bool is_black = (x & 1) != (y & 1);
            unsigned char* pixel = tint_roughness_pixels + (x + y * size) * 4u;
            pixel[0] = pixel[1] = pixel[2] = is_black ? 1 : 255;
            pixel[3] = is_black ? 15 : 255;
        }
    Image tint_roughness_image = Image::create2D("Floor color", PixelFormat::RGBA32, true, size, size, tint_roughness_pixels);

    Materials::Data material_data = Materials::Data::create_dielectric(RGB(1.0f), 0.4f, 0.04f);
    material_data.tint_roughness_texture_ID = Textures::create2D(tint_roughness_image.get_ID(), MagnificationFilter::None, MinificationFilter::Trilinear);
    material_data.flags = MaterialFlag::ThinWalled;
    Material material = Material("Floor", material_data);

    // Mesh scaled to the floor size, texture coordinates to match the checker size.
    MeshFlags buffers = MeshFlag::Position; buffers |= MeshFlag::Texcoord;
    Mesh plane_mesh = MeshCreation::plane(2, buffers);
    for (unsigned v = 0; v < plane_mesh.get_vertex_count(); ++v) plane_mesh.get_positions()[v] = plane_mesh.get_positions()[v] * floor_size;
    const float uv_scale = floor_size / (2 * checker_size);     // a texture is 2 x 2 checkers
    for (unsigned v = 0; v < plane_mesh.get_vertex_count(); ++v) {
        Vector2f& texcoord = plane_mesh.get_texcoords()[v];
        texcoord = Vector2f{(texcoord.x - 0.5f) * uv_scale, (texcoord.y - 0.5f) * uv_scale};     // centred: texcoords stay precise near the middle of the floor
    }
    plane_mesh.compute_bounds();

    SceneNode plane_node = SceneNode("Floor");
    MeshModel(plane_node, plane_mesh, material);
    return plane_node;
}

SceneNode load_shader_ball(const std::string& shader_ball_path, Material material) {
    Material rubber_material = Material::create_dielectric("Rubber", RGB(0.05f), 1);    // the inside
    if (shader_ball_path.empty()) return create_procedural_shader_ball(material, rubber_material);

    printf("Mori knob curtesy of Yasutoshi Mori\n");
    SceneNode shader_ball_node = glTFLoader::load(shader_ball_path);
    if (shader_ball_node == SceneNode::invalid()) return shader_ball_node;

    std::vector<MeshModel> discarded;
    apply_to_children_recursively(shader_ball_node, [&](SceneNode node) {
        MeshModel mesh_model = attached_mesh_model(node);
        if (mesh_model.get_ID() == MeshModelID::invalid_UID()) return;
        if (node.get_name() == "Node5") MeshModels::set_material_ID(mesh_model.get_ID(), material.get_ID());
        else if (node.get_name() == "Node2") MeshModels::set_material_ID(mesh_model.get_ID(), rubber_material.get_ID());
        else discarded.push_back(mesh_model);     // anything but the shader ball
    });
    for (MeshModel mesh_model : discarded) {
        Meshes::destroy(mesh_model.get_mesh().get_ID());
        Materials::destroy(mesh_model.get_material().get_ID());
        MeshModels::destroy(mesh_model.get_ID());
    }
    return shader_ball_node;
}

void create_material_scene(CameraID camera_ID, SceneNode root_node, const std::string& shader_ball_path, bool coat) {
    { // Camera.
        Transform cam_transform = Cameras::get_transform(camera_ID);
        cam_transform.translation = Vector3f(0, 5.5f, -18.5f);
        cam_transform.look_at(Vector3f(0, 0.5f, 0.0f));
        Cameras::set_transform(camera_ID, cam_transform);
    }
    { // A directional light.
        Transform light_transform = Transform(Vector3f(20.0f, 20.0f, -20.0f));
        light_transform.look_at(Vector3f::zero());
        SceneNode light_node = SceneNode("light", light_transform);
        light_node.set_parent(root_node);
        LightSources::create_directional_light(light_node.get_ID(), RGB(3.0f, 2.9f, 2.5f));
    }
    { // Checkered floor.
        SceneNode floor_node = create_checkered_floor(400, 1);
        floor_node.set_global_transform(Transform(Vector3f(0, -1.0f, 0)));
        floor_node.set_parent(root_node);
    }

    // The blended materials (Material.cpp:30-47).
    const int material_count = 7;
    Materials::Data material0_data = Materials::Data::create_dielectric(RGB(0.02f, 0.27f, 0.33f), 1.0f, 0.04f);
    Materials::Data material1_data = Materials::Data::create_metal(gold_tint, 0.02f);
    material1_data.specularity = material0_data.specularity;
    Material materials[material_count];
    for (int m = 0; m < material_count; ++m) {
        const float lerp_t = m / (material_count - 1.0f);
        Materials::Data material_data = {};
        material_data.tint = lerp(material0_data.tint, material1_data.tint, lerp_t);
        material_data.roughness = lerp(material0_data.roughness, material1_data.roughness, lerp_t);
        material_data.specularity = lerp(material0_data.specularity, material1_data.specularity, lerp_t);
        material_data.metallic = lerp(material0_data.metallic, material1_data.metallic, lerp_t);
        material_data.coverage = lerp(material0_data.coverage, material1_data.coverage, lerp_t);
        if (coat) { material_data.coat = 1.0f; material_data.coat_roughness = 0.7f; }
        materials[m] = Material("Lerped material " + std::to_string(m), material_data);
    }

    { // The material models.
        const float shader_ball_distance = 1.2f;
        SceneNode shader_ball_node = load_shader_ball(shader_ball_path, materials[0]);
        if (shader_ball_node == SceneNode::invalid()) return;
        shader_ball_node.set_global_transform(Transform(Vector3f::zero(), Quaternionf::identity(), 2.0f));
        const float shader_ball_pos_x = -shader_ball_distance * 0.5f * (material_count - 1);
        apply_delta_transform(shader_ball_node, Transform(Vector3f(shader_ball_pos_x, 0, 0)));
        shader_ball_node.set_parent(root_node);

        for (int m = 1; m < material_count; ++m) {
            SceneNode shader_ball_node_clone = shallow_clone(shader_ball_node);
            apply_delta_transform(shader_ball_node_clone, Transform(Vector3f(m * shader_ball_distance, 0, 0)));
            shader_ball_node_clone.set_parent(root_node);
            replace_material(materials[m], shader_ball_node_clone, "Node5");     // the outer surface shows the tested material
        }
    }
}

void create_glass_scene(CameraID camera_ID, SceneNode root_node, const std::string& shader_ball_path, const std::string& diamond_path) {
    auto dielectric_specularity = [](float ior_o, float ior_i) { const float r = (ior_o - ior_i) / (ior_o + ior_i); return r * r; };     // BF/Math/Utils.h:202-204
    const float glass_specularity = dielectric_specularity(1.0f, 1.52f), diamond_specularity = dielectric_specularity(1.0f, 2.42f);   // BF/Assets/Material.h:45-58

    { // Camera.
        Transform cam_transform = Cameras::get_transform(camera_ID);
        cam_transform.translation = Vector3f(0, 3.0f, -10.0f);
        cam_transform.look_at(Vector3f(0, 1.0f, 0.0f));
        Cameras::set_transform(camera_ID, cam_transform);
    }
    { // A directional light.
        Transform light_transform = Transform(Vector3f(20.0f, 20.0f, -20.0f));
        light_transform.look_at(Vector3f::zero());
        SceneNode light_node = SceneNode("Directional light", light_transform);
        light_node.set_parent(root_node);
        LightSources::create_directional_light(light_node.get_ID(), RGB(3.0f, 2.9f, 2.5f));
    }
    { // A sphere light.
        SceneNode sphere_light_node = SceneNode("Sphere light", Transform(Vector3f(1, 5, 2)));
        LightSources::create_sphere_light(sphere_light_node.get_ID(), RGB(800.0f, 600.0f, 600.0f), 1.0f);
        sphere_light_node.set_parent(root_node);
    }
    { // Floor.
        SceneNode floor_node = create_checkered_floor(400, 1);
        floor_node.set_global_transform(Transform(Vector3f(0, -1.0f, 0)));
        floor_node.set_parent(root_node);
    }
    { // Glass shader ball.
        Material glass_material = Materials::create("Glass shader ball", Materials::Data::create_transmissive(RGB(0.95f), 0.25f, glass_specularity));
        SceneNode shader_ball_node = load_shader_ball(shader_ball_path, glass_material);
        if (shader_ball_node != SceneNode::invalid()) {
            shader_ball_node.set_global_transform(Transform(Vector3f(0, -0.25f, 0), Quaternionf::from_angle_axis(0.1f * PI<float>(), Vector3f::up()), 1.5f));
            shader_ball_node.set_parent(root_node);
        }
    }
    { // Magnifying glass: the lens and a handle.
        Material lens_material = Materials::create("Magnifying glass", Materials::Data::create_transmissive(RGB(0.975f), 0.0f, glass_specularity));
        Mesh lens_mesh = revolved_sphere("Lens", 64, 32);
        for (unsigned v = 0; v < lens_mesh.get_vertex_count(); ++v) {
            lens_mesh.get_positions()[v].z *= 0.1f;
            Vector3f& normal = lens_mesh.get_normals()[v];
            normal = normalize(Vector3f(normal.x * 0.1f, normal.y * 0.1f, normal.z));
        }
        lens_mesh.compute_bounds();
        SceneNode lens_node = SceneNode("Glass", Transform(Vector3f(3, 0.1f, 0)));
        MeshModel(lens_node, lens_mesh, lens_material);
        lens_node.set_parent(root_node);

        Material frame_material = Material::create_metal("Magnifying glass frame", gold_tint, 0.5f);
        MeshFlags buffers = MeshFlag::Position; buffers |= MeshFlag::Normal;
        Mesh handle_mesh = MeshCreation::box(1, Vector3f(0.08f, 0.6f, 0.08f), buffers);
        SceneNode handle_node = SceneNode("Magnifying glass handle", Transform(Vector3f(3, 0.1f - 0.8f, 0)));
        MeshModel(handle_node, handle_mesh, frame_material);
        handle_node.set_parent(root_node);
    }
    { // Diamond.
        Material diamond_material = Materials::create("Diamond", Materials::Data::create_transmissive(RGB(0.94f), 0.0f, diamond_specularity));
        if (!diamond_path.empty()) {
            SceneNode diamond_node = glTFLoader::load(diamond_path);
            if (diamond_node != SceneNode::invalid()) {
                diamond_node.set_global_transform(Transform(Vector3f(-3, 0, 0), Quaternionf::from_angle_axis(-0.5f * PI<float>(), Vector3f::right())));
                diamond_node.set_parent(root_node);
                if (diamond_node.get_name() == "pCone1_DiamondOutside_0") {     // a single-node file is its own root
                    MeshModel model = attached_mesh_model(diamond_node);
                    if (model.get_ID() != MeshModelID::invalid_UID()) MeshModels::set_material_ID(model.get_ID(), diamond_material.get_ID());
                }
                replace_material(diamond_material, diamond_node, "pCone1_DiamondOutside_0");
            }
        } else {
            // A brilliant's silhouette from a coarse revolved sphere: flat table, wide girdle, pointed pavilion; flat facets (no vertex normals).
            Mesh sphere = revolved_sphere("Diamond", 8, 4);
            MeshFlags buffers = MeshFlag::Position;
            Mesh gem = Mesh("Diamond", sphere.get_primitive_count(), sphere.get_vertex_count(), buffers);
            for (unsigned t = 0; t < sphere.get_primitive_count(); ++t) gem.get_primitives()[t] = sphere.get_primitives()[t];
            for (unsigned v = 0; v < sphere.get_vertex_count(); ++v) {
                Vector3f p = sphere.get_positions()[v] * 2.0f;        // unit radius
                if (p.y > 0.0f) p.y *= 0.35f;                          // crown: low and wide
                p.y = p.y > 0.3f ? 0.3f : p.y;                         // table
                gem.get_positions()[v] = Vector3f(p.x * 0.8f, p.y * 0.9f, p.z * 0.8f);
            }
            gem.compute_bounds();
            Meshes::destroy(sphere.get_ID());
            SceneNode diamond_node = SceneNode("pCone1_DiamondOutside_0", Transform(Vector3f(-3, -0.1f, 0)));
            MeshModel(diamond_node, gem, diamond_material);
            diamond_node.set_parent(root_node);
        }
    }
}

// apps/SimpleViewer/Scenes/Opacity.h:27-104. `quads_per_edge` tessellates the box and the two planes (same surfaces and texture
// coordinates, more triangles) so that the scene can be served by each of the three searches.
void create_opacity_scene(CameraID camera_ID, SceneNode root_node, unsigned quads_per_edge) {
    if (quads_per_edge == 0) quads_per_edge = 1;
    { // Camera.
        Transform cam_transform = Cameras::get_transform(camera_ID);
        cam_transform.translation = Vector3f(0, 1, -6);
        Cameras::set_transform(camera_ID, cam_transform);
    }
    { // Floor.
        SceneNode floor_node = create_checkered_floor(400, 1);
        floor_node.set_global_transform(Transform(Vector3f(0, -0.0005f, 0)));
        floor_node.set_parent(root_node);
        Material floor_material = attached_mesh_model(floor_node).get_material();
        floor_material.set_tint(RGB(0.02f, 0.27f, 0.33f));
        floor_material.set_roughness(0.3f);
    }
    { // Sphere light.
        SceneNode light_node = SceneNode("Light", Transform(Vector3f(0.0f, 0.5f, 0.0f)));
        light_node.set_parent(root_node);
        LightSources::create_sphere_light(light_node.get_ID(), RGB(50.0f), 0.1f);
    }
    { // Cut-out box around the light: a 17 x 17 grid texture, every odd texel in both directions is a hole.
        const unsigned width = 17, height = 17;
        unsigned char pixels[width * height];
        for (unsigned y = 0; y < height; ++y)
            for (unsigned x = 0; x < width; ++x) pixels[x + y * width] = ((x & 1) == 0 || (y & 1) == 0) ? 255 : 0;
        Image image = Image::create2D("Grid", PixelFormat::Alpha8, false, width, height, pixels);

        Materials::Data material_data = Materials::Data::create_dielectric(RGB(0.005f, 0.01f, 0.25f), 0.05f, 0.04f);
        material_data.coverage_texture_ID = Textures::create2D(image.get_ID(), MagnificationFilter::None, MinificationFilter::None);
        material_data.flags = MaterialFlag::Cutout;
        Material material = Materials::create("Plastic", material_data);

        SceneNode box_node = SceneNode("Swizz box", Transform(Vector3f(0.0f, 0.5f, 0.0f)));
        MeshFlags buffers = MeshFlag::Position; buffers |= MeshFlag::Texcoord;
        Mesh box_mesh = MeshCreation::box(quads_per_edge, Vector3f::one(), buffers);
        MeshModel(box_node, box_mesh, material);
        box_node.set_parent(root_node);
    }
    { // Two partially covering, thin-walled planes in front of the box.
        Materials::Data transparent_material_data = Materials::Data::create_dielectric(RGB(0.25f), 0.95f, 0.04f);
        transparent_material_data.coverage = 0.75f;
        transparent_material_data.flags = MaterialFlag::ThinWalled;
        Material transparent_material = Materials::create("Transparent", transparent_material_data);
        Mesh plane_mesh = MeshCreation::plane(quads_per_edge, MeshFlag::Position);
        const Quaternionf rotation = Quaternionf::from_angle_axis(PI<float>() * 0.5f, Vector3f::right());
        const Transform transforms[2] = {Transform(Vector3f(1.0f, 1.0f, -2.0f), rotation, 2.0f), Transform(Vector3f(0.0f, 0.25f, -3.0f), rotation, 1.0f)};
        for (const Transform& transform : transforms) {
            SceneNode plane_node = SceneNode("Plane", transform);
            MeshModel(plane_node, plane_mesh, transparent_material);
            plane_node.set_parent(root_node);
        }
    }
}

// ---- the atrium through the Bifrost managers --------------------------------------------------------------------------------------------
namespace {
struct BifrostSceneSink {
    static constexpr bool takes_textures = false;       // the textured atrium is built in the flat builder only
    SceneNode root;
    SceneRootID scene_root;
    std::vector<Mesh> meshes;
    std::vector<Material> materials = {Material()};       // index 0 = the invalid material, as in SceneBuilder
    HIPRenderer::CameraDescription camera;

    uint32_t add_mesh(HIPRenderer::MeshData data) {
        MeshFlags buffers = MeshFlag::Position;
        if (!data.normals.empty()) buffers |= MeshFlag::Normal;
        if (!data.texcoords.empty()) buffers |= MeshFlag::Texcoord;
        Mesh mesh("Atrium part", unsigned(data.primitives.size()), unsigned(data.positions.size()), buffers);
        std::memcpy(mesh.get_primitives(), data.primitives.data(), data.primitives.size() * sizeof(Vector3ui));
        std::memcpy(mesh.get_positions(), data.positions.data(), data.positions.size() * sizeof(Vector3f));
        if (!data.normals.empty()) std::memcpy(mesh.get_normals(), data.normals.data(), data.normals.size() * sizeof(Vector3f));
        if (!data.texcoords.empty()) std::memcpy(mesh.get_texcoords(), data.texcoords.data(), data.texcoords.size() * sizeof(Vector2f));
        mesh.compute_bounds();
        meshes.push_back(mesh);
        return uint32_t(meshes.size() - 1);
    }
    uint32_t add_material(const HiprMaterial& m) {
        Materials::Data d = {};
        if (m.flags & HIPR_MATERIAL_THIN_WALLED) d.flags |= MaterialFlag::ThinWalled;
        if (m.flags & HIPR_MATERIAL_CUTOUT) d.flags |= MaterialFlag::Cutout;
        d.shading_model = m.shading_model == HIPR_SHADING_DIFFUSE ? ShadingModel::Diffuse : (m.shading_model == HIPR_SHADING_TRANSMISSIVE ? ShadingModel::Transmissive : ShadingModel::Default);
        d.tint = RGB(m.tint[0], m.tint[1], m.tint[2]);
        d.roughness = m.roughness; d.specularity = m.specularity; d.metallic = m.metallic;
        d.coat = m.coat / 65535.0f; d.coat_roughness = m.coat_roughness / 65535.0f;
        d.coverage = m.coverage;
        d.emission = RGB(m.emission[0], m.emission[1], m.emission[2]);
        materials.push_back(Materials::create("Atrium material", d));
        return uint32_t(materials.size() - 1);
    }
    uint32_t add_model(uint32_t mesh, uint32_t material, const Transform& transform, uint32_t = 0) {
        SceneNode node = SceneNode("Atrium model", transform);
        node.set_parent(root);
        MeshModel model(node, meshes[mesh], materials[material]);
        return model.get_ID().get_index();
    }
    void add_light(const HiprLight& l) {
        const uint32_t type = l.flags & HIPR_LIGHT_TYPE_MASK;
        if (type == HIPR_LIGHT_DIRECTIONAL) {
            SceneNode node = SceneNode("Directional light", Transform(Vector3f::zero(), Quaternionf::look_in(Vector3f(l.data[3], l.data[4], l.data[5]))));
            node.set_parent(root);
            LightSources::create_directional_light(node.get_ID(), RGB(l.data[0], l.data[1], l.data[2]));
        } else if (type == HIPR_LIGHT_SPHERE) {
            SceneNode node = SceneNode("Sphere light", Transform(Vector3f(l.data[3], l.data[4], l.data[5])));
            node.set_parent(root);
            LightSources::create_sphere_light(node.get_ID(), RGB(l.data[0], l.data[1], l.data[2]), l.data[6]);
        }
    }
    void set_environment_tint(RGB tint) { SceneRoots::set_environment_tint(scene_root, tint); }
};
} // namespace

AtriumCamera create_atrium_scene(CameraID camera_ID, SceneNode root_node, unsigned target_triangles, unsigned seed) {
    BifrostSceneSink sink;
    sink.root = root_node;
    sink.scene_root = Cameras::get_scene_ID(camera_ID);
    HIPRenderer::Scenes::build_atrium(sink, target_triangles, seed);
    Cameras::set_transform(camera_ID, sink.camera.transform);
    return {sink.camera.near_plane, sink.camera.far_plane, sink.camera.field_of_view, sink.camera.max_bounce_count};
}

} // namespace ViewerScenes
