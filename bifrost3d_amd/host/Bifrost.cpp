// host/Bifrost.cpp -- out-of-line parts of the Bifrost mirror (mesh creation, camera projection utilities).
#include "Bifrost.h"
#include "Scenes.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace Bifrost {

namespace Assets {
namespace MeshUtils {

Mesh deep_clone(Mesh mesh) {
    MeshFlags buffers = MeshFlag::Position;
    if (mesh.get_normals()) buffers |= MeshFlag::Normal;
    if (mesh.get_texcoords()) buffers |= MeshFlag::Texcoord;
    if (mesh.get_tint_and_roughness()) buffers |= MeshFlag::TintAndRoughness;
    if (mesh.get_emission()) buffers |= MeshFlag::Emissive;
    const unsigned primitives = mesh.get_primitive_count(), vertices = mesh.get_vertex_count();
    Mesh clone(mesh.get_name() + "_clone", primitives, vertices, buffers);
    std::copy_n(mesh.get_primitives(), primitives, clone.get_primitives());
    std::copy_n(mesh.get_positions(), vertices, clone.get_positions());
    if (mesh.get_normals()) std::copy_n(mesh.get_normals(), vertices, clone.get_normals());
    if (mesh.get_texcoords()) std::copy_n(mesh.get_texcoords(), vertices, clone.get_texcoords());
    if (mesh.get_tint_and_roughness()) std::copy_n(mesh.get_tint_and_roughness(), vertices, clone.get_tint_and_roughness());
    if (mesh.get_emission()) std::copy_n(mesh.get_emission(), vertices, clone.get_emission());
    clone.set_bounds(mesh.get_bounds());
    return clone;
}

void transform_mesh(Mesh mesh, Matrix3x4f affine) {
    const float (&a)[3][4] = affine.m;
    AABB bounds = {Vector3f(std::numeric_limits<float>::infinity()), Vector3f(-std::numeric_limits<float>::infinity())};
    Vector3f* positions = mesh.get_positions();
    for (unsigned v = 0; v < mesh.get_vertex_count(); ++v) {
        const Vector3f p = positions[v];
        positions[v] = Vector3f(a[0][0] * p.x + a[0][1] * p.y + a[0][2] * p.z + a[0][3],
                                a[1][0] * p.x + a[1][1] * p.y + a[1][2] * p.z + a[1][3],
                                a[2][0] * p.x + a[2][1] * p.y + a[2][2] * p.z + a[2][3]);
        bounds.grow_to_contain(positions[v]);
    }
    mesh.set_bounds(bounds);

    if (Vector3f* normals = mesh.get_normals()) {
        // transpose(invert(L)) is the cofactor matrix over the determinant.
        const float c[3][3] = {{a[1][1] * a[2][2] - a[1][2] * a[2][1], a[1][2] * a[2][0] - a[1][0] * a[2][2], a[1][0] * a[2][1] - a[1][1] * a[2][0]},
                               {a[0][2] * a[2][1] - a[0][1] * a[2][2], a[0][0] * a[2][2] - a[0][2] * a[2][0], a[0][1] * a[2][0] - a[0][0] * a[2][1]},
                               {a[0][1] * a[1][2] - a[0][2] * a[1][1], a[0][2] * a[1][0] - a[0][0] * a[1][2], a[0][0] * a[1][1] - a[0][1] * a[1][0]}};
        const float inv_det = 1.0f / (a[0][0] * c[0][0] + a[0][1] * c[0][1] + a[0][2] * c[0][2]);
        for (unsigned v = 0; v < mesh.get_vertex_count(); ++v) {
            const Vector3f n = normals[v];
            normals[v] = normalize(Vector3f((c[0][0] * n.x + c[0][1] * n.y + c[0][2] * n.z) * inv_det,
                                            (c[1][0] * n.x + c[1][1] * n.y + c[1][2] * n.z) * inv_det,
                                            (c[2][0] * n.x + c[2][1] * n.y + c[2][2] * n.z) * inv_det));
        }
    }
}

} // namespace MeshUtils

namespace MeshCreation {

static Mesh from_data(const std::string& name, const HIPRenderer::MeshData& d, MeshFlags buffers, AABB bounds) {
    Mesh mesh(name, unsigned(d.primitives.size()), unsigned(d.positions.size()), buffers);
    std::memcpy(mesh.get_primitives(), d.primitives.data(), d.primitives.size() * sizeof(Vector3ui));
    std::memcpy(mesh.get_positions(), d.positions.data(), d.positions.size() * sizeof(Vector3f));
    if (mesh.get_normals() && !d.normals.empty()) std::memcpy(mesh.get_normals(), d.normals.data(), d.normals.size() * sizeof(Vector3f));
    if (mesh.get_texcoords() && !d.texcoords.empty()) std::memcpy(mesh.get_texcoords(), d.texcoords.data(), d.texcoords.size() * sizeof(Vector2f));
    // default_initialize_shading (BF/Assets/MeshCreation.cpp:20-28): white tint, emission 2 * red
    if (mesh.get_emission())
        for (unsigned i = 0; i < mesh.get_vertex_count(); ++i) mesh.get_emission()[i] = Vector3f(2.0f, 0.0f, 0.0f);
    mesh.set_bounds(bounds);
    return mesh;
}

Mesh plane(unsigned quads_per_edge, MeshFlags buffers) {
    if (quads_per_edge == 0) return Mesh();
    auto data = HIPRenderer::Scenes::plane(quads_per_edge, buffers.is_set(MeshFlag::Normal), buffers.is_set(MeshFlag::Texcoord));
    return from_data("Plane", data, buffers, AABB{Vector3f(-0.5f, 0.0f, -0.5f), Vector3f(0.5f, 0.0f, 0.5f)});
}

Mesh box(unsigned quads_per_edge, Vector3f size, MeshFlags buffers) {
    if (quads_per_edge == 0) return Mesh();
    auto data = HIPRenderer::Scenes::box(quads_per_edge, size, false, buffers.is_set(MeshFlag::Texcoord));
    return from_data("Box", data, buffers, AABB{size * -0.5f, size * 0.5f});
}

} // namespace MeshCreation
} // namespace Assets

namespace Scene {
namespace CameraUtils {

void compute_perspective_projection(float near_distance, float far_distance, float field_of_view_in_radians, float aspect_ratio, Matrix4x4f& projection,
                                    Matrix4x4f& inverse_projection) {
    // BF/Scene/Camera.cpp:237-266
    float f = 1.0f / std::tan(field_of_view_in_radians * 0.5f);
    float a = (far_distance + near_distance) / (near_distance - far_distance);
    float b = (2.0f * far_distance * near_distance) / (near_distance - far_distance);
    projection = {};
    projection[0][0] = f / aspect_ratio;
    projection[1][1] = f;
    projection[2][2] = -a;
    projection[2][3] = b;
    projection[3][2] = 1.0f;
    inverse_projection = {};
    inverse_projection[0][0] = 1.0f / projection[0][0];
    inverse_projection[1][1] = 1.0f / projection[1][1];
    inverse_projection[2][3] = 1.0f;
    inverse_projection[3][2] = 1.0f / projection[2][3];
    inverse_projection[3][3] = -projection[2][2] / projection[2][3];
}

void compute_orthographic_projection(float width, float height, float depth, Matrix4x4f& projection, Matrix4x4f& inverse_projection) {
    // BF/Scene/Camera.cpp:268-286
    projection = {};
    projection[0][0] = 2 / width;
    projection[1][1] = 2 / height;
    projection[2][2] = 2 / depth;
    projection[2][3] = -1;
    projection[3][3] = 1;
    inverse_projection = {};
    inverse_projection[0][0] = 0.5f * width;
    inverse_projection[1][1] = 0.5f * height;
    inverse_projection[2][2] = 0.5f * depth;
    inverse_projection[2][3] = 0.5f * depth;
    inverse_projection[3][3] = 1.0f;
}

} // namespace CameraUtils
} // namespace Scene

} // namespace Bifrost
