// host/AtriumScene.h -- the procedural atrium (Scenes::create_atrium) as a template over the scene builder it feeds.
#pragma once

#include "SceneBuilder.h"

#include <cmath>

namespace HIPRenderer {
namespace Scenes {

// ---------------------------------------------------------------------------------------------
// Procedural atrium: a triangle-count-controlled stand-in for Sponza-class scenes.
// ---------------------------------------------------------------------------------------------
namespace atrium_detail {
struct Lcg {
    uint32_t s;
    float next() { s = 1664525u * s + 1013904223u; return float(s >> 8) * (1.0f / 16777216.0f); }
};

// Parametric grid surface: f(u, v) -> position; smooth normals from finite differences.
template <typename F>
MeshData grid_surface(unsigned nu, unsigned nv, F f, bool texcoords) {
    MeshData m;
    for (unsigned j = 0; j <= nv; ++j)
        for (unsigned i = 0; i <= nu; ++i) {
            float u = float(i) / nu, v = float(j) / nv;
            const float e = 1e-3f;
            Vector3f p = f(u, v);
            Vector3f du = f(u + e, v) - f(u - e, v), dv = f(u, v + e) - f(u, v - e);
            Vector3f n = cross(dv, du);
            float len = magnitude(n);
            m.positions.push_back(p);
            m.normals.push_back(len > 0 ? n / len : Vector3f(0, 1, 0));
            if (texcoords) m.texcoords.push_back({u * 4.0f, v * 4.0f});
        }
    for (unsigned j = 0; j < nv; ++j)
        for (unsigned i = 0; i < nu; ++i) {
            unsigned a = i + j * (nu + 1), b = a + 1, c = a + nu + 1, d = c + 1;
            m.primitives.push_back({a, c, b});
            m.primitives.push_back({b, c, d});
        }
    return m;
}
} // namespace atrium_detail

// `Builder`: SceneBuilder (flat scene, directly) or ViewerScenes::BifrostSceneSink (the same calls recorded as Bifrost meshes, materials,
// models and lights, i.e. the route a host application takes through HIPRenderer::Renderer).
// `textured` (builders that take textures: SceneBuilder): every material gets one of eight tileable tint / roughness textures and the cloth banners become cut-outs with
// a lace pattern -- what the real Sponza brings (a texture on every material, alpha-masked plants and chains) and the plain stand-in does not. With it the scene is
// rendered by the FULL kernels: texture samplers in the shade kernel, coverage lookups for shadow rays in the traversal.
template <typename Builder>
void build_atrium(Builder& sb, unsigned target_triangles, unsigned seed, bool textured = false) {
    using namespace atrium_detail;
    Lcg rng = {seed * 747796405u + 2891336453u};
    uint32_t tint_textures[8] = {}, lace_texture = 0;
    if constexpr (Builder::takes_textures) if (textured) {
        Lcg texture_rng = {seed * 2654435761u + 12345u};
        for (uint32_t t = 0; t < 8; ++t) {       // value noise over an 8 x 8 lattice, wrapped: tint scale 0.55 .. 1 per channel, roughness scale 0.7 .. 1 in alpha
            const uint32_t size = 128, cells = 8;
            float lattice[4][cells * cells];
            for (auto& channel : lattice) for (float& v : channel) v = texture_rng.next();
            ImageData image;
            image.width = image.height = size; image.format = HIPR_TEXEL_RGBA8; image.is_sRGB = false;
            image.pixels.resize(size_t(size) * size * 4);
            for (uint32_t y = 0; y < size; ++y)
                for (uint32_t x = 0; x < size; ++x) {
                    const float fx = float(x) * cells / size, fy = float(y) * cells / size;
                    const uint32_t x0 = uint32_t(fx) % cells, y0 = uint32_t(fy) % cells, x1 = (x0 + 1) % cells, y1 = (y0 + 1) % cells;
                    const float tx = fx - std::floor(fx), ty = fy - std::floor(fy);
                    for (int c = 0; c < 4; ++c) {
                        const float* l = lattice[c];
                        const float v = (l[x0 + y0 * cells] * (1 - tx) + l[x1 + y0 * cells] * tx) * (1 - ty) + (l[x0 + y1 * cells] * (1 - tx) + l[x1 + y1 * cells] * tx) * ty;
                        const float lo = c == 3 ? 0.7f : 0.55f;
                        image.pixels[(size_t(y) * size + x) * 4 + size_t(c)] = uint8_t(255.0f * (lo + (1.0f - lo) * v) + 0.5f);
                    }
                }
            tint_textures[t] = sb.add_texture(image, true, true, true, true);
        }
        ImageData lace;                           // 64 x 64 coverage: a lattice of round holes, a quarter of the area open
        lace.width = lace.height = 64; lace.format = HIPR_TEXEL_R8; lace.is_sRGB = false;
        lace.pixels.resize(64 * 64);
        for (uint32_t y = 0; y < 64; ++y)
            for (uint32_t x = 0; x < 64; ++x) {
                const float dx = float(x % 16) - 7.5f, dy = float(y % 16) - 7.5f;
                lace.pixels[y * 64 + x] = dx * dx + dy * dy < 4.6f * 4.6f ? 0 : 255;
            }
        lace_texture = sb.add_texture(lace, true, true, false, false);
    }
    // 25 metal-rough materials: a spread of dielectrics, metals and coated plastics.
    std::vector<uint32_t> materials;
    for (int i = 0; i < 25; ++i) {
        RGB tint(0.25f + 0.7f * rng.next(), 0.25f + 0.7f * rng.next(), 0.25f + 0.7f * rng.next());
        float roughness = 0.08f + 0.9f * rng.next();
        bool metal = (i % 5) == 0;
        HiprMaterial m = SceneBuilder::make_material(tint, roughness, metal ? 1.0f : 0.04f, metal ? 1.0f : 0.0f);
        if ((i % 7) == 3) { m.coat = SceneBuilder::unorm16(1.0f); m.coat_roughness = SceneBuilder::unorm16(0.1f + 0.5f * rng.next()); }
        m.tint_roughness_texture_ID = int32_t(tint_textures[i % 8]);      // 0: none
        materials.push_back(sb.add_material(m));
    }
    auto material = [&](int i) { return materials[size_t(i) % materials.size()]; };

    // Triangle budget: floor 2%, walls 8%, columns 40%, arches 20%, cloth 30%.
    const float budget = float(target_triangles);
    const float W = 30.0f, D = 14.0f, H = 10.0f;
    auto dims = [](float triangles, float aspect, unsigned& nu, unsigned& nv) {
        float quads = triangles * 0.5f;
        nv = unsigned(std::fmax(2.0f, std::sqrt(quads / aspect)));
        nu = unsigned(std::fmax(2.0f, nv * aspect));
    };
    unsigned nu, nv;

    dims(budget * 0.02f, W / D, nu, nv);
    sb.add_model(sb.add_mesh(grid_surface(nu, nv, [&](float u, float v) {
        return Vector3f((u - 0.5f) * W, 0.02f * std::sin(u * 40.0f) * std::sin(v * 20.0f), (v - 0.5f) * D); }, true)), material(1), Transform::identity());

    // two long walls + two end walls, slightly bumpy so the BVH cannot trivially cull them
    const float wall_share = budget * 0.08f / 4.0f;
    for (int side = 0; side < 2; ++side) {
        const float z = side ? 0.5f * D : -0.5f * D;
        dims(wall_share, W / H, nu, nv);
        sb.add_model(sb.add_mesh(grid_surface(nu, nv, [&](float u, float v) {
            return Vector3f((side ? (0.5f - u) : (u - 0.5f)) * W, v * H, z + (side ? -1.f : 1.f) * 0.05f * std::sin(u * 60.0f + v * 30.0f)); }, true)),
            material(2 + side), Transform::identity());
        const float x = side ? 0.5f * W : -0.5f * W;
        dims(wall_share, D / H, nu, nv);
        sb.add_model(sb.add_mesh(grid_surface(nu, nv, [&](float u, float v) {
            return Vector3f(x + (side ? -1.f : 1.f) * 0.05f * std::sin(u * 30.0f + v * 30.0f), v * H, (side ? (u - 0.5f) : (0.5f - u)) * D); }, true)),
            material(4 + side), Transform::identity());
    }

    // colonnades: 2 rows x 12 fluted columns, one shared mesh instanced 24 times (flattened by the builder)
    const int columns_per_row = 12;
    dims(budget * 0.40f / (2 * columns_per_row), 0.35f, nu, nv);
    const uint32_t column_mesh = sb.add_mesh(grid_surface(nu, nv, [&](float u, float v) {
        float a = u * 2.0f * PI<float>();
        float r = 0.45f * (1.0f + 0.06f * std::cos(a * 16.0f)) * (1.0f - 0.15f * v) + (v < 0.06f || v > 0.94f ? 0.12f : 0.0f);
        return Vector3f(r * std::cos(a), v * 6.5f, r * std::sin(a)); }, true));
    for (int row = 0; row < 2; ++row)
        for (int c = 0; c < columns_per_row; ++c) {
            float x = (float(c) + 0.5f) / columns_per_row * (W - 4.0f) - 0.5f * (W - 4.0f);
            float z = row ? 3.2f : -3.2f;
            Transform t(Vector3f(x, 0.0f, z), Quaternionf::from_angle_axis(rng.next() * 6.28f, Vector3f::up()), 0.9f + 0.2f * rng.next());
            sb.add_model(column_mesh, material(6 + (c + row * 5) % 9), t);
        }

    // arches spanning the nave between the two rows
    const int arch_count = 11;
    dims(budget * 0.20f / arch_count, 6.0f, nu, nv);
    const uint32_t arch_mesh = sb.add_mesh(grid_surface(nu, nv, [&](float u, float v) {
        float a = u * PI<float>();
        float tube = v * 2.0f * PI<float>();
        float R = 3.2f, r = 0.28f * (1.0f + 0.1f * std::cos(tube * 6.0f));
        return Vector3f(r * std::sin(tube), 6.3f + (R + r * std::cos(tube)) * std::sin(a) * 0.8f, (R + r * std::cos(tube)) * std::cos(a)); }, true));
    for (int a = 0; a < arch_count; ++a) {
        float x = (float(a) + 1.0f) / (arch_count + 1) * (W - 4.0f) - 0.5f * (W - 4.0f);
        sb.add_model(arch_mesh, material(15 + a % 6), Transform(Vector3f(x, 0.0f, 0.0f)));
    }

    // draped cloth banners: fine, wavy, thin walled grids (long thin triangles near the folds)
    const int cloth_count = 6;
    dims(budget * 0.30f / cloth_count, 0.6f, nu, nv);
    for (int cidx = 0; cidx < cloth_count; ++cidx) {
        float phase = rng.next() * 6.28f, folds = 9.0f + 6.0f * rng.next();
        uint32_t mesh = sb.add_mesh(grid_surface(nu, nv, [&](float u, float v) {
            float sag = 0.6f * std::sin(v * PI<float>());
            return Vector3f((u - 0.5f) * 2.4f, 8.5f - v * 5.0f - 0.15f * sag, 0.25f * std::sin(u * folds + phase) * (0.3f + v) + sag * 0.4f); }, true));
        HiprMaterial m = SceneBuilder::make_material(RGB(0.3f + 0.6f * rng.next(), 0.15f + 0.3f * rng.next(), 0.15f + 0.5f * rng.next()), 0.85f, 0.04f, 0.0f, HIPR_MATERIAL_THIN_WALLED);
        m.tint_roughness_texture_ID = int32_t(tint_textures[(cidx + 3) % 8]);
        if (lace_texture) { m.flags |= HIPR_MATERIAL_CUTOUT; m.coverage = 0.5f; m.coverage_texture_ID = int32_t(lace_texture); }      // cut-out threshold 0.5 (upload_material, OR/Renderer.cpp:754-812)
        float x = (float(cidx) + 0.5f) / cloth_count * (W - 8.0f) - 0.5f * (W - 8.0f);
        sb.add_model(mesh, sb.add_material(m), Transform(Vector3f(x, 0.0f, (cidx & 1) ? 1.2f : -1.2f), Quaternionf::from_angle_axis(0.5f * PI<float>(), Vector3f::up())));
    }

    // Loader defaults of SimpleViewer for files without lights (main.cpp:419-426) and its camera rule (:411-416).
    Quaternionf light_rotation = Quaternionf::look_in(normalize(Vector3f(-0.1f, -10.0f, -0.1f)));
    sb.add_light(SceneBuilder::directional_light(light_rotation.forward(), RGB(15.0f)));
    sb.add_light(SceneBuilder::sphere_light(Vector3f(0.0f, 5.0f, 0.0f), RGB(400.0f, 360.0f, 300.0f), 0.4f));
    sb.set_environment_tint(RGB(0.68f, 0.92f, 1.0f));
    AABB bounds = {Vector3f(-0.5f * W, 0.0f, -0.5f * D), Vector3f(0.5f * W, H, 0.5f * D)};
    sb.camera.transform = Transform(Vector3f(-0.42f * W, 2.2f, -0.5f));
    sb.camera.transform.look_at(Vector3f(0.3f * W, 4.0f, 0.6f));
    float scene_size = magnitude(bounds.size());
    sb.camera.near_plane = scene_size / 10000.0f;
    sb.camera.far_plane = scene_size * 3.0f;
    sb.camera.max_bounce_count = 4;   // loaded files, main.cpp:380
}


} // namespace Scenes
} // namespace HIPRenderer
