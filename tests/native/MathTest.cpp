// MathTest.cpp -- the reference's tests of the math the host side mirrors (tests/BifrostTests/Math/{Transform,Quaternion,
// OctahedralNormal,RNG}Test.h), on host/Math.h and host/RNG.h. Transforms and quaternions place every instance and the camera,
// octahedral normals are how vertex normals travel to the device, the blue-noise points presample the environment.
#include "MiniTest.h"

#include "../../bifrost3d_amd/host/Math.h"
#include "../../bifrost3d_amd/host/RNG.h"

#include <algorithm>
#include <cmath>

using namespace Bifrost::Math;

namespace {

struct MathFixture {
    void SetUp() {}
    void TearDown() {}
    bool usable() const { return true; }
};

// almost_equal with a ulp budget, as the reference compares (BF/Math/Utils.h almost_equal)
bool almost_equal(float a, float b, int max_ulps) {
    if (a == b) return true;
    if (std::signbit(a) != std::signbit(b)) return std::fabs(a - b) <= 1e-6f;      // straddling zero
    int ia, ib;
    std::memcpy(&ia, &a, 4); std::memcpy(&ib, &b, 4);
    return std::abs(ia - ib) <= max_ulps || std::fabs(a - b) <= 1e-6f;
}
bool almost_equal(Vector3f a, Vector3f b, int max_ulps) { return almost_equal(a.x, b.x, max_ulps) && almost_equal(a.y, b.y, max_ulps) && almost_equal(a.z, b.z, max_ulps); }
bool almost_equal(Quaternionf a, Quaternionf b, int max_ulps) {
    return almost_equal(a.x, b.x, max_ulps) && almost_equal(a.y, b.y, max_ulps) && almost_equal(a.z, b.z, max_ulps) && almost_equal(a.w, b.w, max_ulps);
}
bool same_transform(Transform a, Transform b) { return almost_equal(a.translation, b.translation, 10) && almost_equal(a.rotation, b.rotation, 10) && almost_equal(a.scale, b.scale, 10); }
float degrees_to_radians(float d) { return d * PI<float>() / 180.0f; }

} // namespace

CPU_TEST_F(MathFixture, transform_applies_translation_rotation_and_scale) {       // TransformTest.h:43-75
    {
        Transform t = Transform(Vector3f(3, -4, 1));
        EXPECT_TRUE(same_transform(t.apply(invert(t)), Transform::identity()));
        EXPECT_TRUE(almost_equal(t.apply(Vector3f::zero()), t.translation, 10));
        EXPECT_TRUE(almost_equal(t.apply(t.translation * -1.0f), Vector3f::zero(), 10));
        EXPECT_TRUE(almost_equal(t.apply(Vector3f(7, 3, -1)), Vector3f(10, -1, 0), 10));
    }
    {
        Transform t = Transform(Vector3f::zero(), Quaternionf::from_angle_axis(degrees_to_radians(45.0f), Vector3f::up()));
        EXPECT_TRUE(almost_equal(t.apply(Vector3f::zero()), Vector3f::zero(), 10));
        EXPECT_TRUE(same_transform(t.apply(invert(t)), Transform::identity()));
        EXPECT_TRUE(almost_equal(t.apply(Vector3f::forward()), Vector3f(std::sqrt(0.5f), 0.0f, std::sqrt(0.5f)), 10));      // forward turned 45 degrees about up
    }
    {
        Transform t = Transform(Vector3f::zero(), Quaternionf::identity(), 0.5f);
        EXPECT_TRUE(same_transform(t.apply(invert(t)), Transform::identity()));
        EXPECT_TRUE(almost_equal(t.apply(Vector3f::zero()), Vector3f::zero(), 10));
        EXPECT_TRUE(almost_equal(t.apply(Vector3f::one()), Vector3f::one() * 0.5f, 10));
    }
}

CPU_TEST_F(MathFixture, transform_matrix_representation) {       // TransformTest.h:77-94
    Transform t = Transform(Vector3f(3, -4, 1), normalize(Quaternionf(4, 2, -7, 3)), 0.75f);
    const Matrix3x4f m3x4 = to_matrix3x4(t);
    const Matrix4x4f m4x4 = to_matrix4x4(t);
    for (Vector3f p : {Vector3f::forward(), Vector3f::right(), Vector3f::up()}) {
        const Vector3f by_quaternion = t * p;
        const float w = m4x4.m[3][0] * p.x + m4x4.m[3][1] * p.y + m4x4.m[3][2] * p.z + m4x4.m[3][3];
        const Vector3f by_4x4 = Vector3f(m4x4.m[0][0] * p.x + m4x4.m[0][1] * p.y + m4x4.m[0][2] * p.z + m4x4.m[0][3], m4x4.m[1][0] * p.x + m4x4.m[1][1] * p.y + m4x4.m[1][2] * p.z + m4x4.m[1][3],
                                         m4x4.m[2][0] * p.x + m4x4.m[2][1] * p.y + m4x4.m[2][2] * p.z + m4x4.m[2][3]) / w;
        const Vector3f by_3x4 = Vector3f(m3x4.m[0][0] * p.x + m3x4.m[0][1] * p.y + m3x4.m[0][2] * p.z + m3x4.m[0][3], m3x4.m[1][0] * p.x + m3x4.m[1][1] * p.y + m3x4.m[1][2] * p.z + m3x4.m[1][3],
                                         m3x4.m[2][0] * p.x + m3x4.m[2][1] * p.y + m3x4.m[2][2] * p.z + m3x4.m[2][3]);
        EXPECT_TRUE(almost_equal(by_quaternion, by_4x4, 10));
        EXPECT_TRUE(almost_equal(by_quaternion, by_3x4, 10));
    }
}

CPU_TEST_F(MathFixture, quaternion_axis_helpers_and_matrix_representation) {       // QuaternionTest.h:32-63
    Quaternionf quat = Quaternionf::from_angle_axis(degrees_to_radians(25.0f), Vector3f::up());
    EXPECT_TRUE(almost_equal(quat.forward(), quat * Vector3f::forward(), 10));
    EXPECT_TRUE(almost_equal(quat.up(), quat * Vector3f::up(), 10));
    EXPECT_TRUE(almost_equal(quat.right(), quat * Vector3f::right(), 10));

    for (int i = 0; i < 10; ++i) {
        const float angle = (i * 360.0f) / 10;
        Vector3f axis = i % 2 ? Vector3f::up() : Vector3f::up() * -1.0f;
        axis = axis + (i % 4 ? Vector3f::forward() : Vector3f::zero());
        axis = axis + (i % 8 ? Vector3f::right() : Vector3f::zero());
        const Quaternionf q0 = Quaternionf::from_angle_axis(degrees_to_radians(angle), normalize(axis));
        const Matrix3x3f m = to_matrix3x3(q0);
        const Quaternionf q1 = to_quaternion(m);
        for (Vector3f v : {Vector3f::forward(), Vector3f::right(), Vector3f::up()}) {
            const Vector3f by_matrix = Vector3f(m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z, m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z);
            EXPECT_TRUE(almost_equal(q0 * v, by_matrix, 20));
            EXPECT_TRUE(almost_equal(q1 * v, by_matrix, 30));
        }
    }
}

CPU_TEST_F(MathFixture, quaternion_look_in) {       // QuaternionTest.h:65-84 (in f32: the host mirror has no double quaternion)
    for (int y = -1; y < 2; ++y)
        for (int i = 0; i < 8; ++i) {
            const float phi = i * 2.0f * PI<float>() / 8.0f;
            const Vector3f direction = normalize(Vector3f(std::cos(phi), float(y), std::sin(phi)));
            const Quaternionf q = Quaternionf::look_in(direction, Vector3f::up());
            EXPECT_TRUE(dot(q.forward(), direction) > 0.99999f);      // looks along the direction
            EXPECT_TRUE(0.0f < q.up().y);                              // local up does not point down
            EXPECT_TRUE(std::fabs(q.right().y) < 0.0000005f);         // local right lies in the xz plane
        }
}

CPU_TEST_F(MathFixture, octahedral_normal_encode_decode) {       // OctahedralNormalTest.h:19-37
    const float max_error = 0.000047f;
    auto close = [&](Vector3f a, Vector3f b) { return std::fabs(a.x - b.x) < max_error && std::fabs(a.y - b.y) < max_error && std::fabs(a.z - b.z) < max_error; };
    int failures = 0;
    for (int x = -10; x < 11; ++x)
        for (int y = -10; y < 11; ++y)
            for (int z = -10; z < 11; ++z) {
                if (x == 0 && y == 0 && z == 0) continue;
                const Vector3f normal = normalize(Vector3f(float(x), float(y), float(z)));
                failures += close(normal, OctahedralNormal::encode_precise(normal).decode()) ? 0 : 1;
            }
    for (int s = 0; s < 10000; ++s) {
        const Vector2f u = RNG::sample02(unsigned(s));
        const float z = 1.0f - 2.0f * u.x, r = std::sqrt(std::fmax(0.0f, 1.0f - z * z)), phi = 2.0f * PI<float>() * u.y;      // Distributions::Sphere::sample_direction
        const Vector3f normal = normalize(Vector3f(r * std::cos(phi), r * std::sin(phi), z));
        failures += close(normal, OctahedralNormal::encode_precise(normal).decode()) ? 0 : 1;
    }
    EXPECT_EQ(0, failures);
}

CPU_TEST_F(MathFixture, blue_noise_points_fill_exactly_the_requested_range) {       // RNGTest.h:18-40
    const int sample_count = 16;
    Vector2f samples[sample_count + 4];
    const Vector2f sentinel = {1e10f, 1e20f};
    std::fill_n(samples, sample_count + 4, sentinel);
    RNG::fill_progressive_multijittered_bluenoise_samples(samples, samples + sample_count);
    for (int i = 0; i < sample_count; ++i) {
        EXPECT_TRUE(samples[i].x >= 0.0f && samples[i].y >= 0.0f);
        EXPECT_TRUE(samples[i].x < 1.0f && samples[i].y < 1.0f);
    }
    for (int i = sample_count; i < sample_count + 4; ++i) EXPECT_TRUE(samples[i].x == sentinel.x && samples[i].y == sentinel.y);
    // a count that is no power of two
    Vector2f odd[11];
    std::fill_n(odd, 11, sentinel);
    RNG::fill_progressive_multijittered_bluenoise_samples(odd, odd + 7);
    for (int i = 0; i < 7; ++i) EXPECT_TRUE(odd[i].x >= 0.0f && odd[i].x < 1.0f && odd[i].y >= 0.0f && odd[i].y < 1.0f);
    for (int i = 7; i < 11; ++i) EXPECT_TRUE(odd[i].x == sentinel.x);
}

// A transform that MIRRORS (negative uniform scale): the reference carries the object-space geometric normal through the inverse transpose
// (rtTransformNormal, OptiXRenderer/Shading/MonteCarlo.cu:147), so the front of the triangle stays its object-space front. The flattened world-space
// corners must wind accordingly: SceneBuilder gives such an instance its own index triples with two corners exchanged, and every consumer of the
// triple order (positions, normals, texture coordinates) follows.
#include "../../bifrost3d_amd/host/SceneBuilder.h"

CPU_TEST_F(MathFixture, a_mirroring_instance_keeps_its_front_side) {
    using namespace HIPRenderer;
    SceneBuilder scene;
    MeshData mesh;
    mesh.name = "triangle";
    mesh.positions = {Vector3f(0, 0, 0), Vector3f(1, 0, 0), Vector3f(0, 1, 0)};      // counter-clockwise seen from +z: geometric normal +z
    mesh.normals = {Vector3f(0, 0, 1), Vector3f(0, 0, 1), Vector3f(0, 0, 1)};
    mesh.texcoords = {Vector2f{0, 0}, Vector2f{1, 0}, Vector2f{0, 1}};
    mesh.primitives = {Vector3ui{0, 1, 2}};
    const uint32_t mesh_index = scene.add_mesh(mesh);
    HiprMaterial material = {};
    material.tint[0] = material.tint[1] = material.tint[2] = 0.5f;
    material.roughness = 0.5f; material.coverage = 1.0f;
    const uint32_t material_index = scene.add_material(material);
    const Quaternionf turned = Quaternionf::from_angle_axis(degrees_to_radians(40.0f), normalize(Vector3f(0.3f, 1.0f, -0.2f)));
    scene.add_model(mesh_index, material_index, Transform(Vector3f(0, 0, 0), turned, 2.0f));
    scene.add_model(mesh_index, material_index, Transform(Vector3f(5, 0, 0), turned, -2.0f));
    scene.add_model(mesh_index, material_index, Transform(Vector3f(-5, 0, 0), turned, -0.5f));      // a second mirrored instance shares the copy
    scene.finalize();
    const HiprSceneDesc& d = scene.desc();
    EXPECT_EQ(3u, d.triangle_count);
    EXPECT_EQ(6u, d.index_count);       // the mesh's triple and ONE mirrored copy
    EXPECT_EQ(d.instances[1].index_offset, d.instances[2].index_offset);
    EXPECT_TRUE(d.instances[0].index_offset != d.instances[1].index_offset);
    for (uint32_t t = 0; t < d.triangle_count; ++t) {
        const HiprTriangle& tri = d.triangles[t];
        const HiprInstance& inst = d.instances[tri.instance_index];
        const float* M = inst.object_to_world;
        const Vector3f p0(tri.v0[0], tri.v0[1], tri.v0[2]), p1(tri.v1[0], tri.v1[1], tri.v1[2]), p2(tri.v2[0], tri.v2[1], tri.v2[2]);
        const Vector3f world_normal = normalize(cross(p1 - p0, p2 - p0));
        // rtTransformNormal: the object-space normal (0, 0, 1) through the inverse transpose of M = s R, i.e. (1 / s) R n
        const float s = tri.instance_index == 0 ? 2.0f : (tri.instance_index == 1 ? -2.0f : -0.5f);
        const Vector3f expected = normalize(Vector3f(M[2], M[6], M[10]) / (s * s));      // M n / s^2 = R n / s
        EXPECT_TRUE(almost_equal(world_normal, expected, 64));
        // the corners are the instance's triple through the instance's matrix, in that order
        const uint32_t* idx = d.indices + 3 * size_t(inst.index_offset + tri.primitive_index);
        const HiprVertexGeometry& g1 = d.geometry[inst.vertex_offset + idx[1]];
        const Vector3f from_triple(M[0] * g1.position[0] + M[1] * g1.position[1] + M[2] * g1.position[2] + M[3], M[4] * g1.position[0] + M[5] * g1.position[1] + M[6] * g1.position[2] + M[7],
                                   M[8] * g1.position[0] + M[9] * g1.position[1] + M[10] * g1.position[2] + M[11]);
        EXPECT_TRUE(almost_equal(p1, from_triple, 4));
    }
    // a transform-only update that turns an instance inside out cannot be a refit: the triples change
    // -- and "false" means REBUILT: desc() carries the new pose of every model of the batch (the flipped one and the one updated before it), the flipped
    // instance points at the mirrored triples, and the world-space triangles agree with the instance matrices.
    EXPECT_TRUE(!scene.update_model_transforms({{3u, Transform(Vector3f(-5, 2, 0), turned, -0.5f)}, {1u, Transform(Vector3f(0, 1, 0), turned, -2.0f)}}));
    {
        const HiprSceneDesc& r = scene.desc();
        EXPECT_EQ(3u, r.triangle_count);
        EXPECT_EQ(r.instances[0].index_offset, r.instances[1].index_offset);      // model 1 now shares the mirrored copy
        EXPECT_FLOAT_EQ_EPS(1.0f, r.instances[0].object_to_world[7], 1e-6);                 // translation y of model 1
        EXPECT_FLOAT_EQ_EPS(2.0f, r.instances[2].object_to_world[7], 1e-6);                 // and of model 3, updated in the same batch
        for (uint32_t t = 0; t < r.triangle_count; ++t) {
            const HiprTriangle& tri = r.triangles[t];
            const HiprInstance& inst = r.instances[tri.instance_index];
            const float* M = inst.object_to_world;
            const uint32_t* idx = r.indices + 3 * size_t(inst.index_offset + tri.primitive_index);
            const float* corners[3] = {tri.v0, tri.v1, tri.v2};
            for (int k = 0; k < 3; ++k) {
                const HiprVertexGeometry& g = r.geometry[inst.vertex_offset + idx[k]];
                const Vector3f expected(M[0] * g.position[0] + M[1] * g.position[1] + M[2] * g.position[2] + M[3], M[4] * g.position[0] + M[5] * g.position[1] + M[6] * g.position[2] + M[7],
                                        M[8] * g.position[0] + M[9] * g.position[1] + M[10] * g.position[2] + M[11]);
                EXPECT_TRUE(almost_equal(Vector3f(corners[k][0], corners[k][1], corners[k][2]), expected, 4));
            }
            // the front side stays the object-space front side: geometric normal = R n / s
            const float s = tri.instance_index == 0 ? -2.0f : (tri.instance_index == 1 ? -2.0f : -0.5f);
            const Vector3f p0(tri.v0[0], tri.v0[1], tri.v0[2]), p1(tri.v1[0], tri.v1[1], tri.v1[2]), p2(tri.v2[0], tri.v2[1], tri.v2[2]);
            EXPECT_TRUE(almost_equal(normalize(cross(p1 - p0, p2 - p0)), normalize(Vector3f(M[2], M[6], M[10]) / (s * s)), 64));
        }
    }
    EXPECT_TRUE(scene.update_model_transforms({{2u, Transform(Vector3f(5, 1, 0), turned, -3.0f)}}));
}

// csrc/fast_divide.h: the multiply-high division the kernels use to turn a path slot into (pixel, sample) and a tile into (column, row) must be exact for
// every 32-bit dividend -- the slot of a path decides its sample sequence. Checked against the operator on the divisors a frame can bring (samples per pass,
// tiles per row), on the awkward ones (2^k - 1, 2^k, 2^k + 1, the largest), at multiples of the divisor and their neighbours, and on random pairs.
#include "../../bifrost3d_amd/csrc/fast_divide.h"

CPU_TEST_F(MathFixture, multiply_high_division_is_exact) {
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { state = state * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(state >> 32); };
    unsigned long long mismatches = 0;
    auto check = [&](uint32_t n, uint32_t d, const hipr::Divisor& divisor) { if (hipr::divide(n, divisor) != n / d) ++mismatches; };
    const uint32_t divisors[] = {1, 2, 3, 5, 6, 7, 9, 10, 12, 15, 17, 24, 31, 32, 33, 60, 63, 64, 100, 120, 135, 160, 240, 255, 256, 257, 480, 641, 1000, 4096, 65535, 65536, 65537,
                                 1000003, 0x7FFFFFFFu, 0x80000000u, 0x80000001u, 0xFFFFFFFEu, 0xFFFFFFFFu};
    for (uint32_t d : divisors) {
        const hipr::Divisor divisor = hipr::make_divisor(d);
        for (int i = 0; i < 20000; ++i) check(next(), d, divisor);
        for (uint32_t n : {0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x80000000u, 0x7FFFFFFFu}) check(n, d, divisor);
        for (uint64_t q = 1; q < 300 && q * d <= 0xFFFFFFFFull; ++q) { check(uint32_t(q * d), d, divisor); check(uint32_t(q * d - 1u), d, divisor); }
    }
    for (int i = 0; i < 20000; ++i) {
        uint32_t d = next() >> (next() % 32u);
        if (d == 0) d = 1;
        const hipr::Divisor divisor = hipr::make_divisor(d);
        for (int j = 0; j < 8; ++j) check(next(), d, divisor);
        check(0xFFFFFFFFu, d, divisor);
    }
    EXPECT_EQ(0ull, mismatches);
}
