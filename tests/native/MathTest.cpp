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
