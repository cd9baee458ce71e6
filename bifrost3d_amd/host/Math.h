// host/Math.h -- the slice of Bifrost::Math the renderer's host side consumes.
//
// Mirrors the names and conventions of core/Bifrost/Bifrost/Math (+Z forward, +Y up, +X right,
// Vector.h:82-84; row-major matrices, Matrix.h; Transform = rotation quaternion + translation +
// uniform scale, Transform.h:28-98) so host code written against Bifrost reads the same here.
// Only what OptiXRenderer/Renderer.cpp and the SimpleViewer scenes touch is provided.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace Bifrost {
namespace Math {

template <typename T> constexpr T PI() { return T(3.14159265358979323846); }

struct Vector2f { float x, y; };
struct Vector2i { int x, y; Vector2i() = default; constexpr Vector2i(int x, int y) : x(x), y(y) {} };
struct Vector2s { short x, y; };
struct Vector3ui { unsigned int x, y, z; };

const float nearly_one = 0xffffff / float(1 << 24);   // Constants.h:19

struct RGBA { float r, g, b, a; };

struct Vector3f {
    float x, y, z;
    Vector3f() = default;
    constexpr Vector3f(float x, float y, float z) : x(x), y(y), z(z) {}
    explicit constexpr Vector3f(float v) : x(v), y(v), z(v) {}
    static constexpr Vector3f zero() { return Vector3f(0, 0, 0); }
    static constexpr Vector3f one() { return Vector3f(1, 1, 1); }
    static constexpr Vector3f forward() { return Vector3f(0, 0, 1); }
    static constexpr Vector3f up() { return Vector3f(0, 1, 0); }
    static constexpr Vector3f right() { return Vector3f(1, 0, 0); }
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline Vector3f operator+(Vector3f a, Vector3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3f operator-(Vector3f a, Vector3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector3f operator-(Vector3f a) { return {-a.x, -a.y, -a.z}; }
inline Vector3f operator*(Vector3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector3f operator*(float s, Vector3f a) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector3f operator*(Vector3f a, Vector3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vector3f operator/(Vector3f a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(Vector3f a, Vector3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector3f cross(Vector3f a, Vector3f b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float magnitude_squared(Vector3f v) { return dot(v, v); }
inline float magnitude(Vector3f v) { return std::sqrt(dot(v, v)); }
inline Vector3f normalize(Vector3f v) { float m = magnitude(v); return v / m; }

// Latitude-longitude environment map parameterisation (BF/Math/Utils.h:90-101).
inline Vector2f direction_to_latlong_texcoord(Vector3f direction) {
    const float u = (std::atan2(direction.z, direction.x) + PI<float>()) * 0.5f / PI<float>();
    const float v = (std::asin(direction.y) + PI<float>() * 0.5f) / PI<float>();
    return {u, v};
}
inline Vector3f latlong_texcoord_to_direction(Vector2f uv) {
    const float phi = uv.x * 2.0f * PI<float>(), theta = uv.y * PI<float>();
    const float sin_theta = std::sin(theta);
    return -Vector3f(sin_theta * std::cos(phi), std::cos(theta), sin_theta * std::sin(phi));
}

struct RGB {
    float r, g, b;
    RGB() = default;
    constexpr RGB(float r, float g, float b) : r(r), g(g), b(b) {}
    explicit constexpr RGB(float v) : r(v), g(v), b(v) {}
    static constexpr RGB black() { return RGB(0, 0, 0); }
    static constexpr RGB white() { return RGB(1, 1, 1); }
};

struct Quaternionf {
    float x, y, z, w;
    Quaternionf() = default;
    constexpr Quaternionf(float x, float y, float z, float w) : x(x), y(y), z(z), w(w) {}
    Quaternionf(Vector3f imaginary, float real) : x(imaginary.x), y(imaginary.y), z(imaginary.z), w(real) {}
    static constexpr Quaternionf identity() { return Quaternionf(0, 0, 0, 1); }
    Vector3f imaginary() const { return {x, y, z}; }

    // Quaternion.h:66-72
    static Quaternionf from_angle_axis(float angle_in_radians, Vector3f axis) {
        float half = angle_in_radians * 0.5f;
        return Quaternionf(axis * std::sin(half), std::cos(half));
    }

    // Quaternion.h:76-112: rotation whose forward is `direction`.
    static Quaternionf look_in(Vector3f direction, Vector3f up = Vector3f::up()) {
        Vector3f right = normalize(cross(up, direction));
        up = cross(direction, right);
        float trace = right.x + up.y + direction.z;
        if (trace > 0.0f) {
            float s = std::sqrt(trace + 1.0f);
            float real = s * 0.5f;
            s = 0.5f / s;
            return Quaternionf(Vector3f(up.z - direction.y, direction.x - right.z, right.y - up.x) * s, real);
        }
        Vector3f m[3] = {right, up, direction};
        const int next[3] = {1, 2, 0};
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        int j = next[i], k = next[j];
        float s = std::sqrt((m[i][i] - (m[j][j] + m[k][k])) + 1.0f);
        Vector3f imaginary;
        imaginary[i] = s * 0.5f;
        if (s != 0.0f) s = 0.5f / s;
        float real = (m[j][k] - m[k][j]) * s;
        imaginary[j] = (m[i][j] + m[j][i]) * s;
        imaginary[k] = (m[i][k] + m[k][i]) * s;
        return Quaternionf(imaginary, real);
    }

    Quaternionf operator*(Quaternionf rhs) const {
        float real = w * rhs.w - dot(imaginary(), rhs.imaginary());
        Vector3f img = cross(imaginary(), rhs.imaginary()) + rhs.imaginary() * w + imaginary() * rhs.w;
        return Quaternionf(img, real);
    }
    // Rotate a vector, Quaternion.h:150-157
    Vector3f operator*(Vector3f rhs) const {
        Vector3f img = imaginary();
        Vector3f uv = cross(img, rhs);
        Vector3f uuv = cross(img, uv);
        Vector3f half_res = (uv * w) + uuv;
        return rhs + half_res * 2.0f;
    }
    Vector3f forward() const { return *this * Vector3f::forward(); }
    Vector3f up() const { return *this * Vector3f::up(); }
    Vector3f right() const { return *this * Vector3f::right(); }
};
inline Quaternionf conjugate(Quaternionf q) { return Quaternionf(-q.x, -q.y, -q.z, q.w); }
inline Quaternionf inverse_unit(Quaternionf q) { return conjugate(q); }
inline float magnitude(Quaternionf q) { return std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); }
inline Quaternionf normalize(Quaternionf q) {
    float m = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Quaternionf(q.x / m, q.y / m, q.z / m, q.w / m);
}

struct Transform {
    Quaternionf rotation;
    Vector3f translation;
    float scale;
    Transform() = default;
    Transform(Vector3f translation, Quaternionf rotation = Quaternionf::identity(), float scale = 1.0f)
        : rotation(rotation), translation(translation), scale(scale) {}
    static Transform identity() { return Transform(Vector3f::zero()); }
    Vector3f apply(Vector3f v) const { return translation + rotation * v * scale; }
    Vector3f operator*(Vector3f v) const { return apply(v); }
    Transform apply(Transform t) const { return Transform(apply(t.translation), normalize(rotation * t.rotation), scale * t.scale); }
    Transform operator*(Transform t) const { return apply(t); }
    void look_at(Vector3f target, Vector3f up = Vector3f::up()) { rotation = Quaternionf::look_in(normalize(target - translation), up); }
    Transform inverse() const {
        float s = 1.0f / scale;
        Quaternionf r = inverse_unit(rotation);
        Vector3f t = (r * translation) * -s;
        return Transform(t, r, s);
    }
    bool operator==(const Transform& rhs) const { return std::memcmp(this, &rhs, sizeof(rhs)) == 0; }
    bool operator!=(const Transform& rhs) const { return !(*this == rhs); }
};
inline Transform invert(Transform t) { return t.inverse(); }

struct Matrix3x3f { float m[3][3]; float* begin() { return &m[0][0]; } const float* begin() const { return &m[0][0]; } float* operator[](int r) { return m[r]; } const float* operator[](int r) const { return m[r]; } };
struct Matrix3x4f { float m[3][4]; float* begin() { return &m[0][0]; } const float* begin() const { return &m[0][0]; } };
struct Matrix4x4f {
    float m[4][4];
    float* begin() { return &m[0][0]; }
    const float* begin() const { return &m[0][0]; }
    float* operator[](int r) { return m[r]; }
    const float* operator[](int r) const { return m[r]; }
    static Matrix4x4f identity() { Matrix4x4f r = {}; r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f; return r; }
    bool operator==(const Matrix4x4f& rhs) const { return std::memcmp(m, rhs.m, sizeof(m)) == 0; }
    bool operator!=(const Matrix4x4f& rhs) const { return !(*this == rhs); }
};
inline Matrix4x4f operator*(const Matrix4x4f& a, const Matrix4x4f& b) {
    Matrix4x4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

// Conversions.h:22-84
inline Matrix3x3f to_matrix3x3(Quaternionf q) {
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    return {{{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - w * z), 2.0f * (x * z + w * y)},
             {2.0f * (x * y + w * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (z * y - w * x)},
             {2.0f * (x * z - w * y), 2.0f * (z * y + w * x), 1.0f - 2.0f * (x * x + y * y)}}};
}
// Conversions.h:36-70: rotation matrix to quaternion (x, y, z, w), branch on the trace and otherwise on the largest diagonal
// element. Evaluated in T: the glTF loader decomposes its matrices in double and rounds after normalising.
template <typename T>
inline void to_quaternion(const T (&m)[3][3], T (&q)[4]) {
    const T trace = m[0][0] + m[1][1] + m[2][2];
    if (trace > T(0)) {
        T s = std::sqrt(trace + T(1));
        q[3] = s * T(0.5);
        s = T(0.5) / s;
        q[0] = (m[2][1] - m[1][2]) * s; q[1] = (m[0][2] - m[2][0]) * s; q[2] = (m[1][0] - m[0][1]) * s;
        return;
    }
    const int next[3] = {1, 2, 0};
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = next[i], k = next[j];
    T s = std::sqrt((m[i][i] - (m[j][j] + m[k][k])) + T(1));
    q[i] = s * T(0.5);
    s = T(0.5) / s;
    q[3] = (m[k][j] - m[j][k]) * s;
    q[j] = (m[j][i] + m[i][j]) * s;
    q[k] = (m[k][i] + m[i][k]) * s;
}
inline Quaternionf to_quaternion(const Matrix3x3f& m) { float q[4]; to_quaternion(m.m, q); return Quaternionf(q[0], q[1], q[2], q[3]); }
inline Matrix3x4f to_matrix3x4(Transform t) {
    const Matrix3x3f r = to_matrix3x3(t.rotation);
    const float s = t.scale;
    return {{{s * r[0][0], s * r[0][1], s * r[0][2], t.translation.x},
             {s * r[1][0], s * r[1][1], s * r[1][2], t.translation.y},
             {s * r[2][0], s * r[2][1], s * r[2][2], t.translation.z}}};
}
inline Matrix4x4f to_matrix4x4(Transform t) {
    const Matrix3x4f a = to_matrix3x4(t);
    Matrix4x4f r = {};
    std::memcpy(r.m, a.m, sizeof(a.m));
    r.m[3][3] = 1.0f;
    return r;
}

struct AABB {
    Vector3f minimum, maximum;
    static AABB invalid() { return {Vector3f(1e30f), Vector3f(-1e30f)}; }
    Vector3f center() const { return (minimum + maximum) * 0.5f; }
    Vector3f size() const { return maximum - minimum; }
    void grow_to_contain(Vector3f p) {
        minimum = {std::fmin(minimum.x, p.x), std::fmin(minimum.y, p.y), std::fmin(minimum.z, p.z)};
        maximum = {std::fmax(maximum.x, p.x), std::fmax(maximum.y, p.y), std::fmax(maximum.z, p.z)};
    }
    void grow_to_contain(const AABB& b) { grow_to_contain(b.minimum); grow_to_contain(b.maximum); }
};

// OctahedralNormal.h:30-98: SNORM16 octahedral encoding of unit vectors.
struct OctahedralNormal {
    Vector2s encoding;
    static constexpr float max_short_value = 32767.0f;

    Vector3f decode() const {
        float fx = float(encoding.x), fy = float(encoding.y);
        Vector3f n(fx, fy, max_short_value - std::fabs(fx) - std::fabs(fy));
        float t = std::fmax(-n.z, 0.0f);
        n.x += n.x >= 0 ? -t : t;
        n.y += n.y >= 0 ? -t : t;
        return normalize(n);
    }

    static OctahedralNormal encode_precise(Vector3f n) {
        auto clamp1 = [](float v) { return v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v); };
        auto sign = [](float v) { return v >= 0.0f ? 1.0f : -1.0f; };
        float denom = std::fabs(n.x) + std::fabs(n.y) + std::fabs(n.z);
        float px = n.x / denom, py = n.y / denom;
        float p2x = px, p2y = py;
        if (n.z < 0) { p2x = (1.0f - std::fabs(py)) * sign(px); p2y = (1.0f - std::fabs(px)) * sign(py); }
        OctahedralNormal floored = {{short(std::floor(clamp1(p2x) * max_short_value)), short(std::floor(clamp1(p2y) * max_short_value))}};
        OctahedralNormal best = floored;
        float lowest = magnitude_squared(best.decode() - n);
        auto test = [&](short dx, short dy) {
            OctahedralNormal c = {{short(floored.encoding.x + dx), short(floored.encoding.y + dy)}};
            float m = magnitude_squared(c.decode() - n);
            if (m < lowest) { best = c; lowest = m; }
        };
        test(0, 1);
        test(1, 0);
        test(1, 1);
        return best;
    }
};

} // namespace Math
} // namespace Bifrost
