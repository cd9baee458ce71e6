// oracle/vecmath.h -- minimal f32 vector algebra for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY: nothing under oracle/ is linked into or called by the product
// (bifrost3d_amd/, include/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// Semantics follow the optix:: vector helpers the reference's shading headers are written
// against (dot, cross, normalize, lerp, reflect, refract, clamp) -- see the call sites in
// /root/reference/extensions/OptiXRenderer/OptiXRenderer/*.h. Compiled with -ffp-contract=off so
// every operation rounds once, like the device code.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

// The transcendentals of the path (sin, cos, pow, atan2, asin, acos). By default glibc's f32 functions, the independent counterpart of whatever the product's
// fast shade kernel uses (hardware approximations). With oracle_set_f64_transcendentals(1) -- the checker of the EXACT arithmetic mode of the device
// (hipr_set_arithmetic, csrc/device_shading.h HIPR_VERIFY_MATH) -- sin, cos and pow are the SPECIFIED functions of DESIGN.md section 6 (csrc/spec_math.h
// states the specification): fixed sequences of correctly rounded binary64 operations, restated here, which any two IEEE-754 machines evaluate to the same
// bits; atan2, asin and acos (the environment map's lookup) are glibc's f64 functions rounded once to f32, which agree with the device's f64 libm but for
// ~2^-26 of the arguments. Set before a render, read by its threads.
inline bool g_f64_transcendentals = false;

namespace spec {
// Horner evaluation, one fma per coefficient, highest power first.
template <int N> inline double horner(const double (&c)[N], double z) {
    double p = c[0];
    for (int i = 1; i < N; ++i) p = std::fma(p, z, c[i]);
    return p;
}
// Taylor coefficients as quotients of integers: -1/3!, 1/5!, ... for (sin r - r) / r^3 and 1, -1/2!, 1/4!, ... for cos r, in z = r^2, highest power first.
inline const double SIN_TAIL[7] = {-1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0};
inline const double COS_SUM[9] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0, -0.5, 1.0};
inline const double ATANH_SUM[10] = {1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.0};
inline const double EXP_SUM[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                                   1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
inline const double TWO_OVER_PI = 0.63661977236758134308, PIO2_HIGH = 1.57079632679489655800, PIO2_LOW = 6.12323399573676603587e-17;
inline const double SQRT2 = 1.41421356237309514547, ONE_OVER_LN2 = 1.44269504088896338700, LN2 = 0.69314718055994528623;

// sin (which = 0) or cos (which = 1) of x: quadrant k = rint(x * 2/pi), r = x - k * pi/2 in two fused steps, the two Taylor sums, picked and signed by k mod 4.
inline float sin_or_cos(float x, int which) {
    const double xd = x;
    if (!(std::fabs(xd) < 1.0e6)) return std::nanf("");
    const double k = std::nearbyint(xd * TWO_OVER_PI);
    const double r = std::fma(-k, PIO2_LOW, std::fma(-k, PIO2_HIGH, xd));
    const double z = r * r;
    const int quadrant = (int(k) + which) & 3;       // cos x = sin(x + pi/2)
    const double value = (quadrant & 1) ? horner(COS_SUM, z) : std::fma(r * z, horner(SIN_TAIL, z), r);
    return float((quadrant & 2) ? -value : value);
}
inline float pow(float x, float y) {
    if (y != y) return y;
    if (!(x > 0.0f)) return x == 0.0f ? (y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : INFINITY)) : std::nanf("");
    if (std::isinf(x)) return y > 0.0f ? x : (y == 0.0f ? 1.0f : 0.0f);
    const double xd = x;
    uint64_t bits;
    std::memcpy(&bits, &xd, 8);
    int e = int(bits >> 52) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    std::memcpy(&m, &bits, 8);
    if (m > SQRT2) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double ln_m = (s + s) * horner(ATANH_SUM, s * s);
    double t = double(y) * std::fma(ln_m, ONE_OVER_LN2, double(e));
    t = t < -300.0 ? -300.0 : (t > 300.0 ? 300.0 : t);
    const double k = std::nearbyint(t);
    const double power = horner(EXP_SUM, (t - k) * LN2);
    const uint64_t scale_bits = uint64_t(int(k) + 1023) << 52;
    double scale;
    std::memcpy(&scale, &scale_bits, 8);
    return float(power * scale);
}
} // namespace spec

inline float exact_sinf(float x) { return g_f64_transcendentals ? spec::sin_or_cos(x, 0) : sinf(x); }
inline float exact_cosf(float x) { return g_f64_transcendentals ? spec::sin_or_cos(x, 1) : cosf(x); }
inline float exact_powf(float x, float y) { return g_f64_transcendentals ? spec::pow(x, y) : powf(x, y); }
inline float exact_atan2f(float y, float x) { return g_f64_transcendentals ? float(std::atan2(double(y), double(x))) : atan2f(y, x); }
inline float exact_asinf(float x) { return g_f64_transcendentals ? float(std::asin(double(x))) : asinf(x); }
inline float exact_acosf(float x) { return g_f64_transcendentals ? float(std::acos(double(x))) : acosf(x); }

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct double3 { double x, y, z; };

inline float2 make_float2(float x, float y) { return {x, y}; }
inline float2 make_float2(float3 v) { return {v.x, v.y}; }
inline float2 make_float2(float4 v) { return {v.x, v.y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float3 make_float3(float v) { return {v, v, v}; }
inline float3 make_float3(float2 v, float z) { return {v.x, v.y, z}; }
inline float3 make_float3(float4 v) { return {v.x, v.y, v.z}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline float4 make_float4(float3 v, float w) { return {v.x, v.y, v.z, w}; }

inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }
inline float2 operator*(float s, float2 a) { return {a.x * s, a.y * s}; }
inline float2 operator*(float2 a, float2 b) { return {a.x * b.x, a.y * b.y}; }
inline float2 operator/(float2 a, float s) { float inv = 1.0f / s; return {a.x * inv, a.y * inv}; }
inline float2 operator-(float2 a, float s) { return {a.x - s, a.y - s}; }

inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
// optix::operator/(float3, float) multiplies by the reciprocal.
inline float3 operator/(float3 a, float s) { float inv = 1.0f / s; return {a.x * inv, a.y * inv, a.z * inv}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 operator+(float3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline float3 operator+(float s, float3 a) { return {a.x + s, a.y + s, a.z + s}; }
inline float3 operator-(float3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline float3 operator-(float s, float3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
inline float3& operator*=(float3& a, float s) { a = a * s; return a; }
inline float3& operator/=(float3& a, float s) { a = a / s; return a; }

inline float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline float4 operator-(float4 a, float4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline float4 operator*(float4 a, float4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline float4 operator*(float4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(float2 v) { return sqrtf(dot(v, v)); }
inline float length(float3 v) { return sqrtf(dot(v, v)); }
// optix::normalize: v * (1 / sqrt(dot(v, v)))
inline float3 normalize(float3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float3 lerp(float3 a, float3 b, float t) { return a + t * (b - a); }
inline float clampf(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); }
inline float3 fminf3(float3 a, float3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
// optix::reflect(i, n) = i - 2 n dot(n, i)
inline float3 reflect(float3 i, float3 n) { return i - 2.0f * n * dot(n, i); }

// optix::refract(r, i, n, ior) from optixu_math_namespace.h as documented by the reference's
// own copy with n = (0,0,1) (OR/Utils.h:242-256) and pinned by ORT/MiscTest.h:292-322.
inline bool refract(float3& r, float3 i, float3 n, float ior) {
    float3 nn = n;
    float negNdotV = dot(i, nn);
    float eta;
    if (negNdotV > 0.0f) {
        eta = ior;
        nn = -n;
        negNdotV = -negNdotV;
    } else
        eta = 1.0f / ior;
    const float k = 1.0f - eta * eta * (1.0f - negNdotV * negNdotV);
    if (k < 0.0f) {
        r = make_float3(0.0f);
        return false;
    }
    r = normalize(eta * i - (eta * negNdotV + sqrtf(k)) * nn);
    return true;
}

inline uint32_t float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline int32_t float_as_int(float f) { int32_t u; std::memcpy(&u, &f, 4); return u; }
inline float uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float int_as_float(int32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct Matrix2x2 {
    float m[4]; // row major
    void setCol(int c, float2 v) { m[c] = v.x; m[2 + c] = v.y; }
    Matrix2x2 transpose() const { return {{m[0], m[2], m[1], m[3]}}; }
};
inline float2 operator*(const Matrix2x2& M, float2 v) { return {M.m[0] * v.x + M.m[1] * v.y, M.m[2] * v.x + M.m[3] * v.y}; }

} // namespace oracle
