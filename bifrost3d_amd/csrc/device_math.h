// device_math.h -- f32 vector algebra and the random number generators of the HIP path tracer.
//
// Written for gfx950 wave64 execution; everything is __device__ __forceinline__ so that the
// per-lane state stays in VGPRs. The arithmetic (operation order, reciprocal-multiply division of a
// vector by a scalar, normalize = v * (1 / sqrt(dot))) follows the optix:: helpers the reference
// shading code is written against, so results track the CPU oracle to the last bit wherever no
// transcendental is involved. Compiled with -ffp-contract=off: fmaf only where written.
//
// Reference: extensions/OptiXRenderer/OptiXRenderer/RNG.h:39-75,127-157,238-293, Utils.h:331-342.
#pragma once

// HIPR_HOST_DEVICE (tests/native/DeviceShadeHost.hip only): HD = __host__ __device__, so that hipcc's host pass compiles the shared code -- the BSDFs, shading models and
// lights of device_shading.h, shade_path of shade_kernel.h with the samplers and scene structures of kernels.h -- for x86 as well, statement for statement, and the
// CPU test suite can set the device's K3 against the oracle BIT for bit (tests/test_device_code_on_host_cpu.py). With HIPR_VERIFY_MATH and -ffp-contract=off the host
// build evaluates what libhiprenderer_verify.so evaluates on the GPU: IEEE f32 operations in the same order. Test infrastructure: the product libraries are never
// built this way and have no CPU path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef HIPR_HOST_DEVICE
// host overloads of the bit-cast and bit-reversal device functions the shared code uses (clang overloads on the target attribute)
#include <cstring>
__host__ static inline unsigned int __float_as_uint(float f) { unsigned int u; std::memcpy(&u, &f, 4); return u; }
__host__ static inline float __uint_as_float(unsigned int u) { float f; std::memcpy(&f, &u, 4); return f; }
__host__ static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
__host__ static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
__host__ static inline unsigned int __brev(unsigned int v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
#endif
#ifndef HD
#ifdef HIPR_HOST_DEVICE
#define HD __host__ __device__ __forceinline__
#else
#define HD __device__ __forceinline__
#endif
#endif

namespace hipr {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

HD f2 mk2(float x, float y) { return {x, y}; }
HD f3 mk3(float x, float y, float z) { return {x, y, z}; }
HD f3 mk3(float v) { return {v, v, v}; }
HD f3 mk3(float4 v) { return {v.x, v.y, v.z}; }
HD f4 mk4(float4 v) { return {v.x, v.y, v.z, v.w}; }

HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
HD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
HD f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
HD f2 operator*(float s, f2 a) { return {a.x * s, a.y * s}; }
HD f2 operator/(f2 a, float s) { float inv = 1.0f / s; return {a.x * inv, a.y * inv}; }

HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
HD f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
HD f3 operator/(f3 a, float s) { float inv = 1.0f / s; return {a.x * inv, a.y * inv, a.z * inv}; }
HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
HD f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
HD f3 operator+(float s, f3 a) { return {a.x + s, a.y + s, a.z + s}; }
HD f3 operator-(f3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
HD f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
HD f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
HD f3& operator*=(f3& a, float s) { a = a * s; return a; }
HD f3& operator/=(f3& a, float s) { a = a / s; return a; }

HD f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
HD f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
HD f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
HD f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

HD float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
HD float length(f2 v) { return sqrtf(dot(v, v)); }
HD float length(f3 v) { return sqrtf(dot(v, v)); }
HD f3 normalize(f3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
HD float lerp(float a, float b, float t) { return a + t * (b - a); }
HD f3 lerp(f3 a, f3 b, float t) { return a + t * (b - a); }
HD float clampf(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); }
HD f3 min3(f3 a, f3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
HD f3 reflect(f3 i, f3 n) { return i - 2.0f * n * dot(n, i); }
HD float sum(f3 v) { return v.x + v.y + v.z; }
HD bool is_black(f3 c) { return c.x <= 0.0f && c.y <= 0.0f && c.z <= 0.0f; }

// Explicitly fused forms used by the ray/box and ray/triangle tests (DESIGN.md "Intersection arithmetic").
HD float dot_fma(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
HD f3 cross_fma(f3 a, f3 b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}

HD bool refract(f3& r, f3 i, f3 n, float ior) {
    f3 nn = n;
    float neg_n_dot_v = dot(i, nn);
    float eta;
    if (neg_n_dot_v > 0.0f) { eta = ior; nn = -n; neg_n_dot_v = -neg_n_dot_v; }
    else eta = 1.0f / ior;
    const float k = 1.0f - eta * eta * (1.0f - neg_n_dot_v * neg_n_dot_v);
    if (k < 0.0f) { r = mk3(0.0f); return false; }
    r = normalize(eta * i - (eta * neg_n_dot_v + sqrtf(k)) * nn);
    return true;
}

// ---------------------------------------------------------------------------------------------
// Random numbers
// ---------------------------------------------------------------------------------------------
#define HIPR_UINT_NORMALIZER (1.0f / 4294967296.0f)

constexpr uint32_t SOBOL_DIRECTIONS[3][32] = {
    // dimension 1..3 (dimension 0 is the bit reversal of the index)
    {0x80000000u, 0xc0000000u, 0xa0000000u, 0xf0000000u, 0x88000000u, 0xcc000000u, 0xaa000000u, 0xff000000u,
     0x80800000u, 0xc0c00000u, 0xa0a00000u, 0xf0f00000u, 0x88880000u, 0xcccc0000u, 0xaaaa0000u, 0xffff0000u,
     0x80008000u, 0xc000c000u, 0xa000a000u, 0xf000f000u, 0x88008800u, 0xcc00cc00u, 0xaa00aa00u, 0xff00ff00u,
     0x80808080u, 0xc0c0c0c0u, 0xa0a0a0a0u, 0xf0f0f0f0u, 0x88888888u, 0xccccccccu, 0xaaaaaaaau, 0xffffffffu},
    {0x80000000u, 0xc0000000u, 0x60000000u, 0x90000000u, 0xe8000000u, 0x5c000000u, 0x8e000000u, 0xc5000000u,
     0x68800000u, 0x9cc00000u, 0xee600000u, 0x55900000u, 0x80680000u, 0xc09c0000u, 0x60ee0000u, 0x90550000u,
     0xe8808000u, 0x5cc0c000u, 0x8e606000u, 0xc5909000u, 0x6868e800u, 0x9c9c5c00u, 0xeeee8e00u, 0x5555c500u,
     0x8000e880u, 0xc0005cc0u, 0x60008e60u, 0x9000c590u, 0xe8006868u, 0x5c009c9cu, 0x8e00eeeeu, 0xc5005555u},
    {0x80000000u, 0xc0000000u, 0x20000000u, 0x50000000u, 0xf8000000u, 0x74000000u, 0xa2000000u, 0x93000000u,
     0xd8800000u, 0x25400000u, 0x59e00000u, 0xe6d00000u, 0x78080000u, 0xb40c0000u, 0x82020000u, 0xc3050000u,
     0x208f8000u, 0x51474000u, 0xfbea2000u, 0x75d93000u, 0xa0858800u, 0x914e5400u, 0xdbe79e00u, 0x25db6d00u,
     0x58800080u, 0xe54000c0u, 0x79e00020u, 0xb6d00050u, 0x800800f8u, 0xc00c0074u, 0x200200a2u, 0x50050093u},
};

HD uint32_t pcg2d_x(uint32_t x, uint32_t y) {
    const uint32_t m = 1664525u, c = 1013904223u;
    x = x * m + c;
    y = y * m + c;
    x += y * m;
    y += x * m;
    x ^= x >> 16;
    y ^= y >> 16;
    x += y * m;
    y += x * m;   // kept: x's final xorshift does not depend on it, the compiler drops it
    x ^= x >> 16;
    return x;
}

HD uint32_t cessen_owen_hash(uint32_t x, uint32_t seed) {
    x ^= x * 0x3d20adeau;
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return x;
}
HD uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
HD uint32_t owen_scramble(uint32_t x, uint32_t seed) { return __brev(cessen_owen_hash(__brev(x), seed)); }

struct u4 { uint32_t x, y, z, w; };

// PracticalScrambledSobol::sample4ui(accumulation, pixel_hash, dimension), OR/RNG.h:269-287.
// The direction numbers sit in constant memory and the loop counter is wave-uniform, so the
// table reads are scalar loads and only the per-lane index bits cost VALU work.
HD u4 sobol4ui(uint32_t accumulation, uint32_t pixel_hash, uint32_t dimension) {
    const uint32_t seed = pcg2d_x(pixel_hash, dimension);
    const uint32_t index = owen_scramble(accumulation, seed);
    uint32_t r1 = 0, r2 = 0, r3 = 0;
#pragma unroll
    for (int bit = 0; bit < 32; ++bit) {
        const uint32_t mask = 0u - ((index >> bit) & 1u);
        r1 ^= mask & SOBOL_DIRECTIONS[0][bit];
        r2 ^= mask & SOBOL_DIRECTIONS[1][bit];
        r3 ^= mask & SOBOL_DIRECTIONS[2][bit];
    }
    u4 s;
    s.x = owen_scramble(__brev(index), hash_combine(seed, 0));
    s.y = owen_scramble(r1, hash_combine(seed, 1));
    s.z = owen_scramble(r2, hash_combine(seed, 2));
    s.w = owen_scramble(r3, hash_combine(seed, 3));
    return s;
}

// Same sampler with the 32-term XOR of each dimension folded into four byte-indexed tables (3 dimensions x 4 bytes x
// 256 entries, 12 KiB, staged in LDS by the shade kernel): 12 ds_read_b32 + 12 XOR instead of ~250 VALU operations.
constexpr uint32_t SOBOL_TABLE_WORDS = 3 * 4 * 256;
HD u4 sobol4ui_tables(uint32_t accumulation, uint32_t pixel_hash, uint32_t dimension, const uint32_t* tables) {
    const uint32_t seed = pcg2d_x(pixel_hash, dimension);
    const uint32_t index = owen_scramble(accumulation, seed);
    const uint32_t b0 = index & 255u, b1 = (index >> 8) & 255u, b2 = (index >> 16) & 255u, b3 = index >> 24;
    const uint32_t r1 = tables[b0] ^ tables[256 + b1] ^ tables[512 + b2] ^ tables[768 + b3];
    const uint32_t r2 = tables[1024 + b0] ^ tables[1280 + b1] ^ tables[1536 + b2] ^ tables[1792 + b3];
    const uint32_t r3 = tables[2048 + b0] ^ tables[2304 + b1] ^ tables[2560 + b2] ^ tables[2816 + b3];
    u4 s;
    s.x = owen_scramble(__brev(index), hash_combine(seed, 0));
    s.y = owen_scramble(r1, hash_combine(seed, 1));
    s.z = owen_scramble(r2, hash_combine(seed, 2));
    s.w = owen_scramble(r3, hash_combine(seed, 3));
    return s;
}
HD f4 sobol4f_tables(uint32_t accumulation, uint32_t pixel_hash, uint32_t dimension, const uint32_t* tables) {
    u4 s = sobol4ui_tables(accumulation, pixel_hash, dimension, tables);
    return {float(s.x) * HIPR_UINT_NORMALIZER, float(s.y) * HIPR_UINT_NORMALIZER, float(s.z) * HIPR_UINT_NORMALIZER, float(s.w) * HIPR_UINT_NORMALIZER};
}

HD f4 sobol4f(uint32_t accumulation, uint32_t pixel_hash, uint32_t dimension) {
    u4 s = sobol4ui(accumulation, pixel_hash, dimension);
    return {float(s.x) * HIPR_UINT_NORMALIZER, float(s.y) * HIPR_UINT_NORMALIZER, float(s.z) * HIPR_UINT_NORMALIZER, float(s.w) * HIPR_UINT_NORMALIZER};
}

} // namespace hipr
