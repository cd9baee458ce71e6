// oracle/smallpt.cpp -- headless CPU restatement of the reference's SmallPT integrator, the CPU
// baseline BASELINE.json names (config 1: 256x256, 64 accumulations).
// TEST INFRASTRUCTURE ONLY (see oracle/vecmath.h).
//
// Follows /root/reference/apps/SmallPT/smallpt.h:22-147 (scene :47-57, intersect :40-45,59-63,
// radiance :65-118, accumulate_radiance :120-147) and core/Bifrost/Bifrost/Math/RNG.h:58-66,131-149.
// f64 geometry, f32 colour, LCG seeded with jenkins_hash(index) ^ reverse_bits(accumulations),
// OpenMP schedule(dynamic, 16) over rows like the reference (:129).
// Parity status: no reference test pins SmallPT output -- "parity unpinned"; where C++ leaves the
// evaluation order of the two recursive calls at smallpt.h:116 unspecified, this restatement
// evaluates left to right.
#include "rng.h"

#include <cmath>
#include <cstdint>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace smallpt_oracle {

struct V3 { double x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 normalize(V3 v) { double m = std::sqrt(dot(v, v)); return {v.x / m, v.y / m, v.z / m}; }

struct RGB { float r, g, b; };
static inline RGB operator+(RGB a, RGB b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
static inline RGB operator-(RGB a, RGB b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
static inline RGB operator*(RGB a, RGB b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
static inline RGB operator*(RGB a, float s) { return {a.r * s, a.g * s, a.b * s}; }

enum class BSDF { Diffuse, Specular, Glass };
struct Sphere { double radius; V3 position; RGB emission, color; BSDF bsdf; };

static const Sphere scene[9] = {
    {1e5, {1e5 + 1, 40.8, 81.6}, {0, 0, 0}, {.75f, .25f, .25f}, BSDF::Diffuse},
    {1e5, {-1e5 + 99, 40.8, 81.6}, {0, 0, 0}, {.25f, .25f, .75f}, BSDF::Diffuse},
    {1e5, {50, 40.8, 1e5}, {0, 0, 0}, {.75f, .75f, .75f}, BSDF::Diffuse},
    {1e5, {50, 40.8, -1e5 + 170}, {0, 0, 0}, {0, 0, 0}, BSDF::Diffuse},
    {1e5, {50, 1e5, 81.6}, {0, 0, 0}, {.75f, .75f, .75f}, BSDF::Diffuse},
    {1e5, {50, -1e5 + 81.6, 81.6}, {0, 0, 0}, {.75f, .75f, .75f}, BSDF::Diffuse},
    {16.5, {27, 16.5, 47}, {0, 0, 0}, {.999f, .999f, .999f}, BSDF::Specular},
    {16.5, {73, 16.5, 78}, {0, 0, 0}, {.999f, .999f, .999f}, BSDF::Glass},
    {600, {50, 681.6 - .27, 81.6}, {12.0f, 12.0f, 12.0f}, {0, 0, 0}, BSDF::Diffuse},
};

struct Ray { V3 origin, direction; };

static inline double intersect_sphere(const Sphere& s, const Ray& r) {
    V3 op = s.position - r.origin;
    double t, eps = 1e-4, b = dot(op, r.direction), det = b * b - dot(op, op) + s.radius * s.radius;
    if (det < 0) return 0;
    det = std::sqrt(det);
    return (t = b - det) > eps ? t : ((t = b + det) > eps ? t : 0);
}

static inline bool intersect(const Ray& r, double& t, int& id) {
    double d, inf = t = 1e20;
    for (int i = 9; i--;)
        if ((d = intersect_sphere(scene[i], r)) && d < t) { t = d; id = i; }
    return t < inf;
}

static RGB radiance(const Ray& ray, int depth, oracle::rng::LinearCongruential& rng, uint64_t& rays) {
    ++rays;
    double t;
    int id = 0;
    if (depth > 20 || !intersect(ray, t, id)) return {0, 0, 0};
    const Sphere& obj = scene[id];
    V3 pos = ray.origin + ray.direction * t;
    V3 norm = normalize(pos - obj.position);
    V3 nl = dot(norm, ray.direction) < 0 ? norm : norm * -1;
    RGB f = obj.color;
    float max_refl = f.r > f.g && f.r > f.b ? f.r : f.g > f.b ? f.g : f.b;
    if (++depth > 5) {
        if (rng.sample1f() < max_refl) f = f * (1 / max_refl);
        else return obj.emission;
    }
    const float PI = 3.14159265358979323846f;
    if (obj.bsdf == BSDF::Diffuse) {
        double r1 = 2.0f * PI * rng.sample1f();
        double r2 = rng.sample1f();
        double r2s = std::sqrt(r2);
        V3 w = nl;
        V3 u = normalize(cross(std::fabs(w.x) > 0.1 ? V3{0, 1, 0} : V3{1, 0, 0}, w));
        V3 v = cross(w, u);
        V3 dir = normalize(u * std::cos(r1) * r2s + v * std::sin(r1) * r2s + w * std::sqrt(1 - r2));
        return obj.emission + f * radiance({pos, dir}, depth, rng, rays);
    } else if (obj.bsdf == BSDF::Specular) {
        V3 refl = ray.direction - nl * 2 * dot(nl, ray.direction);
        return obj.emission + f * radiance({pos, refl}, depth, rng, rays);
    }
    Ray refl_ray = {pos, ray.direction - norm * 2 * dot(norm, ray.direction)};
    bool into = dot(norm, nl) > 0;
    const float nc = 1, nt = 1.5;
    double nnt = into ? nc / nt : nt / nc, ddn = dot(ray.direction, nl), cos2t;
    if ((cos2t = 1 - nnt * nnt * (1 - ddn * ddn)) < 0)
        return obj.emission + f * radiance(refl_ray, depth, rng, rays);
    V3 tdir = normalize(ray.direction * nnt - norm * ((into ? 1 : -1) * (ddn * nnt + std::sqrt(cos2t))));
    float a = nt - nc, b = nt + nc;
    float R0 = a * a / (b * b);
    float c = 1.0f - float(into ? -ddn : dot(tdir, norm));
    float Re = R0 + (1.0f - R0) * c * c * c * c * c;
    float Tr = 1.0f - Re;
    float P = .25f + .5f * Re;
    float RP = Re / P;
    float TP = Tr / (1.0f - P);
    if (depth > 2) {
        if (rng.sample1f() < P)
            return obj.emission + f * (radiance(refl_ray, depth, rng, rays) * RP);
        return obj.emission + f * (radiance({pos, tdir}, depth, rng, rays) * TP);
    }
    RGB first = radiance(refl_ray, depth, rng, rays) * Re;
    RGB second = radiance({pos, tdir}, depth, rng, rays) * Tr;
    return obj.emission + f * (first + second);
}

} // namespace smallpt_oracle

extern "C" {

// One accumulation over a w x h backbuffer (RGB f32). Returns the number of radiance() invocations.
uint64_t oracle_smallpt_accumulate(int w, int h, float* backbuffer_rgb, int* accumulations) {
    using namespace smallpt_oracle;
    Ray cam = {{50, 52, 295.6}, normalize(V3{0, -0.042612, -1})};
    int acc = ++(*accumulations);
    float blend = 1.0f / acc;
    V3 cx = {w * 0.5135 / h, 0, 0}, cy = normalize(cross(cx, cam.direction)) * 0.5135;
    uint64_t total_rays = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total_rays)
    for (int y = 0; y < h; ++y) {
        uint64_t rays = 0;
        for (int x = 0; x < w; ++x) {
            int sx = acc % 2;
            int sy = (acc >> 1) % 2;
            int index = (y * 2 + sy) * (w * 2) + x * 2 + sx;
            oracle::rng::LinearCongruential rng(oracle::rng::jenkins_hash(uint32_t(index)) ^ oracle::rng::reverse_bits(uint32_t(acc)));
            double r1 = 2 * rng.sample1f(), dx = r1 < 1 ? std::sqrt(r1) - 1 : 1 - std::sqrt(2 - r1);
            double r2 = 2 * rng.sample1f(), dy = r2 < 1 ? std::sqrt(r2) - 1 : 1 - std::sqrt(2 - r2);
            V3 d = cx * (((sx + .5 + dx) / 2 + x) / w - .5) + cy * (((sy + .5 + dy) / 2 + y) / h - .5) + cam.direction;
            RGB r = radiance({cam.origin + d * 140, normalize(d)}, 0, rng, rays);
            float* px = backbuffer_rgb + 3 * (size_t(y) * w + x);
            px[0] = px[0] + (r.r - px[0]) * blend;
            px[1] = px[1] + (r.g - px[1]) * blend;
            px[2] = px[2] + (r.b - px[2]) * blend;
        }
        total_rays += rays;
    }
    return total_rays;
}

int oracle_smallpt_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"
