// spec_math.h -- the SPECIFIED transcendentals of the exact arithmetic mode (hipr_set_arithmetic(ctx, HIPR_ARITHMETIC_EXACT)).
//
// sin / cos / pow are the only transcendentals on the shade stage's hot path (the sampled azimuths and the two roughness remappings; the reference gets them from
// CUDA's --use_fast_math approximations, extensions/OptiXRenderer/CMakeLists.txt:82-83, call sites OR/Distributions.h:304-461, ORS/ShadingModels/DefaultShading.h:68-132).
// A renderer whose frames are to be compared with a CPU restatement bit for bit needs functions that BOTH sides can evaluate to the same bits, and a GPU
// libm and glibc do not agree in f32. Round 5 evaluated every one of them with the f64 libm of either side and rounded once (two f64 results a few ulp from
// the truth round to the same f32 but for ~2^-26 of the arguments): exact enough, and 31 ms per atrium step -- ocml's f64 sin, cos and pow are hundreds of
// instructions each with Payne-Hanek and table paths nobody takes here.
//
// Round 6 specifies the functions instead: each is a FIXED sequence of IEEE-754 binary64 operations (+, *, fma, /, round-to-nearest-even integer, conversion),
// every one of them correctly rounded on gfx950 and on x86 alike, so any two implementations of the sequence agree in every bit by construction;
// oracle/vecmath.h restates the same sequences for the CPU. The polynomials are plain Taylor sums whose coefficients are written as quotients of integers
// (folded by any IEEE compiler to the same constant); their truncation error is below 2^-55 of the result on the reduced range, so the f32 results are the
// correctly rounded ones except where the true value lies within ~1e-14 relative of a rounding boundary (tests/test_spec_math_cpu.py: against glibc's f64
// functions on 4 M arguments, at most 0.5 + 1e-6 ulp away, none differing by more than the last bit).
//
//   spec_sincos(x): k = rint(x * 2/pi); r = fma(-k, PIO2_LO, fma(-k, PIO2_HI, x)) (|r| <= pi/4 + 1e-16); sin r and cos r by their Taylor sums to r^15 and r^16 in
//                   Horner form over z = r * r, fma at every step; the quadrant k & 3 picks and signs the pair. |x| >= 1e6 (never a sampled azimuth) or NaN: NaN, NaN.
//   spec_pow(x, y): x = m * 2^e with m in (sqrt(1/2), sqrt(2)]; s = (m - 1) / (m + 1), z = s * s; ln m = 2 s (1 + z/3 + ... + z^9/19);
//                   t = y * (e + ln m * (1/ln 2)) clamped to [-300, 300]; k = rint(t), g = (t - k) * ln 2; e^g by its Taylor sum to g^13; result = float(e^g * 2^k).
//                   x == 0: 0 for y > 0, 1 for y == 0, +inf for y < 0; x < 0 or NaN, y NaN: NaN; x = +inf: +inf, 1, 0 for y >, ==, < 0.
#pragma once

#include "device_math.h"

namespace hipr {

HD double spec_poly_sin(double r, double z) {
    double p = -1.0 / 1307674368000.0;
    p = fma(p, z, 1.0 / 6227020800.0);
    p = fma(p, z, -1.0 / 39916800.0);
    p = fma(p, z, 1.0 / 362880.0);
    p = fma(p, z, -1.0 / 5040.0);
    p = fma(p, z, 1.0 / 120.0);
    p = fma(p, z, -1.0 / 6.0);
    return fma(r * z, p, r);
}
HD double spec_poly_cos(double z) {
    double p = 1.0 / 20922789888000.0;
    p = fma(p, z, -1.0 / 87178291200.0);
    p = fma(p, z, 1.0 / 479001600.0);
    p = fma(p, z, -1.0 / 3628800.0);
    p = fma(p, z, 1.0 / 40320.0);
    p = fma(p, z, -1.0 / 720.0);
    p = fma(p, z, 1.0 / 24.0);
    p = fma(p, z, -0.5);
    return fma(p, z, 1.0);
}

HD void spec_sincos(float x, float& s, float& c) {
    const double xd = double(x);
    if (!(fabs(xd) < 1.0e6)) { s = c = __builtin_nanf(""); return; }
    const double k = rint(xd * 0.63661977236758134308);                   // 2 / pi
    double r = fma(-k, 1.57079632679489655800, xd);                       // pi / 2, the double nearest to it ...
    r = fma(-k, 6.12323399573676603587e-17, r);                           // ... and what it leaves
    const double z = r * r;
    const double sr = spec_poly_sin(r, z), cr = spec_poly_cos(z);
    const int q = int(k) & 3;
    const double sq = (q & 1) ? cr : sr, cq = (q & 1) ? sr : cr;
    s = float((q & 2) ? -sq : sq);
    c = float(((q + 1) & 2) ? -cq : cq);
}

HD float spec_pow(float x, float y) {
    if (!(y == y)) return y;
    if (!(x > 0.0f)) {
        if (x == 0.0f) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : __builtin_inff());
        return __builtin_nanf("");
    }
    if (x == __builtin_inff()) return y > 0.0f ? x : (y == 0.0f ? 1.0f : 0.0f);
    // a float, denormal or not, is a normal double: exponent and mantissa straight from its bits
    unsigned long long bits = __builtin_bit_cast(unsigned long long, double(x));
    int e = int(bits >> 52) - 1023;
    double m = __builtin_bit_cast(double, (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);      // [1, 2)
    if (m > 1.41421356237309514547) { m *= 0.5; e += 1; }
    const double sm = (m - 1.0) / (m + 1.0), z = sm * sm;
    double p = 1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    const double ln_m = (sm + sm) * p;
    double t = double(y) * fma(ln_m, 1.44269504088896338700, double(e));        // 1 / ln 2
    t = t < -300.0 ? -300.0 : (t > 300.0 ? 300.0 : t);
    const double k = rint(t);
    const double g = (t - k) * 0.69314718055994528623;                           // ln 2
    double q = 1.0 / 6227020800.0;
    q = fma(q, g, 1.0 / 479001600.0);
    q = fma(q, g, 1.0 / 39916800.0);
    q = fma(q, g, 1.0 / 3628800.0);
    q = fma(q, g, 1.0 / 362880.0);
    q = fma(q, g, 1.0 / 40320.0);
    q = fma(q, g, 1.0 / 5040.0);
    q = fma(q, g, 1.0 / 720.0);
    q = fma(q, g, 1.0 / 120.0);
    q = fma(q, g, 1.0 / 24.0);
    q = fma(q, g, 1.0 / 6.0);
    q = fma(q, g, 0.5);
    q = fma(q, g, 1.0);
    q = fma(q, g, 1.0);
    const double scale = __builtin_bit_cast(double, (unsigned long long)(int(k) + 1023) << 52);
    return float(q * scale);
}

} // namespace hipr
