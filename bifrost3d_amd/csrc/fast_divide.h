// fast_divide.h -- exact unsigned 32-bit division by a divisor known to the host (the samples per pass, the tiles per row): one multiply-high and two or
// three shifts / adds on the device instead of the ~30 instructions of a hardware-assisted n / d with a wave-uniform d. The usual round-up multiplier
// (Granlund & Montgomery 1994): floor(n / d) = mulhi(n, m) >> s, with the 33rd bit of the multiplier handled by the "add" form where it is needed.
// Host and device code (tests/native/MathTest.cpp checks it against the operator).
#pragma once

#include <cstdint>

#ifdef __HIPCC__
#define HIPR_FAST_DIVIDE_HD __host__ __device__ __forceinline__
#else
#define HIPR_FAST_DIVIDE_HD inline
#endif

namespace hipr {

struct Divisor {
    uint32_t multiplier;    // 0: a power of two, the quotient is n >> shift
    uint32_t shift;
    uint32_t add;           // the multiplier has 33 bits: q = (((n - t) >> 1) + t) >> shift with t = mulhi(n, multiplier)
};

inline Divisor make_divisor(uint32_t d) {       // d > 0 (0 is taken as 1: the callers validate their frames before they get here)
    Divisor r = {0u, 0u, 0u};
    if (d == 0u) return r;
    uint32_t log2_floor = 31u;
    while (!((d >> log2_floor) & 1u)) --log2_floor;
    if ((d & (d - 1u)) == 0u) { r.shift = log2_floor; return r; }
    const uint64_t two_k = uint64_t(1) << (32u + log2_floor);
    uint64_t m = two_k / d;
    const uint64_t rem = two_k - m * d;
    const uint64_t e = d - rem;
    if (e < (uint64_t(1) << log2_floor)) r.shift = log2_floor;
    else {      // one more bit of the multiplier
        m *= 2u;
        const uint64_t twice_rem = rem * 2u;
        if (twice_rem >= d) m += 1u;
        r.shift = log2_floor;
        r.add = 1u;
    }
    r.multiplier = uint32_t(m + 1u);
    return r;
}

HIPR_FAST_DIVIDE_HD uint32_t multiply_high(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return uint32_t((uint64_t(a) * uint64_t(b)) >> 32);
#endif
}

HIPR_FAST_DIVIDE_HD uint32_t divide(uint32_t n, const Divisor& d) {
    if (d.multiplier == 0u) return n >> d.shift;
    const uint32_t t = multiply_high(n, d.multiplier);
    if (d.add) return (((n - t) >> 1) + t) >> d.shift;
    return t >> d.shift;
}

} // namespace hipr
