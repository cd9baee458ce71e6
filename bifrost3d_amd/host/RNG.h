// RNG.h -- the pieces of Bifrost::Math::RNG the host side needs (core/Bifrost/Bifrost/Math/RNG.h:20-66, RNG.cpp:21-199):
// the (0,2)-sequence used by the reference's tests and the progressive multi-jittered blue-noise point set the environment
// light is presampled with.
#pragma once

#include "Math.h"

#include <cstdint>

namespace Bifrost {
namespace Math {
namespace RNG {

inline uint32_t reverse_bits(uint32_t n) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    return n;
}

const float uint_normalizer = 1.0f / 4294967296.0f;

inline float van_der_corput(uint32_t n, uint32_t scramble) { return float(reverse_bits(n) ^ scramble) * uint_normalizer; }
inline float sobol2(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 1u) scramble ^= v;
    return float(scramble) * uint_normalizer;
}
inline Vector2f sample02(uint32_t n, uint32_t scramble_x = 5569u, uint32_t scramble_y = 95597u) { return {van_der_corput(n, scramble_x), sobol2(n, scramble_y)}; }

// Fills [begin, end) with progressive multi-jittered (0,2) samples, each picked as the best of `blue_noise_samples` candidates
// by toroidal distance to the points placed so far (Christensen et al. 2018, as implemented in RNG.cpp:21-199; at most 65535 points).
void fill_progressive_multijittered_bluenoise_samples(Vector2f* begin, Vector2f* end, unsigned int blue_noise_samples = 8);

} // namespace RNG
} // namespace Math
} // namespace Bifrost
