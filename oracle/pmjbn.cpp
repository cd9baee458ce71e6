// oracle/pmjbn.cpp -- progressive multi-jittered blue-noise point set the reference's BSDF tests
// integrate with (ORT/BSDFTestUtils.h:31-33 -> Bifrost::Math::RNG::PmjbRNG(16384)).
// TEST INFRASTRUCTURE ONLY (see oracle/vecmath.h).
//
// Restates core/Bifrost/Bifrost/Math/RNG.cpp:21-199 (Christensen et al. 2018 pmj02 with a
// best-of-8 candidate pick): LCG seed 19349669, 1-D strata occupancy tables, toroidal nearest
// neighbour search along the strata, alternating even / odd power-of-two extension. Needed so the
// statistical goldens G3-G5 of SURVEY.md 8c are replayed on the same points the reference used.
#include "rng.h"

#include <cmath>
#include <vector>

namespace {

struct P2 { float x, y; };

inline float dist2(P2 a, P2 b) { float dx = a.x - b.x, dy = a.y - b.y; return dx * dx + dy * dy; }

struct Generator {
    P2* samples;
    unsigned total;
    oracle::rng::LinearCongruential lcg{19349669u};
    unsigned candidates;
    unsigned short next_index = 0;
    static constexpr unsigned short FREE = 65535;
    std::vector<unsigned short> sx, sy;

    float rnd() { return lcg.sample1f(); }

    void place(P2 oldpt, int i, int j, int xhalf, int yhalf, int prev_grid, int prev_count) {
        int next_count = 2 * prev_count;
        P2 best = {NAN, NAN};
        float best_distance = 0;
        for (unsigned s = 0; s < candidates; ++s) {
            P2 pt;
            do { pt.x = (i + 0.5f * (xhalf + rnd())) / prev_grid; } while (sx[int(next_count * pt.x)] != FREE);
            do { pt.y = (j + 0.5f * (yhalf + rnd())) / prev_grid; } while (sy[int(next_count * pt.y)] != FREE);
            int xs = int(next_count * pt.x), ys = int(next_count * pt.y);
            float d = dist2(oldpt, pt);
            int max_search = int(next_count * std::sqrt(d));
            auto test = [&](unsigned short idx) {
                if (idx == FREE) return;
                P2 nb = samples[idx];
                if (nb.x < pt.x - 0.5f) nb.x += 1.0f; else if (nb.x > pt.x + 0.5f) nb.x -= 1.0f;
                if (nb.y < pt.y - 0.5f) nb.y += 1.0f; else if (nb.y > pt.y + 0.5f) nb.y -= 1.0f;
                float ld = dist2(nb, pt);
                if (ld < d) { d = ld; max_search = int(next_count * std::sqrt(d)); }
            };
            for (int off = 1; off <= max_search; ++off) {
                test(sx[(xs + off) % next_count]);
                test(sx[(xs + next_count - off) % next_count]);
                test(sy[(ys + off) % next_count]);
                test(sy[(ys + next_count - off) % next_count]);
            }
            if (best_distance < d) { best_distance = d; best = pt; }
        }
        sx[int(next_count * best.x)] = sy[int(next_count * best.y)] = next_index;
        samples[next_index++] = best;
    }

    void mark(unsigned prev_count) {
        unsigned next_count = 2 * prev_count;
        for (unsigned i = 0; i < next_count; ++i) sx[i] = sy[i] = FREE;
        for (unsigned s = 0; s < prev_count; ++s) {
            sx[int(next_count * samples[s].x)] = (unsigned short)s;
            sy[int(next_count * samples[s].y)] = (unsigned short)s;
        }
    }

    void extend_even(unsigned prev_count) {
        unsigned grid = (unsigned)std::sqrt(prev_count);
        mark(prev_count);
        for (unsigned s = 0; s < prev_count && next_index < total; ++s) {
            P2 o = samples[s];
            int i = int(grid * o.x), j = int(grid * o.y);
            int xhalf = 1 - int(2 * (grid * o.x - i)), yhalf = 1 - int(2 * (grid * o.y - j));
            place(o, i, j, xhalf, yhalf, grid, prev_count);
        }
    }

    void extend_odd(unsigned prev_count) {
        unsigned grid = (unsigned)std::sqrt(prev_count / 2);
        mark(prev_count);
        for (unsigned s = 0; s < prev_count / 2 && next_index < total; ++s) {
            P2 o = samples[s];
            int i = int(grid * o.x), j = int(grid * o.y);
            int xhalf = int(2 * (grid * o.x - i)), yhalf = int(2 * (grid * o.y - j));
            if (rnd() > 0.5) xhalf = 1 - xhalf; else yhalf = 1 - yhalf;
            place(o, i, j, xhalf, yhalf, grid, prev_count);
        }
        for (unsigned s = 0; s < prev_count / 2 && next_index < total; ++s) {
            P2 o = samples[s + prev_count];
            int i = int(grid * o.x), j = int(grid * o.y);
            int xhalf = 1 - int(2 * (grid * o.x - i)), yhalf = 1 - int(2 * (grid * o.y - j));
            place(o, i, j, xhalf, yhalf, grid, prev_count);
        }
    }

    void run() {
        unsigned cap = 1;
        while (cap < total) cap *= 2;
        sx.assign(cap, FREE);
        sy.assign(cap, FREE);
        float x = rnd();
        float y = rnd();
        samples[next_index++] = {x, y};
        unsigned count = 1;
        while (count < total) {
            extend_even(count);
            if (2 * count < total) extend_odd(2 * count);
            count *= 4;
        }
    }
};

} // namespace

extern "C" void oracle_pmjbn_samples(float* out_xy, unsigned count, unsigned blue_noise_candidates) {
    Generator g;
    g.samples = reinterpret_cast<P2*>(out_xy);
    g.total = count;
    g.candidates = blue_noise_candidates ? blue_noise_candidates : 1u;
    g.run();
}
