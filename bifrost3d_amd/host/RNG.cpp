// RNG.cpp -- progressive multi-jittered blue-noise samples (see RNG.h). Follows the construction of
// core/Bifrost/Bifrost/Math/RNG.cpp:21-199: an LCG seeded with 19349669 drives everything; the point count quadruples per round
// (one "diagonal" pass that doubles it, one pass that fills the remaining two sub-quadrants); every new point must fall in
// free 1-D strata of both axes at the doubled resolution, and of `blue_noise_samples` candidates the one farthest (toroidally)
// from all earlier points is kept.
#include "RNG.h"

#include <cmath>
#include <limits>
#include <vector>

namespace Bifrost {
namespace Math {
namespace RNG {

namespace {

class BlueNoisePointSet {
public:
    BlueNoisePointSet(Vector2f* points, unsigned int total, unsigned int candidates) : m_points(points), m_total(total), m_candidates(candidates ? candidates : 1u) {
        unsigned int capacity = 1;
        while (capacity < total) capacity *= 2;
        m_stratum_x.assign(capacity, EMPTY);
        m_stratum_y.assign(capacity, EMPTY);
    }

    void generate() {
        const float x = uniform(), y = uniform();
        m_points[m_placed++] = {x, y};
        for (unsigned int count = 1; count < m_total; count *= 4) {
            diagonal_pass(count);
            if (2 * count < m_total) remaining_pass(2 * count);
        }
    }

private:
    static constexpr uint16_t EMPTY = 65535;

    float uniform() { m_state = 1664525u * m_state + 1013904223u; return float(m_state) * uint_normalizer; }

    static float squared_distance(Vector2f a, Vector2f b) { const float dx = a.x - b.x, dy = a.y - b.y; return dx * dx + dy * dy; }

    // Occupancy of the 1-D strata at twice the current resolution.
    void rebuild_strata(unsigned int placed) {
        const unsigned int strata = 2 * placed;
        for (unsigned int i = 0; i < strata; ++i) m_stratum_x[i] = m_stratum_y[i] = EMPTY;
        for (unsigned int p = 0; p < placed; ++p) {
            m_stratum_x[int(strata * m_points[p].x)] = uint16_t(p);
            m_stratum_y[int(strata * m_points[p].y)] = uint16_t(p);
        }
    }

    // Places one point in sub-cell (x_half, y_half) of cell (i, j) of the grid the parent point lives in.
    void place(Vector2f parent, int i, int j, int x_half, int y_half, int grid, int placed) {
        const int strata = 2 * placed;
        Vector2f best = {std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::quiet_NaN()};
        float best_distance = 0.0f;
        for (unsigned int c = 0; c < m_candidates; ++c) {
            Vector2f candidate;
            do { candidate.x = (i + 0.5f * (x_half + uniform())) / grid; } while (m_stratum_x[int(strata * candidate.x)] != EMPTY);
            do { candidate.y = (j + 0.5f * (y_half + uniform())) / grid; } while (m_stratum_y[int(strata * candidate.y)] != EMPTY);
            const int sx = int(strata * candidate.x), sy = int(strata * candidate.y);

            // nearest earlier point on the torus; walking outwards along the strata bounds the search
            float nearest = squared_distance(parent, candidate);
            int reach = int(strata * std::sqrt(nearest));
            auto consider = [&](uint16_t index) {
                if (index == EMPTY) return;
                Vector2f other = m_points[index];
                if (other.x < candidate.x - 0.5f) other.x += 1.0f; else if (other.x > candidate.x + 0.5f) other.x -= 1.0f;
                if (other.y < candidate.y - 0.5f) other.y += 1.0f; else if (other.y > candidate.y + 0.5f) other.y -= 1.0f;
                const float d = squared_distance(other, candidate);
                if (d < nearest) { nearest = d; reach = int(strata * std::sqrt(nearest)); }
            };
            for (int offset = 1; offset <= reach; ++offset) {
                consider(m_stratum_x[(sx + offset) % strata]);
                consider(m_stratum_x[(sx + strata - offset) % strata]);
                consider(m_stratum_y[(sy + offset) % strata]);
                consider(m_stratum_y[(sy + strata - offset) % strata]);
            }
            if (best_distance < nearest) { best_distance = nearest; best = candidate; }
        }
        m_stratum_x[int(strata * best.x)] = m_stratum_y[int(strata * best.y)] = uint16_t(m_placed);
        m_points[m_placed++] = best;
    }

    // count -> 2 * count points: every point gets a partner in the diagonally opposite sub-cell.
    void diagonal_pass(unsigned int count) {
        const int grid = int(std::sqrt(float(count)));
        rebuild_strata(count);
        for (unsigned int p = 0; p < count && m_placed < m_total; ++p) {
            const Vector2f parent = m_points[p];
            const int i = int(grid * parent.x), j = int(grid * parent.y);
            const int x_half = 1 - int(2 * (grid * parent.x - i)), y_half = 1 - int(2 * (grid * parent.y - j));
            place(parent, i, j, x_half, y_half, grid, int(count));
        }
    }

    // count -> 2 * count points (count is twice a square): one of the two remaining sub-cells at random, then the last one.
    void remaining_pass(unsigned int count) {
        const int grid = int(std::sqrt(float(count / 2)));
        rebuild_strata(count);
        for (unsigned int p = 0; p < count / 2 && m_placed < m_total; ++p) {
            const Vector2f parent = m_points[p];
            const int i = int(grid * parent.x), j = int(grid * parent.y);
            int x_half = int(2 * (grid * parent.x - i)), y_half = int(2 * (grid * parent.y - j));
            if (uniform() > 0.5f) x_half = 1 - x_half; else y_half = 1 - y_half;
            place(parent, i, j, x_half, y_half, grid, int(count));
        }
        for (unsigned int p = 0; p < count / 2 && m_placed < m_total; ++p) {
            const Vector2f parent = m_points[p + count];
            const int i = int(grid * parent.x), j = int(grid * parent.y);
            const int x_half = 1 - int(2 * (grid * parent.x - i)), y_half = 1 - int(2 * (grid * parent.y - j));
            place(parent, i, j, x_half, y_half, grid, int(count));
        }
    }

    Vector2f* m_points;
    unsigned int m_total, m_candidates, m_placed = 0;
    uint32_t m_state = 19349669u;
    std::vector<uint16_t> m_stratum_x, m_stratum_y;
};

} // namespace

void fill_progressive_multijittered_bluenoise_samples(Vector2f* begin, Vector2f* end, unsigned int blue_noise_samples) {
    const unsigned int total = unsigned(end - begin);
    if (total == 0) return;
    BlueNoisePointSet(begin, total, blue_noise_samples).generate();
}

} // namespace RNG
} // namespace Math
} // namespace Bifrost
