// Distribution2D.h -- piecewise constant 2-D distribution (marginal + conditional CDFs) for importance sampling images.
// Mirror of core/Bifrost/Bifrost/Math/Distribution2D.h:20-210 (same class surface: the constructor builds the CDFs, then
// sample_discrete / sample_continuous / PDF_discrete / PDF_continuous / evaluate), storage in std::vector.
#pragma once

#include "Math.h"

#include <vector>

namespace Bifrost {
namespace Math {

template <typename T>
class Distribution2D final {
public:
    template <typename I>
    struct Sample { I x, y; T PDF; };

    // function: width * height non-negative values, row major.
    template <typename U>
    Distribution2D(const U* function, int width, int height)
        : m_width(width), m_height(height), m_marginal_CDF(height + 1), m_conditional_CDF(size_t(width + 1) * height) {
        // running sums per row, then over the row totals
        for (int y = 0; y < height; ++y) {
            T* row = conditional_row(y);
            row[0] = T(0);
            for (int x = 0; x < width; ++x) row[x + 1] = row[x] + T(function[x + y * width]);
        }
        m_marginal_CDF[0] = T(0);
        for (int y = 0; y < height; ++y) m_marginal_CDF[y + 1] = m_marginal_CDF[y] + conditional_row(y)[width];
        m_integral = m_marginal_CDF[height] / (width * height);   // integral of the function over [0, 1)^2

        for (int y = 1; y < height; ++y) m_marginal_CDF[y] /= m_marginal_CDF[height];
        m_marginal_CDF[height] = T(1);
        for (int y = 0; y < height; ++y) {
            T* row = conditional_row(y);
            if (row[width] > T(0))
                for (int x = 1; x < width; ++x) row[x] /= row[width];
            row[width] = T(1);
        }
    }

    int get_width() const { return m_width; }
    int get_height() const { return m_height; }
    T get_integral() const { return m_integral; }
    const T* get_marginal_CDF() const { return m_marginal_CDF.data(); }
    int get_marginal_CDF_size() const { return m_height + 1; }
    const T* get_conditional_CDF() const { return m_conditional_CDF.data(); }   // (width + 1) entries per row

    T evaluate(int x, int y) const { return PDF_discrete(x, y) * m_width * m_height * m_integral; }
    T evaluate(Vector2f uv) const { return evaluate(int(uv.x * m_width), int(uv.y * m_height)); }

    Sample<int> sample_discrete(Vector2f random_sample) const {
        const int y = find_interval(random_sample.y, m_marginal_CDF.data(), m_height);
        const int x = find_interval(random_sample.x, conditional_row(y), m_width);
        return {x, y, PDF_discrete(x, y)};
    }

    Sample<float> sample_continuous(Vector2f random_sample) const {
        const int y = find_interval(random_sample.y, m_marginal_CDF.data(), m_height);
        const T dy = (random_sample.y - m_marginal_CDF[y]) / (m_marginal_CDF[y + 1] - m_marginal_CDF[y]);   // inverse lerp
        const T* row = conditional_row(y);
        const int x = find_interval(random_sample.x, row, m_width);
        const T dx = (random_sample.x - row[x]) / (row[x + 1] - row[x]);
        const T PDF = (m_marginal_CDF[y + 1] - m_marginal_CDF[y]) * (row[x + 1] - row[x]) * m_width * m_height;
        return {float(x + dx) / m_width, float(y + dy) / m_height, PDF};
    }

    T PDF_discrete(int x, int y) const {
        const T* row = conditional_row(y);
        return (m_marginal_CDF[y + 1] - m_marginal_CDF[y]) * (row[x + 1] - row[x]);
    }
    T PDF_continuous(Vector2f uv) const { return PDF_discrete(int(uv.x * m_width), int(uv.y * m_height)) * m_width * m_height; }

private:
    // Largest i in [0, count) with CDF[i] <= random_sample (binary search as the reference does it).
    static int find_interval(float random_sample, const T* CDF, int count) {
        int lower = 0, upper = count;
        while (lower + 1 != upper) {
            const int middle = (lower + upper) / 2;
            if (random_sample < CDF[middle]) upper = middle;
            else lower = middle;
        }
        return lower;
    }
    T* conditional_row(int y) { return m_conditional_CDF.data() + size_t(y) * (m_width + 1); }
    const T* conditional_row(int y) const { return m_conditional_CDF.data() + size_t(y) * (m_width + 1); }

    int m_width, m_height;
    T m_integral = T(0);
    std::vector<T> m_marginal_CDF, m_conditional_CDF;
};

} // namespace Math
} // namespace Bifrost
