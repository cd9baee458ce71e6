// host/BvhBuilder.cpp -- deterministic binned-SAH BVH2 builder for the HIP traversal kernels.
//
// Replaces the OptiX "Trbvh" acceleration build the reference requests at
// OptiXRenderer/Renderer.cpp:161-182,471-476 (closed source, so the tree itself is new design).
// Output: HiprBvhNode[] (64 B, both child boxes in the parent, Aila-Laine layout) in depth-first
// order with node 0 the root, and the triangle permutation in leaf order. Guarantees:
//   * every triangle is referenced by exactly one leaf, leaves hold 1..4 triangles;
//   * the deepest leaf is at most `max_depth` levels below the root (median splits take over when
//     the SAH tree would exceed the LDS stack of the kernels);
//   * the result depends only on the input order and values (single threaded, stable partition).
#include "BvhBuilder.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cmath>

namespace HIPRenderer {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int BIN_COUNT = 16;
// Triangles per leaf. The traversal kernels spend one loop iteration per node AND per triangle, so a leaf is worth splitting
// as long as the split culls triangles; HIPR_BVH_LEAF_SIZE overrides for experiments.
static uint32_t leaf_max() {
    static const uint32_t value = [] { const char* v = std::getenv("HIPR_BVH_LEAF_SIZE"); int n = v ? std::atoi(v) : 3; return uint32_t(n < 1 ? 1 : (n > 8 ? 8 : n)); }();
    return value;
}
#define LEAF_MAX leaf_max()

struct Builder {
    const std::vector<HiprTriangle>& tris;
    std::vector<Box> boxes;
    std::vector<float> centroids;   // 3 per triangle
    std::vector<uint32_t> order;
    std::vector<HiprBvhNode> nodes;
    uint32_t max_depth_limit;
    uint32_t deepest = 0;

    explicit Builder(const std::vector<HiprTriangle>& t, uint32_t limit) : tris(t), max_depth_limit(limit) {}

    Box bounds_of(uint32_t begin, uint32_t end) const {
        Box b; b.reset();
        for (uint32_t i = begin; i < end; ++i) b.grow(boxes[order[i]]);
        return b;
    }

    static uint32_t levels_needed(uint32_t count) {   // levels a balanced median tree needs below this node
        uint32_t leaves = (count + LEAF_MAX - 1) / LEAF_MAX, levels = 0;
        while ((1u << levels) < leaves) ++levels;
        return levels;
    }

    // Returns the split position in [begin+1, end-1].
    uint32_t split(uint32_t begin, uint32_t end, uint32_t depth) {
        const uint32_t count = end - begin;
        Box cb; cb.reset();
        for (uint32_t i = begin; i < end; ++i) cb.grow(&centroids[3 * order[i]]);

        // Depth budget: once only enough levels remain for a balanced tree, split at the median.
        const bool force_median = depth + levels_needed(count) >= max_depth_limit;

        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        if (!force_median) {
            for (int axis = 0; axis < 3; ++axis) {
                const float extent = cb.hi[axis] - cb.lo[axis];
                if (!(extent > 0.0f)) continue;
                Box bin_box[BIN_COUNT];
                uint32_t bin_n[BIN_COUNT] = {};
                for (auto& b : bin_box) b.reset();
                const float scale = BIN_COUNT / extent;
                for (uint32_t i = begin; i < end; ++i) {
                    const uint32_t t = order[i];
                    int b = int((centroids[3 * t + axis] - cb.lo[axis]) * scale);
                    b = std::min(std::max(b, 0), BIN_COUNT - 1);
                    bin_box[b].grow(boxes[t]);
                    bin_n[b]++;
                }
                float right_area[BIN_COUNT];
                uint32_t right_n[BIN_COUNT];
                Box acc; acc.reset();
                uint32_t n = 0;
                for (int b = BIN_COUNT - 1; b > 0; --b) {
                    if (bin_n[b]) acc.grow(bin_box[b]);
                    n += bin_n[b];
                    right_area[b] = n ? acc.half_area() : 0.0f;
                    right_n[b] = n;
                }
                acc.reset();
                n = 0;
                for (int b = 0; b < BIN_COUNT - 1; ++b) {
                    if (bin_n[b]) acc.grow(bin_box[b]);
                    n += bin_n[b];
                    if (n == 0 || right_n[b + 1] == 0) continue;
                    const float cost = acc.half_area() * float(n) + right_area[b + 1] * float(right_n[b + 1]);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }

        if (best_axis >= 0) {
            const float extent = cb.hi[best_axis] - cb.lo[best_axis];
            const float scale = BIN_COUNT / extent;
            const float lo = cb.lo[best_axis];
            auto mid = std::stable_partition(order.begin() + begin, order.begin() + end, [&](uint32_t t) {
                int b = int((centroids[3 * t + best_axis] - lo) * scale);
                b = std::min(std::max(b, 0), BIN_COUNT - 1);
                return b <= best_bin;
            });
            const uint32_t m = uint32_t(mid - order.begin());
            if (m > begin && m < end) return m;
        }

        // Median split along the widest centroid axis (also the fallback for coincident centroids).
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
        const uint32_t m = begin + count / 2;
        std::stable_sort(order.begin() + begin, order.begin() + end,
                         [&](uint32_t a, uint32_t b) { return centroids[3 * a + axis] < centroids[3 * b + axis]; });
        return m;
    }

    static int32_t leaf_ref(uint32_t first, uint32_t count) { return ~int32_t((first << 3) | (count - 1)); }

    static void store_child(HiprBvhNode& n, int c, const Box& b, int32_t ref) {
        float* xy = c == 0 ? n.c0xy : n.c1xy;
        xy[0] = b.lo[0]; xy[1] = b.hi[0]; xy[2] = b.lo[1]; xy[3] = b.hi[1];
        n.cz[2 * c] = b.lo[2]; n.cz[2 * c + 1] = b.hi[2];
        n.child[c] = ref;
    }

    // Builds the node covering order[begin, end) (count > LEAF_MAX) and returns its index.
    uint32_t build(uint32_t begin, uint32_t end, uint32_t depth) {
        const uint32_t index = uint32_t(nodes.size());
        nodes.emplace_back();
        HiprBvhNode scratch = {};
        const uint32_t mid = split(begin, end, depth);
        const uint32_t range[2][2] = {{begin, mid}, {mid, end}};
        for (int c = 0; c < 2; ++c) {
            const uint32_t b = range[c][0], e = range[c][1];
            const Box box = bounds_of(b, e);
            if (e - b <= LEAF_MAX) {
                store_child(scratch, c, box, leaf_ref(b, e - b));
                deepest = std::max(deepest, depth + 1);
            } else
                store_child(scratch, c, box, int32_t(build(b, e, depth + 1)));
        }
        nodes[index] = scratch;
        return index;
    }
};

// The quantised child boxes of a wide node: origin = the low corner of the union, one power-of-two grid per axis, 8 bit bounds rounded
// outwards (checked in f64) so that the decoded boxes contain the exact ones. Shared by the build and the refit.
static void quantise_children(const Box* boxes, size_t count, HiprWideNode& w) {
    Box all; all.reset();
    for (size_t k = 0; k < count; ++k) all.grow(boxes[k]);
    uint32_t q[2][3] = {{0, 0, 0}, {0, 0, 0}};
    w.exponents = 0;
    for (int a = 0; a < 3; ++a) {
        w.origin[a] = all.lo[a];
        const double origin = all.lo[a], extent = double(all.hi[a]) - origin;
        int e = extent > 0.0 ? int(std::ceil(std::log2(extent / 255.0))) : -126;
        e = std::max(e, -126);
        for (;; ++e) {   // grow the grid until every bound fits in [0, 255] after rounding outwards
            const double scale = std::ldexp(1.0, e);
            bool fits = true;
            uint32_t qa[2] = {0, 0};
            for (size_t k = 0; k < count && fits; ++k) {
                double lo = std::floor((double(boxes[k].lo[a]) - origin) / scale), hi = std::ceil((double(boxes[k].hi[a]) - origin) / scale);
                while (lo > 0.0 && origin + lo * scale > double(boxes[k].lo[a])) lo -= 1.0;
                while (origin + hi * scale < double(boxes[k].hi[a])) hi += 1.0;
                lo = std::max(lo, 0.0);
                if (hi > 255.0) { fits = false; break; }
                qa[0] |= uint32_t(lo) << (8 * k);
                qa[1] |= uint32_t(hi) << (8 * k);
            }
            if (fits) { q[0][a] = qa[0]; q[1][a] = qa[1]; break; }
        }
        w.exponents |= uint32_t(e + 127) << (8 * a);
    }
    for (int a = 0; a < 3; ++a) { w.qlo[a] = q[0][a]; w.qhi[a] = q[1][a]; }
}

// ---------------------------------------------------------------------------------------------
// Collapse to compressed 4-wide nodes: a node adopts the children of its largest inner child until it has four, then the
// child boxes are quantised to 8 bits per bound on the node's own grid, rounding outwards.
// ---------------------------------------------------------------------------------------------
struct WideCollapse {
    const std::vector<HiprBvhNode>& nodes;
    std::vector<HiprWideNode> wide;
    std::vector<uint32_t> binary_need;   // stack entries the BVH2 subtree of a node needs when nothing below it is widened

    uint32_t compute_binary_need(uint32_t node_index) {
        uint32_t need = 0;
        for (int c = 0; c < 2; ++c) {
            const int32_t ref = nodes[node_index].child[c];
            need = std::max(need, 1u + (ref >= 0 ? compute_binary_need(uint32_t(ref)) : 0u));
        }
        return binary_need[node_index] = need;
    }
    uint32_t need_of(int32_t ref) const { return ref >= 0 ? binary_need[ref] : 0u; }

    struct Child { Box box; int32_t ref; };

    static Child child_of(const HiprBvhNode& n, int c) {
        Child r;
        const float* xy = c == 0 ? n.c0xy : n.c1xy;
        r.box.lo[0] = xy[0]; r.box.hi[0] = xy[1]; r.box.lo[1] = xy[2]; r.box.hi[1] = xy[3];
        r.box.lo[2] = n.cz[2 * c]; r.box.hi[2] = n.cz[2 * c + 1];
        r.ref = n.child[c];
        return r;
    }

    // Returns the wide node index and, through `stack_need`, the worst case number of stack entries below it. `budget`: the
    // entries this subtree may need. A node with c children keeps c - 1 entries on the stack while a child is traversed, and a
    // subtree can always fall back to binary nodes (binary_need), so a node is only widened while every child still fits.
    uint32_t collapse(uint32_t node_index, uint32_t budget, uint32_t& stack_need) {
        std::vector<Child> children = {child_of(nodes[node_index], 0), child_of(nodes[node_index], 1)};
        if (children[0].ref == children[1].ref && children[0].ref < 0) children.pop_back();   // the single-leaf root references its leaf twice
        while (children.size() < 4) {
            int widest = -1;
            float widest_area = -1.0f;
            for (size_t i = 0; i < children.size(); ++i)
                if (children[i].ref >= 0 && children[i].box.half_area() > widest_area) { widest = int(i); widest_area = children[i].box.half_area(); }
            if (widest < 0) break;
            const HiprBvhNode& inner = nodes[children[widest].ref];
            const uint32_t held = uint32_t(children.size());   // entries held once this node has one more child
            bool fits = held + std::max(need_of(inner.child[0]), need_of(inner.child[1])) <= budget;
            for (size_t i = 0; i < children.size() && fits; ++i)
                if (int(i) != widest) fits = held + need_of(children[i].ref) <= budget;
            if (!fits) break;
            children[widest] = child_of(inner, 0);
            children.insert(children.begin() + widest + 1, child_of(inner, 1));
        }

        const uint32_t index = uint32_t(wide.size());
        wide.emplace_back();
        HiprWideNode w = {};
        Box child_boxes[4];
        for (size_t k = 0; k < children.size(); ++k) child_boxes[k] = children[k].box;
        quantise_children(child_boxes, children.size(), w);

        stack_need = 0;
        for (int k = 0; k < 4; ++k) w.child[k] = HIPR_WIDE_EMPTY;
        for (size_t k = 0; k < children.size(); ++k) {
            uint32_t below = 0;
            const uint32_t held = uint32_t(children.size() - 1);
            w.child[k] = children[k].ref >= 0 ? int32_t(collapse(uint32_t(children[k].ref), budget > held ? budget - held : 0u, below)) : children[k].ref;
            stack_need = std::max(stack_need, uint32_t(children.size() - 1) + below);
        }
        wide[index] = w;
        return index;
    }
};

} // namespace

BvhBuildResult build_bvh(const std::vector<HiprTriangle>& triangles, uint32_t max_depth) {
    BvhBuildResult result;
    const uint32_t n = uint32_t(triangles.size());
    result.max_depth = 0;
    if (n == 0) return result;

    Builder b(triangles, std::max(max_depth, 8u));
    b.boxes.resize(n);
    b.centroids.resize(3 * size_t(n));
    b.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const HiprTriangle& t = triangles[i];
        Box& box = b.boxes[i];
        box.reset();
        box.grow(t.v0); box.grow(t.v1); box.grow(t.v2);
        for (int a = 0; a < 3; ++a) b.centroids[3 * i + a] = 0.5f * (box.lo[a] + box.hi[a]);
        b.order[i] = i;
    }

    if (n <= LEAF_MAX) {
        // A single leaf: both children reference it (the duplicate test cannot change the result).
        HiprBvhNode root = {};
        Box box = b.bounds_of(0, n);
        Builder::store_child(root, 0, box, Builder::leaf_ref(0, n));
        Builder::store_child(root, 1, box, Builder::leaf_ref(0, n));
        b.nodes.push_back(root);
        b.deepest = 1;
    } else {
        b.nodes.reserve(n);
        b.build(0, n, 1);
    }

    result.nodes = std::move(b.nodes);
    result.order = std::move(b.order);
    WideCollapse collapse{result.nodes, {}, {}};
    collapse.wide.reserve(result.nodes.size() / 2 + 1);
    collapse.binary_need.assign(result.nodes.size(), 0u);
    // Keep the worst case within the 32 entry LDS stack of the traversal kernels whenever the BVH2 itself allows it; deeper
    // trees are collapsed freely and traversed by the kernels that back the LDS stack with scratch memory.
    const uint32_t budget = collapse.compute_binary_need(0) <= 32u ? 32u : 0xFFFFu;
    collapse.collapse(0, budget, result.wide_stack_entries);
    result.wide_nodes = std::move(collapse.wide);
    result.max_depth = b.deepest + 1;   // stack entries needed is bounded by the node depth; keep one spare
    return result;
}

// Refit: same topology and triangle order, new triangle positions (a transform-only scene update; the reference refits its root
// acceleration structure the same way, OR/Renderer.cpp:472,1010-1041). Nodes are stored parent before children, so one sweep from the
// last node to the first sees every child box up to date. Returns the sum of the BVH2 child box half-areas, a SAH-style quality
// figure the caller compares with the build's to decide when a moved scene deserves a new tree.
double refit_bvh(BvhBuildResult& bvh, const std::vector<HiprTriangle>& triangles) {
    auto triangle_bounds = [&](int32_t leaf) {
        const uint32_t code = uint32_t(~leaf), first = code >> 3, count = (code & 7u) + 1u;
        Box b; b.reset();
        for (uint32_t t = first; t < first + count; ++t) { b.grow(triangles[t].v0); b.grow(triangles[t].v1); b.grow(triangles[t].v2); }
        return b;
    };
    auto node_bounds = [&](const HiprBvhNode& n) {
        Box b;
        b.lo[0] = std::min(n.c0xy[0], n.c1xy[0]); b.hi[0] = std::max(n.c0xy[1], n.c1xy[1]);
        b.lo[1] = std::min(n.c0xy[2], n.c1xy[2]); b.hi[1] = std::max(n.c0xy[3], n.c1xy[3]);
        b.lo[2] = std::min(n.cz[0], n.cz[2]); b.hi[2] = std::max(n.cz[1], n.cz[3]);
        return b;
    };
    double area = 0.0;
    for (size_t i = bvh.nodes.size(); i-- > 0;) {
        HiprBvhNode& n = bvh.nodes[i];
        for (int c = 0; c < 2; ++c) {
            const int32_t ref = n.child[c];
            const Box box = ref < 0 ? triangle_bounds(ref) : node_bounds(bvh.nodes[size_t(ref)]);
            Builder::store_child(n, c, box, ref);
            area += box.half_area();
        }
    }
    std::vector<Box> exact(bvh.wide_nodes.size());
    for (size_t i = bvh.wide_nodes.size(); i-- > 0;) {
        HiprWideNode& w = bvh.wide_nodes[i];
        Box boxes[4];
        size_t count = 0;
        int32_t refs[4];
        for (int k = 0; k < 4; ++k) {
            if (w.child[k] == HIPR_WIDE_EMPTY) continue;
            refs[count] = w.child[k];
            boxes[count++] = w.child[k] < 0 ? triangle_bounds(w.child[k]) : exact[size_t(w.child[k])];
        }
        exact[i].reset();
        for (size_t k = 0; k < count; ++k) exact[i].grow(boxes[k]);
        quantise_children(boxes, count, w);      // the non-empty children occupy the first `count` slots, as the build left them
        (void)refs;
    }
    return area;
}

double bvh_child_area(const BvhBuildResult& bvh) {
    double area = 0.0;
    for (const HiprBvhNode& n : bvh.nodes) {
        Box a, b;
        a.lo[0] = n.c0xy[0]; a.hi[0] = n.c0xy[1]; a.lo[1] = n.c0xy[2]; a.hi[1] = n.c0xy[3]; a.lo[2] = n.cz[0]; a.hi[2] = n.cz[1];
        b.lo[0] = n.c1xy[0]; b.hi[0] = n.c1xy[1]; b.lo[1] = n.c1xy[2]; b.hi[1] = n.c1xy[3]; b.lo[2] = n.cz[2]; b.hi[2] = n.cz[3];
        area += a.half_area() + b.half_area();
    }
    return area;
}

} // namespace HIPRenderer
