// host/BvhOptimizer.cpp -- see BvhOptimizer.h.
#include "BvhOptimizer.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <numeric>

namespace HIPRenderer {

namespace {

struct Box3 {
    float lo[3], hi[3];
    void grow(const Box3& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    bool operator==(const Box3& b) const { for (int a = 0; a < 3; ++a) if (lo[a] != b.lo[a] || hi[a] != b.hi[a]) return false; return true; }
};
inline Box3 merged(Box3 a, const Box3& b) { a.grow(b); return a; }

Box3 child_box(const HiprBvhNode& n, int c) {
    const float* xy = c == 0 ? n.c0xy : n.c1xy;
    return {{xy[0], xy[2], n.cz[2 * c]}, {xy[1], xy[3], n.cz[2 * c + 1]}};
}
void store_child(HiprBvhNode& n, int c, const Box3& b, int32_t ref) {
    float* xy = c == 0 ? n.c0xy : n.c1xy;
    xy[0] = b.lo[0]; xy[1] = b.hi[0]; xy[2] = b.lo[1]; xy[3] = b.hi[1];
    n.cz[2 * c] = b.lo[2]; n.cz[2 * c + 1] = b.hi[2];
    n.child[c] = ref;
}

// The binary tree with a box in every node and parent links: what the reinsertion works on. Node 0 is the root; a leaf keeps the BVH2's leaf reference.
struct Tree {
    struct Node {
        Box3 box;
        int32_t parent = -1, child[2] = {-1, -1};     // child[0] < 0: a leaf
        int32_t leaf_ref = 0;
        bool is_leaf() const { return child[0] < 0; }
    };
    std::vector<Node> nodes;

    int32_t sibling(int32_t n) const { const Node& p = nodes[size_t(nodes[size_t(n)].parent)]; return p.child[0] == n ? p.child[1] : p.child[0]; }
    float area(int32_t n) const { return nodes[size_t(n)].box.half_area(); }

    // SAH cost with unit node and leaf costs: the sum of the half-areas of all nodes but the root (each is paid when its parent is entered).
    double cost() const {
        double sum = 0.0;
        for (size_t i = 1; i < nodes.size(); ++i) sum += nodes[i].box.half_area();
        return sum;
    }

    void refit_upwards(int32_t n) {
        while (n >= 0) {
            Node& node = nodes[size_t(n)];
            const Box3 box = merged(nodes[size_t(node.child[0])].box, nodes[size_t(node.child[1])].box);
            if (box == node.box) return;      // nothing above can change either
            node.box = box;
            n = node.parent;
        }
    }

    // The best place to hang the subtree `x` instead of where it is (Bittner et al. 2013 as restated by the insertion search of Meister & Bittner 2018):
    // removing x deletes its parent (gain: the parent's area, and every ancestor above the PIVOT shrinks to the union of what is left below it);
    // inserting next to `to` adds a node of area(to U x) and grows every node between the pivot's other subtree and `to`. Returns the gain, 0 for none.
    float find_reinsertion(int32_t x, int32_t& best_to, std::vector<std::pair<float, int32_t>>& stack) const {
        const Node& X = nodes[size_t(x)];
        const float x_area = X.box.half_area();
        float best_gain = 0.0f;
        best_to = -1;
        const int32_t parent = X.parent;
        float gain_base = area(parent);
        int32_t explored = sibling(x), pivot = parent;
        Box3 pivot_box = nodes[size_t(explored)].box;
        for (;;) {
            stack.clear();
            stack.emplace_back(gain_base, explored);
            while (!stack.empty()) {
                const auto [gain, to] = stack.back();
                stack.pop_back();
                if (gain - x_area <= best_gain) continue;         // even a destination that x fits into entirely costs area(x)
                const Node& T = nodes[size_t(to)];
                const float here = gain - merged(T.box, X.box).half_area();
                if (here > best_gain) { best_gain = here; best_to = to; }
                if (!T.is_leaf()) {
                    const float below = here + T.box.half_area();   // `to` itself grows to (to U x) when x goes anywhere below it
                    stack.emplace_back(below, T.child[0]);
                    stack.emplace_back(below, T.child[1]);
                }
            }
            if (pivot != parent) {        // x leaves the pivot's subtree altogether: the pivot shrinks to what is left below it
                pivot_box.grow(nodes[size_t(explored)].box);
                gain_base += area(pivot) - pivot_box.half_area();
            }
            if (nodes[size_t(pivot)].parent < 0) break;
            explored = sibling(pivot);
            pivot = nodes[size_t(pivot)].parent;
        }
        if (best_to == sibling(x) || best_to == parent) { best_to = -1; return 0.0f; }      // where it already is
        return best_gain;
    }

    void reinsert(int32_t x, int32_t to) {
        const int32_t p = nodes[size_t(x)].parent, s = sibling(x), g = nodes[size_t(p)].parent;
        // take x and its parent out: the sibling moves up
        Node& G = nodes[size_t(g)];
        (G.child[0] == p ? G.child[0] : G.child[1]) = s;
        nodes[size_t(s)].parent = g;
        refit_upwards(g);
        // the parent node is reused above `to`
        const int32_t tp = nodes[size_t(to)].parent;
        Node& TP = nodes[size_t(tp)];
        (TP.child[0] == to ? TP.child[0] : TP.child[1]) = p;
        Node& P = nodes[size_t(p)];
        P.parent = tp;
        P.child[0] = to; P.child[1] = x;
        nodes[size_t(to)].parent = p;
        nodes[size_t(x)].parent = p;
        P.box = merged(nodes[size_t(to)].box, nodes[size_t(x)].box);
        refit_upwards(tp);
    }
};

} // namespace

ReinsertionStatistics optimise_by_reinsertion(std::vector<HiprBvhNode>& bvh, std::vector<uint32_t>& order, uint32_t depth_limit, uint32_t& deepest_leaf, int passes) {
    ReinsertionStatistics stats;
    if (bvh.size() < 4 || passes <= 0) return stats;
    // ---- the BVH2 as a tree with parent links (inner node i of the BVH2 is tree node i; the leaves follow)
    Tree tree;
    tree.nodes.resize(bvh.size());
    for (size_t i = 0; i < bvh.size(); ++i)
        for (int c = 0; c < 2; ++c) {
            const int32_t ref = bvh[i].child[c];
            int32_t index = ref;
            if (ref < 0) {
                index = int32_t(tree.nodes.size());
                tree.nodes.emplace_back();
                tree.nodes.back().leaf_ref = ref;
            }
            tree.nodes[size_t(index)].box = child_box(bvh[i], c);
            tree.nodes[size_t(index)].parent = int32_t(i);
            tree.nodes[i].child[c] = index;
        }
    tree.nodes[0].box = merged(tree.nodes[size_t(tree.nodes[0].child[0])].box, tree.nodes[size_t(tree.nodes[0].child[1])].box);
    stats.cost_before = tree.cost();

    // ---- passes: the candidates with the largest boxes first, each moved to the best place found for it at that moment
    std::vector<int32_t> candidates;
    std::vector<std::pair<float, int32_t>> stack;
    for (int pass = 0; pass < passes; ++pass) {
        candidates.clear();
        for (int32_t n = 1; n < int32_t(tree.nodes.size()); ++n)
            if (tree.nodes[size_t(n)].parent > 0) candidates.push_back(n);      // children of the root stay: their parent cannot be taken out
        std::stable_sort(candidates.begin(), candidates.end(), [&](int32_t a, int32_t b) { return tree.area(a) > tree.area(b); });
        size_t moved = 0;
        for (int32_t x : candidates) {
            if (tree.nodes[size_t(x)].parent <= 0) continue;       // an earlier move made it a child of the root
            int32_t to;
            const float gain = tree.find_reinsertion(x, to, stack);
            if (to < 0 || !(gain > 1e-6f * tree.area(0))) continue;
            // never below itself (the search cannot get there: it walks siblings of ancestors) and never next to its own parent (excluded above)
            tree.reinsert(x, to);
            ++moved;
        }
        stats.moves += moved;
        if (moved * 200 < candidates.size()) break;      // fewer than half a percent moved: converged
    }
    stats.cost_after = tree.cost();

    // ---- depth of the result; a tree deeper than the traversal stacks allow is not taken
    std::vector<uint32_t> depth(tree.nodes.size(), 0);
    uint32_t deepest = 0;
    {
        std::vector<int32_t> walk = {0};
        depth[0] = 1;
        while (!walk.empty()) {
            const int32_t n = walk.back();
            walk.pop_back();
            const Tree::Node& N = tree.nodes[size_t(n)];
            if (N.is_leaf()) { deepest = std::max(deepest, depth[size_t(n)]); continue; }      // root = 1, as the builder counts
            for (int c = 0; c < 2; ++c) { depth[size_t(N.child[c])] = depth[size_t(n)] + 1u; walk.push_back(N.child[c]); }
        }
    }
    stats.deepest_leaf = deepest;
    if (deepest > depth_limit || !(stats.cost_after < stats.cost_before)) { stats.taken = false; return stats; }

    // ---- back to the array: depth first, parent before children, the triangles in the order the leaves are met
    std::vector<HiprBvhNode> out;
    out.reserve(bvh.size());
    std::vector<uint32_t> new_order;
    new_order.reserve(order.size());
    struct Pending { int32_t node, parent_out; int side; };
    std::vector<Pending> pending = {{0, -1, 0}};
    while (!pending.empty()) {
        const Pending p = pending.back();
        pending.pop_back();
        const Tree::Node& N = tree.nodes[size_t(p.node)];
        int32_t ref;
        if (N.is_leaf()) {
            const uint32_t code = uint32_t(~N.leaf_ref), first = code >> 3, count = (code & 7u) + 1u;
            ref = ~int32_t((uint32_t(new_order.size()) << 3) | (count - 1u));
            for (uint32_t k = 0; k < count; ++k) new_order.push_back(order[first + k]);
        } else {
            ref = int32_t(out.size());
            out.emplace_back();
            out.back() = HiprBvhNode{};
            pending.push_back({N.child[1], ref, 1});     // the first child is laid out (and its triangles listed) first
            pending.push_back({N.child[0], ref, 0});
        }
        if (p.parent_out >= 0) store_child(out[size_t(p.parent_out)], p.side, N.box, ref);
    }
    bvh.swap(out);
    order.swap(new_order);
    deepest_leaf = deepest;
    stats.taken = true;
    return stats;
}

} // namespace HIPRenderer
