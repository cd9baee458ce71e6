// host/Wide8Builder.cpp -- the 8-wide compressed BVH the persistent traversal kernels walk (include/hiprenderer_c.h "wide8").
//
// Replaces, like BvhBuilder.cpp, the OptiX "Trbvh" acceleration build of OptiXRenderer/Renderer.cpp:161-182,471-476 (closed source: the tree is new
// design). Input: the binned-SAH BVH2 and the triangles in its leaf order. Output: one array of 64-byte slots,
//   * leaf records: the triangles of a BVH2 leaf, paired where two of them (same instance) share an edge with bit-identical corners -- the record
//     stores the shared corner a and the edges to the other three corners, so both triangles are tested from 64 bytes (HiprLeaf8);
//   * inner nodes of up to eight children (HiprNode8): a node adopts the children of its largest inner child while they fit; the children are dealt to
//     the eight positions so that position bit `axis` says on which side of the node's centre the child lies (greedy assignment on the signed centroid
//     offsets, as in Ylitie et al., "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017), which is what lets a ray
//     order the children by XOR-ing the position with its direction's octant instead of sorting distances;
//   * child boxes quantised to 8 bits per bound on the node's own power-of-two grid, rounded outwards (checked in f64), the node origin snapped down
//     to a 21-bit grid over the scene bounds.
// Slots are laid out depth first with the children of a node contiguous; a child's slot is always greater than its parent's (the refit sweeps backwards).
// Deterministic: the result depends only on the input.
#include "Wide8Builder.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace HIPRenderer {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

inline const float* corner(const HiprTriangle& t, int k) { return k % 3 == 0 ? t.v0 : (k % 3 == 1 ? t.v1 : t.v2); }
inline bool same_point(const float* p, const float* q) { return std::memcmp(p, q, 12) == 0; }

Box triangle_box(const HiprTriangle& t) {
    Box b; b.reset();
    b.grow(t.v0); b.grow(t.v1); b.grow(t.v2);
    return b;
}

// The record of triangle A alone (index_b == HIPR_LEAF8_NONE) or of A and B, which must share exactly two bit-identical corners. `rotation_a`: the
// record's first corner is A's vertex `rotation_a` (A is stored rotated, never mirrored); B's corners are matched by position.
bool make_record(const std::vector<HiprTriangle>& triangles, uint32_t index_a, uint32_t index_b, int rotation_a, HiprLeaf8& out) {
    const HiprTriangle& A = triangles[index_a];
    const float *a = corner(A, rotation_a), *b = corner(A, rotation_a + 1), *c = corner(A, rotation_a + 2);
    out = {};
    for (int k = 0; k < 3; ++k) { out.a[k] = a[k]; out.e1[k] = b[k] - a[k]; out.e2[k] = c[k] - a[k]; }
    out.triangle[0] = index_a;
    out.triangle[1] = HIPR_LEAF8_NONE;
    // record corner k of A is A's vertex (rotation + k) % 3, so A's vertex j is record corner (j - rotation) mod 3: its weight is (w, u, v)[that]
    uint32_t flags = (A.flags & HIPR_TRIANGLE_OPAQUE ? 1u : 0u) | uint32_t((1 - rotation_a + 3) % 3) << 8 | uint32_t((2 - rotation_a + 3) % 3) << 10;
    if (index_b != HIPR_LEAF8_NONE) {
        const HiprTriangle& B = triangles[index_b];
        int where[3] = {-1, -1, -1};      // record corner (0 = a, 1 = c, 2 = d) of B's vertex j
        const float* d = nullptr;
        for (int j = 0; j < 3; ++j) {
            if (same_point(corner(B, j), a)) where[j] = 0;
            else if (same_point(corner(B, j), c)) where[j] = 1;
            else { where[j] = 2; d = corner(B, j); }
        }
        if (!d || where[0] == where[1] || where[0] == where[2] || where[1] == where[2]) return false;   // not two shared corners + one own
        for (int k = 0; k < 3; ++k) out.e3[k] = d[k] - a[k];
        out.triangle[1] = index_b;
        flags |= (B.flags & HIPR_TRIANGLE_OPAQUE ? 2u : 0u) | uint32_t(where[1]) << 12 | uint32_t(where[2]) << 14;
    }
    out.flags = flags;
    return true;
}

// The binary tree the collapse works on: the BVH2 with every leaf turned into its records (a leaf of several records becomes a short chain), so
// that a leaf of THIS tree is exactly one record. Numbered parent before children.
struct TreeNode {
    Box box;
    int32_t left = -1, right = -1;      // both -1: a leaf
    int32_t record = -1;                // leaf: index into Collapse::records
};

struct Collapse {
    const std::vector<HiprBvhNode>& nodes;
    const std::vector<HiprTriangle>& triangles;
    Wide8Result& out;
    bool overflow = false;
    std::vector<TreeNode> tree;
    std::vector<HiprLeaf8> records;
    Collapse(const std::vector<HiprBvhNode>& n, const std::vector<HiprTriangle>& t, Wide8Result& o) : nodes(n), triangles(t), out(o) {}
    // Ylitie et al. 2017, section 3.1: cost[n][i - 1] = the lowest SAH cost of the subtree of n when it appears in its parent wide node as a forest of at
    // most i roots (i = 1 .. 7); a node that becomes a wide node itself hands its two subtrees up to 8 roots in total. split[n][j - 2] = how many of j roots
    // go to the left subtree in the best distribution; roots_used[n][i - 1] = the i' <= i at which cost[n][i - 1] is attained (1 = the node itself).
    static constexpr float NODE_COST = 1.0f;      // a node visit is four loads + ~230 instructions, a leaf record four loads + ~130
    const float LEAF_COST = [] { const char* v = std::getenv("HIPR_WIDE8_LEAF_COST"); return v ? float(std::atof(v)) : 0.6f; }();
    std::vector<float> cost;
    std::vector<uint8_t> split, roots_used;

    static Box child_box(const HiprBvhNode& n, int c) {
        Box b;
        const float* xy = c == 0 ? n.c0xy : n.c1xy;
        b.lo[0] = xy[0]; b.hi[0] = xy[1]; b.lo[1] = xy[2]; b.hi[1] = xy[3];
        b.lo[2] = n.cz[2 * c]; b.hi[2] = n.cz[2 * c + 1];
        return b;
    }

    // The records of a BVH2 leaf (at most 8): every triangle pairs with the first later one of the same instance that shares exactly two corners with it.
    // The records go to `records`; `leaves` receives one tree leaf per record. Returns their number.
    size_t records_of_leaf(int32_t ref, TreeNode* leaves) {
        const uint32_t code = uint32_t(~ref), first = code >> 3, count = (code & 7u) + 1u;
        bool used[8] = {false, false, false, false, false, false, false, false};
        size_t made = 0;
        for (uint32_t i = 0; i < count; ++i) {
            if (used[i]) continue;
            used[i] = true;
            const HiprTriangle& A = triangles[first + i];
            TreeNode& leaf = leaves[made++];
            leaf = TreeNode();
            HiprLeaf8 record;
            bool paired = false;
            for (uint32_t j = i + 1; j < count && !paired; ++j) {
                if (used[j] || triangles[first + j].instance_index != A.instance_index) continue;
                const HiprTriangle& B = triangles[first + j];
                int shared = 0, own_a = -1;
                for (int x = 0; x < 3; ++x) {
                    bool found = false;
                    for (int y = 0; y < 3; ++y) found = found || same_point(corner(A, x), corner(B, y));
                    if (found) ++shared; else own_a = x;
                }
                if (shared != 2) continue;
                // A as (a, b, c) with b its own corner: a = the vertex before b, c = the one after
                if (make_record(triangles, first + i, first + j, (own_a + 2) % 3, record)) { used[j] = true; paired = true; leaf.box = triangle_box(A); leaf.box.grow(triangle_box(B)); }
            }
            if (!paired) { make_record(triangles, first + i, HIPR_LEAF8_NONE, 0, record); leaf.box = triangle_box(A); }
            leaf.record = int32_t(records.size());
            records.push_back(record);
        }
        return made;
    }

    // Appends the subtree of a BVH2 child reference (inner node or leaf) and returns its index.
    int32_t add_subtree(int32_t ref, const Box& box) {
        // explicit stack (BVH2 trees can be deep); children are linked to their parent by index, `tree` grows underneath
        int32_t result = -1;
        struct Pending { int32_t ref; Box box; int32_t parent; int side; };
        std::vector<Pending> pending = {{ref, box, -1, 0}};
        while (!pending.empty()) {
            const Pending p = pending.back();
            pending.pop_back();
            int32_t index;
            if (p.ref >= 0) {
                index = int32_t(tree.size());
                tree.emplace_back();
                tree[size_t(index)].box = p.box;
                const HiprBvhNode& n = nodes[size_t(p.ref)];
                // right first so that the left subtree is numbered (and later laid out) first
                pending.push_back({n.child[1], child_box(n, 1), index, 1});
                pending.push_back({n.child[0], child_box(n, 0), index, 0});
            } else {
                TreeNode leaves[8];
                const size_t r = records_of_leaf(p.ref, leaves);
                // r records -> a chain: ((r0, r1), r2) ... numbered parent first
                index = int32_t(tree.size());
                if (r == 1) tree.push_back(leaves[0]);
                else {
                    // inner nodes of the chain, outermost first
                    Box prefix[8];
                    prefix[0] = leaves[0].box;
                    for (size_t k = 1; k < r; ++k) { prefix[k] = prefix[k - 1]; prefix[k].grow(leaves[k].box); }
                    int32_t parent = -1;
                    for (size_t k = r - 1; k >= 1; --k) {     // node covering records 0..k: left = node covering 0..k-1 (or record 0), right = record k
                        const int32_t inner = int32_t(tree.size());
                        tree.emplace_back();
                        tree[size_t(inner)].box = prefix[k];
                        if (parent >= 0) tree[size_t(parent)].left = inner;
                        const int32_t right_leaf = int32_t(tree.size());
                        tree.push_back(leaves[k]);
                        tree[size_t(inner)].right = right_leaf;
                        parent = inner;
                    }
                    const int32_t first_leaf = int32_t(tree.size());
                    tree.push_back(leaves[0]);
                    tree[size_t(parent)].left = first_leaf;
                }
            }
            if (p.parent >= 0) (p.side == 0 ? tree[size_t(p.parent)].left : tree[size_t(p.parent)].right) = index;
            else result = index;
        }
        return result;
    }

    void build_tree() {
        tree.reserve(triangles.size() + triangles.size() / 8 + 16);
        records.reserve(triangles.size() / 2 + triangles.size() / 8 + 16);
        const HiprBvhNode& root = nodes[0];
        if (root.child[0] == root.child[1] && root.child[0] < 0) {      // the single-leaf root references its leaf twice
            Box b = child_box(root, 0);
            add_subtree(root.child[0], b);
        } else {
            Box all = child_box(root, 0);
            all.grow(child_box(root, 1));
            add_subtree(0, all);
        }
    }

    float& cost_of(size_t n, int i) { return cost[7 * n + size_t(i - 1)]; }
    void optimise() {
        const size_t count = tree.size();
        cost.assign(7 * count, 0.0f);
        split.assign(7 * count, 0);
        roots_used.assign(7 * count, 1);
        for (size_t n = count; n-- > 0;) {      // children are numbered after their parent
            const TreeNode& t = tree[n];
            const float area = t.box.half_area();
            if (t.left < 0) {
                for (int i = 1; i <= 7; ++i) cost_of(n, i) = area * LEAF_COST;
                continue;
            }
            const size_t l = size_t(t.left), r = size_t(t.right);
            float distribute[9];     // [j], j = 2 .. 8
            for (int j = 2; j <= 8; ++j) {
                float best = FLT_MAX;
                int best_k = 1;
                for (int k = 1; k < j; ++k) {
                    const float c = cost_of(l, std::min(k, 7)) + cost_of(r, std::min(j - k, 7));
                    if (c < best) { best = c; best_k = k; }
                }
                distribute[j] = best;
                split[7 * n + size_t(j - 2)] = uint8_t(best_k);
            }
            cost_of(n, 1) = area * NODE_COST + distribute[8];
            roots_used[7 * n] = 1;
            for (int i = 2; i <= 7; ++i) {
                if (distribute[i] < cost_of(n, i - 1)) { cost_of(n, i) = distribute[i]; roots_used[7 * n + size_t(i - 1)] = uint8_t(i); }
                else { cost_of(n, i) = cost_of(n, i - 1); roots_used[7 * n + size_t(i - 1)] = roots_used[7 * n + size_t(i - 2)]; }
            }
        }
    }
    // The roots (tree node indices) that represent the subtree of n in a parent wide node that grants it at most `allowance` positions.
    void collect_roots(int32_t n, int allowance, std::vector<int32_t>& roots) const {
        const TreeNode& t = tree[size_t(n)];
        if (t.left < 0) { roots.push_back(n); return; }
        const int used = roots_used[7 * size_t(n) + size_t(std::min(allowance, 7) - 1)];
        if (used <= 1) { roots.push_back(n); return; }
        const int k = split[7 * size_t(n) + size_t(used - 2)];
        collect_roots(t.left, k, roots);
        collect_roots(t.right, used - k, roots);
    }

    // Deals the children to the eight positions: position bit `axis` set = towards +axis of the node's centre. Greedy on the signed centroid offsets.
    void assign_positions(const std::vector<int32_t>& children, const Box& all, int position_of[8]) const {
        float cost_matrix[8][8];
        for (size_t c = 0; c < children.size(); ++c) {
            const Box& box = tree[size_t(children[c])].box;
            float d[3];
            for (int a = 0; a < 3; ++a) d[a] = 0.5f * (box.lo[a] + box.hi[a]) - 0.5f * (all.lo[a] + all.hi[a]);
            for (int s = 0; s < 8; ++s) cost_matrix[c][s] = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
        }
        bool child_done[8] = {false, false, false, false, false, false, false, false}, position_taken[8] = {false, false, false, false, false, false, false, false};
        for (size_t round = 0; round < children.size(); ++round) {
            int best_c = -1, best_s = -1;
            for (size_t c = 0; c < children.size(); ++c) {
                if (child_done[c]) continue;
                for (int s = 0; s < 8; ++s)
                    if (!position_taken[s] && (best_c < 0 || cost_matrix[c][s] > cost_matrix[best_c][best_s])) { best_c = int(c); best_s = s; }
            }
            child_done[best_c] = true; position_taken[best_s] = true;
            position_of[best_c] = best_s;
        }
    }

    // Writes the wide node of tree node n into `slot` and its subtree behind the current end of the array; returns the height below (and including) it.
    // `queue` (breadth-first layout): the inner children are not descended into here but appended for the caller's loop.
    struct Queued { uint32_t slot; int32_t node; uint32_t depth; };
    uint32_t emit_node(uint32_t slot, int32_t n, uint32_t depth, std::vector<Queued>* queue = nullptr) {
        std::vector<int32_t> children;
        const TreeNode& t = tree[size_t(n)];
        if (t.left < 0) children.push_back(n);      // a scene of a single record: the root node holds it
        else {
            const int k = split[7 * size_t(n) + 6];     // the distribution of 8 roots
            collect_roots(t.left, k, children);
            collect_roots(t.right, 8 - k, children);
        }

        Box all; all.reset();
        for (int32_t c : children) all.grow(tree[size_t(c)].box);
        int position_of[8];
        assign_positions(children, all, position_of);
        int child_at[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        for (size_t c = 0; c < children.size(); ++c) child_at[position_of[c]] = int(c);

        const size_t base = out.slots.size();
        if (base + children.size() > 0xFFFFFFu) { overflow = true; return depth; }
        out.slots.resize(base + children.size());
        HiprNode8 node = {};
        Box boxes[8];
        uint32_t valid = 0, inner_mask = 0;
        for (int s = 0; s < 8; ++s) {
            if (child_at[s] < 0) continue;
            valid |= 1u << s;
            const TreeNode& child = tree[size_t(children[size_t(child_at[s])])];
            boxes[s] = child.box;
            if (child.left >= 0) inner_mask |= 1u << s;
        }
        node.inner_mask = uint8_t(inner_mask);
        node.base_valid = uint32_t(base) | valid << 24;
        quantise(boxes, valid, all, node);
        out.slots[slot].node = node;
        out.node_count += 1;

        uint32_t height = depth;
        uint32_t next = uint32_t(base);
        for (int s = 0; s < 8 && !overflow; ++s) {
            if (child_at[s] < 0) continue;
            const int32_t child = children[size_t(child_at[s])];
            const uint32_t child_slot = next++;
            if (tree[size_t(child)].left >= 0) {
                if (queue) { queue->push_back({child_slot, child, depth + 1}); height = std::max(height, depth + 1); }
                else height = std::max(height, emit_node(child_slot, child, depth + 1));
            } else {
                out.slots[child_slot].leaf = records[size_t(tree[size_t(child)].record)];
                out.leaf_count += 1;
                out.paired_leaves += out.slots[child_slot].leaf.triangle[1] != HIPR_LEAF8_NONE;
            }
        }
        return height;
    }

    // Origin on the scene grid (snapped down), one power-of-two grid per axis, bounds rounded outwards; empty positions get qlo = 255, qhi = 0.
    void quantise(const Box* boxes, uint32_t valid, const Box& all, HiprNode8& node) const { quantise_node(boxes, valid, all, out.grid_min, out.grid_cell, node); }

    static void quantise_node(const Box* boxes, uint32_t valid, const Box& all, const float* grid_min, const float* grid_cell, HiprNode8& node) {
        unsigned long long packed_origin = 0;
        for (int a = 0; a < 3; ++a) {
            double m = std::floor((double(all.lo[a]) - double(grid_min[a])) / double(grid_cell[a]));
            m = std::min(std::max(m, 0.0), 2097151.0);
            while (m > 0.0 && std::fmaf(float(m), grid_cell[a], grid_min[a]) > all.lo[a]) m -= 1.0;
            packed_origin |= (unsigned long long)(m) << (21 * a);
            const float origin_f = std::fmaf(float(m), grid_cell[a], grid_min[a]);      // what the traversal computes
            const double origin = origin_f, extent = double(all.hi[a]) - origin;
            int e = extent > 0.0 ? int(std::ceil(std::log2(extent / 255.0))) : -126;
            e = std::min(std::max(e, -126), 127);
            for (;; ++e) {
                const double scale = std::ldexp(1.0, e);
                bool fits = true;
                uint8_t lo8[8], hi8[8];
                for (int s = 0; s < 8 && fits; ++s) {
                    if (!(valid >> s & 1u)) { lo8[s] = 255; hi8[s] = 0; continue; }
                    double lo = std::floor((double(boxes[s].lo[a]) - origin) / scale), hi = std::ceil((double(boxes[s].hi[a]) - origin) / scale);
                    while (lo > 0.0 && origin + lo * scale > double(boxes[s].lo[a])) lo -= 1.0;
                    while (origin + hi * scale < double(boxes[s].hi[a])) hi += 1.0;
                    lo = std::max(lo, 0.0);
                    if (hi > 255.0) { fits = false; break; }
                    lo8[s] = uint8_t(lo); hi8[s] = uint8_t(hi);
                }
                if (fits || e >= 127) {
                    for (int s = 0; s < 8; ++s) { node.qlo[a][s] = lo8[s]; node.qhi[a][s] = hi8[s]; }
                    break;
                }
            }
            node.exponent[a] = uint8_t(e + 127);
        }
        node.origin[0] = uint32_t(packed_origin);
        node.origin[1] = uint32_t(packed_origin >> 32);
    }
};

void set_grid(const std::vector<HiprTriangle>& triangles, Wide8Result& r) {
    Box all; all.reset();
    for (const HiprTriangle& t : triangles) { all.grow(t.v0); all.grow(t.v1); all.grow(t.v2); }
    for (int a = 0; a < 3; ++a) {
        r.grid_min[a] = all.lo[a];
        const float extent = all.hi[a] - all.lo[a];
        float cell = extent > 0.0f ? extent / 2097151.0f : 1.0f;
        while (double(cell) * 2097151.0 < double(all.hi[a]) - double(all.lo[a])) cell = std::nextafter(cell, FLT_MAX);
        r.grid_cell[a] = cell > 0.0f ? cell : FLT_MIN;
    }
}

} // namespace

Wide8Result build_wide8(const std::vector<HiprBvhNode>& nodes, const std::vector<HiprTriangle>& triangles) {
    Wide8Result result;
    if (nodes.empty() || triangles.empty()) return result;
    set_grid(triangles, result);
    result.slots.reserve(triangles.size() / 2 + triangles.size() / 6 + 16);
    result.slots.resize(1);
    Collapse collapse(nodes, triangles, result);
    const auto t0 = std::chrono::steady_clock::now();
    collapse.build_tree();
    const auto t1 = std::chrono::steady_clock::now();
    collapse.optimise();
    const auto t2 = std::chrono::steady_clock::now();
    // Layout: depth first (a subtree's slots stay together) or, HIPR_WIDE8_LAYOUT=bfs, level by level (the upper levels, mostly inner nodes, stay dense).
    const char* layout = std::getenv("HIPR_WIDE8_LAYOUT");
    if (layout && !std::strcmp(layout, "bfs")) {
        std::vector<Collapse::Queued> queue = {{0u, 0, 1u}};
        for (size_t next = 0; next < queue.size() && !collapse.overflow; ++next) {
            const Collapse::Queued item = queue[next];
            result.height = std::max(result.height, collapse.emit_node(item.slot, item.node, item.depth, &queue));
        }
    } else
        result.height = collapse.emit_node(0, 0, 1);
    if (std::getenv("HIPR_BVH_TIMING"))
        fprintf(stderr, "[hipr]   8-wide: records + binary tree %.3f s, collapse optimisation %.3f s, layout + quantisation %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                std::chrono::duration<double>(t2 - t1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count());
    if (collapse.overflow) return Wide8Result();      // more than 2^24 slots: the caller keeps the 4-wide tree
    return result;
}

void refit_wide8(Wide8Result& tree, const std::vector<HiprTriangle>& triangles) {
    if (tree.slots.empty()) return;
    set_grid(triangles, tree);      // the moved scene's bounds: node origins must not be clamped at the ends of a stale grid
    std::vector<uint8_t> is_node(tree.slots.size(), 0);
    is_node[0] = 1;
    for (size_t i = 0; i < tree.slots.size(); ++i) {
        if (!is_node[i]) continue;
        const HiprNode8& n = tree.slots[i].node;
        const uint32_t base = n.base_valid & 0xFFFFFFu, valid = n.base_valid >> 24;
        uint32_t rank = 0;
        for (int s = 0; s < 8; ++s) {
            if (!(valid >> s & 1u)) continue;
            if (n.inner_mask >> s & 1u) is_node[base + rank] = 1;
            ++rank;
        }
    }
    std::vector<Box> exact(tree.slots.size());
    for (size_t i = tree.slots.size(); i-- > 0;) {
        if (!is_node[i]) {
            HiprLeaf8& leaf = tree.slots[i].leaf;
            const uint32_t index_a = leaf.triangle[0], index_b = leaf.triangle[1];
            const int rotation_a = int((1 - int((leaf.flags >> 8) & 3u) + 3) % 3);
            HiprLeaf8 rebuilt;
            if (!make_record(triangles, index_a, index_b, rotation_a, rebuilt)) make_record(triangles, index_a, HIPR_LEAF8_NONE, 0, rebuilt);   // cannot happen for a rigid move of one instance
            leaf = rebuilt;
            exact[i] = triangle_box(triangles[index_a]);
            if (index_b != HIPR_LEAF8_NONE) exact[i].grow(triangle_box(triangles[index_b]));
            continue;
        }
        HiprNode8& n = tree.slots[i].node;
        const uint32_t base = n.base_valid & 0xFFFFFFu, valid = n.base_valid >> 24;
        Box boxes[8], all; all.reset();
        uint32_t rank = 0;
        for (int s = 0; s < 8; ++s) {
            if (!(valid >> s & 1u)) continue;
            boxes[s] = exact[base + rank++];
            all.grow(boxes[s]);
        }
        exact[i] = all;
        Collapse::quantise_node(boxes, valid, all, tree.grid_min, tree.grid_cell, n);
    }
}

} // namespace HIPRenderer
