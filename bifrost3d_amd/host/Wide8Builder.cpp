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
// Deterministic: the result depends only on the input, not on the number of host threads the three phases run on (HIPR_BVH_THREADS).
#include "Wide8Builder.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <utility>

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
bool make_record(const OrderedTriangles& triangles, uint32_t index_a, uint32_t index_b, int rotation_a, HiprLeaf8& out) {
    const HiprTriangle& A = triangles[index_a];
    const float *a = corner(A, rotation_a), *b = corner(A, rotation_a + 1), *c = corner(A, rotation_a + 2);
    out = {};
    for (int k = 0; k < 3; ++k) { out.a[k] = a[k]; out.e1[k] = b[k] - a[k]; out.e2[k] = c[k] - a[k]; }
    out.triangle[0] = index_a;
    out.triangle[1] = HIPR_LEAF8_NONE;
    // record corner k of A is A's vertex (rotation + k) % 3, so A's vertex j is record corner (j - rotation) mod 3: its weight is (w, u, v)[that]
    uint32_t flags = (A.flags & HIPR_TRIANGLE_OPAQUE ? 1u : 0u) | (A.flags & HIPR_TRIANGLE_ONE_SIDED ? 4u : 0u) | uint32_t((1 - rotation_a + 3) % 3) << 8 | uint32_t((2 - rotation_a + 3) % 3) << 10;
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
        flags |= (B.flags & HIPR_TRIANGLE_OPAQUE ? 2u : 0u) | (B.flags & HIPR_TRIANGLE_ONE_SIDED ? 8u : 0u) | uint32_t(where[1]) << 12 | uint32_t(where[2]) << 14;
        // B's vertices 0, 1, 2 sit on the record's corners where[0], where[1], where[2] of (a, c, d): an odd permutation means B is wound the other way round
        const bool even = (where[0] == 0 && where[1] == 1) || (where[0] == 1 && where[1] == 2) || (where[0] == 2 && where[1] == 0);
        if (!even) flags |= 16u;
    }
    out.flags = flags;
    float squares = 0.0f;
    for (int k = 0; k < 3; ++k) squares += out.e1[k] * out.e1[k] + out.e2[k] * out.e2[k] + out.e3[k] * out.e3[k];
    out.facing_margin = squares * (1.0f / 8192.0f);
    return true;
}

// The binary tree the collapse works on: the BVH2 with every leaf turned into its records (a leaf of several records becomes a short chain), so
// that a leaf of THIS tree is exactly one record. Numbered parent before children.
struct TreeNode {
    Box box;
    int32_t left = -1, right = -1;      // both -1: a leaf
    int32_t record = -1;                // leaf: index into Collapse::records
};

// An array whose elements are NOT touched at allocation: the worker threads write (and thereby page in) their own stretches side by side, which a
// std::vector's value-initialisation would do on the caller alone.
template <typename T>
struct RawArray {
    T* data = nullptr;
    size_t count = 0;
    RawArray() = default;
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
    ~RawArray() { std::free(data); }
    bool allocate(size_t n) { std::free(data); data = static_cast<T*>(std::malloc(std::max<size_t>(n, 1) * sizeof(T))); count = data ? n : 0; return data != nullptr; }
    size_t size() const { return count; }
    T& operator[](size_t i) { return data[i]; }
    const T& operator[](size_t i) const { return data[i]; }
};

// Worker threads of the collapse: HIPR_BVH_THREADS, else the hardware's (at most 16); small scenes stay on the caller.
unsigned collapse_threads(size_t triangle_count) {
    if (triangle_count < (1u << 16)) return 1;
    if (const char* v = std::getenv("HIPR_BVH_THREADS")) return unsigned(std::max(1, std::atoi(v)));
    const unsigned n = std::thread::hardware_concurrency();
    return n == 0 ? 1u : std::min(n, 16u);
}
// f(i) for i in [0, count), tasks handed out in order to `threads` threads (the caller is one of them).
template <typename F>
void run_tasks(size_t count, unsigned threads, F f) {
    if (threads <= 1 || count <= 1) { for (size_t i = 0; i < count; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr failure;      // an exception (out of memory) must not leave a worker thread: the first one is handed to the caller
    std::mutex failure_lock;
    auto worker = [&] {
        try {
            for (size_t i = next.fetch_add(1); i < count; i = next.fetch_add(1)) f(i);
        } catch (...) {
            next.store(count);
            std::lock_guard<std::mutex> guard(failure_lock);
            if (!failure) failure = std::current_exception();
        }
    };
    std::vector<std::thread> workers;
    for (unsigned t = 1; t < threads; ++t) workers.emplace_back(worker);
    worker();
    for (std::thread& w : workers) w.join();
    if (failure) std::rethrow_exception(failure);
}

// The three phases -- records + binary tree, dynamic program, layout -- each run the same way: the top of the tree on the caller, the subtrees below it as
// tasks on the worker threads into arrays of their own, which are then joined IN DEPTH-FIRST ORDER with their indices shifted. What comes out is the array a
// single depth-first pass writes, whatever the number of threads and wherever the top ends.
struct Collapse {
    const std::vector<HiprBvhNode>& nodes;
    const OrderedTriangles& triangles;
    Wide8Result& out;
    const unsigned threads;
    bool overflow = false;
    RawArray<TreeNode> tree;
    RawArray<HiprLeaf8> records;
    Collapse(const std::vector<HiprBvhNode>& n, const OrderedTriangles& t, Wide8Result& o) : nodes(n), triangles(t), out(o), threads(collapse_threads(t.size())) {}
    // Ylitie et al. 2017, section 3.1: cost[n][i - 1] = the lowest SAH cost of the subtree of n when it appears in its parent wide node as a forest of at
    // most i roots (i = 1 .. 7); a node that becomes a wide node itself hands its two subtrees up to 8 roots in total. split[n][j - 2] = how many of j roots
    // go to the left subtree in the best distribution; roots_used[n][i - 1] = the i' <= i at which cost[n][i - 1] is attained (1 = the node itself).
    static constexpr float NODE_COST = 1.0f;      // a node visit is four loads + ~230 instructions, a leaf record four loads + ~130
    const float LEAF_COST = [] { const char* v = std::getenv("HIPR_WIDE8_LEAF_COST"); return v ? float(std::atof(v)) : 0.6f; }();
    RawArray<float> cost;               // the three tables are written by optimise_node() before anything reads them: inner nodes all entries, leaves `cost` only
    RawArray<uint8_t> split, roots_used;
    RawArray<uint32_t> subtree_size;       // tree nodes in the subtree of a tree node (itself included): they are [n, n + subtree_size[n])

    static Box child_box(const HiprBvhNode& n, int c) {
        Box b;
        const float* xy = c == 0 ? n.c0xy : n.c1xy;
        b.lo[0] = xy[0]; b.hi[0] = xy[1]; b.lo[1] = xy[2]; b.hi[1] = xy[3];
        b.lo[2] = n.cz[2 * c]; b.hi[2] = n.cz[2 * c + 1];
        return b;
    }

    // ---- phase 1: records + binary tree --------------------------------------------------------------------------------------------------------------
    struct TreePart { std::vector<TreeNode> tree; std::vector<HiprLeaf8> records; };

    // The records of a BVH2 leaf (at most 8): every triangle pairs with the first later one of the same instance that shares exactly two corners with it.
    // The records go to `part.records`; `leaves` receives one tree leaf per record. Returns their number.
    size_t records_of_leaf(int32_t ref, TreeNode* leaves, TreePart& part) const {
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
            leaf.record = int32_t(part.records.size());
            part.records.push_back(record);
        }
        return made;
    }

    // Appends the subtree of a BVH2 child reference (inner node or leaf) to `part` and returns its index there.
    int32_t add_subtree(int32_t ref, const Box& box, TreePart& part) const {
        // explicit stack (BVH2 trees can be deep); children are linked to their parent by index, `tree` grows underneath
        std::vector<TreeNode>& tree = part.tree;
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
                const size_t r = records_of_leaf(p.ref, leaves, part);
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

    // One step of the top's depth-first walk: a BVH2 node of the top (its tree node is numbered when the walk gets to it) or a subtree task.
    struct TreeEvent { int32_t ref; Box box; int32_t parent_event; int side; int32_t task; size_t index; };
    std::vector<TreeEvent> tree_events;
    std::vector<TreePart> tree_parts;
    std::vector<size_t> part_offset;      // of a task's tree nodes in `tree`

    void build_tree() {
        // triangles below every BVH2 node (children are referenced by index only: post-order over an explicit stack)
        std::vector<uint32_t> below(nodes.size(), 0);
        auto count_of = [&](int32_t ref) { return ref >= 0 ? below[size_t(ref)] : (uint32_t(~ref) & 7u) + 1u; };
        {
            std::vector<std::pair<int32_t, bool>> stack = {{0, false}};
            while (!stack.empty()) {
                const auto [n, expanded] = stack.back();
                stack.pop_back();
                const HiprBvhNode& node = nodes[size_t(n)];
                if (expanded) { below[size_t(n)] = count_of(node.child[0]) + (node.child[1] != node.child[0] || node.child[0] >= 0 ? count_of(node.child[1]) : 0u); continue; }
                stack.push_back({n, true});
                for (int c = 0; c < 2; ++c) if (node.child[c] >= 0) stack.push_back({node.child[c], false});
            }
        }
        const uint32_t cutoff = uint32_t(std::max<size_t>(2048, triangles.size() / (size_t(threads) * 8)));
        const HiprBvhNode& root = nodes[0];
        struct Pending { int32_t ref; Box box; int32_t parent_event; int side; };
        std::vector<Pending> pending;
        if (root.child[0] == root.child[1] && root.child[0] < 0) pending.push_back({root.child[0], child_box(root, 0), -1, 0});      // the single-leaf root references its leaf twice
        else {
            Box all = child_box(root, 0);
            all.grow(child_box(root, 1));
            pending.push_back({0, all, -1, 0});
        }
        int32_t task_count = 0;
        while (!pending.empty()) {
            const Pending p = pending.back();
            pending.pop_back();
            const int32_t event = int32_t(tree_events.size());
            if (threads > 1 && p.ref >= 0 && count_of(p.ref) > cutoff) {
                tree_events.push_back({p.ref, p.box, p.parent_event, p.side, -1, 0});
                const HiprBvhNode& n = nodes[size_t(p.ref)];
                pending.push_back({n.child[1], child_box(n, 1), event, 1});
                pending.push_back({n.child[0], child_box(n, 0), event, 0});
            } else
                tree_events.push_back({p.ref, p.box, p.parent_event, p.side, task_count++, 0});
        }
        tree_parts.resize(size_t(task_count));
        run_tasks(tree_events.size(), threads, [&](size_t e) {
            const TreeEvent& ev = tree_events[e];
            if (ev.task < 0) return;
            TreePart& part = tree_parts[size_t(ev.task)];
            const size_t below_it = count_of(ev.ref);       // at most one record per triangle, and a binary tree over them
            part.tree.reserve(2 * below_it);
            part.records.reserve(below_it);
            add_subtree(ev.ref, ev.box, part);
        });
        // numbering: the walk's order
        size_t tree_cursor = 0, record_cursor = 0;
        part_offset.assign(size_t(task_count), 0);
        std::vector<size_t> record_offset(size_t(task_count), 0);
        for (TreeEvent& ev : tree_events) {
            ev.index = tree_cursor;
            if (ev.task < 0) { tree_cursor += 1; continue; }
            part_offset[size_t(ev.task)] = tree_cursor;
            record_offset[size_t(ev.task)] = record_cursor;
            tree_cursor += tree_parts[size_t(ev.task)].tree.size();
            record_cursor += tree_parts[size_t(ev.task)].records.size();
        }
        if (!tree.allocate(tree_cursor) || !records.allocate(record_cursor)) { overflow = true; return; }
        for (const TreeEvent& ev : tree_events) {
            if (ev.task < 0) { tree[ev.index] = TreeNode(); tree[ev.index].box = ev.box; }
            if (ev.parent_event >= 0) {
                TreeNode& parent = tree[tree_events[size_t(ev.parent_event)].index];
                (ev.side == 0 ? parent.left : parent.right) = int32_t(ev.index);
            }
        }
        run_tasks(tree_parts.size(), threads, [&](size_t t) {
            TreePart& part = tree_parts[t];
            const int32_t shift = int32_t(part_offset[t]), record_shift = int32_t(record_offset[t]);
            for (size_t i = 0; i < part.tree.size(); ++i) {
                TreeNode n = part.tree[i];
                if (n.left >= 0) { n.left += shift; n.right += shift; }
                if (n.record >= 0) n.record += record_shift;
                tree[part_offset[t] + i] = n;
            }
            std::copy(part.records.begin(), part.records.end(), records.data + record_offset[t]);
            std::vector<TreeNode>().swap(part.tree);
            std::vector<HiprLeaf8>().swap(part.records);
        });
    }

    // ---- phase 2: the dynamic program ----------------------------------------------------------------------------------------------------------------
    float& cost_of(size_t n, int i) { return cost[7 * n + size_t(i - 1)]; }
    void optimise_node(size_t n) {
        const TreeNode& t = tree[n];
        const float area = t.box.half_area();
        if (t.left < 0) {
            for (int i = 1; i <= 7; ++i) cost_of(n, i) = area * LEAF_COST;
            subtree_size[n] = 1;
            return;
        }
        const size_t l = size_t(t.left), r = size_t(t.right);
        subtree_size[n] = 1 + subtree_size[l] + subtree_size[r];
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
    void optimise() {
        const size_t count = tree.size();
        if (!cost.allocate(7 * count) || !split.allocate(7 * count) || !roots_used.allocate(7 * count) || !subtree_size.allocate(count)) { overflow = true; return; }
        // children are numbered after their parent: every task's stretch backwards, then the top backwards
        run_tasks(tree_events.size(), threads, [&](size_t e) {
            const TreeEvent& ev = tree_events[e];
            if (ev.task < 0) return;
            const size_t end = e + 1 < tree_events.size() ? tree_events[e + 1].index : count;
            for (size_t n = end; n-- > ev.index;) optimise_node(n);
        });
        for (size_t e = tree_events.size(); e-- > 0;) if (tree_events[e].task < 0) optimise_node(tree_events[e].index);
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

    // ---- phase 3: layout + quantisation --------------------------------------------------------------------------------------------------------------
    // The wide node of tree node n without its base: its children by position (tree node index, -1 = empty), masks and quantised boxes.
    struct Prepared { int32_t child_at[8]; uint32_t child_count; HiprNode8 node; };
    Prepared prepare_node(int32_t n) const {
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
        Prepared p;
        for (int s = 0; s < 8; ++s) p.child_at[s] = -1;
        for (size_t c = 0; c < children.size(); ++c) p.child_at[position_of[c]] = children[c];
        p.child_count = uint32_t(children.size());
        p.node = {};
        Box boxes[8];
        uint32_t valid = 0, inner_mask = 0;
        for (int s = 0; s < 8; ++s) {
            if (p.child_at[s] < 0) continue;
            valid |= 1u << s;
            const TreeNode& child = tree[size_t(p.child_at[s])];
            boxes[s] = child.box;
            if (child.left >= 0) inner_mask |= 1u << s;
        }
        p.node.inner_mask = uint8_t(inner_mask);
        p.node.base_valid = valid << 24;
        quantise(boxes, valid, all, p.node);
        return p;
    }

    // A stretch of slots with indices of its own, and what was written to it.
    struct SlotPart {
        std::vector<HiprSlot8> slots;
        std::vector<uint32_t> node_slots;      // which of them are nodes (their bases are shifted when the stretch is placed)
        uint32_t node_count = 0, leaf_count = 0, paired_leaves = 0;
    };
    void put_leaf(SlotPart& part, uint32_t slot, int32_t tree_leaf) const {
        part.slots[slot].leaf = records[size_t(tree[size_t(tree_leaf)].record)];
        part.leaf_count += 1;
        part.paired_leaves += part.slots[slot].leaf.triangle[1] != HIPR_LEAF8_NONE;
    }
    // Writes the wide node of tree node n into `slot` of the part and its subtree behind the part's current end; returns the height below (and including) it.
    // `queue` (breadth-first layout): the inner children are not descended into here but appended for the caller's loop.
    struct Queued { uint32_t slot; int32_t node; uint32_t depth; };
    uint32_t emit_node(SlotPart& part, uint32_t slot, int32_t n, uint32_t depth, std::vector<Queued>* queue = nullptr) const {
        Prepared p = prepare_node(n);
        const size_t base = part.slots.size();
        part.slots.resize(base + p.child_count);
        p.node.base_valid |= uint32_t(base) & 0xFFFFFFu;
        part.slots[slot].node = p.node;
        part.node_slots.push_back(slot);
        part.node_count += 1;
        uint32_t height = depth;
        uint32_t next = uint32_t(base);
        for (int s = 0; s < 8; ++s) {
            if (p.child_at[s] < 0) continue;
            const int32_t child = p.child_at[s];
            const uint32_t child_slot = next++;
            if (tree[size_t(child)].left >= 0) {
                if (queue) { queue->push_back({child_slot, child, depth + 1}); height = std::max(height, depth + 1); }
                else height = std::max(height, emit_node(part, child_slot, child, depth + 1));
            } else put_leaf(part, child_slot, child);
        }
        return height;
    }

    // Depth-first layout. The top is planned first -- which stretches follow each other: the children block of a top node, or the whole subtree of a task --
    // then the tasks write their stretches side by side, then everything is placed.
    uint32_t layout_depth_first() {
        struct Stretch { int32_t task; uint32_t size; uint32_t offset; };
        struct Place { int32_t stretch; uint32_t rank; };       // slot = offset of the stretch + rank; stretch -1: slot 0, the root
        struct TopNode { HiprNode8 node; int32_t stretch; Place where; };
        struct TopLeaf { int32_t tree_leaf; Place where; };
        struct Task { int32_t node; Place where; uint32_t depth; SlotPart part; uint32_t height = 0; };
        std::vector<Stretch> stretches;
        std::vector<TopNode> top_nodes;
        std::vector<TopLeaf> top_leaves;
        std::vector<Task> tasks;
        const uint32_t cutoff = uint32_t(std::max<size_t>(4096, tree.size() / (size_t(threads) * 8)));
        uint32_t height = 0;
        struct Pending { int32_t node; Place where; uint32_t depth; };
        // explicit stack; a node's inner children are pushed in reverse so that they are planned in position order, each with everything below it first
        std::vector<Pending> pending = {{0, {-1, 0}, 1u}};
        while (!pending.empty()) {
            const Pending item = pending.back();
            pending.pop_back();
            if (item.node < 0) {       // a task's place in the order (pushed below)
                stretches.push_back({~item.node, 0, 0});
                continue;
            }
            const Prepared p = prepare_node(item.node);
            const int32_t stretch = int32_t(stretches.size());
            stretches.push_back({-1, p.child_count, 0});
            top_nodes.push_back({p.node, stretch, item.where});
            height = std::max(height, item.depth);
            std::vector<Pending> inner;
            uint32_t rank = 0;
            for (int s = 0; s < 8; ++s) {
                if (p.child_at[s] < 0) continue;
                const int32_t child = p.child_at[s];
                const Place where = {stretch, rank++};
                if (tree[size_t(child)].left < 0) { top_leaves.push_back({child, where}); continue; }
                if (threads <= 1 || subtree_size[size_t(child)] <= cutoff) {
                    const int32_t task = int32_t(tasks.size());
                    tasks.push_back({child, where, item.depth + 1, {}, 0});
                    inner.push_back({~task, where, item.depth + 1});
                } else inner.push_back({child, where, item.depth + 1});
            }
            for (size_t i = inner.size(); i-- > 0;) pending.push_back(inner[i]);
        }
        run_tasks(tasks.size(), threads, [&](size_t t) {
            Task& task = tasks[t];
            task.part.slots.reserve(size_t(subtree_size[size_t(task.node)]) + 1);
            task.part.slots.resize(1);      // [0] = the task's own node, which lives in its parent's children block
            task.height = emit_node(task.part, 0, task.node, task.depth);
        });
        uint64_t cursor = 1;
        for (Stretch& st : stretches) {
            if (st.task >= 0) st.size = uint32_t(tasks[size_t(st.task)].part.slots.size() - 1);
            st.offset = uint32_t(cursor);
            cursor += st.size;
        }
        if (cursor > 0xFFFFFFu) { overflow = true; return 0; }
        out.slots.resize(size_t(cursor));
        auto slot_of = [&](const Place& where) { return where.stretch < 0 ? 0u : stretches[size_t(where.stretch)].offset + where.rank; };
        for (const TopNode& top : top_nodes) {
            HiprNode8 node = top.node;
            node.base_valid |= stretches[size_t(top.stretch)].offset;
            out.slots[slot_of(top.where)].node = node;
        }
        out.node_count += uint32_t(top_nodes.size());
        SlotPart top_part;      // the leaves of top nodes: written in place
        top_part.slots.swap(out.slots);
        for (const TopLeaf& leaf : top_leaves) put_leaf(top_part, slot_of(leaf.where), leaf.tree_leaf);
        top_part.slots.swap(out.slots);
        out.leaf_count += top_part.leaf_count;
        out.paired_leaves += top_part.paired_leaves;
        std::vector<uint32_t> task_offset(tasks.size(), 0);
        for (const Stretch& st : stretches) if (st.task >= 0) task_offset[size_t(st.task)] = st.offset;
        run_tasks(tasks.size(), threads, [&](size_t t) {
            Task& task = tasks[t];
            const uint32_t shift = task_offset[t] - 1u;      // local slot i >= 1 -> offset + i - 1
            for (uint32_t slot : task.part.node_slots) {
                HiprNode8& node = task.part.slots[slot].node;
                node.base_valid = (node.base_valid & 0xFF000000u) | (((node.base_valid & 0xFFFFFFu) + shift) & 0xFFFFFFu);
            }
            out.slots[slot_of(task.where)] = task.part.slots[0];
            std::copy(task.part.slots.begin() + 1, task.part.slots.end(), out.slots.begin() + ptrdiff_t(task_offset[t]));
            std::vector<HiprSlot8>().swap(task.part.slots);
        });
        for (const Task& task : tasks) {
            height = std::max(height, task.height);
            out.node_count += task.part.node_count;
            out.leaf_count += task.part.leaf_count;
            out.paired_leaves += task.part.paired_leaves;
        }
        return height;
    }

    // Level by level (HIPR_WIDE8_LAYOUT=bfs, an experiment): single-threaded.
    uint32_t layout_breadth_first() {
        SlotPart part;
        part.slots.resize(1);
        uint32_t height = 0;
        std::vector<Queued> queue = {{0u, 0, 1u}};
        for (size_t next = 0; next < queue.size(); ++next) {
            const Queued item = queue[next];
            height = std::max(height, emit_node(part, item.slot, item.node, item.depth, &queue));
            if (part.slots.size() > 0xFFFFFFu) { overflow = true; return 0; }
        }
        out.slots.swap(part.slots);
        out.node_count = part.node_count; out.leaf_count = part.leaf_count; out.paired_leaves = part.paired_leaves;
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

void set_grid(const OrderedTriangles& triangles, Wide8Result& r) {
    Box all; all.reset();
    for (size_t k = 0; k < triangles.size(); ++k) { const HiprTriangle& t = triangles.triangles[k]; all.grow(t.v0); all.grow(t.v1); all.grow(t.v2); }      // in storage order
    for (int a = 0; a < 3; ++a) {
        r.grid_min[a] = all.lo[a];
        const float extent = all.hi[a] - all.lo[a];
        float cell = extent > 0.0f ? extent / 2097151.0f : 1.0f;
        while (double(cell) * 2097151.0 < double(all.hi[a]) - double(all.lo[a])) cell = std::nextafter(cell, FLT_MAX);
        r.grid_cell[a] = cell > 0.0f ? cell : FLT_MIN;
    }
}

} // namespace

Wide8Result build_wide8(const std::vector<HiprBvhNode>& nodes, const OrderedTriangles& triangles) {
    Wide8Result result;
    if (nodes.empty() || triangles.empty()) return result;
    const auto t_entry = std::chrono::steady_clock::now();
    set_grid(triangles, result);
    Collapse collapse(nodes, triangles, result);
    const auto t0 = std::chrono::steady_clock::now();
    collapse.build_tree();
    const auto t1 = std::chrono::steady_clock::now();
    if (!collapse.overflow) collapse.optimise();
    const auto t2 = std::chrono::steady_clock::now();
    // Layout: depth first (a subtree's slots stay together) or, HIPR_WIDE8_LAYOUT=bfs, level by level (the upper levels, mostly inner nodes, stay dense).
    const char* layout = std::getenv("HIPR_WIDE8_LAYOUT");
    if (!collapse.overflow) result.height = layout && !std::strcmp(layout, "bfs") ? collapse.layout_breadth_first() : collapse.layout_depth_first();
    if (std::getenv("HIPR_BVH_TIMING"))
        fprintf(stderr, "[hipr]   8-wide on %u threads: scene grid %.3f s, records + binary tree %.3f s, collapse optimisation %.3f s, layout + quantisation %.3f s\n", collapse.threads, std::chrono::duration<double>(t0 - t_entry).count(), std::chrono::duration<double>(t1 - t0).count(),
                std::chrono::duration<double>(t2 - t1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count());
    if (collapse.overflow) return Wide8Result();      // more than 2^24 slots (or out of memory): the caller keeps the 4-wide tree
    return result;
}

bool refit_wide8(Wide8Result& tree, const OrderedTriangles& triangles) {
    if (tree.slots.empty()) return true;
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
            // Pairing was decided on bit-identical WORLD positions: two distinct object-space vertices that rounded to one world position at build time may
            // part after the move. The record cannot hold both triangles then; the caller rebuilds (a record of A alone would drop B from this tree only).
            if (!make_record(triangles, index_a, index_b, rotation_a, rebuilt)) return false;
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
    return true;
}

} // namespace HIPRenderer
