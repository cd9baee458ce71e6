// host/BvhBuilder.cpp -- deterministic binned-SAH BVH2 builder for the HIP traversal kernels.
//
// Replaces the OptiX "Trbvh" acceleration build the reference requests at
// OptiXRenderer/Renderer.cpp:161-182,471-476 (closed source, so the tree itself is new design).
// Output: HiprBvhNode[] (64 B, both child boxes in the parent, Aila-Laine layout) in depth-first
// order with node 0 the root, and the triangle permutation in leaf order. Guarantees:
//   * every triangle is referenced by exactly one leaf, leaves hold 1..4 triangles;
//   * the deepest leaf is at most `max_depth` levels below the root (median splits take over when
//     the SAH tree would exceed the LDS stack of the kernels);
//   * the result depends only on the input order and values: the build runs on several host threads for large scenes (the top levels
//     bin and partition in parallel, the subtrees below them build side by side) and produces the tree the single-threaded build does,
//     node for node -- bounds and bin counts are order-independent reductions, the partition is stable, and subtrees are stitched in
//     depth-first order. HIPR_BVH_THREADS overrides the thread count (1 = single threaded).
#include "BvhBuilder.h"
#include "BvhOptimizer.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cfloat>
#include <cstdlib>
#include <cmath>
#include <exception>
#include <mutex>
#include <thread>

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

#ifndef HIPR_BVH_BINS
#define HIPR_BVH_BINS 16
#endif
constexpr int BIN_COUNT = HIPR_BVH_BINS;
// Triangles per leaf. The traversal kernels spend one loop iteration per node AND per triangle, so a leaf is worth splitting
// as long as the split culls triangles; HIPR_BVH_LEAF_SIZE overrides for experiments.
static uint32_t leaf_max() {
    static const uint32_t value = [] { const char* v = std::getenv("HIPR_BVH_LEAF_SIZE"); int n = v ? std::atoi(v) : 3; return uint32_t(n < 1 ? 1 : (n > 8 ? 8 : n)); }();
    return value;
}
#define LEAF_MAX leaf_max()

static unsigned build_threads() {
    static const unsigned value = [] {
        if (const char* v = std::getenv("HIPR_BVH_THREADS")) return unsigned(std::max(1, std::atoi(v)));
        const unsigned n = std::thread::hardware_concurrency();
        return n == 0 ? 1u : std::min(n, 16u);
    }();
    return value;
}

// An exception (out of memory) must not leave a worker thread: the first one is kept and thrown again on the caller once the threads are joined.
struct Failure {
    std::exception_ptr first;
    std::mutex lock;
    template <typename F> void guard(F f) {
        try { f(); } catch (...) { std::lock_guard<std::mutex> g(lock); if (!first) first = std::current_exception(); }
    }
    void rethrow() { if (first) std::rethrow_exception(first); }
};

// f(chunk, begin, end) over `chunks` equal slices of [begin, end), on as many threads; chunk 0 runs on the caller.
template <typename F>
static void for_chunks(uint32_t begin, uint32_t end, unsigned chunks, F f) {
    const uint64_t count = end - begin;
    auto slice = [&](unsigned c) { return uint32_t(begin + count * c / chunks); };
    Failure failure;
    std::vector<std::thread> workers;
    workers.reserve(chunks > 0 ? chunks - 1 : 0);
    for (unsigned c = 1; c < chunks; ++c) workers.emplace_back([&, c] { failure.guard([&] { f(c, slice(c), slice(c + 1)); }); });
    failure.guard([&] { f(0u, slice(0), slice(1)); });
    for (std::thread& w : workers) w.join();
    failure.rethrow();
}

constexpr uint32_t PARALLEL_RANGE = 1u << 17;      // ranges at least this long are binned and partitioned on several threads
constexpr int32_t SUBTREE_MARK = 0x40000000;       // child reference of the top tree: a subtree built as its own task (| task index)

struct Builder {
    const std::vector<HiprTriangle>& tris;
    const Box* boxes = nullptr;           // per input triangle
    const float* centroids = nullptr;     // 3 per input triangle
    uint32_t* order = nullptr;            // the permutation being built; a (sub)builder touches only its own range
    std::vector<HiprBvhNode> nodes;
    uint32_t max_depth_limit;
    uint32_t deepest = 0;
    unsigned threads = 1;                 // worker threads of the build (the top builder only)
    unsigned range_threads = 1;           // > 1: long ranges are binned and partitioned in this many slices (measured: beyond 8 the top levels get slower)

    // Subtrees handed to worker threads (top builder only): ranges at most `subtree_cutoff` long.
    struct Task { uint32_t begin, end, depth; std::vector<HiprBvhNode> nodes; uint32_t deepest = 0; };
    std::vector<Task> tasks;
    uint32_t subtree_cutoff = 0;          // 0: build everything here

    Builder(const std::vector<HiprTriangle>& t, uint32_t limit) : tris(t), max_depth_limit(limit) {}

    Box bounds_of(uint32_t begin, uint32_t end) const {
        Box b; b.reset();
        if (range_threads > 1 && end - begin >= PARALLEL_RANGE) {
            std::vector<Box> partial(range_threads);
            for_chunks(begin, end, range_threads, [&](unsigned c, uint32_t cb, uint32_t ce) {
                Box p; p.reset();
                for (uint32_t i = cb; i < ce; ++i) p.grow(boxes[order[i]]);
                partial[c] = p;
            });
            for (const Box& p : partial) b.grow(p);      // min / max: the same box in any order
            return b;
        }
        for (uint32_t i = begin; i < end; ++i) b.grow(boxes[order[i]]);
        return b;
    }

    static uint32_t levels_needed(uint32_t count) {   // levels a balanced median tree needs below this node
        uint32_t leaves = (count + LEAF_MAX - 1) / LEAF_MAX, levels = 0;
        while ((1u << levels) < leaves) ++levels;
        return levels;
    }

    struct Bins {
        Box box[3][BIN_COUNT];
        uint32_t n[3][BIN_COUNT];
        void reset() { for (int a = 0; a < 3; ++a) for (int b = 0; b < BIN_COUNT; ++b) { box[a][b].reset(); n[a][b] = 0; } }
    };
    static int bin_of(float centroid, float lo, float scale) {
        int b = int((centroid - lo) * scale);
        return std::min(std::max(b, 0), BIN_COUNT - 1);
    }
    // The triangles of [begin, end) sorted into the bins of every axis with an extent (scale[axis] > 0).
    void fill_bins(uint32_t begin, uint32_t end, const Box& cb, const float scale[3], Bins& bins) const {
        bins.reset();
        for (uint32_t i = begin; i < end; ++i) {
            const uint32_t t = order[i];
            for (int axis = 0; axis < 3; ++axis) {
                if (!(scale[axis] > 0.0f)) continue;
                const int b = bin_of(centroids[3 * t + axis], cb.lo[axis], scale[axis]);
                bins.box[axis][b].grow(boxes[t]);
                bins.n[axis][b]++;
            }
        }
    }

    // Returns the split position in [begin+1, end-1].
    uint32_t split(uint32_t begin, uint32_t end, uint32_t depth) {
        const uint32_t count = end - begin;
        const bool parallel = range_threads > 1 && count >= PARALLEL_RANGE;
        Box cb; cb.reset();
        if (parallel) {
            std::vector<Box> partial(range_threads);
            for_chunks(begin, end, range_threads, [&](unsigned c, uint32_t b, uint32_t e) {
                Box p; p.reset();
                for (uint32_t i = b; i < e; ++i) p.grow(&centroids[3 * order[i]]);
                partial[c] = p;
            });
            for (const Box& p : partial) cb.grow(p);
        } else
            for (uint32_t i = begin; i < end; ++i) cb.grow(&centroids[3 * order[i]]);

        // Depth budget: once only enough levels remain for a balanced tree, split at the median.
        const bool force_median = depth + levels_needed(count) >= max_depth_limit;

        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        if (!force_median) {
            float scale[3];
            for (int axis = 0; axis < 3; ++axis) {
                const float extent = cb.hi[axis] - cb.lo[axis];
                scale[axis] = extent > 0.0f ? BIN_COUNT / extent : 0.0f;
            }
            Bins bins;
            if (parallel) {
                std::vector<Bins> partial(range_threads);
                for_chunks(begin, end, range_threads, [&](unsigned c, uint32_t b, uint32_t e) { fill_bins(b, e, cb, scale, partial[c]); });
                bins.reset();
                for (const Bins& p : partial)
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < BIN_COUNT; ++b) { if (p.n[a][b]) bins.box[a][b].grow(p.box[a][b]); bins.n[a][b] += p.n[a][b]; }
            } else
                fill_bins(begin, end, cb, scale, bins);
            for (int axis = 0; axis < 3; ++axis) {
                if (!(scale[axis] > 0.0f)) continue;
                const Box* bin_box = bins.box[axis];
                const uint32_t* bin_n = bins.n[axis];
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
            auto left = [&](uint32_t t) { return bin_of(centroids[3 * t + best_axis], lo, scale) <= best_bin; };
            uint32_t m;
            if (parallel) {
                // stable partition in two passes: count the left elements of every slice, then scatter both sides in slice order
                std::vector<uint32_t> lefts(range_threads, 0u);
                for_chunks(begin, end, range_threads, [&](unsigned c, uint32_t b, uint32_t e) {
                    uint32_t k = 0;
                    for (uint32_t i = b; i < e; ++i) k += left(order[i]) ? 1u : 0u;
                    lefts[c] = k;
                });
                uint32_t total_left = 0;
                for (uint32_t k : lefts) total_left += k;
                std::vector<uint32_t> left_at(range_threads), right_at(range_threads);
                uint32_t l = 0, r = total_left;
                for (unsigned c = 0; c < range_threads; ++c) {
                    const uint32_t slice = uint32_t(begin + uint64_t(count) * (c + 1) / range_threads) - uint32_t(begin + uint64_t(count) * c / range_threads);
                    left_at[c] = l; right_at[c] = r;
                    l += lefts[c]; r += slice - lefts[c];
                }
                std::vector<uint32_t> moved(count);
                for_chunks(begin, end, range_threads, [&](unsigned c, uint32_t b, uint32_t e) {
                    uint32_t lw = left_at[c], rw = right_at[c];
                    for (uint32_t i = b; i < e; ++i) { const uint32_t t = order[i]; if (left(t)) moved[lw++] = t; else moved[rw++] = t; }
                });
                for_chunks(begin, end, range_threads, [&](unsigned, uint32_t b, uint32_t e) { std::copy(moved.begin() + (b - begin), moved.begin() + (e - begin), order + b); });
                m = begin + total_left;
            } else
                m = uint32_t(std::stable_partition(order + begin, order + end, left) - order);
            if (m > begin && m < end) return m;
        }

        // Median split along the widest centroid axis (also the fallback for coincident centroids).
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
        const uint32_t m = begin + count / 2;
        std::stable_sort(order + begin, order + end, [&](uint32_t a, uint32_t b) { return centroids[3 * a + axis] < centroids[3 * b + axis]; });
        return m;
    }

    static int32_t leaf_ref(uint32_t first, uint32_t count) { return ~int32_t((first << 3) | (count - 1)); }

    static void store_child(HiprBvhNode& n, int c, const Box& b, int32_t ref) {
        float* xy = c == 0 ? n.c0xy : n.c1xy;
        xy[0] = b.lo[0]; xy[1] = b.hi[0]; xy[2] = b.lo[1]; xy[3] = b.hi[1];
        n.cz[2 * c] = b.lo[2]; n.cz[2 * c + 1] = b.hi[2];
        n.child[c] = ref;
    }

    // Builds the node covering order[begin, end) (count > LEAF_MAX) and returns its index. With a subtree cutoff (the top builder of a
    // parallel build), a child range at most that long becomes a task instead and its reference carries SUBTREE_MARK.
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
            } else if (subtree_cutoff && e - b <= subtree_cutoff) {
                store_child(scratch, c, box, SUBTREE_MARK | int32_t(tasks.size()));
                tasks.push_back({b, e, depth + 1, {}, 0});
            } else
                store_child(scratch, c, box, int32_t(build(b, e, depth + 1)));
        }
        nodes[index] = scratch;
        return index;
    }

    // The top tree and the finished subtrees laid out in the depth-first order the single-threaded build produces: a layout pass gives every
    // top node and every subtree its place, then the subtrees are copied to theirs side by side.
    void layout(uint32_t top_index, const std::vector<HiprBvhNode>& top, std::vector<uint32_t>& top_at, std::vector<uint32_t>& task_at, uint32_t& next) const {
        top_at[top_index] = next++;
        for (int c = 0; c < 2; ++c) {
            const int32_t ref = top[top_index].child[c];
            if (ref < 0) continue;                                     // a leaf
            if (ref & SUBTREE_MARK) { task_at[size_t(ref & ~SUBTREE_MARK)] = next; next += uint32_t(tasks[size_t(ref & ~SUBTREE_MARK)].nodes.size()); }
            else layout(uint32_t(ref), top, top_at, task_at, next);
        }
    }

    void build_all(uint32_t n) {
        threads = n >= 2 * PARALLEL_RANGE ? build_threads() : 1u;
        range_threads = std::min(threads, 8u);
        if (threads == 1) { nodes.reserve(n); build(0, n, 1); return; }
        subtree_cutoff = std::max<uint32_t>(n / (threads * 8u), 4096u);
        const auto x0 = std::chrono::steady_clock::now();
        build(0, n, 1);
        const auto x1 = std::chrono::steady_clock::now();
        // the subtrees, longest first, pulled from a shared counter
        std::vector<size_t> by_size(tasks.size());
        for (size_t i = 0; i < by_size.size(); ++i) by_size[i] = i;
        std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return tasks[a].end - tasks[a].begin > tasks[b].end - tasks[b].begin; });
        std::atomic<size_t> next{0};
        auto worker = [&] {
            for (size_t k = next.fetch_add(1); k < by_size.size(); k = next.fetch_add(1)) {
                Task& task = tasks[by_size[k]];
                Builder sub(tris, max_depth_limit);
                sub.boxes = boxes; sub.centroids = centroids; sub.order = order;
                sub.nodes.reserve(task.end - task.begin);
                sub.build(task.begin, task.end, task.depth);
                task.nodes = std::move(sub.nodes);
                task.deepest = sub.deepest;
            }
        };
        Failure failure;
        std::vector<std::thread> workers;
        for (unsigned t = 1; t < threads; ++t) workers.emplace_back([&] { failure.guard(worker); });
        failure.guard(worker);
        for (std::thread& w : workers) w.join();
        failure.rethrow();
        const auto x2 = std::chrono::steady_clock::now();
        for (const Task& task : tasks) deepest = std::max(deepest, task.deepest);
        std::vector<HiprBvhNode> top = std::move(nodes);
        std::vector<uint32_t> top_at(top.size()), task_at(tasks.size());
        uint32_t total = 0;
        layout(0, top, top_at, task_at, total);
        std::vector<HiprBvhNode> out(total);
        for (size_t i = 0; i < top.size(); ++i) {
            HiprBvhNode node = top[i];
            for (int c = 0; c < 2; ++c)
                if (node.child[c] >= 0) node.child[c] = int32_t((node.child[c] & SUBTREE_MARK) ? task_at[size_t(node.child[c] & ~SUBTREE_MARK)] : top_at[size_t(node.child[c])]);
            out[top_at[i]] = node;
        }
        std::atomic<size_t> next_copy{0};
        auto copier = [&] {
            for (size_t k = next_copy.fetch_add(1); k < tasks.size(); k = next_copy.fetch_add(1)) {
                const int32_t offset = int32_t(task_at[k]);
                HiprBvhNode* target = out.data() + offset;
                for (HiprBvhNode node : tasks[k].nodes) {
                    for (int c = 0; c < 2; ++c) if (node.child[c] >= 0) node.child[c] += offset;
                    *target++ = node;
                }
                std::vector<HiprBvhNode>().swap(tasks[k].nodes);
            }
        };
        Failure copy_failure;
        std::vector<std::thread> copiers;
        for (unsigned t = 1; t < threads; ++t) copiers.emplace_back([&] { copy_failure.guard(copier); });
        copy_failure.guard(copier);
        for (std::thread& w : copiers) w.join();
        copy_failure.rethrow();
        nodes = std::move(out);
        if (std::getenv("HIPR_BVH_TIMING"))
            fprintf(stderr, "[hipr]   top tree %.3f s (%zu subtree tasks of <= %u triangles), subtrees %.3f s, stitch %.3f s\n", std::chrono::duration<double>(x1 - x0).count(), tasks.size(), subtree_cutoff,
                    std::chrono::duration<double>(x2 - x1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - x2).count());
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
    std::vector<uint32_t> subtree_size;  // BVH2 nodes in the subtree of a node
    const uint32_t* need = nullptr;      // = binary_need.data() of the collapse that computed it (shared with the subtree collapses)

    // Subtrees collapsed as tasks of their own (the top collapse of a parallel run): BVH2 subtrees of at most `subtree_cutoff` nodes.
    struct Task { uint32_t node_index, budget; std::vector<HiprWideNode> wide; uint32_t stack_need = 0; };
    std::vector<Task> tasks;
    uint32_t subtree_cutoff = 0;
    static bool is_task(int32_t ref) { return ref >= SUBTREE_MARK && ref != int32_t(HIPR_WIDE_EMPTY); }
    explicit WideCollapse(const std::vector<HiprBvhNode>& n, const uint32_t* shared_need = nullptr) : nodes(n), need(shared_need) {}

    // Nodes are stored parent before children: one sweep from the last node to the first sees every child done.
    uint32_t compute_binary_need() {
        binary_need.assign(nodes.size(), 0u);
        subtree_size.assign(nodes.size(), 1u);
        for (size_t i = nodes.size(); i-- > 0;) {
            uint32_t deepest = 0;
            for (int c = 0; c < 2; ++c) {
                const int32_t ref = nodes[i].child[c];
                deepest = std::max(deepest, 1u + (ref >= 0 ? binary_need[size_t(ref)] : 0u));
                if (ref >= 0) subtree_size[i] += subtree_size[size_t(ref)];
            }
            binary_need[i] = deepest;
        }
        need = binary_need.data();
        return binary_need[0];
    }
    uint32_t need_of(int32_t ref) const { return ref >= 0 ? need[ref] : 0u; }

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
            const uint32_t child_budget = budget > held ? budget - held : 0u;
            if (children[k].ref < 0) w.child[k] = children[k].ref;
            else if (subtree_cutoff && subtree_size[size_t(children[k].ref)] <= subtree_cutoff) {
                w.child[k] = SUBTREE_MARK | int32_t(tasks.size());      // its stack need joins in finish()
                tasks.push_back({uint32_t(children[k].ref), child_budget, {}, 0u});
            } else
                w.child[k] = int32_t(collapse(uint32_t(children[k].ref), child_budget, below));
            stack_need = std::max(stack_need, uint32_t(children.size() - 1) + below);
        }
        wide[index] = w;
        return index;
    }

    // ---- the parallel run: top collapse here, subtrees on worker threads, then the depth-first layout the sequential collapse produces ----
    void layout(uint32_t top_index, const std::vector<HiprWideNode>& top, std::vector<uint32_t>& top_at, std::vector<uint32_t>& task_at, uint32_t& next, uint32_t& stack_need) const {
        top_at[top_index] = next++;
        uint32_t children = 0;
        for (int k = 0; k < 4; ++k) children += top[top_index].child[k] != int32_t(HIPR_WIDE_EMPTY);
        stack_need = 0;
        for (int k = 0; k < 4; ++k) {
            const int32_t ref = top[top_index].child[k];
            if (ref == int32_t(HIPR_WIDE_EMPTY)) continue;
            uint32_t below = 0;
            if (is_task(ref)) {
                const Task& task = tasks[size_t(ref - SUBTREE_MARK)];
                task_at[size_t(ref - SUBTREE_MARK)] = next;
                next += uint32_t(task.wide.size());
                below = task.stack_need;
            } else if (ref >= 0)
                layout(uint32_t(ref), top, top_at, task_at, next, below);
            stack_need = std::max(stack_need, children - 1u + below);
        }
    }

    void run(uint32_t budget, unsigned threads, uint32_t& stack_need) {
        if (threads <= 1 || nodes.size() < PARALLEL_RANGE) { wide.reserve(nodes.size() / 2 + 1); collapse(0, budget, stack_need); return; }
        subtree_cutoff = std::max<uint32_t>(uint32_t(nodes.size() / (threads * 8u)), 2048u);
        uint32_t ignored = 0;
        collapse(0, budget, ignored);
        std::vector<size_t> by_size(tasks.size());
        for (size_t i = 0; i < by_size.size(); ++i) by_size[i] = i;
        std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return subtree_size[tasks[a].node_index] > subtree_size[tasks[b].node_index]; });
        std::atomic<size_t> next{0};
        auto worker = [&] {
            for (size_t k = next.fetch_add(1); k < by_size.size(); k = next.fetch_add(1)) {
                Task& task = tasks[by_size[k]];
                WideCollapse sub(nodes, need);
                sub.wide.reserve(subtree_size[task.node_index] / 2 + 1);
                sub.collapse(task.node_index, task.budget, task.stack_need);
                task.wide = std::move(sub.wide);
            }
        };
        Failure failure;
        std::vector<std::thread> workers;
        for (unsigned t = 1; t < threads; ++t) workers.emplace_back([&] { failure.guard(worker); });
        failure.guard(worker);
        for (std::thread& w : workers) w.join();
        failure.rethrow();

        std::vector<HiprWideNode> top = std::move(wide);
        std::vector<uint32_t> top_at(top.size()), task_at(tasks.size());
        uint32_t total = 0;
        layout(0, top, top_at, task_at, total, stack_need);
        std::vector<HiprWideNode> out(total);
        for (size_t i = 0; i < top.size(); ++i) {
            HiprWideNode node = top[i];
            for (int k = 0; k < 4; ++k) {
                const int32_t ref = node.child[k];
                if (is_task(ref)) node.child[k] = int32_t(task_at[size_t(ref - SUBTREE_MARK)]);
                else if (ref >= 0 && ref != int32_t(HIPR_WIDE_EMPTY)) node.child[k] = int32_t(top_at[size_t(ref)]);
            }
            out[top_at[i]] = node;
        }
        std::atomic<size_t> next_copy{0};
        auto copier = [&] {
            for (size_t k = next_copy.fetch_add(1); k < tasks.size(); k = next_copy.fetch_add(1)) {
                const int32_t offset = int32_t(task_at[k]);
                HiprWideNode* target = out.data() + offset;
                for (HiprWideNode node : tasks[k].wide) {
                    for (int c = 0; c < 4; ++c) if (node.child[c] >= 0 && node.child[c] != int32_t(HIPR_WIDE_EMPTY)) node.child[c] += offset;
                    *target++ = node;
                }
                std::vector<HiprWideNode>().swap(tasks[k].wide);
            }
        };
        Failure copy_failure;
        std::vector<std::thread> copiers;
        for (unsigned t = 1; t < threads; ++t) copiers.emplace_back([&] { copy_failure.guard(copier); });
        copy_failure.guard(copier);
        for (std::thread& w : copiers) w.join();
        copy_failure.rethrow();
        wide = std::move(out);
    }
};

} // namespace

BvhBuildResult build_bvh(const std::vector<HiprTriangle>& triangles, uint32_t max_depth) {
    BvhBuildResult result;
    const uint32_t n = uint32_t(triangles.size());
    result.max_depth = 0;
    if (n == 0) return result;

    const auto t_start = std::chrono::steady_clock::now();
    Builder b(triangles, std::max(max_depth, 8u));
    std::vector<Box> boxes(n);
    std::vector<float> centroids(3 * size_t(n));
    result.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const HiprTriangle& t = triangles[i];
        Box& box = boxes[i];
        box.reset();
        box.grow(t.v0); box.grow(t.v1); box.grow(t.v2);
        for (int a = 0; a < 3; ++a) centroids[3 * i + a] = 0.5f * (box.lo[a] + box.hi[a]);
        result.order[i] = i;
    }
    b.boxes = boxes.data(); b.centroids = centroids.data(); b.order = result.order.data();

    if (n <= LEAF_MAX) {
        // A single leaf: both children reference it (the duplicate test cannot change the result).
        HiprBvhNode root = {};
        Box box = b.bounds_of(0, n);
        Builder::store_child(root, 0, box, Builder::leaf_ref(0, n));
        Builder::store_child(root, 1, box, Builder::leaf_ref(0, n));
        b.nodes.push_back(root);
        b.deepest = 1;
    } else
        b.build_all(n);
    // Insertion-based optimisation of the finished BVH2 (BvhOptimizer.h), opt-in: HIPR_BVH_REINSERTION = passes (3 converge), default 0. Measured in round 4
    // (profiles/r04_ab_bvh_reinsertion.txt): the binned-SAH tree of the atrium's evenly tessellated surfaces is close to a local optimum already -- 2 925 of
    // 375 k subtrees move, SAH cost -4.1 %, a closest-hit ray visits 14.3 nodes instead of 14.8 while a shadow ray tests 12.1 triangles instead of 10.9 -- and
    // the step times move by +1.0 % (atrium), -0.9 % (material scene), -0.6 % (1 M triangles): nothing to adopt. Sequential, so never beyond 4 M triangles.
    static const int reinsertion_passes = [] { const char* v = std::getenv("HIPR_BVH_REINSERTION"); return v ? std::atoi(v) : 0; }();
    if (reinsertion_passes > 0 && n > LEAF_MAX && n <= 4000000u) {
        const ReinsertionStatistics stats = optimise_by_reinsertion(b.nodes, result.order, b.max_depth_limit, b.deepest, reinsertion_passes);
        if (std::getenv("HIPR_BVH_TIMING"))
            fprintf(stderr, "[hipr] build_bvh: reinsertion %s: SAH cost %.4g -> %.4g (%.1f %%), %zu moves, deepest leaf %u\n", stats.taken ? "taken" : "not taken", stats.cost_before,
                    stats.cost_after, 100.0 * (stats.cost_after / stats.cost_before - 1.0), stats.moves, stats.deepest_leaf);
    }
    const auto t_built = std::chrono::steady_clock::now();

    result.nodes = std::move(b.nodes);
    WideCollapse collapse(result.nodes);
    // Keep the worst case within the 32 entry LDS stack of the traversal kernels whenever the BVH2 itself allows it; deeper
    // trees are collapsed freely and traversed by the kernels that back the LDS stack with scratch memory.
    const uint32_t budget = collapse.compute_binary_need() <= 32u ? 32u : 0xFFFFu;
    collapse.run(budget, b.threads, result.wide_stack_entries);
    result.wide_nodes = std::move(collapse.wide);
    result.max_depth = b.deepest + 1;   // stack entries needed is bounded by the node depth; keep one spare
    const auto t_wide = std::chrono::steady_clock::now();
    result.wide8 = build_wide8(result.nodes, OrderedTriangles{triangles.data(), result.order.data(), n});
    if (std::getenv("HIPR_BVH_TIMING")) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[hipr] build_bvh: %u triangles, %u threads: BVH2 %.3f s, 4-wide collapse %.3f s, 8-wide collapse %.3f s (%u nodes, %u leaf records of which %u hold two triangles, height %u)\n", n,
                b.threads, std::chrono::duration<double>(t_built - t_start).count(), std::chrono::duration<double>(t_wide - t_built).count(),
                std::chrono::duration<double>(t_end - t_wide).count(), result.wide8.node_count, result.wide8.leaf_count, result.wide8.paired_leaves, result.wide8.height);
    }
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
    if (!refit_wide8(bvh.wide8, OrderedTriangles{triangles.data(), nullptr, triangles.size()})) return -1.0;      // the 8-wide tree asks for a rebuild
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
