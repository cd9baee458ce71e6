// host/BvhOptimizer.h -- insertion-based optimisation of the binned-SAH BVH2 before it is collapsed to the wide trees (round 4).
//
// The top-down build decides every split once, on 16 bins per axis, and never revisits it. This pass does: every subtree, largest box first, is
// taken out and hung where the tree's SAH cost (sum of the node areas) falls most -- the search of Bittner, Hapala and Havran, "Fast Insertion-Based
// Optimization of Bounding Volume Hierarchies" (CGF 2013) in the pruned form of Meister and Bittner, "Parallel Reinsertion for Bounding Volume Hierarchy
// Optimization" (CGF 2018), run sequentially here: the result depends on the input tree only, not on the thread count of the build that made it.
// The leaves (up to LEAF_MAX triangles each) stay as they are; nodes and triangle order are re-emitted depth first, parent before children.
// Nothing to match in the reference: OptiX's "Trbvh" build is closed (OptiXRenderer/Renderer.cpp:161-182, 471-476). The oracle and the device walk the
// same tree, so hits, transmittance and counters stay bit-identical whatever this pass does; what it changes is how many nodes a ray visits.
#pragma once

#include "../../include/hiprenderer_c.h"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace HIPRenderer {

struct ReinsertionStatistics {
    double cost_before = 0.0, cost_after = 0.0;      // sum of the half-areas of all nodes below the root
    size_t moves = 0;
    uint32_t deepest_leaf = 0;
    bool taken = false;                              // false: the input is returned untouched (no gain, or deeper than `depth_limit`)
};

// `bvh`: HiprBvhNode[] with node 0 the root, `order`: the triangle permutation its leaf references index. `deepest_leaf` is updated when the result is taken.
ReinsertionStatistics optimise_by_reinsertion(std::vector<HiprBvhNode>& bvh, std::vector<uint32_t>& order, uint32_t depth_limit, uint32_t& deepest_leaf, int passes);

} // namespace HIPRenderer
