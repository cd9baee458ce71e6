// host/BvhBuilder.h -- see BvhBuilder.cpp.
#pragma once

#include "../../include/hiprenderer_c.h"
#include "Wide8Builder.h"

#include <cstdint>
#include <vector>

namespace HIPRenderer {

struct BvhBuildResult {
    std::vector<HiprBvhNode> nodes;   // node 0 is the root; empty for an empty scene
    std::vector<uint32_t> order;      // order[k] = index of the input triangle stored at leaf slot k
    uint32_t max_depth = 0;           // upper bound of the traversal stack entries the tree needs
    // The same tree collapsed to compressed 4-wide nodes (HiprWideNode); empty when the scene is empty.
    std::vector<HiprWideNode> wide_nodes;
    uint32_t wide_stack_entries = 0;  // most entries a traversal of wide_nodes can have on its stack
    // The same tree collapsed to the 8-wide slots of include/hiprenderer_c.h "wide8" (Wide8Builder.cpp): what the persistent kernels walk.
    Wide8Result wide8;
};

// `max_depth`: the deepest leaf the builder may produce (root = 1). 62 fits the 64 entry LDS stack.
BvhBuildResult build_bvh(const std::vector<HiprTriangle>& world_triangles, uint32_t max_depth = 62);

// Transform-only update: refits every box of `bvh` (BVH2 child boxes and the wide nodes' quantised child boxes) to `triangles`, which are
// the build's triangles in leaf order with new positions; topology and triangle order are kept. Returns bvh_child_area() of the result, or a negative value when
// the 8-wide tree could not be refitted (refit_wide8) and the caller has to rebuild.
double refit_bvh(BvhBuildResult& bvh, const std::vector<HiprTriangle>& triangles_in_leaf_order);
// Sum of the half-areas of all BVH2 child boxes: grows when a refit leaves the tree with overlapping, stretched boxes.
double bvh_child_area(const BvhBuildResult& bvh);

} // namespace HIPRenderer
