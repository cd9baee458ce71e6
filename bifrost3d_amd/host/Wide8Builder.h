// host/Wide8Builder.h -- see Wide8Builder.cpp.
#pragma once

#include "../../include/hiprenderer_c.h"

#include <cstdint>
#include <vector>

namespace HIPRenderer {

struct Wide8Result {
    std::vector<HiprSlot8> slots;       // slot 0 = the root node; empty for an empty scene
    uint32_t height = 0;                // nodes on the longest root-to-leaf chain
    float grid_min[3] = {0, 0, 0}, grid_cell[3] = {1, 1, 1};
    uint32_t node_count = 0, leaf_count = 0, paired_leaves = 0;
};

// Collapses a BVH2 (HiprBvhNode[], leaves referencing ranges of `triangles_in_leaf_order`) into the 8-wide tree of include/hiprenderer_c.h "wide8".
Wide8Result build_wide8(const std::vector<HiprBvhNode>& nodes, const std::vector<HiprTriangle>& triangles_in_leaf_order);

// Transform-only update: same topology, new triangle positions. Rewrites every leaf record from `triangles_in_leaf_order` and requantises every node.
void refit_wide8(Wide8Result& tree, const std::vector<HiprTriangle>& triangles_in_leaf_order);

} // namespace HIPRenderer
