// host/Wide8Builder.h -- see Wide8Builder.cpp.
#pragma once

#include "../../include/hiprenderer_c.h"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace HIPRenderer {

struct Wide8Result {
    std::vector<HiprSlot8> slots;       // slot 0 = the root node; empty for an empty scene
    uint32_t height = 0;                // nodes on the longest root-to-leaf chain
    float grid_min[3] = {0, 0, 0}, grid_cell[3] = {1, 1, 1};
    uint32_t node_count = 0, leaf_count = 0, paired_leaves = 0;
};

// The scene's triangles in the order the BVH2's leaves reference them, without a copy: element k is triangles[order[k]] (order == nullptr: triangles[k]).
struct OrderedTriangles {
    const HiprTriangle* triangles = nullptr;
    const uint32_t* order = nullptr;
    size_t count = 0;
    const HiprTriangle& operator[](size_t k) const { return triangles[order ? order[k] : k]; }
    size_t size() const { return count; }
    bool empty() const { return count == 0; }
};

// Collapses a BVH2 (HiprBvhNode[], leaves referencing ranges of `triangles_in_leaf_order`) into the 8-wide tree of include/hiprenderer_c.h "wide8".
Wide8Result build_wide8(const std::vector<HiprBvhNode>& nodes, const OrderedTriangles& triangles_in_leaf_order);

// Transform-only update: same topology, new triangle positions. Rewrites every leaf record from `triangles_in_leaf_order` and requantises every node.
// false: a paired record's shared corners are no longer bit-identical after the move -- the tree is stale and must be rebuilt (nothing else is wrong).
bool refit_wide8(Wide8Result& tree, const OrderedTriangles& triangles_in_leaf_order);

} // namespace HIPRenderer
