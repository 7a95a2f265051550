// bvh_build.h -- host-side SAH tree builder producing the flattened Node64 layout.
//
// Takes the place of BVHBuilder::buildBVH + BVH::pack (shape/bvh/BVHBuilder.h:340-361,
// shape/bvh/BVH.h:70-113).  Tree shape is free (hits, not trees, are compared with the
// reference), so this builder is its own design: binned SAH (32 bins x 3 axes) for large
// ranges, exact sweep SAH for small ones, std::thread fan-out over the top subtrees, a hard
// depth bound so the kernels' fixed-size stack can never overflow, depth-first flattening.
#pragma once
#include <cstdint>
#include <vector>

#include "rt_types.h"

namespace rt {

struct PrimBox {
    float lo[3], hi[3];
};

struct BuildParams {
    float traversalCost = 1.0f;   // relative to one primitive test
    int maxLeaf = 4;              // split whenever a range is larger than this (if it can be split)
    int maxLeafHard = kMaxLeafHard; // largest leaf the depth bound may force (1 for the top-level tree)
    int threads = 0;              // 0 = hardware concurrency
    int maxDepth = 0;             // bound on inner nodes along any root-to-leaf path (0 = kMaxTreeDepth)
};

struct BuiltTree {
    std::vector<Node64> nodes;    // nodes[0] is the root; child indices are relative to nodes[0]
    std::vector<uint32_t> order;  // leaf slot -> primitive index
    uint32_t maxDepth = 0;
    uint32_t maxLeafSize = 0;
    uint64_t leafCount = 0;
};

// Builds a tree over n primitive boxes.  n >= 1.  A single-leaf tree still gets a root node
// (child 1 is an empty leaf with an inverted box that no ray can hit).
BuiltTree build_tree(const PrimBox* boxes, uint32_t n, const BuildParams& params);

// The four-way form of a built binary tree: every inner node at an even level absorbs its two children (see Node128).
// depth4 receives the number of four-way inner nodes on the longest root-to-leaf path (= ceil(binary depth / 2)).
std::vector<Node128> collapse_tree4(const std::vector<Node64>& binary, uint32_t& depth4);

// The quantised form the kernels fetch (see NodeQ64): every decoded child box contains the exact one, with room for the
// rounding of the kernels' decode arithmetic.
std::vector<NodeQ64> quantize_tree4(const std::vector<Node128>& nodes, int threads);

} // namespace rt
