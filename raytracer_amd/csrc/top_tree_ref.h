// top_tree_ref.h -- top-level tree over instances built exactly like the reference builds it.
//
// Needed only for scenes that contain analytic spheres next to other instances: the reference
// declares a sphere's box as +-0.5 although its radius is 1 (shape/Sphere.cpp:11-14), so top-level
// boxes are not conservative and which sphere hits are found depends on the tree (multi-instance
// leaves test every member once the merged leaf box is hit) and on the traversal order.  For such
// scenes the tree, the child order and the early-exit rule of the reference are reproduced; all
// other scenes use the faster builder of bvh_build.h, which gives the same hits whenever every
// box is conservative.
#pragma once
#include <cstdint>
#include <vector>

#include "bvh_build.h"

namespace rt {

struct RefTopTree {
    std::vector<Node64> nodes;    // inner nodes; leaf refs use encode_top_leaf
    std::vector<uint32_t> order;  // leaf slot -> instance index
    bool rootIsLeaf = false;      // the whole list is one leaf (entered without any box test)
    uint32_t maxDepth = 0;        // inner nodes on the longest root-to-leaf path
    uint32_t maxLeaf = 0;
};

// BVHBuilder<SceneRayHitInfo>::buildBVH over an InstancedModelList
// (shape/bvh/BVHBuilder.h:144-361, shape/list/InstancedModelList.cpp:21-58):
// boxes[i] = world box of instance i, keys[i] = its transformed shape centroid.
RefTopTree build_top_tree_reference(const PrimBox* boxes, const float (*keys)[3], uint32_t n);

} // namespace rt
