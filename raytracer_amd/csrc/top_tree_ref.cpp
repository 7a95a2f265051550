// top_tree_ref.cpp -- see top_tree_ref.h.
#include "top_tree_ref.h"

#include <algorithm>
#include <memory>

namespace rt {
namespace {

struct Bx {
    float lo[3], hi[3];
};
// AABB::merge, shape/AABB.cpp:64-78
Bx merge(const Bx& a, const Bx& b)
{
    Bx r;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = std::min({a.lo[k], a.hi[k], b.lo[k], b.hi[k]});
        r.hi[k] = std::max({a.lo[k], a.hi[k], b.lo[k], b.hi[k]});
    }
    return r;
}
// AABB::getSurfaceArea, shape/AABB.cpp:55-62: float products, doubled in double
double area(const Bx& a)
{
    float x = a.hi[0] - a.lo[0], y = a.hi[1] - a.lo[1], z = a.hi[2] - a.lo[2];
    return 2.0 * (x * y + z * y + x * z);
}
Bx from(const PrimBox& p)
{
    Bx b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = p.lo[k]; b.hi[k] = p.hi[k]; }
    return b;
}

struct N {
    Bx box;
    std::unique_ptr<N> c[2];
    uint32_t begin = 0, end = 0;
    uint8_t axis = 0;
    bool leaf = false;
};

struct B {
    const PrimBox* boxes;
    const float (*keys)[3];
    std::vector<uint32_t>& order;

    void sortRange(uint32_t b, uint32_t e, int axis)
    {
        // InstancedModelList::sortByCentroid: std::sort on the transformed centroid, :48-58
        const float (*k)[3] = keys;
        std::sort(order.begin() + b, order.begin() + e, [k, axis](uint32_t p, uint32_t q) { return k[p][axis] < k[q][axis]; });
    }
    // BVHBuilder::buildNode / buildTree, shape/bvh/BVHBuilder.h:144-294, costs 1 and 4 (:347-348).
    // Kept as the reference has them: leftArea[leftSideShapeCount] (one shape too many on the
    // left, :198), and the node's child-order axis / the children's presorted axis are the last
    // axis of the sweep even when the list was re-sorted by the best axis afterwards (:265-272).
    std::unique_ptr<N> build(uint32_t b, uint32_t e, int presorted)
    {
        const uint32_t count = e - b;
        auto n = std::make_unique<N>();
        if (count == 1) {
            n->box = from(boxes[order[b]]);
            n->leaf = true; n->begin = b; n->end = e;
            return n;
        }
        std::vector<float> leftArea(count);
        double bestCost = count * 1.0;
        bool leavesAreCheapest = true;
        uint32_t bestSplit = 0;
        int bestAxis = 0, cur = presorted;
        Bx total{};
        int axes[3] = {presorted, 0, 0};
        for (int a = 0, k = 1; a < 3; ++a)
            if (a != presorted) axes[k++] = a;
        for (int ai = 0; ai < 3; ++ai) {
            const int axis = axes[ai];
            if (cur != axis) { sortRange(b, e, axis); cur = axis; }
            Bx acc = from(boxes[order[b]]);
            leftArea[0] = (float)area(acc);
            for (uint32_t i = 1; i < count; ++i) {
                acc = merge(acc, from(boxes[order[b + i]]));
                leftArea[i] = (float)area(acc);
            }
            total = acc;
            const double totalSurf = leftArea[count - 1];
            acc = from(boxes[order[b + count - 1]]);
            for (uint32_t left = count - 1; left > 0; --left) {
                const double cost = 4.0 + ((leftArea[left] / totalSurf) * left * 1.0) + ((area(acc) / totalSurf) * (count - left) * 1.0);
                if (cost < bestCost) { leavesAreCheapest = false; bestSplit = left; bestAxis = axis; bestCost = cost; }
                acc = merge(acc, from(boxes[order[b + left - 1]]));
            }
        }
        n->box = total;
        if (leavesAreCheapest) {
            n->leaf = true; n->begin = b; n->end = e;
            return n;
        }
        if (cur != bestAxis) sortRange(b, e, bestAxis);
        n->axis = (uint8_t)cur;
        n->c[0] = build(b, b + bestSplit, cur);
        n->c[1] = build(b + bestSplit, e, cur);
        return n;
    }
};

int32_t emit(const N* n, RefTopTree& t, uint32_t depth)
{
    if (n->leaf) {
        t.maxLeaf = std::max(t.maxLeaf, n->end - n->begin);
        return encode_top_leaf(n->begin, n->end - n->begin);
    }
    const int32_t idx = (int32_t)t.nodes.size();
    t.nodes.emplace_back();
    t.maxDepth = std::max(t.maxDepth, depth + 1);
    for (int k = 0; k < 3; ++k) {
        t.nodes[idx].lo0[k] = n->c[0]->box.lo[k]; t.nodes[idx].hi0[k] = n->c[0]->box.hi[k];
        t.nodes[idx].lo1[k] = n->c[1]->box.lo[k]; t.nodes[idx].hi1[k] = n->c[1]->box.hi[k];
    }
    t.nodes[idx].axis = n->axis;
    t.nodes[idx].pad = 0;
    const int32_t r0 = emit(n->c[0].get(), t, depth + 1);
    t.nodes[idx].c0 = r0;
    const int32_t r1 = emit(n->c[1].get(), t, depth + 1);
    t.nodes[idx].c1 = r1;
    return idx;
}

} // namespace

RefTopTree build_top_tree_reference(const PrimBox* boxes, const float (*keys)[3], uint32_t n)
{
    RefTopTree t;
    t.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) t.order[i] = i;
    B b{boxes, keys, t.order};
    b.sortRange(0, n, 0); // buildBVH sorts by x first, BVHBuilder.h:343
    auto root = b.build(0, n, 0);
    if (root->leaf) {
        t.rootIsLeaf = true;
        t.maxLeaf = n;
        return t;
    }
    emit(root.get(), t, 0);
    return t;
}

} // namespace rt
