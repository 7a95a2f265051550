// bvh_build.cpp -- see bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <future>
#include <memory>
#include <thread>

namespace rt {
namespace {

struct Box3 {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* l, const float* h)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], l[a]);
            hi[a] = std::max(hi[a], h[a]);
        }
    }
    void grow(const Box3& b) { grow(b.lo, b.hi); }
    void growPoint(const float* p) { grow(p, p); }
    float area() const
    {
        float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        if (x < 0 || y < 0 || z < 0) return 0.0f;
        return 2.0f * (x * y + y * z + x * z);
    }
};

struct TNode {
    Box3 box;
    TNode* c[2] = {nullptr, nullptr};
    uint32_t first = 0, count = 0; // leaf range when c[0] == nullptr
    uint8_t axis = 0;
};

struct Pool {
    std::vector<std::unique_ptr<TNode[]>> blocks;
    size_t used = kBlock;
    static constexpr size_t kBlock = 4096;
    TNode* alloc()
    {
        if (used == kBlock) {
            blocks.emplace_back(new TNode[kBlock]);
            used = 0;
        }
        return &blocks.back()[used++];
    }
};

int levels_needed(uint32_t n, int hard)
{
    int l = 0;
    while (n > (uint32_t)hard) {
        n = (n + 1) / 2;
        ++l;
    }
    return l;
}

struct Ctx {
    const PrimBox* boxes;
    std::vector<float> cen[3]; // centroids per axis
    uint32_t* order;
    BuildParams params;
    int maxDepth;
    int parallelDepth;
};

struct SubResult {
    TNode* node;
    std::vector<std::unique_ptr<Pool>> pools;
};

constexpr int kBins = 32;

TNode* build_range(Ctx& cx, Pool& pool, std::vector<std::unique_ptr<Pool>>& childPools, uint32_t b, uint32_t e, int depth);

// Chooses the split position (returns mid in (b,e)) and partitions order[b,e) around it.
uint32_t choose_split(Ctx& cx, uint32_t b, uint32_t e, const Box3& bounds, const Box3& cbounds, int depth, bool mustSplit, uint8_t& axisOut)
{
    const uint32_t n = e - b;
    uint32_t* ord = cx.order;
    const float leafCost = (float)n;
    float bestCost = FLT_MAX;
    int bestAxis = -1;
    uint32_t bestMid = 0;
    const float invArea = bounds.area() > 0 ? 1.0f / bounds.area() : 0.0f;

    if (n <= 64) {
        // exact sweep on each axis
        std::vector<uint32_t> tmp(ord + b, ord + e), bestOrder;
        std::vector<float> leftArea(n);
        for (int a = 0; a < 3; ++a) {
            if (!(cbounds.hi[a] > cbounds.lo[a])) continue;
            const float* cen = cx.cen[a].data();
            std::sort(tmp.begin(), tmp.end(), [cen](uint32_t p, uint32_t q) { return cen[p] < cen[q] || (cen[p] == cen[q] && p < q); });
            Box3 acc;
            for (uint32_t i = 0; i < n; ++i) {
                acc.grow(cx.boxes[tmp[i]].lo, cx.boxes[tmp[i]].hi);
                leftArea[i] = acc.area();
            }
            Box3 racc;
            for (uint32_t left = n - 1; left > 0; --left) {
                racc.grow(cx.boxes[tmp[left]].lo, cx.boxes[tmp[left]].hi);
                float cost = cx.params.traversalCost + (leftArea[left - 1] * left + racc.area() * (n - left)) * invArea;
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = a;
                    bestMid = left;
                    bestOrder = tmp;
                }
            }
        }
        if (bestAxis >= 0 && (mustSplit || bestCost < leafCost)) {
            std::copy(bestOrder.begin(), bestOrder.end(), ord + b);
            axisOut = (uint8_t)bestAxis;
            uint32_t mid = b + bestMid;
            // depth guard
            uint32_t big = std::max(bestMid, n - bestMid);
            if (depth + 1 + levels_needed(big, cx.params.maxLeafHard) <= cx.maxDepth) return mid;
            return b + n / 2; // already sorted along bestAxis: median split
        }
        if (!mustSplit) return 0; // leaf
    } else {
        for (int a = 0; a < 3; ++a) {
            const float lo = cbounds.lo[a], ext = cbounds.hi[a] - cbounds.lo[a];
            if (!(ext > 0)) continue;
            const float* cen = cx.cen[a].data();
            const float scale = (float)kBins * (1.0f - 1e-6f) / ext;
            uint32_t cnt[kBins] = {};
            Box3 bb[kBins];
            for (uint32_t i = b; i < e; ++i) {
                uint32_t p = ord[i];
                int k = (int)((cen[p] - lo) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                cnt[k]++;
                bb[k].grow(cx.boxes[p].lo, cx.boxes[p].hi);
            }
            float la[kBins];
            uint32_t lc[kBins];
            Box3 acc;
            uint32_t c = 0;
            for (int k = 0; k < kBins; ++k) {
                if (cnt[k]) acc.grow(bb[k]);
                c += cnt[k];
                la[k] = acc.area();
                lc[k] = c;
            }
            Box3 racc;
            uint32_t rc = 0;
            for (int k = kBins - 1; k > 0; --k) {
                if (cnt[k]) racc.grow(bb[k]);
                rc += cnt[k];
                if (lc[k - 1] == 0 || rc == 0) continue;
                float cost = cx.params.traversalCost + (la[k - 1] * lc[k - 1] + racc.area() * rc) * invArea;
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = a;
                    bestMid = (uint32_t)k; // split before bin k
                }
            }
        }
        if (bestAxis >= 0) {
            const int a = bestAxis;
            const float lo = cbounds.lo[a], ext = cbounds.hi[a] - cbounds.lo[a];
            const float scale = (float)kBins * (1.0f - 1e-6f) / ext;
            const float* cen = cx.cen[a].data();
            const int kb = (int)bestMid;
            uint32_t* midp = std::partition(ord + b, ord + e, [&](uint32_t p) {
                int k = (int)((cen[p] - lo) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                return k < kb;
            });
            uint32_t mid = (uint32_t)(midp - ord);
            axisOut = (uint8_t)a;
            if (mid > b && mid < e) {
                uint32_t big = std::max(mid - b, e - mid);
                if (depth + 1 + levels_needed(big, cx.params.maxLeafHard) <= cx.maxDepth) return mid;
            }
        }
    }
    // Fallback: median split along the widest centroid axis (or by index when all centroids coincide).
    int a = 0;
    float best = -1;
    for (int k = 0; k < 3; ++k) {
        float ext = cbounds.hi[k] - cbounds.lo[k];
        if (ext > best) { best = ext; a = k; }
    }
    axisOut = (uint8_t)a;
    uint32_t mid = b + n / 2;
    const float* cen = cx.cen[a].data();
    std::nth_element(ord + b, ord + mid, ord + e, [cen](uint32_t p, uint32_t q) { return cen[p] < cen[q] || (cen[p] == cen[q] && p < q); });
    return mid;
}

TNode* build_range(Ctx& cx, Pool& pool, std::vector<std::unique_ptr<Pool>>& childPools, uint32_t b, uint32_t e, int depth)
{
    const uint32_t n = e - b;
    TNode* node = pool.alloc();
    Box3 bounds, cbounds;
    for (uint32_t i = b; i < e; ++i) {
        uint32_t p = cx.order[i];
        bounds.grow(cx.boxes[p].lo, cx.boxes[p].hi);
        float c[3] = {cx.cen[0][p], cx.cen[1][p], cx.cen[2][p]};
        cbounds.growPoint(c);
    }
    node->box = bounds;
    const bool canBeLeaf = n <= (uint32_t)cx.params.maxLeafHard;
    const bool depthExhausted = depth >= cx.maxDepth;
    if (n == 1 || (canBeLeaf && depthExhausted)) {
        node->first = b;
        node->count = n;
        return node;
    }
    const bool mustSplit = n > (uint32_t)cx.params.maxLeaf;
    uint8_t axis = 0;
    uint32_t mid = choose_split(cx, b, e, bounds, cbounds, depth, mustSplit, axis);
    if (mid == 0) {
        node->first = b;
        node->count = n;
        return node;
    }
    node->axis = axis;
    if (depth < cx.parallelDepth && n > 65536) {
        auto fut = std::async(std::launch::async, [&cx, b, mid, depth]() {
            SubResult r;
            r.pools.emplace_back(new Pool());
            Pool* p = r.pools.back().get();
            std::vector<std::unique_ptr<Pool>> sub;
            r.node = build_range(cx, *p, sub, b, mid, depth + 1);
            for (auto& s : sub) r.pools.push_back(std::move(s));
            return r;
        });
        node->c[1] = build_range(cx, pool, childPools, mid, e, depth + 1);
        SubResult r = fut.get();
        node->c[0] = r.node;
        for (auto& s : r.pools) childPools.push_back(std::move(s));
    } else {
        node->c[0] = build_range(cx, pool, childPools, b, mid, depth + 1);
        node->c[1] = build_range(cx, pool, childPools, mid, e, depth + 1);
    }
    return node;
}

struct Flattener {
    std::vector<Node64>& out;
    BuiltTree& tree;
    int32_t ref(const TNode* n, int depth)
    {
        if (!n->c[0]) {
            tree.leafCount++;
            tree.maxLeafSize = std::max(tree.maxLeafSize, n->count);
            return encode_leaf(n->first, n->count);
        }
        return emit(n, depth);
    }
    int32_t emit(const TNode* n, int depth)
    {
        const int32_t idx = (int32_t)out.size();
        out.emplace_back();
        tree.maxDepth = std::max(tree.maxDepth, (uint32_t)(depth + 1));
        {
            Node64& d = out[idx];
            for (int a = 0; a < 3; ++a) {
                d.lo0[a] = n->c[0]->box.lo[a]; d.hi0[a] = n->c[0]->box.hi[a];
                d.lo1[a] = n->c[1]->box.lo[a]; d.hi1[a] = n->c[1]->box.hi[a];
            }
            d.axis = n->axis;
            d.pad = 0;
        }
        int32_t r0 = ref(n->c[0], depth + 1);
        out[idx].c0 = r0;
        int32_t r1 = ref(n->c[1], depth + 1);
        out[idx].c1 = r1;
        return idx;
    }
};

} // namespace

BuiltTree build_tree(const PrimBox* boxes, uint32_t n, const BuildParams& params)
{
    BuiltTree tree;
    tree.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) tree.order[i] = i;
    Ctx cx;
    cx.boxes = boxes;
    for (int a = 0; a < 3; ++a) {
        cx.cen[a].resize(n);
        for (uint32_t i = 0; i < n; ++i) cx.cen[a][i] = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    }
    cx.order = tree.order.data();
    cx.params = params;
    cx.maxDepth = params.maxDepth > 0 ? params.maxDepth : kMaxTreeDepth;
    if (levels_needed(n, params.maxLeafHard) > cx.maxDepth) cx.maxDepth = levels_needed(n, params.maxLeafHard); // callers size the bound so this never triggers
    int threads = params.threads > 0 ? params.threads : (int)std::thread::hardware_concurrency();
    cx.parallelDepth = 0;
    while ((1 << cx.parallelDepth) < threads * 2 && cx.parallelDepth < 6) cx.parallelDepth++;

    Pool pool;
    std::vector<std::unique_ptr<Pool>> childPools;
    TNode* root = build_range(cx, pool, childPools, 0, n, 0);

    Flattener fl{tree.nodes, tree};
    tree.nodes.reserve(n);
    if (!root->c[0]) {
        // single-leaf tree: synthesise a root whose second child is an empty leaf no ray can hit
        Node64 r{};
        for (int a = 0; a < 3; ++a) {
            r.lo0[a] = root->box.lo[a]; r.hi0[a] = root->box.hi[a];
            r.lo1[a] = FLT_MAX; r.hi1[a] = -FLT_MAX;
        }
        r.c0 = encode_leaf(root->first, root->count);
        r.c1 = encode_leaf(root->first, 1); // never visited (inverted box)
        r.axis = 0;
        tree.nodes.push_back(r);
        tree.maxDepth = 1;
        tree.maxLeafSize = root->count;
        tree.leafCount = 1;
    } else {
        fl.emit(root, 0);
    }
    tree.nodes.shrink_to_fit();
    return tree;
}

namespace {

struct Collapser {
    const std::vector<Node64>& in;
    std::vector<Node128>& out;
    uint32_t depth4 = 0;
    struct Child { const float* lo; const float* hi; int32_t ref; };
    static bool empty_box(const float* lo, const float* hi) { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }
    int32_t emit(int32_t i, uint32_t depth)
    {
        const int32_t idx = (int32_t)out.size();
        out.emplace_back();
        depth4 = std::max(depth4, depth + 1);
        const Node64& n = in[(size_t)i];
        Child kids[4];
        int nk = 0;
        const Child two[2] = {{n.lo0, n.hi0, n.c0}, {n.lo1, n.hi1, n.c1}};
        for (const Child& c : two) {
            if (empty_box(c.lo, c.hi)) continue;          // the padding child of a single-leaf tree
            if (c.ref >= 0) {
                const Node64& g = in[(size_t)c.ref];
                if (!empty_box(g.lo0, g.hi0)) kids[nk++] = Child{g.lo0, g.hi0, g.c0};
                if (!empty_box(g.lo1, g.hi1)) kids[nk++] = Child{g.lo1, g.hi1, g.c1};
            } else {
                kids[nk++] = c;
            }
        }
        for (int k = 0; k < 4; ++k) {
            Node128& d = out[(size_t)idx];
            if (k < nk) {
                for (int a = 0; a < 3; ++a) { d.box[k][2 * a] = kids[k].lo[a]; d.box[k][2 * a + 1] = kids[k].hi[a]; }
            } else {
                for (int a = 0; a < 3; ++a) { d.box[k][2 * a] = FLT_MAX; d.box[k][2 * a + 1] = -FLT_MAX; }
                d.c[k] = kEmptyChild;
            }
            d.pad[k] = 0;
        }
        for (int k = 0; k < nk; ++k) {
            const int32_t r = kids[k].ref >= 0 ? emit(kids[k].ref, depth + 1) : kids[k].ref;
            out[(size_t)idx].c[k] = r;
        }
        return idx;
    }
};

} // namespace

std::vector<Node128> collapse_tree4(const std::vector<Node64>& binary, uint32_t& depth4)
{
    std::vector<Node128> out;
    out.reserve(binary.size() / 2 + 1);
    Collapser c{binary, out};
    c.emit(0, 0);
    depth4 = c.depth4;
    out.shrink_to_fit();
    return out;
}

namespace {

// One node.  Per axis: origin = the smallest low plane of the node's children, step = a little more than 1/255 of the
// extent, low planes rounded down and high planes rounded up on that grid.  The kernels decode a plane's distance along a
// ray as fma(q, step / d, (origin - ray origin) / d).  Against the exact (plane - ray origin) / d that is off by two kinds
// of rounding: one proportional to the node's extent (q * step, at most 2^-23 of the extent) -- every plane off the grid's
// first point is moved out by 2^-22 of the extent before it is rounded, which covers it (the first point decodes without
// it: q = 0 adds nothing) -- and one proportional to the distance between the ray's origin and the node (about 3 * 2^-24 of
// it: the subtraction, the product with 1 / d and the fma).  The second kind is not the builder's to cover: the kernels
// rely on the 1e-4 padding of triangle boxes for rays that start near the tree and widen the decoded distances themselves
// for the others (plane_offsets, rt_trace_inl.h).  An axis on which some child box is not finite gets a grid that spans all
// finite numbers: the child is entered whatever the ray.
NodeQ64 quantize_node(const Node128& n)
{
    NodeQ64 q{};
    uint8_t plane[4][6];
    float step[3];
    for (int a = 0; a < 3; ++a) {
        double lo = DBL_MAX, hi = -DBL_MAX;
        bool finite = true;
        for (int k = 0; k < 4; ++k) {
            if (n.c[k] == kEmptyChild) continue;
            const double l = n.box[k][2 * a], h = n.box[k][2 * a + 1];
            if (!std::isfinite(l) || !std::isfinite(h) || l > h) finite = false;
            lo = std::min(lo, l);
            hi = std::max(hi, h);
        }
        if (!finite || !(hi - lo < 1e37)) {
            q.origin[a] = -1.5e38f;
            step[a] = 1.2e36f;   // 255 steps = 3.06e38
            for (int k = 0; k < 4; ++k) { plane[k][2 * a] = 0; plane[k][2 * a + 1] = 255; }
            continue;
        }
        const float origin = (float)lo;   // exact: a child's plane
        const double extent = hi - lo;
        float st = (float)(extent / 255.0 * (1.0 + 1e-6));
        while ((double)st * 255.0 < extent * (1.0 + 5e-7)) st = std::nextafter(st, FLT_MAX);
        const double margin = std::ldexp(extent, -22);
        q.origin[a] = origin;
        step[a] = st;
        for (int k = 0; k < 4; ++k) {
            if (n.c[k] == kEmptyChild) { plane[k][2 * a] = 255; plane[k][2 * a + 1] = 0; continue; }
            const double l = n.box[k][2 * a], h = n.box[k][2 * a + 1];
            int ql = 0, qh = 255;
            if (st > 0.0f) {
                ql = (int)std::floor((l - margin - lo) / st);
                qh = (int)std::ceil((h + margin - lo) / st);
                ql = std::min(std::max(ql, 0), 255);
                qh = std::min(std::max(qh, 0), 255);
                while (ql > 0 && lo + (double)ql * st > l - margin) --ql;
                while (qh < 255 && lo + (double)qh * st < h + margin) ++qh;
            } else {
                qh = 0;   // a flat node: every plane is the origin
            }
            plane[k][2 * a] = (uint8_t)ql;
            plane[k][2 * a + 1] = (uint8_t)qh;
        }
    }
    q.sx = step[0]; q.sy = step[1]; q.sz = step[2];
    auto word = [&](int a, int k0) {
        return (uint32_t)plane[k0][2 * a] | ((uint32_t)plane[k0][2 * a + 1] << 8) | ((uint32_t)plane[k0 + 1][2 * a] << 16) | ((uint32_t)plane[k0 + 1][2 * a + 1] << 24);
    };
    q.x01 = word(0, 0); q.x23 = word(0, 2);
    q.y01 = word(1, 0); q.y23 = word(1, 2);
    q.z01 = word(2, 0); q.z23 = word(2, 2);
    for (int k = 0; k < 4; ++k) q.c[k] = n.c[k];
    return q;
}

} // namespace

std::vector<NodeQ64> quantize_tree4(const std::vector<Node128>& nodes, int threads)
{
    std::vector<NodeQ64> out(nodes.size());
    const size_t n = nodes.size();
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), n / 65536 + 1));
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t]() {
            for (size_t i = n * (size_t)t / (size_t)T, e = n * (size_t)(t + 1) / (size_t)T; i < e; ++i) out[i] = quantize_node(nodes[i]);
        });
    for (std::thread& th : pool) th.join();
    return out;
}

} // namespace rt
