/*
 * rt_oracle.cpp -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY:
 * see rt_oracle.h for what it is for, who may use it and how far its parity is pinned.
 *
 * Every function cites the reference lines it restates (paths relative to
 * raytracer/src/lib/ of Wouterdek/raytracer).  Arithmetic follows the reference's types:
 * fp32 vectors (Eigen::Vector3f), fp64 4x4 transforms (Eigen::Matrix4d), fp64 sphere test.
 * Build with -ffp-contract=off: the product kernels are built the same way so that hit
 * decisions agree bit for bit.
 *
 * Assumed Eigen evaluation orders (Eigen is not vendored by the reference and not installed
 * here, so these are restated from Eigen's published sources, unpinned):
 *   - 3-vector reductions (dot, squaredNorm): x + (y + z)   (unrolled non-vectorised redux)
 *   - 3x3 determinant: m00*(m11*m22-m12*m21) - m01*(m10*m22-m12*m20) + m02*(m10*m21-m11*m20)
 *   - Matrix4d * Vector4d: ((m0*v0 + m1*v1) + m2*v2) + m3*v3 per row, unfused
 *   - vector * double scalar on a float vector narrows the scalar to float first
 *   - normalize(): v / sqrt(squaredNorm) by true division
 *
 * The reference RNG (thread_local pcg32 seeded from std::random_device, math/FastRandom.h:6-67)
 * cannot be seeded, so every Rand::unit() call site is replaced by a counter-based draw keyed
 * by (seed, pixel, sample, path vertex, dimension); the HIP integrator uses the same keys, which
 * makes the two images comparable sample for sample.
 */
#include "rt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

constexpr double PI = 3.14159265358979323; // math/Constants.h

// ------------------------------------------------------------------------------------------
// fp32 3-vectors (math/Vector3.h: Eigen::Vector3f)
// ------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 mk(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a)
{
    float z = dot(a, a);
    if (z > 0.0f) return a / std::sqrt(z);
    return a;
}

struct RGBf { // film/RGB.h
    float r = 0, g = 0, b = 0;
};
inline RGBf rgb_add(RGBf a, RGBf b) { return RGBf{a.r + b.r, a.g + b.g, a.b + b.b}; }
inline RGBf rgb_scale_d(RGBf a, double s) { return RGBf{(float)(a.r * s), (float)(a.g * s), (float)(a.b * s)}; } // RGB.h:44-47,125-135
inline RGBf rgb_scale_f(RGBf a, float s) { return RGBf{a.r * s, a.g * s, a.b * s}; }                              // RGB.h:49-52
inline RGBf rgb_div_d(RGBf a, double d) { return rgb_scale_d(a, 1.0 / d); }                                      // RGB.h:60-63
inline RGBf rgb_div_f(RGBf a, float d) { return rgb_scale_f(a, 1.0f / d); }                                      // RGB.h:65-68
inline RGBf rgb_div_i(RGBf a, int d) { return rgb_scale_f(a, 1.0f / (float)d); }                                 // RGB.h:70-73
inline RGBf rgb_mul(RGBf a, RGBf b) { return RGBf{a.r * b.r, a.g * b.g, a.b * b.b}; }

// ------------------------------------------------------------------------------------------
// fp64 4x4 transforms (math/Transformation.h:71-113), row-major storage
// ------------------------------------------------------------------------------------------
struct Xf {
    double m[16];
    double inv[16];
};
inline void mat_vec4(const double* M, double x, double y, double z, double w, double out[4])
{
    for (int i = 0; i < 4; ++i)
        out[i] = ((M[i * 4 + 0] * x + M[i * 4 + 1] * y) + M[i * 4 + 2] * z) + M[i * 4 + 3] * w;
}
// Transformation::transform(Point) :71-76 and transformInverse(Point) :78-81 (hnormalized + cast)
inline V3 xf_point(const double* M, V3 p)
{
    double v[4];
    mat_vec4(M, p.x, p.y, p.z, 1.0, v);
    return V3{(float)(v[0] / v[3]), (float)(v[1] / v[3]), (float)(v[2] / v[3])};
}
// Transformation::transform(Vector3) :83-87 and transformInverse(Vector3) :89-93
inline V3 xf_vector(const double* M, V3 p)
{
    double v[4];
    mat_vec4(M, p.x, p.y, p.z, 0.0, v);
    return V3{(float)v[0], (float)v[1], (float)v[2]};
}
// Transformation::transformNormal :95-99 (row vector times inverse)
inline V3 xf_normal(const double* INV, V3 n)
{
    double r[3];
    for (int j = 0; j < 3; ++j)
        r[j] = (((double)n.x * INV[0 * 4 + j] + (double)n.y * INV[1 * 4 + j]) + (double)n.z * INV[2 * 4 + j]) + 0.0 * INV[3 * 4 + j];
    return V3{(float)r[0], (float)r[1], (float)r[2]};
}

// ------------------------------------------------------------------------------------------
// Triangle (math/Triangle.h)
// ------------------------------------------------------------------------------------------
inline float det3(V3 r0, V3 r1, V3 r2)
{
    float h0 = r0.x * (r1.y * r2.z - r1.z * r2.y);
    float h1 = r0.y * (r1.x * r2.z - r1.z * r2.x);
    float h2 = r0.z * (r1.x * r2.y - r1.y * r2.x);
    return (h0 - h1) + h2;
}

// Triangle::intersect, math/Triangle.h:34-101
inline bool tri_intersect(V3 o, V3 d, V3 a, V3 b, V3 c, float& beta_out, float& gamma_out, double& t_out)
{
    V3 x = a - o;
    V3 e1 = a - b;
    V3 e2 = a - c;
    float pDet = det3(d, e1, e2);
    if (pDet == 0) return false;
    float invDet = 1.0f / pDet;
    float p0Det = det3(x, e1, e2);
    float t = p0Det * invDet;
    if (t < 0) return false;
    float p1Det = det3(d, x, e2);
    float beta = p1Det * invDet;
    if (beta < 0) return false;
    float p2Det = det3(d, e1, x);
    float gamma = p2Det * invDet;
    if (gamma < 0) return false;
    if (gamma + beta > 1) return false;
    beta_out = beta;
    gamma_out = gamma;
    t_out = t;
    return true;
}

// Triangle::getSurfaceArea, math/Triangle.h:7-18
inline float tri_area(V3 a, V3 b, V3 c)
{
    V3 v1 = b - a;
    float v1Length = norm(v1);
    V3 v1n = v1 / v1Length;
    V3 v2 = c - a;
    V3 proj = v1n * dot(v2, v1n);
    float height = norm(v2 - proj);
    return v1Length * height / 2.0f;
}

// ------------------------------------------------------------------------------------------
// AABB (shape/AABB.h, shape/AABB.cpp)
// ------------------------------------------------------------------------------------------
struct AABB {
    V3 s, e;
};
inline float fmax_ref(float a, float b) { return (a < b) ? b : a; } // std::max
inline float fmin_ref(float a, float b) { return (b < a) ? b : a; } // std::min

// AABB::getIntersections, shape/AABB.h:67-121
inline bool aabb_intersections(const AABB& bx, V3 o, V3 d, float& t0, float& t1)
{
    float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
    float a = 1.0f / d.x;
    if (a >= 0) { tx_min = (bx.s.x - o.x) * a; tx_max = (bx.e.x - o.x) * a; }
    else        { tx_min = (bx.e.x - o.x) * a; tx_max = (bx.s.x - o.x) * a; }
    float b = 1.0f / d.y;
    if (b >= 0) { ty_min = (bx.s.y - o.y) * b; ty_max = (bx.e.y - o.y) * b; }
    else        { ty_min = (bx.e.y - o.y) * b; ty_max = (bx.s.y - o.y) * b; }
    float c = 1.0f / d.z;
    if (c >= 0) { tz_min = (bx.s.z - o.z) * c; tz_max = (bx.e.z - o.z) * c; }
    else        { tz_min = (bx.e.z - o.z) * c; tz_max = (bx.s.z - o.z) * c; }
    t0 = fmax_ref(tx_min, ty_min);
    t0 = fmax_ref(t0, tz_min);
    t1 = fmin_ref(tx_max, ty_max);
    t1 = fmin_ref(t1, tz_max);
    bool isIntersecting = t0 <= t1;
    t0 = fmax_ref(0.0f, t0);
    return isIntersecting;
}
// AABB::getIntersection, shape/AABB.h:123-131
inline float aabb_intersection(const AABB& bx, V3 o, V3 d)
{
    float t0, t1;
    if (!aabb_intersections(bx, o, d, t0, t1)) return -1.0f;
    return t0;
}
// AABB::merge, shape/AABB.cpp:64-78
inline AABB aabb_merge(const AABB& a, const AABB& b)
{
    AABB r;
    r.s.x = std::min({a.s.x, a.e.x, b.s.x, b.e.x}); r.e.x = std::max({a.s.x, a.e.x, b.s.x, b.e.x});
    r.s.y = std::min({a.s.y, a.e.y, b.s.y, b.e.y}); r.e.y = std::max({a.s.y, a.e.y, b.s.y, b.e.y});
    r.s.z = std::min({a.s.z, a.e.z, b.s.z, b.e.z}); r.e.z = std::max({a.s.z, a.e.z, b.s.z, b.e.z});
    return r;
}
// AABB::getSurfaceArea, shape/AABB.cpp:55-62
inline double aabb_area(const AABB& a)
{
    V3 dg = a.e - a.s;
    float x = dg.x, y = dg.y, z = dg.z;
    return 2.0 * (x * y + z * y + x * z);
}
// AABB::getAABBOfTransformed, shape/AABB.cpp:19-46
// Box::getCorners, shape/Box.cpp (the eight corners AABB::getAABBOfTransformed maps, shape/AABB.cpp:33-46)
inline void aabb_corners(const AABB& a, V3 c[8])
{
    const V3 k[8] = {a.s, V3{a.s.x, a.s.y, a.e.z}, V3{a.s.x, a.e.y, a.s.z}, V3{a.e.x, a.s.y, a.s.z},
                     V3{a.s.x, a.e.y, a.e.z}, V3{a.e.x, a.s.y, a.e.z}, V3{a.e.x, a.e.y, a.s.z}, a.e};
    for (int i = 0; i < 8; ++i) c[i] = k[i];
}

inline AABB aabb_of_transformed(const AABB& a, const double* M)
{
    V3 c[8];
    aabb_corners(a, c);
    AABB r;
    r.s = V3{INFINITY, INFINITY, INFINITY};
    r.e = V3{-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < 8; ++i) {
        V3 p = xf_point(M, c[i]);
        r.s.x = std::min(r.s.x, p.x); r.s.y = std::min(r.s.y, p.y); r.s.z = std::min(r.s.z, p.z);
        r.e.x = std::max(r.e.x, p.x); r.e.y = std::max(r.e.y, p.y); r.e.z = std::max(r.e.z, p.z);
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// Sphere (shape/Sphere.cpp)
// ------------------------------------------------------------------------------------------
// Sphere::intersect, shape/Sphere.cpp:16-59, including largestT = min(t1,t0) (:38): a ray that
// starts inside the sphere misses.
inline bool sphere_intersect(V3 o, V3 dir, double& t_out)
{
    double a = dot(dir, dir);
    double b = 2.0 * (dot(dir, o));
    double c = dot(o, o) - 1.0;
    double d = b * b - 4.0 * a * c;
    if (d < 0) return false;
    double dr = std::sqrt(d);
    double q = -0.5 * (b < 0 ? (b - dr) : (b + dr));
    double t0 = q / a;
    double t1 = c / q;
    double smallestT = std::min(t0, t1);
    double largestT = std::min(t1, t0);
    double t;
    if (smallestT >= 0) t = smallestT;
    else if (largestT >= 0) t = largestT;
    else return false;
    t_out = t;
    return true;
}
// Sphere::getAABB, shape/Sphere.cpp:11-14 (half the true extent; reproduced)
inline AABB sphere_aabb() { return AABB{V3{-0.5f, -0.5f, -0.5f}, V3{0.5f, 0.5f, 0.5f}}; }

// ------------------------------------------------------------------------------------------
// Generic binary SAH BVH (shape/bvh/BVHBuilder.h, BVHNode.h)
// ------------------------------------------------------------------------------------------
struct Node {
    AABB box;
    int child[2] = {-1, -1};
    uint32_t begin = 0, end = 0; // leaf: range in the sorted order array
    uint8_t axis = 0;            // childOrder
    bool leaf = false;
};

struct BuildList { // IShapeList: count / getAABB(i) / sortByCentroid over a sub-range
    std::vector<AABB> boxes;        // per original element
    std::vector<V3> sortKey;        // per original element: the value compared by sortByCentroid
    std::vector<uint32_t> order;    // current arrangement (position -> original element)
};

struct PNode {
    AABB box;
    std::unique_ptr<PNode> c[2];
    uint32_t begin = 0, end = 0;
    uint8_t axis = 0;
    bool leaf = false;
};

struct Builder {
    BuildList& L;
    const double costPerIntersection = 1; // BVHBuilder.h:347
    const double costPerTraversal = 4;    // BVHBuilder.h:348

    explicit Builder(BuildList& l) : L(l) {}

    void sortRange(uint32_t b, uint32_t e, int axis)
    {
        const auto& key = L.sortKey;
        std::sort(L.order.begin() + b, L.order.begin() + e,
                  [&key, axis](uint32_t p, uint32_t q) { return key[p][axis] < key[q][axis]; });
    }
    // BVHBuilder::calculateSAH, BVHBuilder.h:135-139
    double sah(size_t n1, double a1, size_t n2, double a2, double total) const
    {
        return costPerTraversal + ((a1 / total) * n1 * costPerIntersection) + ((a2 / total) * n2 * costPerIntersection);
    }

    // BVHBuilder::buildNode + buildTree, BVHBuilder.h:144-294.  Quirks kept: leftArea is indexed
    // by leftSideShapeCount (one element too many on the left, :198); the node's childOrder and
    // the children's presorted axis are the LAST axis sorted in the sweep, not the axis the list
    // was finally re-sorted by (:265-272).
    std::unique_ptr<PNode> build(uint32_t b, uint32_t e, int presortedAxis, int depth)
    {
        const uint32_t count = e - b;
        auto node = std::make_unique<PNode>();
        if (count == 1) {
            node->box = L.boxes[L.order[b]];
            node->leaf = true;
            node->begin = b;
            node->end = e;
            return node;
        }
        std::vector<float> leftArea(count);
        double bestCost = count * costPerIntersection;
        bool leavesAreCheapest = true;
        uint32_t bestSplit = 0;
        int bestAxis = 0;
        int currentSorting = presortedAxis;
        AABB totalAABB{};
        int axes[3];
        axes[0] = presortedAxis; // getAxesStartingWith, math/Axis.h:13-29
        {
            int k = 1;
            for (int a = 0; a < 3; ++a)
                if (a != presortedAxis) axes[k++] = a;
        }
        for (int ai = 0; ai < 3; ++ai) {
            int axis = axes[ai];
            if (currentSorting != axis) {
                sortRange(b, e, axis);
                currentSorting = axis;
            }
            AABB acc = L.boxes[L.order[b]];
            leftArea[0] = (float)aabb_area(acc);
            for (uint32_t i = 1; i < count; ++i) {
                acc = aabb_merge(acc, L.boxes[L.order[b + i]]);
                leftArea[i] = (float)aabb_area(acc);
            }
            totalAABB = acc;
            const double totalSurf = leftArea[count - 1];
            acc = L.boxes[L.order[b + count - 1]];
            for (uint32_t left = count - 1; left > 0; --left) {
                double cur = sah(left, leftArea[left], count - left, aabb_area(acc), totalSurf);
                if (cur < bestCost) {
                    leavesAreCheapest = false;
                    bestSplit = left;
                    bestAxis = axis;
                    bestCost = cur;
                }
                acc = aabb_merge(acc, L.boxes[L.order[b + left - 1]]);
            }
        }
        node->box = totalAABB;
        if (leavesAreCheapest) {
            node->leaf = true;
            node->begin = b;
            node->end = e;
            return node;
        }
        if (currentSorting != bestAxis) sortRange(b, e, bestAxis);
        node->axis = (uint8_t)currentSorting;
        const uint32_t mid = b + bestSplit;
        if (depth < 3 && count > 20000) { // NodeBuildingTask fan-out, BVHBuilder.h:89-122
            auto fut = std::async(std::launch::async, [this, b, mid, currentSorting, depth] { return build(b, mid, currentSorting, depth + 1); });
            node->c[1] = build(mid, e, currentSorting, depth + 1);
            node->c[0] = fut.get();
        } else {
            node->c[0] = build(b, mid, currentSorting, depth + 1);
            node->c[1] = build(mid, e, currentSorting, depth + 1);
        }
        return node;
    }
};

int flatten(const PNode* p, std::vector<Node>& out)
{
    int idx = (int)out.size();
    out.emplace_back();
    out[idx].box = p->box;
    out[idx].leaf = p->leaf;
    out[idx].begin = p->begin;
    out[idx].end = p->end;
    out[idx].axis = p->axis;
    if (!p->leaf) {
        int c0 = flatten(p->c[0].get(), out);
        int c1 = flatten(p->c[1].get(), out);
        out[idx].child[0] = c0;
        out[idx].child[1] = c1;
    }
    return idx;
}

// BVHBuilder::buildBVH, BVHBuilder.h:340-361
std::vector<Node> build_bvh(BuildList& L)
{
    const uint32_t n = (uint32_t)L.order.size();
    Builder bld(L);
    bld.sortRange(0, n, 0);
    auto root = bld.build(0, n, 0, 0);
    std::vector<Node> nodes;
    flatten(root.get(), nodes);
    return nodes;
}

// ------------------------------------------------------------------------------------------
// Scene data
// ------------------------------------------------------------------------------------------
struct Mesh { // TriangleMesh + TriangleMeshData, shape/TriangleMesh.h
    std::vector<V3> vertices, normals;
    std::vector<uint32_t> indices;     // original order, 3 per triangle
    std::vector<uint32_t> triMaterial; // optional
    std::vector<float> texCoords;      // optional, 2 per vertex (TriangleMeshData::texCoords)
    std::vector<uint32_t> order;       // sorted position -> original triangle (= permutation)
    std::vector<Node> nodes;
    AABB box;                          // TriangleMesh::getAABB(), shape/TriangleMesh.cpp:78-97
    V3 centroid;                       // TriangleMesh::getCentroid(), shape/TriangleMesh.cpp:57-76
    uint32_t ntri = 0;
};

struct Instance {
    int shape;
    uint32_t material;
    Xf xf;
    AABB worldBox; // InstancedModelList::getAABB, shape/list/InstancedModelList.cpp:21-26
};

struct Counters {
    uint32_t boxes = 0, tris = 0;
};
thread_local Counters g_cnt;

struct MeshHit {
    bool ok = false;
    double t = 0;
    float beta = 0, gamma = 0;
    uint32_t pos = 0; // sorted position (the reference's triangleIndex)
};

// TriangleMesh::getAABB(i), shape/TriangleMesh.cpp:177-191
AABB tri_box(V3 p1, V3 p2, V3 p3)
{
    float sx = std::min({p1.x, p2.x, p3.x}), ex = std::max({p1.x, p2.x, p3.x});
    float sy = std::min({p1.y, p2.y, p3.y}), ey = std::max({p1.y, p2.y, p3.y});
    float sz = std::min({p1.z, p2.z, p3.z}), ez = std::max({p1.z, p2.z, p3.z});
    return AABB{V3{(float)(sx - 1E-4), (float)(sy - 1E-4), (float)(sz - 1E-4)},
                V3{(float)(ex + 1E-4), (float)(ey + 1E-4), (float)(ez + 1E-4)}};
}

void mesh_build(Mesh& m)
{
    BuildList L;
    L.boxes.resize(m.ntri);
    L.sortKey.resize(m.ntri);
    L.order.resize(m.ntri);
    for (uint32_t i = 0; i < m.ntri; ++i) {
        V3 p1 = m.vertices[m.indices[3 * i]], p2 = m.vertices[m.indices[3 * i + 1]], p3 = m.vertices[m.indices[3 * i + 2]];
        L.boxes[i] = tri_box(p1, p2, p3);
        L.sortKey[i] = (p1 + p2) + p3; // comparator of sortByCentroidImpl, shape/TriangleMesh.cpp:218-237
        L.order[i] = i;
    }
    // TriangleMesh::getAABB() and getCentroid() are evaluated on the unsorted mesh by the caller
    AABB bx = L.boxes[0];
    for (uint32_t i = 1; i < m.ntri; ++i) bx = aabb_merge(bx, L.boxes[i]);
    m.box = bx;
    {
        auto cen = [&](uint32_t i) {
            V3 p1 = m.vertices[m.indices[3 * i]], p2 = m.vertices[m.indices[3 * i + 1]], p3 = m.vertices[m.indices[3 * i + 2]];
            return ((p1 + p2) + p3) / 3.0f;
        };
        const float cnt = (float)m.ntri;
        V3 c = cen(0) / cnt;
        for (uint32_t i = 1; i < m.ntri; ++i) c = c + cen(i) / cnt;
        m.centroid = c;
    }
    m.nodes = build_bvh(L);
    m.order = std::move(L.order);
}

// TriangleMesh::intersect (leaf loop), shape/TriangleMesh.cpp:99-152
MeshHit mesh_leaf_intersect(const Mesh& m, uint32_t begin, uint32_t end, V3 o, V3 d)
{
    MeshHit best;
    for (uint32_t pos = begin; pos < end; ++pos) {
        uint32_t tri = m.order[pos];
        V3 a = m.vertices[m.indices[3 * tri]], b = m.vertices[m.indices[3 * tri + 1]], c = m.vertices[m.indices[3 * tri + 2]];
        float beta, gamma;
        double t;
        g_cnt.tris++;
        if (tri_intersect(o, d, a, b, c, beta, gamma, t) && (!best.ok || best.t > t)) {
            best.ok = true;
            best.t = t;
            best.beta = beta;
            best.gamma = gamma;
            best.pos = pos;
        }
    }
    return best;
}
// TriangleMesh::testVisibility (leaf loop), shape/TriangleMesh.cpp:154-175
bool mesh_leaf_visibility(const Mesh& m, uint32_t begin, uint32_t end, V3 o, V3 d, float maxT)
{
    for (uint32_t pos = begin; pos < end; ++pos) {
        uint32_t tri = m.order[pos];
        V3 a = m.vertices[m.indices[3 * tri]], b = m.vertices[m.indices[3 * tri + 1]], c = m.vertices[m.indices[3 * tri + 2]];
        float beta, gamma;
        double t;
        g_cnt.tris++;
        if (tri_intersect(o, d, a, b, c, beta, gamma, t) && t <= maxT) return true;
    }
    return false;
}

// BVHNode::traceRay, shape/bvh/BVHNode.h:306-397, with LeafFn(begin,end) -> hit
template <typename Hit, typename LeafFn>
Hit bvh_trace(const std::vector<Node>& nodes, int ni, V3 o, V3 d, const LeafFn& leafFn)
{
    const Node& n = nodes[ni];
    if (n.leaf) return leafFn(n.begin, n.end);
    float ts[2];
    for (int i = 0; i < 2; ++i) {
        g_cnt.boxes++;
        ts[i] = aabb_intersection(nodes[n.child[i]].box, o, d);
    }
    Hit best{};
    float order = d[n.axis];
    if (order > 0) {
        for (int i = 0; i < 2; ++i) {
            if (ts[i] >= 0) {
                Hit h = bvh_trace<Hit>(nodes, n.child[i], o, d, leafFn);
                if (h.ok && (!best.ok || best.t > h.t)) {
                    best = h;
                    bool earlyExit = true;
                    for (int j = i + 1; j < 2; ++j)
                        if (ts[j] >= 0 && h.t > ts[j]) { earlyExit = false; break; }
                    if (earlyExit) break;
                }
            }
        }
    } else if (order < 0) {
        for (int i = 1; i >= 0; --i) {
            if (ts[i] >= 0) {
                Hit h = bvh_trace<Hit>(nodes, n.child[i], o, d, leafFn);
                if (h.ok && (!best.ok || best.t > h.t)) {
                    best = h;
                    bool earlyExit = true;
                    for (int j = i - 1; j >= 0; --j)
                        if (ts[j] >= 0 && h.t > ts[j]) { earlyExit = false; break; }
                    if (earlyExit) break;
                }
            }
        }
    } else {
        for (int i = 1; i >= 0; --i) {
            if (ts[i] >= 0) {
                Hit h = bvh_trace<Hit>(nodes, n.child[i], o, d, leafFn);
                if (h.ok && (!best.ok || best.t > h.t)) best = h;
            }
        }
    }
    return best;
}

// BVHNode::testVisibility, shape/bvh/BVHNode.h:399-446
template <typename LeafFn>
bool bvh_visibility(const std::vector<Node>& nodes, int ni, V3 o, V3 d, float maxT, const LeafFn& leafFn)
{
    const Node& n = nodes[ni];
    if (n.leaf) return leafFn(n.begin, n.end);
    bool hitb[2];
    for (int i = 0; i < 2; ++i) {
        float t0, t1;
        g_cnt.boxes++;
        hitb[i] = aabb_intersections(nodes[n.child[i]].box, o, d, t0, t1) && t0 <= maxT;
    }
    float order = d[n.axis];
    if (order < 0) {
        for (int i = 1; i >= 0; --i)
            if (hitb[i] && bvh_visibility(nodes, n.child[i], o, d, maxT, leafFn)) return true;
    } else {
        for (int i = 0; i < 2; ++i)
            if (hitb[i] && bvh_visibility(nodes, n.child[i], o, d, maxT, leafFn)) return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// Materials / lights
// ------------------------------------------------------------------------------------------
struct SceneHit { // SceneRayHitInfo + RayHitInfo
    bool ok = false;
    V3 ro{}, rd{}; // ray
    double t = 0;
    V3 normal{};   // world, normalised (InstancedModelList.cpp:129-130)
    uint32_t instPos = 0; // position in the sorted instance order
    uint32_t inst = RTGPU_INVALID_ID;
    uint32_t prim = RTGPU_INVALID_ID; // original triangle index
    float beta = 0, gamma = 0;
    float u = 0, v = 0;   // RayHitInfo::texCoord
    V3 tangent{};         // RayHitInfo::tangent, world space (InstancedModelList.cpp:131); zero for spheres (unset in the reference, Sphere.cpp:56)
    V3 hitpoint() const { return ro + (rd * (float)t); } // RayHitInfo::getHitpoint, math/RayHitInfo.h:48-51
};

} // namespace

struct orc_scene {
    std::vector<Mesh> meshes;
    std::vector<Instance> instances;
    std::vector<uint32_t> instOrder; // sorted position -> instance index
    std::vector<Node> topNodes;
    std::vector<rtgpu_material> materials;
    std::vector<rtgpu_point_light> pointLights;
    std::vector<rtgpu_area_light> areaLights;
    std::vector<rtgpu_directional_light> dirLights;
    struct Tex { std::vector<uint8_t> rgba; uint32_t width, height; double gamma; };
    std::vector<Tex> textures;
    int hasEnv = 0;   // 0 none, 1 ColorEnvironment, 2 ImageMapEnvironment, 3 SkyEnvironment
    RGBf envRadiance; // color * intensity, ColorEnvironment.cpp:6-9
    std::vector<float> envMap;   // Texture<float>: RGBA, rows top to bottom
    uint32_t envW = 0, envH = 0;
    double envIntensity = 0.0;
    // Scene::photonMap / photonMapMode / photonMapDepth: photons + the tree of photonmapping/KDTreeBuilder.h
    std::vector<rtgpu_photon> photons;
    struct KdNode { int32_t child[2]; uint32_t photon; int32_t axis; };
    std::vector<KdNode> kd;
    int32_t kdRoot = -1;
    int pmMode = RTGPU_PM_NONE, pmDepth = 0;
    // Test hooks (orc_scene_set_closest_rule / orc_scene_set_primary_rays): they let a test show that a pixel in which the
    // GPU and the reference's walk differ is one of the documented cases.
    //   closestRule 1: closest hits by the tree-independent definition the GPU kernels implement -- the smallest t over all
    //   primitives, equal t -> the smaller (instance, triangle) -- instead of the reference's walk, whose pick among
    //   exactly tied hits depends on its tree;
    //   primaryO / primaryD: camera rays taken from these arrays ([sample][y][x][3]) instead of camera_ray().
    int closestRule = 0;
    //   gatherRule 1: the k nearest accepted photons by squared distance (strict <, every photon looked at) instead of the
    //   reference's walk, which compares rounded square roots: photons whose distances differ by less than that rounding
    //   count as tied there, and which of them ends up in the list follows from the walk.
    int gatherRule = 0;
    const float* primaryO = nullptr;
    const float* primaryD = nullptr;
};

namespace {

struct InstHit {
    bool ok = false;
    double t = 0;
    float beta = 0, gamma = 0;
    uint32_t pos = 0;       // triangle sorted position or RTGPU_INVALID_ID
    uint32_t instPos = 0;
    V3 objNormal{};
};

// InstancedModelList::traceRay, shape/list/InstancedModelList.cpp:99-138 (leaf of the top tree)
InstHit instances_trace(const orc_scene& S, uint32_t begin, uint32_t end, V3 o, V3 d)
{
    InstHit closest;
    for (uint32_t ip = begin; ip < end; ++ip) {
        const Instance& in = S.instances[S.instOrder[ip]];
        V3 to = xf_point(in.xf.inv, o);   // Transformation::transformInverse(Ray), Transformation.h:108-113
        V3 td = xf_vector(in.xf.inv, d);
        InstHit h;
        if (in.shape >= 0) {
            const Mesh& m = S.meshes[in.shape];
            MeshHit mh = bvh_trace<MeshHit>(m.nodes, 0, to, td, [&](uint32_t b, uint32_t e) { return mesh_leaf_intersect(m, b, e, to, td); });
            if (mh.ok) {
                h.ok = true; h.t = mh.t; h.beta = mh.beta; h.gamma = mh.gamma; h.pos = mh.pos;
            }
        } else {
            double t;
            if (sphere_intersect(to, td, t)) {
                h.ok = true; h.t = t; h.pos = RTGPU_INVALID_ID;
                h.objNormal = to + ((float)t * td); // Sphere.cpp:52-53
            }
        }
        if (h.ok && (!closest.ok || closest.t > h.t)) {
            closest = h;
            closest.instPos = ip;
        }
    }
    return closest;
}

// InstancedModelList::testVisibility, shape/list/InstancedModelList.cpp:205-233
bool instances_visibility(const orc_scene& S, uint32_t begin, uint32_t end, V3 o, V3 d, float maxT)
{
    for (uint32_t ip = begin; ip < end; ++ip) {
        const Instance& in = S.instances[S.instOrder[ip]];
        V3 to = xf_point(in.xf.inv, o);
        V3 td = xf_vector(in.xf.inv, d);
        if (in.shape >= 0) {
            const Mesh& m = S.meshes[in.shape];
            if (bvh_visibility(m.nodes, 0, to, td, maxT, [&](uint32_t b, uint32_t e) { return mesh_leaf_visibility(m, b, e, to, td, maxT); }))
                return true;
        } else {
            double t; // IShape::testVisibility, shape/IShape.h:17-25
            if (sphere_intersect(to, td, t) && t <= maxT) return true;
        }
    }
    return false;
}

// RayHitInfo::tangent of a triangle, shape/TriangleMesh.cpp:130-146: row 0 of uvMat^-1 * deltaPosMat (2x2 inverse
// = adjugate / determinant), or b - a for a mesh without texture coordinates
V3 mesh_tangent(const Mesh& m, uint32_t tri)
{
    const V3 a = m.vertices[m.indices[3 * tri]], b = m.vertices[m.indices[3 * tri + 1]], c = m.vertices[m.indices[3 * tri + 2]];
    if (m.texCoords.empty()) return b - a;
    const float* ta = &m.texCoords[2 * m.indices[3 * tri]];
    const float* tb = &m.texCoords[2 * m.indices[3 * tri + 1]];
    const float* tc = &m.texCoords[2 * m.indices[3 * tri + 2]];
    const float m00 = tb[0] - ta[0], m01 = tb[1] - ta[1], m10 = tc[0] - ta[0], m11 = tc[1] - ta[1];
    const float det = (m00 * m11) - (m10 * m01);
    const float invdet = 1.0f / det;
    const float i00 = m11 * invdet, i01 = -m01 * invdet;
    const V3 d0 = b - a, d1 = c - a;
    return V3{(i00 * d0.x) + (i01 * d1.x), (i00 * d0.y) + (i01 * d1.y), (i00 * d0.z) + (i01 * d1.z)};
}

// What Scene::traceRay returns for the hit (instance instIdx, triangle `tri` of its mesh or a sphere) at distance t.
SceneHit finish_hit(const orc_scene& S, V3 o, V3 d, uint32_t instPos, uint32_t instIdx, uint32_t tri, double t, float beta, float gamma, V3 sphereNormal)
{
    const Instance& in = S.instances[instIdx];
    SceneHit r;
    r.ok = true;
    r.ro = o; r.rd = d; r.t = t;
    r.instPos = instPos;
    r.inst = instIdx;
    r.beta = beta; r.gamma = gamma;
    V3 objN;
    if (in.shape >= 0) {
        const Mesh& m = S.meshes[in.shape];
        r.prim = tri;
        // normal interpolation, shape/TriangleMesh.cpp:114-120
        float alfa = 1.0f - beta - gamma;
        V3 na = m.normals[m.indices[3 * tri]], nb = m.normals[m.indices[3 * tri + 1]], nc = m.normals[m.indices[3 * tri + 2]];
        objN = ((alfa * na) + (beta * nb)) + (gamma * nc);
        if (!m.texCoords.empty()) { // shape/TriangleMesh.cpp:122-128; (0, 0) otherwise (:143-146)
            const float* ta = &m.texCoords[2 * m.indices[3 * tri]];
            const float* tb = &m.texCoords[2 * m.indices[3 * tri + 1]];
            const float* tc = &m.texCoords[2 * m.indices[3 * tri + 2]];
            r.u = ((alfa * ta[0]) + (beta * tb[0])) + (gamma * tc[0]);
            r.v = ((alfa * ta[1]) + (beta * tb[1])) + (gamma * tc[1]);
        }
        r.tangent = xf_vector(in.xf.m, mesh_tangent(m, tri)); // InstancedModelList.cpp:131
    } else {
        r.prim = RTGPU_INVALID_ID;
        objN = sphereNormal;
    }
    r.normal = normalized(xf_normal(in.xf.inv, objN)); // InstancedModelList.cpp:129-130
    return r;
}

// closestRule 1 (test hook): every primitive tested, first met in (instance, triangle) order wins among equal distances
SceneHit scene_trace_all(const orc_scene& S, V3 o, V3 d)
{
    double bestT = INFINITY;
    bool ok = false;
    uint32_t bestInst = 0, bestTri = RTGPU_INVALID_ID;
    float bestBeta = 0, bestGamma = 0;
    V3 bestN{};
    for (uint32_t ii = 0; ii < S.instances.size(); ++ii) {
        const Instance& in = S.instances[ii];
        V3 to = xf_point(in.xf.inv, o), td = xf_vector(in.xf.inv, d);
        if (in.shape >= 0) {
            const Mesh& m = S.meshes[in.shape];
            for (uint32_t tri = 0; tri < m.ntri; ++tri) {
                float be, ga;
                double t;
                if (!tri_intersect(to, td, m.vertices[m.indices[3 * tri]], m.vertices[m.indices[3 * tri + 1]], m.vertices[m.indices[3 * tri + 2]], be, ga, t)) continue;
                if (!ok || t < bestT) { ok = true; bestT = t; bestInst = ii; bestTri = tri; bestBeta = be; bestGamma = ga; }
            }
        } else {
            double t;
            if (sphere_intersect(to, td, t) && (!ok || t < bestT)) {
                ok = true; bestT = t; bestInst = ii; bestTri = RTGPU_INVALID_ID; bestBeta = 0; bestGamma = 0;
                bestN = to + ((float)t * td);
            }
        }
    }
    if (!ok) return SceneHit{};
    uint32_t pos = 0;
    for (uint32_t k = 0; k < S.instOrder.size(); ++k) if (S.instOrder[k] == bestInst) pos = k;
    return finish_hit(S, o, d, pos, bestInst, bestTri, bestT, bestBeta, bestGamma, bestN);
}

// Scene::traceRay, scene/renderable/Scene.cpp:14-17
SceneHit scene_trace(const orc_scene& S, V3 o, V3 d)
{
    if (S.closestRule == 1) return scene_trace_all(S, o, d);
    InstHit ih = bvh_trace<InstHit>(S.topNodes, 0, o, d, [&](uint32_t b, uint32_t e) { return instances_trace(S, b, e, o, d); });
    if (!ih.ok) return SceneHit{};
    const uint32_t instIdx = S.instOrder[ih.instPos];
    const Instance& in = S.instances[instIdx];
    const uint32_t tri = in.shape >= 0 ? S.meshes[in.shape].order[ih.pos] : RTGPU_INVALID_ID;
    return finish_hit(S, o, d, ih.instPos, instIdx, tri, ih.t, ih.beta, ih.gamma, ih.objNormal);
}

// Scene::testVisibility, scene/renderable/Scene.cpp:19-22
bool scene_visibility(const orc_scene& S, V3 o, V3 d, float maxT)
{
    return bvh_visibility(S.topNodes, 0, o, d, maxT, [&](uint32_t b, uint32_t e) { return instances_visibility(S, b, e, o, d, maxT); });
}

// ------------------------------------------------------------------------------------------
// The render path's calls into libm (transcendental functions: glibc here, the device's math library on the GPU; both are
// accurate to the last bit or so, neither is correctly rounded, so now and then their results differ in that bit).
// Test hook: the result of the `target`-th call of a pixel's evaluation can be moved by `ulps` units in the last place
// (orc_render_pixel_probe, orc_attribute_pixel_to_libm) -- which shows whether a pixel sits on a branch that such a
// last-bit difference flips.  Square roots and divisions are correctly rounded on both sides and are not counted.
// ------------------------------------------------------------------------------------------
struct LibmProbe { long target = -1; int ulps = 0; long calls = 0; int fn = -1; };
thread_local LibmProbe g_libm;
enum { LM_ACOS = 0, LM_ASIN, LM_ATAN2, LM_TAN, LM_SIN, LM_COS, LM_POW };
template <typename T>
inline T libm_result(T v, int fn)
{
    if (g_libm.calls++ == g_libm.target) {
        g_libm.fn = fn;
        for (int k = 0; k < std::abs(g_libm.ulps); ++k) v = std::nextafter(v, g_libm.ulps > 0 ? (T)INFINITY : (T)-INFINITY);
    }
    return v;
}
inline double lm_acos(double x) { return libm_result(std::acos(x), LM_ACOS); }
inline float lm_acos(float x) { return libm_result(std::acos(x), LM_ACOS); }
inline double lm_asin(double x) { return libm_result(std::asin(x), LM_ASIN); }
inline float lm_asin(float x) { return libm_result(std::asin(x), LM_ASIN); }
inline double lm_atan2(double y, double x) { return libm_result(std::atan2(y, x), LM_ATAN2); }
inline float lm_atan2(float y, float x) { return libm_result(std::atan2(y, x), LM_ATAN2); }
inline double lm_tan(double x) { return libm_result(std::tan(x), LM_TAN); }
inline float lm_tan(float x) { return libm_result(std::tan(x), LM_TAN); }
inline double lm_sin(double x) { return libm_result(std::sin(x), LM_SIN); }
inline float lm_sin(float x) { return libm_result(std::sin(x), LM_SIN); }
inline double lm_cos(double x) { return libm_result(std::cos(x), LM_COS); }
inline float lm_cos(float x) { return libm_result(std::cos(x), LM_COS); }
inline double lm_pow(double x, double y) { return libm_result(std::pow(x, y), LM_POW); }
inline float lm_pow(float x, float y) { return libm_result(std::pow(x, y), LM_POW); }
inline double lm_pow(float x, double y) { return libm_result(std::pow((double)x, y), LM_POW); }

// ------------------------------------------------------------------------------------------
// Counter-based RNG shared with the HIP integrator (replaces math/FastRandom.h)
// ------------------------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline float rand_unit(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t vertex, uint32_t dim)
{
    uint64_t k1 = ((uint64_t)pixel << 32) | sample;
    uint64_t k2 = ((uint64_t)seed << 32) | (uint64_t)((vertex << 8) | dim);
    uint32_t x = (uint32_t)(splitmix64(splitmix64(k1) ^ k2) >> 32);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

enum Dim : uint32_t {
    DIM_NEE_CHOICE = 1, DIM_LIGHT_TYPE = 2, DIM_LIGHT_INDEX = 3, DIM_AREA_U = 4, DIM_AREA_V = 5,
    DIM_SQUARE_X = 6, DIM_SQUARE_Y = 7, DIM_RR = 8, DIM_DIR_ALPHA = 9, DIM_DIR_BETA = 10, DIM_GLASS = 11, DIM_MIX0 = 12
};
constexpr uint32_t VERTEX_CAMERA = 255;

struct RandCtx {
    uint32_t seed, pixel, sample, vertex;
    float unit(uint32_t dim) const { return rand_unit(seed, pixel, sample, vertex, dim); }
};

// sampleUniformStratifiedSquare, math/Sampler.h:13-26
inline void stratified_square(int level, float sampleI, float r0, float r1, float& sx, float& sy)
{
    if (level < 4) { sx = r0; sy = r1; return; }
    int binsPerAxis = (int)std::sqrt((double)level);
    float binX = (float)std::fmod((double)sampleI, (double)binsPerAxis);
    float binY = (float)std::floor(std::fmod((double)sampleI, (double)level) / binsPerAxis);
    sx = (binX / binsPerAxis) + (r0 / binsPerAxis);
    sy = (binY / binsPerAxis) + (r1 / binsPerAxis);
}

// OrthonormalBasis(a), math/OrthonormalBasis.cpp:10-31
struct Basis { V3 u, v, w; };
inline Basis basis_from(V3 a)
{
    Basis b;
    float length = norm(a);
    b.w = a / length;
    if (std::fabs(b.w.x) > std::fabs(b.w.y)) {
        float inv_length = (float)(1.0f / std::sqrt((b.w.x * b.w.x) + (b.w.z * b.w.z)));
        b.u = V3{-b.w.z * inv_length, 0.0f, b.w.x * inv_length};
    } else {
        float inv_length = (float)(1.0f / std::sqrt((b.w.y * b.w.y) + (b.w.z * b.w.z)));
        b.u = V3{0.0f, b.w.z * inv_length, -b.w.y * inv_length};
    }
    b.v = cross(b.w, b.u);
    return b;
}
// OrthonormalBasis(a, b), math/OrthonormalBasis.cpp:33-51
inline Basis basis_from2(V3 a, V3 b2)
{
    Basis b;
    V3 cr = cross(b2, a);
    float length = norm(cr);
    b.w = normalized(a);
    b.u = cr / length;
    b.v = cross(b.w, b.u);
    return b;
}

// mapSampleToCosineWeightedHemisphere, math/Sampler.h:51-59 (radius 1, exponent 1)
inline V3 cosine_hemisphere(float r1, float r2)
{
    float cos_theta = lm_pow(r1, 1.0f / (1.0f + 1.0f));
    float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
    double phi = 2.0f * PI * r2;
    double cos_phi = lm_cos(phi);
    double sin_phi = lm_sin(phi);
    return V3{(float)(1.0f * sin_theta * cos_phi), (float)(1.0f * sin_theta * sin_phi), 1.0f * cos_theta};
}

// ---- VNDF GGX sampling, material/VNDFGGXSampler.cpp ---------------------------------------
inline void sampleP22(double theta_i, double U1, double U2, float& sx, float& sy) // :7-43
{
    if (theta_i < 0.0001) {
        const double r = std::sqrt(U1 / (1 - U1));
        const double phi = 6.28318530718 * U2;
        sx = (float)(r * lm_cos(phi));
        sy = (float)(r * lm_sin(phi));
        return;
    }
    const double tan_theta_i = lm_tan(theta_i);
    const double a = 1 / (tan_theta_i);
    const double G1 = 2 / (1 + std::sqrt(1.0 + 1.0 / (a * a)));
    const double A = 2.0 * U1 / G1 - 1.0;
    const double tmp = 1.0 / (A * A - 1.0);
    const double B = tan_theta_i;
    const double D = std::sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
    const double slope_x_1 = B * tmp - D;
    const double slope_x_2 = B * tmp + D;
    double slope_x = (A < 0 || slope_x_2 > 1.0 / tan_theta_i) ? slope_x_1 : slope_x_2;
    double S;
    if (U2 > 0.5) { S = 1.0; U2 = 2.0 * (U2 - 0.5); }
    else          { S = -1.0; U2 = 2.0 * (0.5 - U2); }
    const double z = (U2 * (U2 * (U2 * 0.27385 - 0.73369) + 0.46341)) / (U2 * (U2 * (U2 * 0.093073 + 0.309420) - 1.000000) + 0.597999);
    double slope_y = S * z * std::sqrt(1.0 + slope_x * slope_x);
    sx = (float)slope_x;
    sy = (float)slope_y;
}
inline V3 vndf_sample(V3 smoothNormal, V3 worldIncoming, float roughness, double rand1, double rand2) // :53-90
{
    Basis basis = basis_from(smoothNormal);
    // OrthonormalBasis::applyBasisTo, math/OrthonormalBasis.cpp:53-60
    V3 incoming = V3{dot(basis.u, worldIncoming), dot(basis.v, worldIncoming), dot(basis.w, worldIncoming)};
    incoming = normalized(incoming);
    V3 st = normalized(V3{incoming.x * roughness, incoming.y * roughness, incoming.z});
    float theta = 0.0, phi = 0.0;
    if (st.z < 0.99999) {
        theta = lm_acos(st.z);
        phi = lm_atan2(st.y, st.x);
    }
    float slx, sly;
    sampleP22(theta, rand1, rand2, slx, sly);
    float rx = lm_cos(phi) * slx - lm_sin(phi) * sly;
    float ry = lm_sin(phi) * slx + lm_cos(phi) * sly;
    rx *= roughness;
    ry *= roughness;
    V3 micro = normalized(V3{-rx, -ry, 1.0f});
    // OrthonormalBasis::invertApplyBasisTo, math/OrthonormalBasis.cpp:62-69 (columns u v w)
    return V3{basis.u.x * micro.x + (basis.v.x * micro.y + basis.w.x * micro.z),
              basis.u.y * micro.x + (basis.v.y * micro.y + basis.w.y * micro.z),
              basis.u.z * micro.x + (basis.v.z * micro.y + basis.w.z * micro.z)};
}

// ---- GGX terms, material/GlossyMaterial.cpp:27-66 -----------------------------------------
inline float ggx_g1(float roughness, V3 v, V3 n, V3 m)
{
    float theta_v = lm_acos(dot(v, n));
    if (dot(v, m) / dot(v, n) > 0.0f) {
        float tan_theta = lm_tan(theta_v);
        return 2.0f / (1.0f + std::sqrt(1.0f + (roughness * roughness) * (tan_theta * tan_theta)));
    }
    return 0.0f;
}
inline float ggx_brdf(float roughness, V3 i, V3 o, V3 n, V3 m)
{
    float g = ggx_g1(roughness, i, n, m) * ggx_g1(roughness, o, n, m);
    if (g == 0.0f) return 0.0f;
    float d = 0.0f;
    if (dot(m, n) > 0) {
        float theta_m = lm_acos(dot(m, n));
        float aSqr = roughness * roughness;
        float tan_theta_m = lm_tan(theta_m);
        float cos_theta_m = lm_cos(theta_m);
        float cos_theta_m_sqr = cos_theta_m * cos_theta_m;
        float x = aSqr + tan_theta_m * tan_theta_m;
        d = aSqr / ((float)M_PI * cos_theta_m_sqr * cos_theta_m_sqr * x * x);
    } else {
        return 0.0f;
    }
    return g * d / (4.0f * std::fabs(dot(i, n)) * std::fabs(dot(o, n)));
}

// ---- lights ------------------------------------------------------------------------------
inline RGBf light_energy(const float color[3], double intensity) // RGB * double
{
    return rgb_scale_d(RGBf{color[0], color[1], color[2]}, intensity);
}

struct RenderCtx {
    const orc_scene& S;
    rtgpu_render_stats* stats;
    uint64_t primary = 0, extension = 0, shadow = 0;
    const std::vector<struct TransportNode>* path = nullptr;   // TransportContext::path / curI of the vertex being sampled
    int curI = 0;
};

// neePointLight, material/NextEventEstimation.cpp:8-26
RGBf nee_point(RenderCtx& R, const rtgpu_point_light& light, V3 hitpoint, V3& lightDirection)
{
    V3 objectToLamp = mk(light.pos) - hitpoint;
    float lampT = norm(objectToLamp);
    objectToLamp = normalized(objectToLamp);
    lightDirection = objectToLamp;
    R.shadow++;
    bool isVisible = !scene_visibility(R.S, hitpoint + (objectToLamp * 0.0001f), objectToLamp, lampT);
    if (isVisible) {
        double geometricFactor = 1.0f / (4.0f * PI * (lampT * lampT));
        return rgb_scale_d(RGBf{light.color[0], light.color[1], light.color[2]}, light.intensity * geometricFactor);
    }
    return RGBf{};
}
// AreaLight::generateStratifiedJitteredRandomPoint, light/AreaLight.cpp:26-41
V3 area_light_point(const rtgpu_area_light& L, int level, int i, float r0, float r1)
{
    int n = (int)std::sqrt((double)level);
    V3 a = mk(L.a), b = mk(L.b), c = mk(L.c);
    V3 v1 = (c - a) / (float)n;
    V3 v2 = (b - a) / (float)n;
    float u = 1.0f - std::sqrt(r0);
    float v = (1.0f - u) * r1;
    int y = i / n;
    int x = i % n;
    return (a + ((u + x) * v1)) + ((v + y) * v2);
}
// neeAreaLight, material/NextEventEstimation.cpp:29-53
RGBf nee_area(RenderCtx& R, const rtgpu_area_light& light, V3 hitpoint, int sampleI, int sampleCount, const RandCtx& rc, V3& lightDirection)
{
    V3 lampPoint = area_light_point(light, sampleCount, sampleI, rc.unit(DIM_AREA_U), rc.unit(DIM_AREA_V));
    V3 objectToLamp = lampPoint - hitpoint;
    float lampT = norm(objectToLamp);
    objectToLamp = normalized(objectToLamp);
    lightDirection = objectToLamp;
    R.shadow++;
    bool isVisible = !scene_visibility(R.S, hitpoint + (objectToLamp * 0.0001f), objectToLamp, lampT);
    if (isVisible) {
        V3 a = mk(light.a), b = mk(light.b), c = mk(light.c);
        double area = tri_area(a, b, c);
        RGBf lightEnergy = light_energy(light.color, light.intensity);
        RGBf lightIrradiance = rgb_div_d(lightEnergy, area);
        RGBf lightRadiance = rgb_div_d(lightIrradiance, PI);
        V3 nrm = normalized(cross(b - a, c - a)); // AreaLight::getNormal, light/AreaLight.cpp:43-48
        float lampAngle = std::max(0.0f, dot(nrm, -objectToLamp));
        float geometricFactor = lampAngle / (lampT * lampT);
        RGBf irradianceFromLamp = rgb_scale_d(lightRadiance, geometricFactor);
        return rgb_scale_d(irradianceFromLamp, area);
    }
    return RGBf{};
}
// sampleUniformSteradianSphere, math/Sampler.h:76-88, with Transformation::rotateX/rotateY/append/
// transform(Vector3) (math/Transformation.cpp:35-63, Transformation.h:66-69,83-87): the product matrix's
// third column is (sin b, -sin a cos b, cos a cos b) in fp64, narrowed to float by transform().
V3 sample_steradian_sphere(V3 center, float angle, float r0, float r1)
{
    float alpha = (r0 - 0.5f) * 2.0f * angle;
    float beta = (r1 - 0.5f) * 2.0f * angle;
    double sa = lm_sin((double)alpha), ca = lm_cos((double)alpha), sb = lm_sin((double)beta), cb = lm_cos((double)beta);
    V3 l{(float)sb, (float)(-sa * cb), (float)(ca * cb)};
    Basis basis = basis_from(center);
    return V3{basis.u.x * l.x + (basis.v.x * l.y + basis.w.x * l.z),
              basis.u.y * l.x + (basis.v.y * l.y + basis.w.y * l.z),
              basis.u.z * l.x + (basis.v.z * l.y + basis.w.z * l.z)};
}
// neeDirectionalLight, material/NextEventEstimation.cpp:55-67
RGBf nee_directional(RenderCtx& R, const rtgpu_directional_light& light, V3 hitpoint, const RandCtx& rc, V3& lightDirection)
{
    lightDirection = sample_steradian_sphere(-mk(light.direction), light.angle, rc.unit(DIM_DIR_ALPHA), rc.unit(DIM_DIR_BETA));
    R.shadow++;
    bool isVisible = !scene_visibility(R.S, hitpoint + (lightDirection * 0.0001f), lightDirection, INFINITY);
    if (isVisible) return light_energy(light.color, light.intensity);
    return RGBf{};
}
// NextEventEstimation::sample, material/NextEventEstimation.cpp:69-120
RGBf nee_sample(RenderCtx& R, V3 hitpoint, int sampleI, int sampleCount, const RandCtx& rc, V3& lightDirection)
{
    const orc_scene& S = R.S;
    bool hasPoint = !S.pointLights.empty();
    bool hasArea = !S.areaLights.empty();
    bool hasDir = !S.dirLights.empty();
    float total = (hasArea ? 1.0f : 0.0f) + (hasPoint ? 1.0f : 0.0f) + (hasDir ? 1.0f : 0.0f);
    float pointP = (hasPoint ? 1.0f : 0.0f) / total;
    float areaP = (hasArea ? 1.0f : 0.0f) / total;
    float dirP = (hasDir ? 1.0f : 0.0f) / total;
    float choice = rc.unit(DIM_LIGHT_TYPE);
    if (choice < pointP) {
        size_t n = S.pointLights.size();
        float chosenP = 1.0f / n;
        size_t idx = std::min(n - 1, (size_t)(rc.unit(DIM_LIGHT_INDEX) * (float)n)); // Rand::intInRange(n-1)
        return rgb_div_f(nee_point(R, S.pointLights[idx], hitpoint, lightDirection), pointP * chosenP);
    } else if (choice < pointP + areaP) {
        size_t n = S.areaLights.size();
        float chosenP = 1.0f / n;
        size_t idx = std::min(n - 1, (size_t)(rc.unit(DIM_LIGHT_INDEX) * (float)n));
        return rgb_div_f(nee_area(R, S.areaLights[idx], hitpoint, sampleI, sampleCount, rc, lightDirection), areaP * chosenP);
    } else if (hasDir) {
        size_t n = S.dirLights.size();
        float chosenP = 1.0f / n;
        size_t idx = std::min(n - 1, (size_t)(rc.unit(DIM_LIGHT_INDEX) * (float)n));
        return rgb_div_f(nee_directional(R, S.dirLights[idx], hitpoint, rc, lightDirection), dirP * chosenP);
    }
    return RGBf{};
}

// ---- photon map: tree and gather -----------------------------------------------------------
// KDTreeBuilder::buildNode / buildTree, photonmapping/KDTreeBuilder.h:178-231: the slice arrives sorted along
// `presorted`; three or more photons are sorted along the next axis and split at count / 2, two photons make a
// node from the first with the second as its right child, one photon is a leaf
int32_t kd_build(orc_scene& S, uint32_t* idx, size_t count, int presorted)
{
    if (count == 0) return -1;
    const int32_t me = (int32_t)S.kd.size();
    S.kd.push_back(orc_scene::KdNode{{-1, -1}, 0, 0});
    if (count == 1) {
        S.kd[me].photon = idx[0];
        return me;
    }
    if (count == 2) {
        S.kd[me].photon = idx[0];
        S.kd[me].axis = presorted;
        const int32_t right = kd_build(S, idx + 1, 1, presorted);
        S.kd[me].child[1] = right;
        return me;
    }
    const int cur = (presorted + 1) % 3;
    std::sort(idx, idx + count, [&](uint32_t a, uint32_t b) { return S.photons[a].pos[cur] < S.photons[b].pos[cur]; });
    const size_t mid = count / 2;
    S.kd[me].photon = idx[mid];
    S.kd[me].axis = cur;
    const int32_t left = kd_build(S, idx, mid, cur);
    const int32_t right = kd_build(S, idx + mid + 1, count - mid - 1, cur);
    S.kd[me].child[0] = left;
    S.kd[me].child[1] = right;
    return me;
}
// KDTreeBuilder::build, photonmapping/KDTreeBuilder.h:258-276
void kd_build_all(orc_scene& S)
{
    S.kd.clear();
    S.kdRoot = -1;
    if (S.photons.empty()) return;
    std::vector<uint32_t> idx(S.photons.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return S.photons[a].pos[0] < S.photons[b].pos[0]; });
    S.kd.reserve(idx.size());
    S.kdRoot = kd_build(S, idx.data(), idx.size(), 0);
}

struct Gather { // the state of one KDTree::getElementsNearestTo call
    const orc_scene& S;
    V3 target, dir;
    unsigned count;
    std::vector<int64_t> results; // photon indices, nearest first; -1 = empty (nullptr in the reference)
    V3 pos(uint32_t ph) const { return mk(S.photons[ph].pos); }
    bool filter(uint32_t ph) const { return dot(dir, mk(S.photons[ph].normal)) >= 0; } // DiffuseMaterial.cpp:66-68
    // KDTreeNode::insert, photonmapping/KDTree.h:225-245
    float insert(uint32_t ph)
    {
        auto key = [&](int64_t r) { return r < 0 ? 1E99 : (double)dot(pos((uint32_t)r) - target, pos((uint32_t)r) - target); };
        auto it = std::lower_bound(results.begin(), results.end(), (int64_t)ph, [&](int64_t a, int64_t b) { return key(a) < key(b); });
        if (it != results.end() && *it != (int64_t)ph) {
            results.insert(it, (int64_t)ph);
            results.pop_back();
        }
        const int64_t last = results.back();
        if (last < 0) return std::numeric_limits<float>::max();
        return norm(pos((uint32_t)last) - target);
    }
    // KDTreeNode::getElementsNearestTo, photonmapping/KDTree.h:90-164
    void visit(int32_t ni, bool targetInside, float& searchRadius, unsigned& found, V3 lo, V3 hi)
    {
        const orc_scene::KdNode& n = S.kd[ni];
        const V3 position = pos(n.photon);
        if (n.child[0] < 0 && n.child[1] < 0) {
            if (norm(position - target) < searchRadius && filter(n.photon)) {
                searchRadius = insert(n.photon);
                found = std::min(found + 1, count);
            }
            return;
        }
        const int axis = n.axis;
        auto comp = [&](V3 v) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); };
        const int inChild = comp(position) < comp(target) ? 1 : 0;
        // AABB::split at the node's plane: child 0 keeps the low side
        V3 lo0 = lo, hi0 = hi, lo1 = lo, hi1 = hi;
        if (axis == 0) { hi0.x = position.x; lo1.x = position.x; }
        else if (axis == 1) { hi0.y = position.y; lo1.y = position.y; }
        else { hi0.z = position.z; lo1.z = position.z; }
        if (n.child[inChild] >= 0) visit(n.child[inChild], targetInside, searchRadius, found, inChild ? lo1 : lo0, inChild ? hi1 : hi0);
        if (norm(position - target) < searchRadius && filter(n.photon)) {
            searchRadius = insert(n.photon);
            found = std::min(found + 1, count);
        }
        const int other = (inChild + 1) % 2;
        if (n.child[other] >= 0) {
            float distanceToSplitPlane;
            if (targetInside) distanceToSplitPlane = std::abs(comp(position) - comp(target));
            else {
                V3 projection{std::min(std::max(target.x, lo.x), hi.x), std::min(std::max(target.y, lo.y), hi.y), std::min(std::max(target.z, lo.z), hi.z)};
                distanceToSplitPlane = norm(projection - target);
            }
            if (distanceToSplitPlane <= searchRadius) visit(n.child[other], false, searchRadius, found, other ? lo1 : lo0, other ? hi1 : hi0);
        }
    }
};

// photonMap->getElementsNearestTo(hitpoint, 20, 1E9, filter, photons) and the energy sum of DiffuseMaterial.cpp:62-75
rtgpu_photon_gather gather_photons(const orc_scene& S, V3 target, V3 dir, unsigned k = 20)
{
    rtgpu_photon_gather out{};
    if (S.gatherRule == 1) {   // test hook, see orc_scene
        std::vector<std::pair<float, uint32_t>> best;   // (squared distance, photon), kept sorted, at most k
        for (uint32_t i = 0; i < S.photons.size(); ++i) {
            if (!(dot(dir, mk(S.photons[i].normal)) >= 0)) continue;
            const V3 diff = mk(S.photons[i].pos) - target;
            const float d2 = dot(diff, diff);
            if (best.size() == k && !(d2 < best.back().first)) continue;
            auto it = std::upper_bound(best.begin(), best.end(), std::make_pair(d2, 0xFFFFFFFFu));
            best.insert(it, std::make_pair(d2, i));
            if (best.size() > k) best.pop_back();
        }
        out.count = (uint32_t)best.size();
        out.max_dist = best.size() == k ? std::sqrt(best.back().first) : (best.empty() ? 1E9f : std::numeric_limits<float>::max());
        RGBf value;
        for (const auto& b : best) value = rgb_add(value, RGBf{S.photons[b.second].energy[0], S.photons[b.second].energy[1], S.photons[b.second].energy[2]});
        out.energy[0] = value.r; out.energy[1] = value.g; out.energy[2] = value.b;
        return out;
    }
    float searchRadius = 1E9;
    unsigned found = 0;
    Gather g{S, target, dir, k, std::vector<int64_t>(k, -1)};
    if (S.kdRoot >= 0) {
        const float big = std::numeric_limits<float>::max();
        g.visit(S.kdRoot, true, searchRadius, found, V3{-big, -big, -big}, V3{big, big, big});
    }
    out.count = found;
    out.max_dist = searchRadius;
    RGBf value;
    for (unsigned i = 0; i < found; ++i) {
        const rtgpu_photon& ph = S.photons[(size_t)g.results[i]];
        value = rgb_add(value, RGBf{ph.energy[0], ph.energy[1], ph.energy[2]});
    }
    out.energy[0] = value.r; out.energy[1] = value.g; out.energy[2] = value.b;
    return out;
}

// DiffuseMaterial's test in PhotonMapMode::full (material/DiffuseMaterial.cpp:40-54, 230-244): has the path passed
// `photonMapDepth` vertices with specularity < 0.8 before vertex curI?
struct TransportNode;
bool diffuse_reads_map(const orc_scene& S, const std::vector<TransportNode>& path, int curI);

// ---- transport ---------------------------------------------------------------------------
struct TransportNode { // material/IMaterial.h:17-30 + the per-material TransportMetaData
    SceneHit hit;
    float pathTerminationChance = 0;
    V3 transportDirection{};
    bool isEmissive = false;
    // metadata
    uint32_t leafMat = 0;      // material chosen through CompositeMaterial / MixMaterial
    bool isNEERay = false;
    RGBf directLighting;       // DiffuseMaterial
    RGBf neeRadiance;          // GlossyMaterial
    int lightHit = -1;         // GlossyMaterial, index into areaLights
    float specularity = 0;     // TransportNode::specularity (IMaterial.h:23): what DiffuseMaterial counts in photon-map mode
    bool isPhotonMapRay = false; // DiffuseMaterial: the vertex reads the photon map
    bool typeBounce = true;    // TransportNode::type == TransportType::bounce (IMaterial.h:12-15, 24)
    RGBf photonLighting;
    // GlassMaterial's TransportMetaData (material/GlassMaterial.cpp:11-18).  The reference keeps metadata in
    // a tree with one slot per MixMaterial branch (utility/DynAllocTree.h:118-131: depth 3, two children),
    // and a slot survives every re-sampling pass of the node (renderer/Renderer.cpp:246 keeps the node).
    struct GlassMeta {
        bool used = false;
        float reflectionWeight = 0, t = 0;
        V3 reflectedRayDir{}, transmittedRayDir{};
        int lightHit = -1;
    };
    GlassMeta glass[7];
    int leafSlot = 0;          // metadata slot of the leaf material chosen in the current pass
    // AddMaterial (material/AddMaterial.cpp:5-17): the emissive first children met on the way to the leaf
    uint32_t addFirst[4] = {0, 0, 0, 0};
    int nAdd = 0;
};

bool diffuse_reads_map(const orc_scene& S, const std::vector<TransportNode>& path, int curI)
{
    if (S.pmMode != RTGPU_PM_FULL) return false;
    int curDiffuseI = 0;
    for (int i = 0; i < curI; ++i)
        if (path[i].specularity < 0.8) curDiffuseI++;
    return curDiffuseI >= S.pmDepth;
}

uint32_t hit_material(const orc_scene& S, const SceneHit& h) // Model material / CompositeMaterial::findMaterial
{
    const Instance& in = S.instances[h.inst];
    if (in.shape >= 0 && h.prim != RTGPU_INVALID_ID) {
        const Mesh& m = S.meshes[in.shape];
        if (!m.triMaterial.empty()) return m.triMaterial[h.prim];
    }
    return in.material;
}

// FresnelMixMaterial::calcMixFactor, material/FresnelMixMaterial.cpp:3-41
float fresnel_mix_factor(V3 normal, V3 incomingDir, float IOR)
{
    V3 objToIncoming = -incomingDir;
    double nDotWo = dot(normal, objToIncoming);
    double incomingAngle = nDotWo;
    bool isInternalRay = nDotWo < 0;
    double curRelIor = IOR / 1.0;
    if (isInternalRay) curRelIor = 1.0 / curRelIor;
    double cosTransmissionSqr = 1.0 - ((1.0 - (incomingAngle * incomingAngle)) / (curRelIor * curRelIor));
    bool tir = cosTransmissionSqr < 0;
    if (tir) return 1.0;
    if (isInternalRay) nDotWo = -nDotWo;
    double cosRefractAngle = std::sqrt(cosTransmissionSqr);
    double rs = ((curRelIor * nDotWo) - cosRefractAngle) / ((curRelIor * nDotWo) + cosRefractAngle);
    double rp = (nDotWo - (curRelIor * cosRefractAngle)) / (nDotWo + (curRelIor * cosRefractAngle));
    double kr = ((rs * rs) + (rp * rp)) / 2.0;
    double reflectionWeight = kr;
    double transmissionWeight = 1.0 - kr;
    double totalWeight = reflectionWeight + transmissionWeight;
    return (float)(reflectionWeight / totalWeight);
}

// MixMaterial::calcMixFactor at a hit: ConstMixMaterial (material/MixMaterial.h:19-26) or FresnelMixMaterial
float mat_mix_factor(const rtgpu_material& m, const SceneHit& h)
{
    return m.type == RTGPU_MAT_CONSTMIX ? (float)m.mix_factor : fresnel_mix_factor(h.normal, h.rd, m.roughness);
}

// IMaterial::hasVariance: Diffuse (DiffuseMaterial.cpp:228-247, no photon map) true; Glossy
// (GlossyMaterial.cpp:226-229) roughness > 0; Emissive false; Transparent
// (TransparentMaterial.cpp:23-26) false; Mix (MixMaterial.cpp:59-65)
bool mat_has_variance(const orc_scene& S, uint32_t mi, const SceneHit& h, bool diffuseReadsMap = false)
{
    const rtgpu_material& m = S.materials[mi];
    switch (m.type) {
    case RTGPU_MAT_DIFFUSE: return !diffuseReadsMap;   // DiffuseMaterial.cpp:228-247
    case RTGPU_MAT_GLOSSY: return m.roughness > 0.0f;
    case RTGPU_MAT_EMISSIVE: return false;
    case RTGPU_MAT_TRANSPARENT: return false;
    case RTGPU_MAT_GLASS: return true;  // GlassMaterial.cpp:254-257
    case RTGPU_MAT_ADD: return mat_has_variance(S, m.first, h, diffuseReadsMap) || mat_has_variance(S, m.second, h, diffuseReadsMap); // AddMaterial.cpp:25-28
    case RTGPU_MAT_CONSTMIX:
    case RTGPU_MAT_FRESNELMIX: {
        float mf = mat_mix_factor(m, h);
        bool noVar = (mf == 0.0f && !mat_has_variance(S, m.first, h, diffuseReadsMap)) || (mf == 1.0f && !mat_has_variance(S, m.second, h, diffuseReadsMap));
        return !noVar;
    }
    }
    return false;
}

struct SampleOut {
    bool hasNextHit = false;
    SceneHit nextHit;
    bool tracedMiss = false;   // the material traced the ray samplePath would trace (same origin) and it left the scene
    bool causticsCallback = false;   // DiffuseMaterial left a nextNodeCallback (material/DiffuseMaterial.cpp:104-135)
};

// Triangle::intersect against every area light (renderer/Renderer.cpp:184-204, GlossyMaterial.cpp:126-178)
inline bool area_light_hit(const rtgpu_area_light& L, V3 o, V3 d, double& t)
{
    float be, ga;
    return tri_intersect(o, d, mk(L.a), mk(L.b), mk(L.c), be, ga, t);
}

// Texture::get, material/Texture.h:24-33 (an index equal to the size is clamped; the reference would read past the image)
RGBf texture_get(const orc_scene::Tex& T, float x, float y)
{
    unsigned px = (unsigned)(x * T.width), py = (unsigned)(y * T.height);
    px = std::min(px, T.width - 1);
    py = std::min(py, T.height - 1);
    size_t offset = ((size_t)py * T.width + px) * 4;
    RGBf color{(float)T.rgba[offset] / 255.0f, (float)T.rgba[offset + 1] / 255.0f, (float)T.rgba[offset + 2] / 255.0f};
    return RGBf{(float)lm_pow(color.r, T.gamma), (float)lm_pow(color.g, T.gamma), (float)lm_pow(color.b, T.gamma)}; // RGB::pow(double)
}

// sample_normal_map, material/NormalMapSampler.h:7-21 with Transformation::tangentToObject (math/Transformation.cpp:132-145)
V3 sample_normal_map(const orc_scene& S, const SceneHit& hit, uint32_t texture)
{
    float x = fmodf(hit.u, 1.0f), y = fmodf(hit.v, 1.0f);
    if (x < 0) x = 1.0f + x;
    if (y < 0) y = 1.0f + y;
    RGBf c = texture_get(S.textures[texture], x, y);
    V3 nts{2.0f * (c.r - 0.5f), 2.0f * (c.g - 0.5f), 2.0f * (c.b - 0.5f)};
    V3 tangent = normalized(hit.tangent), normal = normalized(hit.normal);
    V3 bit = cross(normal, tangent);
    double M[16] = {tangent.x, bit.x, normal.x, 0, tangent.y, bit.y, normal.y, 0, tangent.z, bit.z, normal.z, 0, 0, 0, 0, 1};
    return normalized(xf_vector(M, nts));
}
// the normal a diffuse / glossy closure shades with (DiffuseMaterial.cpp:33-37, 169-173; GlossyMaterial.cpp:81-85)
V3 shading_normal(const orc_scene& S, const rtgpu_material& m, const SceneHit& hit)
{
    return m.normal_map != 0 ? sample_normal_map(S, hit, m.normal_map - 1) : hit.normal;
}

// IMaterial::sampleTransport for the supported closures
void sample_transport(RenderCtx& R, TransportNode& tn, uint32_t mi, int sampleI, int sampleCount, const RandCtx& rc, int mixLevel, SampleOut& out,
                      int metaIndex = 0, int levelsDeep = 0)
{
    const orc_scene& S = R.S;
    const rtgpu_material& m = S.materials[mi];
    if (mixLevel == 0) tn.nAdd = 0;
    switch (m.type) {
    case RTGPU_MAT_CONSTMIX:
    case RTGPU_MAT_FRESNELMIX: { // MixMaterial::sampleTransport, material/MixMaterial.cpp:9-26
        bool choseFirst = rc.unit(DIM_MIX0 + mixLevel) > mat_mix_factor(m, tn.hit);
        // DynAllocTree::branch(choseFirst), utility/DynAllocTree.h:118-131
        int deeper = levelsDeep + 1;
        int childSize = (1 << (3 - deeper)) - 1;
        int childIndex = metaIndex + 1 + ((choseFirst ? 1 : 0) * childSize);
        sample_transport(R, tn, choseFirst ? m.first : m.second, sampleI, sampleCount, rc, mixLevel + 1, out, childIndex, deeper);
        return;
    }
    case RTGPU_MAT_ADD: { // AddMaterial::sampleTransport, material/AddMaterial.cpp:5-10: both children sample the same node, the second last
        if (tn.nAdd >= 4) { std::fprintf(stderr, "oracle: AddMaterial nested more than 4 deep\n"); std::abort(); }
        tn.addFirst[tn.nAdd++] = m.first;
        sample_transport(R, tn, m.second, sampleI, sampleCount, rc, mixLevel + 1, out, metaIndex, levelsDeep);
        return;
    }
    case RTGPU_MAT_GLASS: { // GlassMaterial::sampleTransport, material/GlassMaterial.cpp:21-133 (no photon map)
        if (metaIndex >= 7) { std::fprintf(stderr, "oracle: GlassMaterial below more than two MixMaterial levels\n"); std::abort(); }
        tn.leafMat = mi;
        tn.leafSlot = metaIndex;
        tn.isNEERay = false;
        tn.specularity = 1.0f;
        TransportNode::GlassMeta& meta = tn.glass[metaIndex];
        if (!meta.used) {
            meta = TransportNode::GlassMeta{};
            meta.used = true;
            // caustics mode: behind a diffuse or rough bounce the photon map supplies what would pass the glass (:31-38)
            if (S.pmMode == RTGPU_PM_CAUSTICS && R.curI != 0 && (*R.path)[R.curI - 1].specularity < 0.8 && (*R.path)[R.curI - 1].typeBounce) {
                tn.isEmissive = true;
                tn.pathTerminationChance = 1.0f;
                meta.reflectionWeight = NAN;
                return;
            }
            tn.isEmissive = false;
            tn.pathTerminationChance = 0.1f;
            V3& normal = tn.hit.normal;   // the reference flips the node's normal in place (:45, :77)
            V3 incomingDir = tn.hit.rd;
            V3 objToIncoming = -incomingDir;
            meta.reflectedRayDir = (-objToIncoming) + ((float)(2.0 * dot(normal, objToIncoming)) * normal);
            double nDotWo = dot(normal, objToIncoming);
            double incomingAngle = nDotWo;
            bool isInternalRay = nDotWo < 0;
            double curRelIor = m.intensity / 1.0;
            if (isInternalRay) curRelIor = 1.0 / curRelIor;
            double cosTransmissionSqr = 1.0 - ((1.0 - (incomingAngle * incomingAngle)) / (curRelIor * curRelIor));
            bool tir = cosTransmissionSqr < 0;
            if (tir) {
                meta.reflectionWeight = 1.0;
                meta.t = NAN;
                tn.typeBounce = true;
                tn.transportDirection = meta.reflectedRayDir;
                return;
            }
            if (isInternalRay) {
                nDotWo = -nDotWo;
                incomingAngle = -incomingAngle;
                normal = -normal;
            }
            double cosRefractAngle = std::sqrt(cosTransmissionSqr);
            meta.transmittedRayDir = (incomingDir / (float)curRelIor) - ((float)(cosRefractAngle - (incomingAngle / curRelIor)) * normal);
            double rs = ((curRelIor * nDotWo) - cosRefractAngle) / ((curRelIor * nDotWo) + cosRefractAngle);
            double rp = (nDotWo - (curRelIor * cosRefractAngle)) / (nDotWo + (curRelIor * cosRefractAngle));
            double kr = ((rs * rs) + (rp * rp)) / 2.0;
            double reflectionWeight = kr;
            double transmissionWeight = 1.0 - kr;
            double totalWeight = reflectionWeight + transmissionWeight;
            meta.reflectionWeight = (float)(reflectionWeight / totalWeight);
        }
        tn.typeBounce = rc.unit(DIM_GLASS) < meta.reflectionWeight;
        tn.transportDirection = tn.typeBounce ? meta.reflectedRayDir : meta.transmittedRayDir;
        V3 hitpoint = tn.hit.hitpoint();
        V3 ro = hitpoint + (tn.transportDirection * 0.0001f);
        R.extension++;
        SceneHit result = scene_trace(S, ro, tn.transportDirection);
        double bestT = result.ok ? result.t : 1E99;
        for (size_t li = 0; li < S.areaLights.size(); ++li) {
            double lt;
            if (area_light_hit(S.areaLights[li], ro, tn.transportDirection, lt) && bestT > lt) { meta.lightHit = (int)li; bestT = lt; }
        }
        meta.t = (float)bestT;
        if (meta.lightHit < 0) {   // also a light found by an earlier pass of this node keeps it emissive
            out.hasNextHit = result.ok;
            out.nextHit = result;
            out.tracedMiss = !result.ok;
            tn.isEmissive = false;
            tn.pathTerminationChance = 0.1f;
        } else {
            tn.isEmissive = true;
            tn.pathTerminationChance = 1.0f;
        }
        return;
    }
    case RTGPU_MAT_TRANSPARENT: // TransparentMaterial::sampleTransport, material/TransparentMaterial.cpp:3-11
        tn.leafMat = mi;
        tn.specularity = 1.0f;
        tn.typeBounce = true;
        tn.isEmissive = false;
        tn.pathTerminationChance = 0.01;
        tn.transportDirection = tn.hit.rd;
        return;
    case RTGPU_MAT_EMISSIVE: // EmissiveMaterial::sampleTransport, material/EmissiveMaterial.cpp:5-10
        tn.leafMat = mi;
        tn.pathTerminationChance = 1.0f;
        tn.isEmissive = true;
        return;
    case RTGPU_MAT_DIFFUSE: { // DiffuseMaterial::sampleTransport, material/DiffuseMaterial.cpp:24-138 (no photon map)
        tn.leafMat = mi;
        V3 normal = shading_normal(S, m, tn.hit);
        tn.specularity = 0.0f;
        tn.typeBounce = true;
        tn.isPhotonMapRay = false;
        if (diffuse_reads_map(S, *R.path, R.curI)) { // material/DiffuseMaterial.cpp:56-80
            tn.pathTerminationChance = 1.0f;
            tn.isEmissive = true;
            tn.isPhotonMapRay = true;
            tn.isNEERay = false;
            rtgpu_photon_gather g = gather_photons(S, tn.hit.hitpoint(), -tn.hit.rd);
            RGBf value{g.energy[0], g.energy[1], g.energy[2]};
            tn.photonLighting = rgb_div_d(rgb_div_d(rgb_scale_d(value, m.intensity), PI * g.max_dist * g.max_dist), PI);
            return;
        }
        tn.pathTerminationChance = 0.1f;
        tn.isEmissive = false;
        bool useNEE = rc.unit(DIM_NEE_CHOICE) > 0.5f;
        if (useNEE) {
            tn.directLighting = nee_sample(R, tn.hit.hitpoint(), sampleI, sampleCount, rc, tn.transportDirection);
            tn.pathTerminationChance = 1.0f;
            tn.isEmissive = true;
            tn.isNEERay = true;
        } else {
            Basis basis = basis_from(normal);
            float sx, sy;
            stratified_square((int)std::sqrt((double)sampleCount), (float)sampleI, rc.unit(DIM_SQUARE_X), rc.unit(DIM_SQUARE_Y), sx, sy);
            V3 l = cosine_hemisphere(sx, sy);
            tn.transportDirection = ((basis.u * l.x) + (basis.v * l.y)) + (basis.w * l.z);
            tn.isNEERay = false;
            if (S.pmMode == RTGPU_PM_CAUSTICS) out.causticsCallback = true;
        }
        return;
    }
    case RTGPU_MAT_GLOSSY: { // GlossyMaterial::sampleTransport, material/GlossyMaterial.cpp:68-180
        tn.leafMat = mi;
        V3 hitpoint = tn.hit.hitpoint();
        V3 normal = shading_normal(S, m, tn.hit);   // meta->normal
        const float roughness = m.roughness;
        tn.specularity = 1.0f - roughness;
        tn.typeBounce = true;
        bool neeEnabled = roughness > 0;
        bool useNEE = neeEnabled && rc.unit(DIM_NEE_CHOICE) > 0.5;
        tn.isNEERay = useNEE;
        tn.lightHit = -1;
        if (useNEE) {
            RGBf radiance = nee_sample(R, hitpoint, sampleI, sampleCount, rc, tn.transportDirection);
            V3 microNormal = normalized((-tn.hit.rd) + tn.transportDirection);
            double angle = std::max(0.0f, dot(microNormal, tn.transportDirection));
            float f = ggx_brdf(roughness, -tn.hit.rd, tn.transportDirection, normal, microNormal);
            tn.neeRadiance = rgb_scale_d(rgb_scale_d(radiance, angle), f);
            tn.isEmissive = true;
            tn.pathTerminationChance = 1.0f;
        } else {
            tn.neeRadiance = RGBf{};
            float sx, sy;
            stratified_square(sampleCount, (float)sampleI, rc.unit(DIM_SQUARE_X), rc.unit(DIM_SQUARE_Y), sx, sy);
            V3 micro = normalized(vndf_sample(normal, -tn.hit.rd, roughness, sx, sy));
            // getPerfectReflectionDir, GlossyMaterial.cpp:22-25 (the double factor narrows to float)
            V3 sampleDir = tn.hit.rd + ((float)(2.0 * dot(micro, -tn.hit.rd)) * micro);
            tn.transportDirection = sampleDir;
            if (dot(sampleDir, tn.hit.normal) < 0) {
                tn.isEmissive = true;
                tn.pathTerminationChance = 1.0f;
                return;
            }
            V3 ro = hitpoint + (tn.transportDirection * 0.001f);
            R.extension++;
            SceneHit result = scene_trace(S, ro, tn.transportDirection);
            if (neeEnabled) {
                bool isLightHit = false;
                for (const auto& light : S.areaLights) {
                    double lt;
                    if (area_light_hit(light, ro, tn.transportDirection, lt) && (!result.ok || result.t > lt)) { isLightHit = true; break; }
                }
                if (isLightHit) {
                    tn.isEmissive = true;
                    tn.pathTerminationChance = 1.0f;
                    tn.isNEERay = true;
                } else {
                    tn.isEmissive = false;
                    tn.pathTerminationChance = 0.2f;
                    out.hasNextHit = result.ok;
                    out.nextHit = result;
                    if (!result.ok) { out.hasNextHit = false; }
                }
            } else {
                double bestT = result.ok ? result.t : 1E99;
                for (size_t li = 0; li < S.areaLights.size(); ++li) {
                    double lt;
                    if (area_light_hit(S.areaLights[li], ro, tn.transportDirection, lt) && bestT > lt) { tn.lightHit = (int)li; bestT = lt; }
                }
                if (tn.lightHit < 0) {
                    out.hasNextHit = result.ok;
                    out.nextHit = result;
                    tn.isEmissive = false;
                    tn.pathTerminationChance = 0.2f;
                } else {
                    tn.isEmissive = true;
                    tn.pathTerminationChance = 1.0f;
                }
            }
        }
        return;
    }
    }
}

// IMaterial::bsdf for the supported closures
RGBf leaf_bsdf(const orc_scene& S, const TransportNode& tn, uint32_t mi, RGBf incoming)
{
    const rtgpu_material& m = S.materials[mi];
    RGBf color{m.color[0], m.color[1], m.color[2]};
    if (m.texture != 0 && m.type == RTGPU_MAT_DIFFUSE) { // albedoMap, material/DiffuseMaterial.cpp:144-152
        float x = fmodf(tn.hit.u, 1.0f), y = fmodf(tn.hit.v, 1.0f);
        if (x < 0) x = 1.0f + x;
        if (y < 0) y = 1.0f + y;
        color = texture_get(S.textures[m.texture - 1], x, y);
    } else if (m.texture != 0 && m.type == RTGPU_MAT_EMISSIVE) { // emissionMap, material/EmissiveMaterial.cpp:14-21
        float x = std::abs(std::fmod(tn.hit.u, 1.0f)), y = std::abs(std::fmod(tn.hit.v, 1.0f));
        color = texture_get(S.textures[m.texture - 1], x, y);
    }
    switch (m.type) {
    case RTGPU_MAT_DIFFUSE: { // DiffuseMaterial::bsdf, material/DiffuseMaterial.cpp:140-197
        if (tn.isPhotonMapRay) { // :154-166
            RGBf o = rgb_mul(color, tn.photonLighting);
            return S.pmMode == RTGPU_PM_CAUSTICS ? rgb_scale_f(o, (float)2) : o;
        }
        double angle = std::max(0.0f, dot(shading_normal(S, m, tn.hit), tn.transportDirection));
        RGBf value;
        if (tn.isNEERay) {
            RGBf direct = rgb_scale_d(tn.directLighting, angle);
            value = rgb_scale_f(rgb_scale_d(direct, m.intensity / PI), 2.0f);
        } else {
            RGBf indirect = rgb_scale_d(incoming, angle);
            value = rgb_scale_f(rgb_scale_d(indirect, m.intensity), 2.0f);
        }
        return rgb_mul(color, value);
    }
    case RTGPU_MAT_GLOSSY: { // GlossyMaterial::bsdf, material/GlossyMaterial.cpp:182-207
        bool neeEnabled = m.roughness > 0;
        RGBf radiance = incoming;
        if (neeEnabled) {
            if (tn.isNEERay) radiance = tn.neeRadiance;
            radiance = rgb_scale_f(radiance, 2.0f);
        } else if (tn.lightHit >= 0) {
            const rtgpu_area_light& L = S.areaLights[tn.lightHit];
            RGBf lightEnergy = light_energy(L.color, L.intensity);
            RGBf lightIrradiance = rgb_div_d(lightEnergy, (double)tri_area(mk(L.a), mk(L.b), mk(L.c)));
            radiance = rgb_div_i(lightIrradiance, 2);
        }
        return rgb_mul(radiance, color);
    }
    case RTGPU_MAT_EMISSIVE: // EmissiveMaterial::bsdf, material/EmissiveMaterial.cpp:12-22
        return rgb_scale_d(color, m.intensity);
    case RTGPU_MAT_TRANSPARENT: // TransparentMaterial::bsdf, material/TransparentMaterial.cpp:13-16
        return incoming;
    case RTGPU_MAT_GLASS: { // GlassMaterial::bsdf, material/GlassMaterial.cpp:135-178
        const TransportNode::GlassMeta& meta = tn.glass[tn.leafSlot];
        if (std::isnan(meta.reflectionWeight)) return RGBf{};   // :139-144
        V3 normal = tn.hit.normal;
        V3 direction = tn.transportDirection;
        bool isInternalRay = dot(normal, direction) < 0;
        RGBf val = incoming;
        if (meta.lightHit >= 0) {
            const rtgpu_area_light& L = S.areaLights[meta.lightHit];
            RGBf lightEnergy = light_energy(L.color, L.intensity);
            RGBf lightIrradiance = rgb_div_d(lightEnergy, (double)tri_area(mk(L.a), mk(L.b), mk(L.c)));
            RGBf lightRadiance = rgb_div_d(lightIrradiance, PI);
            V3 lampPoint = tn.transportDirection * meta.t;   // as the reference has it: not offset by the hit point
            V3 objectToLamp = normalized(lampPoint - tn.hit.hitpoint());
            float angle = std::max(0.0f, dot(normal, objectToLamp));
            val = rgb_scale_d(lightRadiance, (double)angle);
        }
        if (isInternalRay) {
            double t = std::isnan(meta.t) ? tn.hit.t : (double)meta.t;
            double power = m.roughness * t;   // attenuationStrength (float) * t
            val = rgb_mul(val, RGBf{(float)lm_pow(m.color[0], power), (float)lm_pow(m.color[1], power), (float)lm_pow(m.color[2], power)});
        }
        return val;
    }
    default: return RGBf{};
    }
}

// The node's material evaluated the way sampleTransport walked it: the leaf, plus the emissive first child of
// every AddMaterial above it (AddMaterial::bsdf, material/AddMaterial.cpp:12-17: result1 + result2)
RGBf mat_bsdf(const orc_scene& S, const TransportNode& tn, RGBf incoming)
{
    RGBf r = leaf_bsdf(S, tn, tn.leafMat, incoming);
    for (int a = tn.nAdd - 1; a >= 0; --a) {
        RGBf e = leaf_bsdf(S, tn, tn.addFirst[a], incoming);
        r = RGBf{e.r + r.r, e.g + r.g, e.b + r.b};
    }
    return r;
}

struct PathKey {
    uint32_t seed, pixel;
    int geomI, materialAA;
};

// samplePath, renderer/Renderer.cpp:14-87.  sampleJ selects the RNG stream: vertices before
// the first variance vertex are only ever sampled in pass 0.
bool sample_path(RenderCtx& R, std::vector<TransportNode>& path, int startIdx, int maxPathLength, int materialAA, int sampleJ, const PathKey& key)
{
    const orc_scene& S = R.S;
    int curI = startIdx;
    bool pathDone = false, pathTerminated = false;
    bool hasPassedVariance = false;
    bool pendingCallback = false;   // curNodeCallback of renderer/Renderer.cpp:22, 40-50
    do {
        TransportNode& cur = path[curI];
        uint32_t mi = hit_material(S, cur.hit);
        R.path = &path;
        R.curI = curI;
        bool curVar = mat_has_variance(S, mi, cur.hit, diffuse_reads_map(S, path, curI));
        bool isFirstVar = curVar && !hasPassedVariance;
        int sampleCount = isFirstVar ? materialAA : 1;
        int sampleI = isFirstVar ? sampleJ : 0;
        hasPassedVariance = hasPassedVariance || curVar;
        RandCtx rc{key.seed, key.pixel, (uint32_t)(key.geomI * key.materialAA + sampleJ), (uint32_t)curI};
        SampleOut so;
        sample_transport(R, cur, mi, sampleI, sampleCount, rc, 0, so);
        if (pendingCallback) {
            // the callback DiffuseMaterial left at the previous vertex (material/DiffuseMaterial.cpp:106-134): this
            // vertex turned out specular, so the previous one ends the path with the caustic map's estimate
            TransportNode& prev = path[curI - 1];
            if (cur.specularity > 0.8) {
                prev.pathTerminationChance = 1.0f;
                prev.isEmissive = true;
                rtgpu_photon_gather g = gather_photons(S, prev.hit.hitpoint(), -prev.hit.rd);
                RGBf value{g.energy[0], g.energy[1], g.energy[2]};
                prev.photonLighting = rgb_div_d(rgb_div_d(value, PI * (g.max_dist * g.max_dist)), PI);
                prev.isPhotonMapRay = true;
            }
            if (prev.isEmissive || prev.pathTerminationChance == 1.0f) break;   // Renderer.cpp:44-47: curI stays, the path ends at prev
        }
        pendingCallback = so.causticsCallback;
        pathTerminated = (curI + 1 == maxPathLength) || (!cur.isEmissive && rc.unit(DIM_RR) < cur.pathTerminationChance);
        pathDone = cur.isEmissive || pathTerminated;
        if (!pathDone) {
            SceneHit hit;
            if (so.hasNextHit) hit = so.nextHit;
            else if (so.tracedMiss) hit = SceneHit{};   // GlassMaterial traced this very ray (offset 1e-4): the repeat of Renderer.cpp:57-61 misses again
            else {
                // renderer/Renderer.cpp:57-61; also reached when GlossyMaterial's own trace (1e-3
                // offset) found nothing: ctx.nextHit is empty, so the reference traces again
                R.extension++;
                hit = scene_trace(S, cur.hit.hitpoint() + (cur.transportDirection * 0.0001f), cur.transportDirection);
            }
            if (hit.ok) {
                TransportNode nn;
                nn.hit = hit;
                if ((int)path.size() < curI + 2) path.push_back(nn);
                else path[curI + 1] = nn;
            } else {
                pathDone = true;
            }
        }
        curI++;
    } while (!pathDone);
    path.resize(curI);
    return pathTerminated;
}

// IEnvironmentMaterial::getRadiance: ColorEnvironment.cpp:6-9 or ImageMapEnvironment.cpp:15-23 (float asin / cos / atan2 of
// the float direction, the rest in double; Texture<float>::get divides by 1 and raises to gamma 1).  Indices that fall
// outside the image (the reference would read past it: straight down, the seam) are clamped; NaN reads column / row 0.
RGBf env_radiance(const orc_scene& S, V3 dir)
{
    if (S.hasEnv == 3) {   // SkyEnvironment::getRadiance, material/environment/SkyEnvironment.cpp:3-10
        const float t = 0.5f * (dir.y + 1.0f);
        return rgb_add(rgb_scale_f(RGBf{0.5f, 0.7f, 1.0f}, t), rgb_scale_f(RGBf{1.0f, 1.0f, 1.0f}, 1.0f - t));
    }
    if (S.hasEnv != 2) return S.envRadiance;
    const float latitude = lm_asin(dir.y);
    const float cl = lm_cos(latitude);
    const double longitude = M_PI + lm_atan2(dir.z / cl, dir.x / cl);
    const double x = S.envW * longitude / (2.0f * M_PI);
    const double y = S.envH * (1.0f - (((M_PI / 2.0f) + latitude) / M_PI));
    int xi = x == x ? (int)x : 0, yi = y == y ? (int)y : 0;
    xi = std::min(std::max(xi, 0), (int)S.envW - 1);
    yi = std::min(std::max(yi, 0), (int)S.envH - 1);
    const float* t = &S.envMap[((size_t)yi * S.envW + (size_t)xi) * 4];
    return rgb_scale_d(RGBf{t[0] / 1.0f, t[1] / 1.0f, t[2] / 1.0f}, S.envIntensity);
}

// calculatePathEnergy, renderer/Renderer.cpp:89-109
RGBf path_energy(const orc_scene& S, const std::vector<TransportNode>& path)
{
    const TransportNode& last = path.back();
    RGBf energy;
    if (S.hasEnv && !last.isEmissive) energy = env_radiance(S, last.transportDirection);
    for (int i = (int)path.size() - 1; i >= 0; --i) {
        const TransportNode& n = path[i];
        energy = mat_bsdf(S, n, energy);
        if (!n.isEmissive) energy = rgb_div_f(energy, 1.0f - n.pathTerminationChance);
    }
    return energy;
}

// sampleUniformStratifiedCube, math/Sampler.h:28-41
V3 stratified_cube(float level, float sampleI, V3 r)
{
    float binsPerAxis = lm_pow(level, 1.0f / 3.0f);
    float oneOverBinsPerAxis = 1.0f / binsPerAxis;
    float binX = std::fmod(sampleI, binsPerAxis);
    float binY = std::floor(std::fmod(sampleI, binsPerAxis * binsPerAxis) * oneOverBinsPerAxis);
    float binZ = std::floor(std::fmod(sampleI, level) * (oneOverBinsPerAxis * oneOverBinsPerAxis));
    return V3{(binX * oneOverBinsPerAxis) + (r.x * oneOverBinsPerAxis),
              (binY * oneOverBinsPerAxis) + (r.y * oneOverBinsPerAxis),
              (binZ * oneOverBinsPerAxis) + (r.z * oneOverBinsPerAxis)};
}
// sampleUniformCircle(radius, rand1, rand2, rand3), math/Sampler.h:90-103
void uniform_circle(float radius, V3 r, float& x, float& y)
{
    float angle = r.x * 2.0f * M_PI;
    float dist = (r.y * radius) + (r.z * radius);
    if (dist > radius) dist = (2.0f * radius) - dist;
    x = lm_cos(angle) * dist;
    y = lm_sin(angle) * dist;
}

// PerspectiveCamera::generateRay, camera/PerspectiveCamera.cpp:26-52
void camera_ray(const rtgpu_camera& cam, int px, int py, int W, int H, int aaLevel, int sampleI, int jitterMode, const RandCtx& rc, V3& o, V3& d)
{
    const bool random = jitterMode != RTGPU_JITTER_CENTER;
    float r0 = 0.5f, r1 = 0.5f;
    if (random) { r0 = rc.unit(0); r1 = rc.unit(1); }
    float apertureSamplesPerSensorSample = cam.aperture > 0 ? std::min(8, aaLevel / 2) : 1.0f;
    float sx, sy;
    stratified_square((int)(aaLevel / apertureSamplesPerSensorSample), sampleI / apertureSamplesPerSensorSample, r0, r1, sx, sy);
    sx = (float)px + sx;
    sy = (float)py + sy;
    double height = H * cam.width / W;
    double u = cam.width * (sx / W - 0.5);
    double v = -height * (sy / H - 0.5);
    V3 U = mk(cam.u), V = mk(cam.v), Wv = mk(cam.w);
    V3 dir = ((U * (float)u) + (V * (float)v)) - Wv;
    d = normalized(dir);
    o = mk(cam.origin);
    if (cam.aperture == 0) return;
    V3 focalPlanePoint = o + (d * (float)cam.focal_distance);
    V3 r{0.5f, 0.5f, 0.5f};
    if (random) r = V3{rc.unit(2), rc.unit(3), rc.unit(4)};
    // sampleUniformStratifiedCircle(float radius, int level, int sampleI), math/Sampler.h:110-114
    V3 cube = stratified_cube((float)(int)apertureSamplesPerSensorSample, (float)(int)std::fmod((double)sampleI, (double)apertureSamplesPerSensorSample), r);
    float ox, oy;
    uniform_circle((float)cam.aperture, cube, ox, oy);
    V3 lensPoint = o;
    lensPoint = lensPoint + (U * ox);
    lensPoint = lensPoint + (V * oy);
    o = lensPoint;
    d = normalized(focalPlanePoint - lensPoint);
}

// RenderTileTask::execute for one pixel, renderer/Renderer.cpp:135-267
RGBf render_pixel(RenderCtx& R, const rtgpu_camera& cam, const rtgpu_render_settings& st, int x, int y)
{
    const orc_scene& S = R.S;
    const int G = st.geometry_aa, M = st.material_aa;
    const uint32_t pixel = (uint32_t)(y * st.width + x);
    RGBf pixelValue;
    std::vector<TransportNode> path;
    path.reserve(st.max_path_length);
    for (int i = 0; i < G; ++i) {
        RandCtx camRc{st.seed, pixel, (uint32_t)(i * M), VERTEX_CAMERA};
        V3 ro, rd;
        camera_ray(cam, x, y, st.width, st.height, G, i, st.jitter_mode, camRc, ro, rd);
        if (S.primaryO && S.primaryD) {   // test hook: the camera rays another implementation generated
            const size_t k = (((size_t)i * st.height + y) * st.width + x) * 3;
            ro = mk(S.primaryO + k); rd = mk(S.primaryD + k);
        }
        R.primary++;
        SceneHit hit = scene_trace(S, ro, rd);
        path.clear();
        { // area light closer than the hit, Renderer.cpp:184-204
            double bestT = hit.ok ? hit.t : (double)INFINITY;
            int light = -1;
            for (size_t li = 0; li < S.areaLights.size(); ++li) {
                double lt;
                if (area_light_hit(S.areaLights[li], ro, rd, lt) && bestT > lt) { light = (int)li; bestT = lt; }
            }
            if (light >= 0) {
                const rtgpu_area_light& L = S.areaLights[light];
                RGBf e = light_energy(L.color, L.intensity);
                RGBf irr = rgb_div_d(e, (double)tri_area(mk(L.a), mk(L.b), mk(L.c)));
                pixelValue = rgb_add(pixelValue, rgb_div_i(irr, 2));
                continue;
            }
        }
        if (!hit.ok) {
            if (S.hasEnv) pixelValue = rgb_add(pixelValue, env_radiance(S, rd));
            continue;
        }
        TransportNode n0;
        n0.hit = hit;
        path.push_back(n0);
        PathKey key{st.seed, pixel, i, M};
        bool terminated = sample_path(R, path, 0, st.max_path_length, M, 0, key);
        int firstVar = 0;
        for (; firstVar < (int)path.size(); ++firstVar)
            if (mat_has_variance(S, hit_material(S, path[firstVar].hit), path[firstVar].hit, diffuse_reads_map(S, path, firstVar))) break;
        RGBf matSample;
        if (!terminated) matSample = path_energy(S, path);
        if (firstVar < (int)path.size()) {
            for (int j = 1; j < M; ++j) {
                terminated = sample_path(R, path, firstVar, st.max_path_length, M, j, key);
                if (!terminated) matSample = rgb_add(matSample, path_energy(S, path));
            }
            matSample = rgb_div_i(matSample, M);
        }
        pixelValue = rgb_add(pixelValue, matSample);
    }
    return rgb_div_i(pixelValue, G);
}

template <typename F>
void parallel_for(uint64_t n, int n_threads, const F& f)
{
    if (n_threads <= 1 || n < 64) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = std::max<uint64_t>(64, n / (uint64_t)(n_threads * 16));
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&] {
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk));
            }
        });
    for (auto& t : th) t.join();
}

// ---- photon tracing --------------------------------------------------------------------------
// Random numbers of a photon: rand_unit(seed, kPhotonPixel | light, ray, vertex, dim); vertex kVertexEmit for the
// emission, the bounce number afterwards.
constexpr uint32_t kPhotonPixel = 0x40000000u;
constexpr uint32_t kVertexEmit = 254;

// Rand::sampleStdNormalDist (math/FastRandom.h) stand-in: Box-Muller on two uniform draws
inline float std_normal(float u1, float u2) { return (float)(std::sqrt(-2.0 * std::log(1.0 - (double)u1)) * std::cos(2.0 * PI * (double)u2)); }

// IMaterial::interactPhoton of the closures: direction, energy, diffuseness
void interact_photon(const orc_scene& S, const SceneHit& hit, uint32_t mi, RGBf incoming, const RandCtx& rc, int level, V3& dir, RGBf& energy, float& diffuseness)
{
    const rtgpu_material& m = S.materials[mi];
    switch (m.type) {
    case RTGPU_MAT_CONSTMIX:
    case RTGPU_MAT_FRESNELMIX: // MixMaterial::interactPhoton, material/MixMaterial.cpp:47-57
        interact_photon(S, hit, rc.unit(DIM_MIX0 + level) > mat_mix_factor(m, hit) ? m.first : m.second, incoming, rc, level + 1, dir, energy, diffuseness);
        return;
    case RTGPU_MAT_ADD: // AddMaterial::interactPhoton, material/AddMaterial.cpp:19-23: the first child
        interact_photon(S, hit, m.first, incoming, rc, level + 1, dir, energy, diffuseness);
        return;
    case RTGPU_MAT_DIFFUSE: { // DiffuseMaterial::interactPhoton, material/DiffuseMaterial.cpp:206-226
        RGBf color{m.color[0], m.color[1], m.color[2]};
        if (m.texture != 0) color = texture_get(S.textures[m.texture - 1], std::abs(std::fmod(hit.u, 1.0f)), std::abs(std::fmod(hit.v, 1.0f)));
        Basis basis = basis_from(shading_normal(S, m, hit));
        V3 l = cosine_hemisphere(rc.unit(DIM_SQUARE_X), rc.unit(DIM_SQUARE_Y));
        dir = ((basis.u * l.x) + (basis.v * l.y)) + (basis.w * l.z);
        energy = rgb_mul(color, incoming);
        diffuseness = 1.0f;
        return;
    }
    case RTGPU_MAT_GLOSSY: { // GlossyMaterial::interactPhoton, material/GlossyMaterial.cpp:209-224
        V3 micro = normalized(vndf_sample(shading_normal(S, m, hit), -hit.rd, m.roughness, rc.unit(DIM_SQUARE_X), rc.unit(DIM_SQUARE_Y)));
        dir = hit.rd + ((float)(2.0 * dot(micro, -hit.rd)) * micro);
        energy = incoming;
        diffuseness = m.roughness;
        return;
    }
    case RTGPU_MAT_GLASS: { // GlassMaterial::interactPhoton + samplePhotonDirection, material/GlassMaterial.cpp:180-252
        V3 normal = hit.normal;
        bool attenuate = dot(normal, hit.rd) < 0;
        V3 incomingDir = hit.rd, objToIncoming = -incomingDir;
        V3 reflectedRayDir = (-objToIncoming) + ((float)(2.0 * dot(normal, objToIncoming)) * normal);
        double nDotWo = dot(normal, objToIncoming);
        double incomingAngle = nDotWo;
        bool isInternalRay = nDotWo < 0;
        double curRelIor = m.intensity / 1.0;
        if (isInternalRay) curRelIor = 1.0 / curRelIor;
        double cosTransmissionSqr = 1.0 - ((1.0 - (incomingAngle * incomingAngle)) / (curRelIor * curRelIor));
        if (cosTransmissionSqr < 0) {
            dir = reflectedRayDir;
        } else {
            if (isInternalRay) { nDotWo = -nDotWo; incomingAngle = -incomingAngle; normal = -normal; }
            double cosRefractAngle = std::sqrt(cosTransmissionSqr);
            V3 transmittedRayDir = (incomingDir / (float)curRelIor) - ((float)(cosRefractAngle - (incomingAngle / curRelIor)) * normal);
            double rs = ((curRelIor * nDotWo) - cosRefractAngle) / ((curRelIor * nDotWo) + cosRefractAngle);
            double rp = (nDotWo - (curRelIor * cosRefractAngle)) / (nDotWo + (curRelIor * cosRefractAngle));
            double kr = ((rs * rs) + (rp * rp)) / 2.0;
            double reflectionWeight = kr / (kr + (1.0 - kr));
            dir = (double)rc.unit(DIM_GLASS) < reflectionWeight ? reflectedRayDir : transmittedRayDir;
        }
        energy = incoming;
        if (attenuate) {
            double power = m.roughness * hit.t;
            energy = rgb_mul(incoming, RGBf{(float)std::pow(m.color[0], power), (float)std::pow(m.color[1], power), (float)std::pow(m.color[2], power)});
        }
        diffuseness = 0.0f;
        return;
    }
    case RTGPU_MAT_TRANSPARENT: // TransparentMaterial.cpp:18-21
        dir = hit.rd; energy = incoming; diffuseness = 0.0f;
        return;
    default: // EmissiveMaterial.cpp:24-27
        dir = -hit.rd; energy = RGBf{}; diffuseness = 0.0f;
        return;
    }
}

// PhotonDriver::trace, photonmapping/PhotonTracer.cpp:12-58
void trace_photon(const orc_scene& S, V3 o, V3 d, RGBf photonEnergy, int mode, uint32_t seed, uint32_t pixel, uint32_t ray, std::vector<rtgpu_photon>& acc)
{
    bool hasPassedSpecular = false;
    for (int i = 0; i < 12; ++i) {
        SceneHit hit = scene_trace(S, o, d);
        if (!hit.ok) break;
        V3 hitpoint = hit.hitpoint();
        RandCtx rc{seed, pixel, ray, (uint32_t)i};
        V3 newDir;
        RGBf newEnergy;
        float diffuseness;
        interact_photon(S, hit, hit_material(S, hit), photonEnergy, rc, 0, newDir, newEnergy, diffuseness);
        newDir = normalized(newDir);
        bool isDiffuseTransport = diffuseness >= 0.2;
        bool isCaustic = isDiffuseTransport && hasPassedSpecular;
        if ((mode == RTGPU_PM_CAUSTICS && isCaustic) || (mode == RTGPU_PM_FULL && isDiffuseTransport)) {
            rtgpu_photon ph{{hitpoint.x, hitpoint.y, hitpoint.z}, {hit.normal.x, hit.normal.y, hit.normal.z}, {photonEnergy.r, photonEnergy.g, photonEnergy.b}, isCaustic ? 1u : 0u};
            acc.push_back(ph);
        }
        hasPassedSpecular = hasPassedSpecular || !isDiffuseTransport;
        o = hitpoint + (newDir * 0.00005f);
        d = newDir;
        photonEnergy = newEnergy;
    }
}

// The rays PhotonTracer::tracePhotons emits (photonmapping/PhotonTracer.cpp:139-172) in task order: for each light the
// batches of createBatches (utility/Batcher.h: 10 000 rays each; a last partial batch is issued twice) -- counted
// over photonsPerPointLight for both kinds of light, as the reference does.
struct PhotonBatch { uint32_t light; bool area; uint64_t start, end, firstRay; };
std::vector<PhotonBatch> photon_batches(const orc_scene& S, uint64_t perPoint)
{
    std::vector<PhotonBatch> out;
    const uint64_t batchSize = 10000;
    auto add = [&](uint32_t light, bool area) {
        uint64_t ray = 0;
        for (uint64_t i = 0; i < perPoint; i += batchSize) { uint64_t e = std::min(i + batchSize, perPoint); out.push_back({light, area, i, e, ray}); ray += e - i; }
        uint64_t rem = perPoint % batchSize;
        if (rem > 0) out.push_back({light, area, perPoint - rem, perPoint, ray});
    };
    for (uint32_t l = 0; l < S.pointLights.size(); ++l) add(l, false);
    for (uint32_t l = 0; l < S.areaLights.size(); ++l) add(l, true);
    return out;
}

void emit_and_trace(const orc_scene& S, const PhotonBatch& b, uint64_t k, int mode, uint64_t perPoint, uint64_t perArea, uint32_t seed, std::vector<rtgpu_photon>& acc)
{
    const uint32_t ray = (uint32_t)(b.firstRay + k);
    if (!b.area) { // PointLightPhotonTracingTask::execute, :84-96
        const rtgpu_point_light& L = S.pointLights[b.light];
        const uint32_t pixel = kPhotonPixel | b.light;
        RandCtx rc{seed, pixel, ray, kVertexEmit};
        RGBf energyPerPhoton = rgb_scale_d(RGBf{L.color[0], L.color[1], L.color[2]}, L.intensity / (double)perPoint);
        float x = std_normal(rc.unit(0), rc.unit(1)), y = std_normal(rc.unit(2), rc.unit(3)), z = std_normal(rc.unit(4), rc.unit(5));
        float len = std::sqrt(x * x + y * y + z * z);   // sampleUniformSphere(1.0), math/Sampler.h:42-49
        trace_photon(S, mk(L.pos), V3{1.0f * x / len, 1.0f * y / len, 1.0f * z / len}, energyPerPhoton, mode, seed, pixel, ray, acc);
    } else { // AreaLightPhotonTracingTask::execute, :121-137
        const rtgpu_area_light& L = S.areaLights[b.light];
        const uint32_t pixel = kPhotonPixel | (uint32_t)(S.pointLights.size() + b.light);
        RandCtx rc{seed, pixel, ray, kVertexEmit};
        RGBf energyPerPhoton = rgb_div_f(light_energy(L.color, L.intensity), (float)perArea);
        V3 a = mk(L.a), bb = mk(L.b), c = mk(L.c);
        Basis basis = basis_from(cross(bb - a, c - a));
        V3 photonPos = area_light_point(L, (int)(b.end - b.start), (int)k, rc.unit(DIM_AREA_U), rc.unit(DIM_AREA_V));
        V3 l = cosine_hemisphere(rc.unit(DIM_SQUARE_X), rc.unit(DIM_SQUARE_Y));
        V3 dir = normalized(((basis.u * l.x) + (basis.v * l.y)) + (basis.w * l.z));
        trace_photon(S, photonPos + (dir * 0.0001f), dir, energyPerPhoton, mode, seed, pixel, ray, acc);
    }
}

} // namespace

// ==========================================================================================
// C interface
// ==========================================================================================
extern "C" {

orc_scene* orc_scene_create(const rtgpu_scene_desc* desc, int n_threads)
{
    (void)n_threads;
    auto* S = new orc_scene();
    S->meshes.resize(desc->n_meshes);
    for (uint32_t mi = 0; mi < desc->n_meshes; ++mi) {
        const rtgpu_mesh_desc& md = desc->meshes[mi];
        Mesh& m = S->meshes[mi];
        m.ntri = md.n_triangles;
        m.vertices.resize(md.n_vertices);
        m.normals.resize(md.n_vertices);
        for (uint32_t v = 0; v < md.n_vertices; ++v) {
            m.vertices[v] = mk(md.vertices + 3 * v);
            m.normals[v] = md.normals ? mk(md.normals + 3 * v) : V3{0, 0, 0};
        }
        m.indices.assign(md.indices, md.indices + 3 * (size_t)md.n_triangles);
        if (md.tri_material) m.triMaterial.assign(md.tri_material, md.tri_material + md.n_triangles);
        if (md.texcoords && desc->n_textures) m.texCoords.assign(md.texcoords, md.texcoords + 2 * (size_t)md.n_vertices);
        mesh_build(m); // InstancedModelList::buildShapeBVHCache, shape/list/InstancedModelList.cpp:60-81
    }
    S->instances.resize(desc->n_instances);
    BuildList L;
    L.boxes.resize(desc->n_instances);
    L.sortKey.resize(desc->n_instances);
    L.order.resize(desc->n_instances);
    for (uint32_t i = 0; i < desc->n_instances; ++i) {
        const rtgpu_instance_desc& id = desc->instances[i];
        Instance& in = S->instances[i];
        in.shape = id.shape;
        in.material = id.material;
        std::memcpy(in.xf.m, id.transform, sizeof(in.xf.m));
        std::memcpy(in.xf.inv, id.inverse, sizeof(in.xf.inv));
        AABB shapeBox = in.shape >= 0 ? S->meshes[in.shape].box : sphere_aabb();
        V3 shapeCentroid = in.shape >= 0 ? S->meshes[in.shape].centroid : V3{0, 0, 0};
        in.worldBox = aabb_of_transformed(shapeBox, in.xf.m);
        L.boxes[i] = in.worldBox;
        L.sortKey[i] = xf_point(in.xf.m, shapeCentroid); // InstancedModelList::sortByCentroid, :48-58
        L.order[i] = i;
    }
    S->topNodes = build_bvh(L);
    S->instOrder = std::move(L.order);
    S->materials.assign(desc->materials, desc->materials + desc->n_materials);
    S->pointLights.assign(desc->point_lights, desc->point_lights + desc->n_point_lights);
    S->areaLights.assign(desc->area_lights, desc->area_lights + desc->n_area_lights);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const rtgpu_texture& t = desc->textures[i];
        S->textures.push_back({std::vector<uint8_t>(t.rgba, t.rgba + (size_t)t.width * t.height * 4), t.width, t.height, t.gamma});
    }
    if (desc->n_directional_lights) S->dirLights.assign(desc->directional_lights, desc->directional_lights + desc->n_directional_lights);
    S->hasEnv = desc->has_environment;
    if (desc->has_environment == 2) {
        S->envW = desc->env_map_width; S->envH = desc->env_map_height;
        S->envMap.assign(desc->env_map, desc->env_map + (size_t)S->envW * S->envH * 4);
        S->envIntensity = desc->env_intensity;
    }
    S->envRadiance = rgb_scale_d(RGBf{desc->env_color[0], desc->env_color[1], desc->env_color[2]}, desc->env_intensity);
    return S;
}

void orc_scene_destroy(orc_scene* s) { delete s; }

uint64_t orc_scene_node_count(const orc_scene* s)
{
    uint64_t n = s->topNodes.size();
    for (const auto& m : s->meshes) n += m.nodes.size();
    return n;
}

void orc_trace_closest(const orc_scene* s, const float* o, const float* d, uint64_t n, rtgpu_hit* out, uint32_t* counts, int n_threads)
{
    parallel_for(n, n_threads, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            g_cnt = Counters{};
            SceneHit h = scene_trace(*s, mk(o + 3 * i), mk(d + 3 * i));
            rtgpu_hit r;
            if (h.ok) { r.t = (float)h.t; r.prim = h.prim; r.inst = h.inst; r.beta = h.beta; r.gamma = h.gamma; }
            else { r.t = INFINITY; r.prim = RTGPU_INVALID_ID; r.inst = RTGPU_INVALID_ID; r.beta = 0; r.gamma = 0; }
            out[i] = r;
            if (counts) { counts[2 * i] = g_cnt.boxes; counts[2 * i + 1] = g_cnt.tris; }
        }
    });
}

void orc_trace_any(const orc_scene* s, const float* o, const float* d, const float* tmax, uint64_t n, uint8_t* occluded, int n_threads)
{
    parallel_for(n, n_threads, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i)
            occluded[i] = scene_visibility(*s, mk(o + 3 * i), mk(d + 3 * i), tmax[i]) ? 1 : 0;
    });
}

void orc_brute_closest(const orc_scene* s, const float* o, const float* d, uint64_t n, rtgpu_hit* out, uint32_t* n_tied, int n_threads)
{
    parallel_for(n, n_threads, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            V3 ro = mk(o + 3 * i), rd = mk(d + 3 * i);
            double bestT = INFINITY;
            rtgpu_hit best{INFINITY, RTGPU_INVALID_ID, RTGPU_INVALID_ID, 0, 0};
            uint32_t tied = 0;
            for (uint32_t ii = 0; ii < s->instances.size(); ++ii) {
                const Instance& in = s->instances[ii];
                V3 to = xf_point(in.xf.inv, ro), td = xf_vector(in.xf.inv, rd);
                if (in.shape >= 0) {
                    const Mesh& m = s->meshes[in.shape];
                    for (uint32_t tri = 0; tri < m.ntri; ++tri) {
                        float be, ga;
                        double t;
                        if (!tri_intersect(to, td, m.vertices[m.indices[3 * tri]], m.vertices[m.indices[3 * tri + 1]], m.vertices[m.indices[3 * tri + 2]], be, ga, t)) continue;
                        if (t < bestT) { bestT = t; best = rtgpu_hit{(float)t, tri, ii, be, ga}; tied = 1; }
                        else if (t == bestT) tied++;
                    }
                } else {
                    double t;
                    if (sphere_intersect(to, td, t)) {
                        if (t < bestT) { bestT = t; best = rtgpu_hit{(float)t, RTGPU_INVALID_ID, ii, 0, 0}; tied = 1; }
                        else if (t == bestT) tied++;
                    }
                }
            }
            out[i] = best;
            if (n_tied) n_tied[i] = tied;
        }
    });
}

// Test hooks, see orc_scene.  The arrays of orc_scene_set_primary_rays stay the caller's and must outlive the renders
// that use them; null pointers restore camera_ray().
void orc_scene_set_closest_rule(orc_scene* s, int rule) { s->closestRule = rule; }
void orc_scene_set_gather_rule(orc_scene* s, int rule) { s->gatherRule = rule; }
void orc_scene_set_primary_rays(orc_scene* s, const float* o, const float* d) { s->primaryO = o; s->primaryD = d; }

// Test hooks on the render path's libm calls (see LibmProbe).  orc_render_pixel_probe renders pixel (x, y) on the calling
// thread with the result of its `target`-th libm call moved by `ulps` units in the last place (target < 0: nothing moved)
// and returns the number of libm calls the pixel made; *fn receives which function the moved call was (LM_*), -1 if none.
int64_t orc_render_pixel_probe(const orc_scene* s, const rtgpu_camera* cam, const rtgpu_render_settings* st, int x, int y, int64_t target, int ulps,
                               float rgb[3], int* fn)
{
    g_libm = LibmProbe{(long)target, ulps, 0, -1};
    RenderCtx R{*s, nullptr};
    const RGBf p = render_pixel(R, *cam, *st, x, y);
    rgb[0] = p.r; rgb[1] = p.g; rgb[2] = p.b;
    const int64_t n = g_libm.calls;
    if (fn) *fn = g_libm.fn;
    g_libm = LibmProbe{};
    return n;
}

// Looks for ONE libm call of pixel (x, y) whose result, moved by one or two units in the last place either way, turns the pixel
// into `want` (every channel within tol_rel of the larger of the pixel's brightest channel and `floor`).  Returns 1 and the
// call / direction / function when there is one, 0 otherwise.
int orc_attribute_pixel_to_libm(const orc_scene* s, const rtgpu_camera* cam, const rtgpu_render_settings* st, int x, int y, const float want[3],
                                float tol_rel, float floor, int64_t* call, int* ulps, int* fn)
{
    float rgb[3];
    const int64_t n = orc_render_pixel_probe(s, cam, st, x, y, -1, 0, rgb, nullptr);
    const float scale = std::max(std::max(want[0], std::max(want[1], want[2])), floor);
    static const int kMoves[4] = {1, -1, 2, -2};
    for (int64_t k = 0; k < n; ++k)
        for (int u : kMoves) {
            int f = -1;
            orc_render_pixel_probe(s, cam, st, x, y, k, u, rgb, &f);
            if (std::fabs(rgb[0] - want[0]) <= tol_rel * scale && std::fabs(rgb[1] - want[1]) <= tol_rel * scale && std::fabs(rgb[2] - want[2]) <= tol_rel * scale) {
                if (call) *call = k;
                if (ulps) *ulps = u;
                if (fn) *fn = f;
                return 1;
            }
        }
    return 0;
}

void orc_ray_hits_instance_box(const orc_scene* s, const float* o, const float* d, uint64_t n, uint32_t inst, uint8_t* out)
{
    const AABB& bx = s->instances[inst].worldBox;
    for (uint64_t i = 0; i < n; ++i) out[i] = aabb_intersection(bx, mk(o + 3 * i), mk(d + 3 * i)) >= 0 ? 1 : 0;
}

void orc_generate_primary(const rtgpu_camera* cam, const rtgpu_render_settings* st, int32_t sample_index, float* o, float* d)
{
    size_t k = 0;
    for (int y = st->y_start; y < st->y_end; ++y)
        for (int x = st->x_start; x < st->x_end; ++x, ++k) {
            uint32_t pixel = (uint32_t)(y * st->width + x);
            RandCtx rc{st->seed, pixel, (uint32_t)(sample_index * st->material_aa), VERTEX_CAMERA};
            V3 ro, rd;
            camera_ray(*cam, x, y, st->width, st->height, st->geometry_aa, sample_index, st->jitter_mode, rc, ro, rd);
            o[3 * k] = ro.x; o[3 * k + 1] = ro.y; o[3 * k + 2] = ro.z;
            d[3 * k] = rd.x; d[3 * k + 1] = rd.y; d[3 * k + 2] = rd.z;
        }
}

void orc_render(const orc_scene* s, const rtgpu_camera* cam, const rtgpu_render_settings* st, float* rgb, rtgpu_render_stats* stats, int n_threads)
{
    // Renderer::render + subdivideTilePerCores (renderer/Renderer.cpp:273-334) and Task::runTasks
    // (utility/Task.cpp:24-53): a (3*threads)^2 grid of tiles pulled by std::threads.
    const int tw = st->x_end - st->x_start, th = st->y_end - st->y_start;
    if (tw <= 0 || th <= 0) return;
    const int taskCount = std::max(1, std::min({n_threads * 3, tw, th}));
    const int subW = tw / taskCount, subH = th / taskCount;
    struct T { int xs, ys, xe, ye; };
    std::vector<T> tiles; // Tile::subdivide, film/Tile.cpp:30-53
    for (int y = st->y_start; y < st->y_end; y += subH)
        for (int x = st->x_start; x < st->x_end; x += subW)
            tiles.push_back(T{x, y, std::min(st->x_end, x + subW), std::min(st->y_end, y + subH)});
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> primary{0}, extension{0}, shadow{0};
    auto worker = [&] {
        RenderCtx R{*s, stats};
        for (;;) {
            size_t ti = next.fetch_add(1);
            if (ti >= tiles.size()) break;
            const T& t = tiles[ti];
            for (int y = t.ys; y < t.ye; ++y)
                for (int x = t.xs; x < t.xe; ++x) {
                    RGBf p = render_pixel(R, *cam, *st, x, y);
                    float* dst = rgb + ((size_t)y * st->width + x) * 3;
                    dst[0] = p.r; dst[1] = p.g; dst[2] = p.b;
                }
        }
        primary += R.primary; extension += R.extension; shadow += R.shadow;
    };
    std::vector<std::thread> ths;
    for (int i = 0; i < std::max(1, n_threads); ++i) ths.emplace_back(worker);
    for (auto& t : ths) t.join();
    if (stats) {
        stats->primary_rays = primary; stats->extension_rays = extension; stats->shadow_rays = shadow;
        stats->device_ms = 0;
    }
}


// PhotonMapBuilder::buildPhotonMap, photonmapping/PhotonMapBuilder.cpp:15-41.  Photons are kept in emission order.
uint64_t orc_build_photon_map(orc_scene* s, int32_t mode, int32_t depth, uint64_t rays_per_point_light, uint64_t rays_per_area_light, uint32_t seed, int n_threads)
{
    s->photons.clear();
    s->pmMode = mode;
    s->pmDepth = depth;
    if (mode != RTGPU_PM_NONE) {
        std::vector<PhotonBatch> batches = photon_batches(*s, rays_per_point_light);
        std::vector<std::vector<rtgpu_photon>> parts(batches.size());
        parallel_for(batches.size(), n_threads, [&](uint64_t b0, uint64_t b1) {
            for (uint64_t b = b0; b < b1; ++b)
                for (uint64_t k = 0; k < batches[b].end - batches[b].start; ++k)
                    emit_and_trace(*s, batches[b], k, mode, rays_per_point_light, rays_per_area_light, seed, parts[b]);
        });
        for (auto& p : parts) s->photons.insert(s->photons.end(), p.begin(), p.end());
    }
    kd_build_all(*s);
    return s->photons.size();
}
uint64_t orc_get_photons(const orc_scene* s, rtgpu_photon* out, uint64_t capacity)
{
    const uint64_t n = std::min<uint64_t>(capacity, s->photons.size());
    if (out && n) std::memcpy(out, s->photons.data(), n * sizeof(rtgpu_photon));
    return s->photons.size();
}
void orc_set_photons(orc_scene* s, const rtgpu_photon* photons, uint64_t n, int32_t mode, int32_t depth)
{
    s->photons.assign(photons, photons + n);
    s->pmMode = mode;
    s->pmDepth = depth;
    kd_build_all(*s);
}
void orc_gather_photons(const orc_scene* s, const float* points, const float* dirs, uint64_t n, rtgpu_photon_gather* out, int n_threads)
{
    parallel_for(n, n_threads, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) out[i] = gather_photons(*s, mk(points + 3 * i), mk(dirs + 3 * i));
    });
}

float orc_triangle_area(const float a[3], const float b[3], const float c[3]) { return tri_area(mk(a), mk(b), mk(c)); }
float orc_fresnel_mix_factor(const float normal[3], const float dir[3], float ior) { return fresnel_mix_factor(mk(normal), mk(dir), ior); }

void orc_aabb_of_transformed(const float lo[3], const float hi[3], const double m[16], float out_lo[3], float out_hi[3])
{
    AABB r = aabb_of_transformed(AABB{mk(lo), mk(hi)}, m);
    out_lo[0] = r.s.x; out_lo[1] = r.s.y; out_lo[2] = r.s.z;
    out_hi[0] = r.e.x; out_hi[1] = r.e.y; out_hi[2] = r.e.z;
}
void orc_aabb_merge(const float lo0[3], const float hi0[3], const float lo1[3], const float hi1[3], float out_lo[3], float out_hi[3])
{
    AABB r = aabb_merge(AABB{mk(lo0), mk(hi0)}, AABB{mk(lo1), mk(hi1)});
    out_lo[0] = r.s.x; out_lo[1] = r.s.y; out_lo[2] = r.s.z;
    out_hi[0] = r.e.x; out_hi[1] = r.e.y; out_hi[2] = r.e.z;
}
double orc_aabb_area(const float lo[3], const float hi[3]) { return aabb_area(AABB{mk(lo), mk(hi)}); }
void orc_aabb_corners(const float lo[3], const float hi[3], float out[24])
{
    V3 c[8];
    aabb_corners(AABB{mk(lo), mk(hi)}, c);
    for (int i = 0; i < 8; ++i) { out[3 * i] = c[i].x; out[3 * i + 1] = c[i].y; out[3 * i + 2] = c[i].z; }
}

// KDTree::getElementsNearestTo (photonmapping/KDTree.h:380-409) on a tree given node by node, the way the reference's
// tests build theirs (test/KDTree.cpp:20-163: KDTreeNode::setChild / setAxis): node i holds point i, children are node
// indices or -1, the root is node 0.  Every element passes the filter (the three-argument overload, KDTree.h:405-409).
uint32_t orc_kd_nearest_explicit(const float* points, const int32_t* child0, const int32_t* child1, const int32_t* axis,
                                 uint32_t n_nodes, const float target[3], uint32_t count, float max_radius,
                                 int64_t* out_results, float* out_max_dist)
{
    orc_scene S;
    S.photons.resize(n_nodes);
    S.kd.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        rtgpu_photon ph{};
        std::memcpy(ph.pos, points + 3 * (size_t)i, 12);
        S.photons[i] = ph;   // normal 0: dot(dir, normal) >= 0 holds for every direction
        S.kd[i] = orc_scene::KdNode{{child0[i], child1[i]}, i, axis[i]};
    }
    S.kdRoot = n_nodes ? 0 : -1;
    float searchRadius = max_radius;
    unsigned found = 0;
    Gather g{S, mk(target), V3{0, 0, 0}, count, std::vector<int64_t>(count, -1)};
    if (S.kdRoot >= 0) {
        const float big = std::numeric_limits<float>::max();
        g.visit(S.kdRoot, true, searchRadius, found, V3{-big, -big, -big}, V3{big, big, big});
    }
    for (uint32_t i = 0; i < count; ++i) out_results[i] = g.results[i];
    *out_max_dist = searchRadius;
    return found;
}
// KDTreeBuilder::build over `n` points (photonmapping/KDTreeBuilder.h:258-276), then the same query: results are point indices.
uint32_t orc_kd_nearest_built(const float* points, uint32_t n, const float target[3], uint32_t count, float max_radius,
                              int64_t* out_results, float* out_max_dist)
{
    orc_scene S;
    S.photons.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        rtgpu_photon ph{};
        std::memcpy(ph.pos, points + 3 * (size_t)i, 12);
        S.photons[i] = ph;
    }
    kd_build_all(S);
    float searchRadius = max_radius;
    unsigned found = 0;
    Gather g{S, mk(target), V3{0, 0, 0}, count, std::vector<int64_t>(count, -1)};
    if (S.kdRoot >= 0) {
        const float big = std::numeric_limits<float>::max();
        g.visit(S.kdRoot, true, searchRadius, found, V3{-big, -big, -big}, V3{big, big, big});
    }
    for (uint32_t i = 0; i < count; ++i) out_results[i] = g.results[i];
    *out_max_dist = searchRadius;
    return found;
}

// Transformation::rotateQuaternion, math/Transformation.cpp:104-117 (inverse = transpose for a rotation)
void orc_rotate_quaternion(double i, double j, double k, double r, double out_m[16], double out_inv[16])
{
    double s = 1.0 / ((r * r + i * i) + (j * j + k * k));
    double m[16] = {
        1 - 2 * s * ((j * j) + (k * k)), 2 * s * (i * j - k * r), 2 * s * (i * k + j * r), 0,
        2 * s * (i * j + k * r), 1.0 - 2 * s * ((i * i) + (k * k)), 2 * s * (j * k - i * r), 0,
        2 * s * (i * k - j * r), 2 * s * (j * k + i * r), 1 - 2 * s * ((i * i) + (j * j)), 0,
        0, 0, 0, 1};
    std::memcpy(out_m, m, sizeof(m));
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) out_inv[a * 4 + b] = m[b * 4 + a];
}
void orc_transform_vector(const double m[16], const float v[3], float out[3])
{
    V3 r = xf_vector(m, mk(v));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_transform_point(const double m[16], const float p[3], float out[3])
{
    V3 r = xf_point(m, mk(p));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_triangle_intersect(const float o[3], const float d[3], const float a[3], const float b[3], const float c[3], float* beta, float* gamma, float* t)
{
    double td;
    float be = 0, ga = 0;
    bool ok = tri_intersect(mk(o), mk(d), mk(a), mk(b), mk(c), be, ga, td);
    if (ok) { *beta = be; *gamma = ga; *t = (float)td; }
    return ok ? 1 : 0;
}
int orc_sphere_intersect(const float o[3], const float d[3], double* t) { return sphere_intersect(mk(o), mk(d), *t) ? 1 : 0; }
float orc_rand_unit(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t vertex, uint32_t dim) { return rand_unit(seed, pixel, sample, vertex, dim); }

// PerspectiveCamera ctor + setTransform, camera/PerspectiveCamera.cpp:7-19
void orc_camera_from_transform(double fov_x_degrees, const double transform[16], rtgpu_camera* out)
{
    out->width = 2.0 * std::tan(0.5 * (fov_x_degrees * PI / 180.0));
    V3 origin = xf_point(transform, V3{0, 0, 0});
    V3 lookat = xf_vector(transform, V3{0, 0, -1});
    V3 up = xf_vector(transform, V3{0, 1, 0});
    Basis b = basis_from2(-lookat, up);
    out->origin[0] = origin.x; out->origin[1] = origin.y; out->origin[2] = origin.z;
    out->u[0] = b.u.x; out->u[1] = b.u.y; out->u[2] = b.u.z;
    out->v[0] = b.v.x; out->v[1] = b.v.y; out->v[2] = b.v.z;
    out->w[0] = b.w.x; out->w[1] = b.w.y; out->w[2] = b.w.z;
    out->aperture = 0;
    out->focal_distance = 0;
}

} // extern "C"
