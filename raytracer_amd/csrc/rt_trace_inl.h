// rt_trace_inl.h -- device functions of the traversal / intersection path (gfx950, wave64).
//
// One lane owns one ray.  The per-lane traversal stack lives in LDS, laid out
// stack[entry][lane] so that a wave's push/pop touches 64 consecutive dwords (no bank
// conflicts); everything else is in VGPRs.  Build with -ffp-contract=off: the triangle test
// must round exactly like the reference's separate multiplies and subtractions.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device.h"

namespace rt {

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }
// Eigen's unrolled 3-vector reduction: x + (y + z)
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float norm3(F3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ F3 normalized3(F3 a)
{
    float z = dot3(a, a);
    if (z > 0.0f) return a / sqrtf(z);
    return a;
}

// std::max / std::min as the reference's slab test uses them (NaN-order sensitive)
__device__ __forceinline__ float max_ref(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float min_ref(float a, float b) { return (b < a) ? b : a; }

// AABB::getIntersections (shape/AABB.h:67-121) with 1/d hoisted out of the per-box work (the
// reference divides per box; the quotient is the same value).  Returns hit and the clamped entry.
// Exact restatement: used where box tests decide results (top-level trees with the reference's
// non-conservative sphere boxes).
__device__ __forceinline__ bool slab_test(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                          F3 o, F3 inv, float& t0)
{
    float tx1 = (lox - o.x) * inv.x, tx2 = (hix - o.x) * inv.x;
    float ty1 = (loy - o.y) * inv.y, ty2 = (hiy - o.y) * inv.y;
    float tz1 = (loz - o.z) * inv.z, tz2 = (hiz - o.z) * inv.z;
    float txmin = inv.x >= 0 ? tx1 : tx2, txmax = inv.x >= 0 ? tx2 : tx1;
    float tymin = inv.y >= 0 ? ty1 : ty2, tymax = inv.y >= 0 ? ty2 : ty1;
    float tzmin = inv.z >= 0 ? tz1 : tz2, tzmax = inv.z >= 0 ? tz2 : tz1;
    float a = max_ref(max_ref(txmin, tymin), tzmin);
    float b = min_ref(min_ref(txmax, tymax), tzmax);
    t0 = max_ref(0.0f, a);
    return a <= b;
}

// Box tests in trees whose boxes are conservative (every mesh tree: triangle boxes are padded by 1e-4,
// shape/TriangleMesh.cpp:177-191; and the library's own top-level tree over such meshes) only prune: the closest hit is
// the same whatever tree and whatever box arithmetic is used, as long as no box containing a hit is rejected.  They are
// therefore written for speed (slab_quantised below): v_min / v_max, which drop a NaN operand (a 0 * inf slab then counts
// as passed, never as a rejection), and -- unlike the reference, which accepts boxes that lie entirely behind the ray
// origin -- boxes whose exit distance is negative are skipped: they cannot contain a hit with t >= 0.
typedef float Pair __attribute__((ext_vector_type(2)));

// Triangle::intersect (math/Triangle.h:34-101) on a TriRec.  c0..c2 are the cofactors
// (e1.y*e2.z-e1.z*e2.y), (e1.x*e2.z-e1.z*e2.x), (e1.x*e2.y-e1.y*e2.x) that both the pDet and
// the p0Det expansion use; precomputing them performs the same fp32 operations once.
__device__ __forceinline__ bool tri_test(const float4 r0, const float4 r1, const float4 r2, F3 o, F3 d,
                                         float& t, float& beta, float& gamma)
{
    const float c0 = r0.w, c1 = r1.w, c2 = r2.w;
    float pDet = (d.x * c0 - d.y * c1) + d.z * c2;
    if (pDet == 0.0f) return false;
    float invDet = 1.0f / pDet;
    float xx = r0.x - o.x, xy = r0.y - o.y, xz = r0.z - o.z;
    float p0Det = (xx * c0 - xy * c1) + xz * c2;
    t = p0Det * invDet;
    if (t < 0.0f) return false;
    // det [d; x; e2]
    float p1Det = (d.x * (xy * r2.z - xz * r2.y) - d.y * (xx * r2.z - xz * r2.x)) + d.z * (xx * r2.y - xy * r2.x);
    beta = p1Det * invDet;
    if (beta < 0.0f) return false;
    // det [d; e1; x]
    float p2Det = (d.x * (r1.y * xz - r1.z * xy) - d.y * (r1.x * xz - r1.z * xx)) + d.z * (r1.x * xy - r1.y * xx);
    gamma = p2Det * invDet;
    if (gamma < 0.0f) return false;
    if (gamma + beta > 1.0f) return false;
    return true;
}

// The same test as straight-line code for the cooperative leaf step, where every lane of the wave tests a triangle at once
// and a lane that fails early would only wait for the others: same operations in the same order, the rejections
// (written as the negations of the comparisons above, so that NaNs decide the same way) combined at the end.
__device__ __forceinline__ bool tri_test_flat(const float4 r0, const float4 r1, const float4 r2, F3 o, F3 d, float& t, float& betaOut, float& gammaOut)
{
    const float c0 = r0.w, c1 = r1.w, c2 = r2.w;
    const float pDet = (d.x * c0 - d.y * c1) + d.z * c2;
    const float invDet = 1.0f / pDet;
    const float xx = r0.x - o.x, xy = r0.y - o.y, xz = r0.z - o.z;
    const float p0Det = (xx * c0 - xy * c1) + xz * c2;
    t = p0Det * invDet;
    const float p1Det = (d.x * (xy * r2.z - xz * r2.y) - d.y * (xx * r2.z - xz * r2.x)) + d.z * (xx * r2.y - xy * r2.x);
    const float beta = p1Det * invDet;
    const float p2Det = (d.x * (r1.y * xz - r1.z * xy) - d.y * (r1.x * xz - r1.z * xx)) + d.z * (r1.x * xy - r1.y * xx);
    const float gamma = p2Det * invDet;
    betaOut = beta;
    gammaOut = gamma;
    return !(pDet == 0.0f) & !(t < 0.0f) & !(beta < 0.0f) & !(gamma < 0.0f) & !(gamma + beta > 1.0f);
}

// Sphere::intersect (shape/Sphere.cpp:16-59): unit sphere, fp64 quadratic, and the reference's
// largestT = min(t1, t0) (a ray starting inside the sphere misses).
__device__ __forceinline__ bool sphere_test(F3 o, F3 dir, double& t)
{
    double a = dot3(dir, dir);
    double b = 2.0 * (double)dot3(dir, o);
    double c = (double)dot3(o, o) - 1.0;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0) return false;
    double dr = sqrt(disc);
    double q = -0.5 * (b < 0 ? (b - dr) : (b + dr));
    double t0 = q / a;
    double t1 = c / q;
    double smallest = fmin(t0, t1); // both operands ordered: same as std::min
    if (smallest >= 0) { t = smallest; return true; }
    return false;
}

// Transformation::transformInverse(Ray) (math/Transformation.h:78-93,108-113) for an affine
// inverse: fp64 row sums in the order ((m0*x + m1*y) + m2*z) + m3*w, narrowed to fp32.
__device__ __forceinline__ F3 xf_point_inv(const double* m, F3 p)
{
    double x = p.x, y = p.y, z = p.z;
    return F3{(float)(((m[0] * x + m[1] * y) + m[2] * z) + m[3]),
              (float)(((m[4] * x + m[5] * y) + m[6] * z) + m[7]),
              (float)(((m[8] * x + m[9] * y) + m[10] * z) + m[11])};
}
__device__ __forceinline__ F3 xf_vector_inv(const double* m, F3 p)
{
    double x = p.x, y = p.y, z = p.z;
    return F3{(float)(((m[0] * x + m[1] * y) + m[2] * z) + m[3] * 0.0),
              (float)(((m[4] * x + m[5] * y) + m[6] * z) + m[7] * 0.0),
              (float)(((m[8] * x + m[9] * y) + m[10] * z) + m[11] * 0.0)};
}

// Both at once for a single-mesh scene.  With an identity transform the fp64 row sums reduce to
// x + 0 per component (which also turns -0 into +0, like the sums do) as long as every component
// is finite (0 * inf would make NaNs in the general form), so that case skips the fp64 arithmetic.
__device__ __forceinline__ void xf_ray_inv(const double* m, bool identity, F3 wo, F3 wd, F3& o, F3& d)
{
    const float s = (wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z);
    if (identity && (s - s) == 0.0f) {
        o = f3(wo.x + 0.0f, wo.y + 0.0f, wo.z + 0.0f);
        d = f3(wd.x + 0.0f, wd.y + 0.0f, wd.z + 0.0f);
    } else {
        o = xf_point_inv(m, wo);
        d = xf_vector_inv(m, wd);
    }
}

// Node / triangle records are addressed with 32-bit byte offsets from the array base (one scalar
// base + one vector offset per fetch instead of 64-bit vector address arithmetic); scene creation
// checks that both arrays are smaller than 4 GiB.
__device__ __forceinline__ const float4* node_ptr(const SceneView& sv, uint32_t node)
{
    return (const float4*)((const char*)sv.nodes + (node << 6));
}

// Whether a ray that starts at `o` (in the tree's space) may be so far from the tree rooted at `root` that the decode's
// rounding needs the widened form (plane_offsets): farther than 256 units from the farthest plane of the root's grid.
__device__ __forceinline__ bool far_from_tree(const float4* root, F3 o)
{
    const float4 q0 = root[0], q1 = root[1];
    const float hx = fmaf(255.0f, q0.w, q0.x), hy = fmaf(255.0f, q1.x, q0.y), hz = fmaf(255.0f, q1.y, q0.z);
    const float dx = fmaxf(fabsf(o.x - q0.x), fabsf(o.x - hx)), dy = fmaxf(fabsf(o.y - q0.y), fabsf(o.y - hy)), dz = fmaxf(fabsf(o.z - q0.z), fabsf(o.z - hz));
    return !(fmaxf(fmaxf(dx, dy), dz) < 256.0f);   // (a NaN distance counts as far)
}

// The slab test on a child box given by its quantised planes (NodeQ64).  A plane's distance along the ray is
//   (origin + q * step - o) / d  =  q * (step / d) + (origin - o) / d  =  fma(q, A, B)
// with A and B formed once per node and axis (node4_hits) and shared by the eight planes of that axis: one packed fma per
// child and axis for both planes.  The plane words arrive with the plane the ray meets first in the lower byte of each pair
// (node4_hits swaps the bytes of an axis the ray runs down), so the entry and exit distances need no min / max per axis.
// (near, far) = bytes `shift` and `shift + 8` of the words; bx.x / bx.y = B for the near / far plane.
__device__ __forceinline__ bool slab_quantised(uint32_t wx, uint32_t wy, uint32_t wz, int shift, Pair ax, Pair ay, Pair az,
                                               Pair bx, Pair by, Pair bz, float tlimit, float& t0)
{
    const Pair qx = {(float)((wx >> shift) & 0xFFu), (float)((wx >> (shift + 8)) & 0xFFu)};
    const Pair qy = {(float)((wy >> shift) & 0xFFu), (float)((wy >> (shift + 8)) & 0xFFu)};
    const Pair qz = {(float)((wz >> shift) & 0xFFu), (float)((wz >> (shift + 8)) & 0xFFu)};
    const Pair tx = __builtin_elementwise_fma(qx, ax, bx);
    const Pair ty = __builtin_elementwise_fma(qy, ay, by);
    const Pair tz = __builtin_elementwise_fma(qz, az, bz);
    // v_max3 / v_min3 drop a NaN operand (0 * inf, inf - inf: the ray runs along the axis' planes): that slab counts as passed
    const float tnear = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0f);
    const float tfar = fminf(fminf(fminf(tx.y, ty.y), tz.y), tlimit);
    t0 = tnear;
    return tnear <= tfar;
}
// byte selector for v_perm_b32 that keeps (d >= 0) or swaps (d < 0) the two bytes of each pair of a plane word
__device__ __forceinline__ uint32_t near_first(uint32_t word, float invd)
{
    const uint32_t sel = 0x03020100u ^ ((uint32_t)(__float_as_int(invd) >> 31) & 0x01010101u);
    return __builtin_amdgcn_perm(word, word, sel);
}
// B = (grid origin - ray origin) / d for the near and the far planes of an axis.  Rounding puts a decoded distance off the
// exact one by up to 2^-22 of B (the subtraction, the two products and the fma), which is nothing next to the 1e-4 the
// reference pads every triangle's box with (shape/TriangleMesh.cpp:177-191) as long as the node is nearer than a few hundred
// units; WIDE (chosen per wave: some ray of the wave may be farther than that from its tree, or the tree's boxes have less
// room) moves the near distances down and the far distances up by 2^-21 of |B|, so that the decoded box contains the exact
// one at any distance.
template <bool WIDE>
__device__ __forceinline__ Pair plane_offsets(float b)
{
    if constexpr (!WIDE) return Pair{b, b};
    constexpr float k = 4.76837158203125e-07f;   // 2^-21
    return Pair{__builtin_fmaf(-fabsf(b), k, b), __builtin_fmaf(fabsf(b), k, b)};
}

// One step at an inner node of a four-way tree (NodeQ64): the boxes of its children against the ray, and the children that
// are hit ordered by entry distance.  c[0 .. n-1] = their references, nearest first; returns n.  `tested` = boxes tested
// (the node's real children).  Child boxes only prune (see above), so the order is a heuristic, not part of the result.
// ORDERED = false (any-hit queries, where the first hit found ends the search and no child is nearer to the answer than
// another): the children that are hit come first, in the node's own order -- the exchanges then move references only and
// their conditions are mask arithmetic, about 20 vector instructions fewer per step.
template <bool ORDERED = true, bool WIDE = true>
__device__ __forceinline__ int node4_hits(const float4* np, F3 o, F3 inv, float tlim, int c[4], uint32_t& tested)
{
    float t0, t1, t2, t3;
    const float4 q0 = np[0], q1 = np[1], q2 = np[2], q6 = np[3];
    int c0 = __float_as_int(q6.x), c1 = __float_as_int(q6.y), c2 = __float_as_int(q6.z), c3 = __float_as_int(q6.w);
    const float ivx = inv.x, ivy = inv.y, ivz = inv.z;   // (as scalars: a pair taken from two fields of the struct goes through private memory)
    const Pair bx = plane_offsets<WIDE>((q0.x - o.x) * ivx), by = plane_offsets<WIDE>((q0.y - o.y) * ivy), bz = plane_offsets<WIDE>((q0.z - o.z) * ivz);
    const float sx = q0.w * ivx, sy = q1.x * ivy, sz = q1.y * ivz;
    const Pair ax = {sx, sx}, ay = {sy, sy}, az = {sz, sz};
    const uint32_t x01 = near_first(__float_as_uint(q1.z), ivx), x23 = near_first(__float_as_uint(q1.w), ivx);
    const uint32_t y01 = near_first(__float_as_uint(q2.x), ivy), y23 = near_first(__float_as_uint(q2.y), ivy);
    const uint32_t z01 = near_first(__float_as_uint(q2.z), ivz), z23 = near_first(__float_as_uint(q2.w), ivz);
    const bool h0 = slab_quantised(x01, y01, z01, 0, ax, ay, az, bx, by, bz, tlim, t0);                      // the first child always exists
    const bool h1 = slab_quantised(x01, y01, z01, 16, ax, ay, az, bx, by, bz, tlim, t1) && c1 != kEmptyChild;
    const bool h2 = slab_quantised(x23, y23, z23, 0, ax, ay, az, bx, by, bz, tlim, t2) && c2 != kEmptyChild;
    const bool h3 = slab_quantised(x23, y23, z23, 16, ax, ay, az, bx, by, bz, tlim, t3) && c3 != kEmptyChild;
    tested = 1u + (c1 != kEmptyChild ? 1u : 0u) + (c2 != kEmptyChild ? 1u : 0u) + (c3 != kEmptyChild ? 1u : 0u);
    const int nh = (h0 ? 1 : 0) + (h1 ? 1 : 0) + (h2 ? 1 : 0) + (h3 ? 1 : 0);
    if constexpr (ORDERED) {
        t0 = h0 ? t0 : INFINITY; t1 = h1 ? t1 : INFINITY; t2 = h2 ? t2 : INFINITY; t3 = h3 ? t3 : INFINITY;
        // children that are not hit sort behind the others: a five-exchange network on (entry distance, reference)
#define RT_CSWAP(ta, ca, tb, cb) { const bool sw = tb < ta; const float tl = sw ? tb : ta, th = sw ? ta : tb; const int cl = sw ? cb : ca, ch = sw ? ca : cb; ta = tl; tb = th; ca = cl; cb = ch; }
        RT_CSWAP(t0, c0, t1, c1)
        RT_CSWAP(t2, c2, t3, c3)
        RT_CSWAP(t0, c0, t2, c2)
        RT_CSWAP(t1, c1, t3, c3)
        RT_CSWAP(t1, c1, t2, c2)
#undef RT_CSWAP
    } else {
        // the same network on (hit or not, reference)
        bool g0 = h0, g1 = h1, g2 = h2, g3 = h3;
#define RT_HSWAP(ga, ca, gb, cb) { const bool sw = gb && !ga; const int cl = sw ? cb : ca, ch = sw ? ca : cb; const bool gl = ga || gb, gh = ga && gb; ca = cl; cb = ch; ga = gl; gb = gh; }
        RT_HSWAP(g0, c0, g1, c1)
        RT_HSWAP(g2, c2, g3, c3)
        RT_HSWAP(g0, c0, g2, c2)
        RT_HSWAP(g1, c1, g3, c3)
        RT_HSWAP(g1, c1, g2, c2)
#undef RT_HSWAP
    }
    c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
    return nh;
}
__device__ __forceinline__ const float4* tri_ptr(const SceneView& sv, uint32_t slot)
{
    return (const float4*)((const char*)sv.tris + slot * 48u);
}

struct TraceResult {
    float t;          // +inf on a miss
    uint32_t slot;    // global leaf slot of the triangle, kInvalid for spheres / miss
    uint32_t inst;    // instance index, kInvalid on a miss
    float beta, gamma;
    uint32_t prim;    // caller's triangle index; filled by trace_wave_queue() only
};

constexpr int kSentinel = 0x7FFFFFFF;

// Closest hit (ANY = false: Scene::traceRay -> BVHNode::traceRay, shape/bvh/BVHNode.h:306-397)
// or any hit within tmax (ANY = true: BVHNode::testVisibility, :399-446), through the top-level
// tree (INSTANCED) and the mesh trees.
//
// Result definition (tree independent): the primitive with the smallest t; equal t -> the
// smaller (instance, triangle index).  The reference's result for equal t depends on its
// (unstable-sort built) tree; with unequal t both agree because boxes are conservative.  A child
// is entered when its box is hit and its entry distance <= the best t so far.  An instance is
// tested when the ray hits the instance's own world box (the reference tests every instance of
// a top-level leaf whose merged box is hit).
template <bool INSTANCED, bool ANY, bool COUNT>
__device__ __forceinline__ TraceResult trace_ray(const SceneView& sv, F3 wo, F3 wd, float tmax,
                                                 int (*stack)[64], const int lane, uint32_t& nBoxes, uint32_t& nTris,
                                                 const uint32_t soloInst = 0)
{
    TraceResult best;
    best.t = ANY ? tmax : INFINITY;
    best.slot = kInvalid;
    best.inst = kInvalid; // kInvalid = nothing found yet
    best.beta = 0.0f;
    best.gamma = 0.0f;

    // A ray with a NaN component finds nothing in the reference (every box comparison is false);
    // the pruning-only slab test below would instead let it through, so it is answered here.
    if (!((wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z) == (wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z))) {
        if (!ANY) best.t = INFINITY;
        return best;
    }

    F3 o = wo, d = wd;
    uint32_t nodeBase, triBase = 0, curInst = 0;
    int cur;
    int sp = 0;
    bool inMesh;

    if (INSTANCED) {
        nodeBase = sv.topRoot;
        // a single instance is the root leaf: entered without any box test, like the reference's root
        cur = sv.nInst > 1 ? 0 : ~0;
        inMesh = false;
    } else {
        const DevInstance& in = sv.inst[soloInst];
        xf_ray_inv(in.inv, sv.identityXf != 0, wo, wd, o, d);
        nodeBase = in.root;
        triBase = in.triBase;
        cur = 0;
        inMesh = true;
    }
    F3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const F3 winv = inv;

    // The stack's first kLdsStack entries live in LDS (stack[entry][lane]: conflict-free), deeper
    // ones -- rarely reached -- in per-lane private memory.
    int spill[kStackDepth - kLdsStack];
    auto push = [&](int v) {
        if (sp < kLdsStack) stack[sp][lane] = v;
        else spill[sp - kLdsStack] = v;
        ++sp;
    };
    // Pops the next pending subtree; leaves a mesh tree when its sentinel surfaces.  False = done.
    auto pop = [&]() -> bool {
        for (;;) {
            if (sp == 0) return false;
            --sp;
            cur = stack[sp < kLdsStack ? sp : 0][lane];            // always an LDS read ...
            asm volatile("" : "+v"(cur));                          // (keeps the two loads apart: merged, they become one slow generic-address load)
            if (sp >= kLdsStack) cur = spill[sp - kLdsStack];      // ... replaced by the private entry when that deep
            if (INSTANCED && cur == kSentinel) {
                inMesh = false;
                o = wo; d = wd; inv = winv;
                nodeBase = sv.topRoot;
                continue;
            }
            return true;
        }
    };

    bool alive = true;
    while (alive) {
        // ---- inner nodes: one record holds the boxes of all (up to four) children ----
        while (cur >= 0) {
            int c[4];
            uint32_t tested;
            const int nh = node4_hits(node_ptr(sv, nodeBase + (uint32_t)cur), o, inv, best.t, c, tested);
            if (COUNT) nBoxes += tested;
            if (nh > 3) push(c[3]);
            if (nh > 2) push(c[2]);
            if (nh > 1) push(c[1]);
            if (nh > 0) {
                cur = c[0];
            } else {
#ifdef RT_DEV_STATS
                if (COUNT) nTris += 1u << 16;   // developer build: node steps that entered no child
#endif
                if (!pop()) {
                    alive = false;
                    break;
                }
            }
        }
        if (!alive) break;
        // ---- leaf -------------------------------------------------------------------------
        const uint32_t leaf = (uint32_t)(~cur);
        const uint32_t first = leaf >> 4, count = (leaf & 15u) + 1u;
        if (INSTANCED && !inMesh) {
            // top-level leaf = one instance (InstancedModelList::traceRay, shape/list/InstancedModelList.cpp:99-138)
            curInst = sv.instOrder[first];
            const DevInstance& in = sv.inst[curInst];
            o = xf_point_inv(in.inv, wo);
            d = xf_vector_inv(in.inv, wd);
            if (in.shape < 0) {
                double ts;
                if (COUNT) nTris += 1;
                if (sphere_test(o, d, ts)) {
                    if (ANY) {
                        if (ts <= (double)tmax) { best.inst = curInst; return best; }
                    } else {
                        const float tf = (float)ts;
                        if (best.inst == kInvalid || tf < best.t || (tf == best.t && curInst < best.inst)) {
                            best.t = tf; best.slot = kInvalid; best.inst = curInst; best.beta = 0.0f; best.gamma = 0.0f;
                        }
                    }
                }
                o = wo; d = wd;
                if (!pop()) break;
                continue;
            }
            inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            nodeBase = in.root;
            triBase = in.triBase;
            inMesh = true;
            push(kSentinel);
            cur = 0;
            continue;
        }
        // triangle leaf (TriangleMesh::intersect / testVisibility, shape/TriangleMesh.cpp:99-175)
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t slot = triBase + first + k;
            const float4* tp = tri_ptr(sv, slot);
            const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
            float t, be, ga;
            if (COUNT) nTris += 1;
            if (tri_test(r0, r1, r2, o, d, t, be, ga)) {
                const uint32_t ci = INSTANCED ? curInst : soloInst;
                if (ANY) {
                    if (t <= tmax) {
                        best.slot = slot; best.inst = ci; best.beta = be; best.gamma = ga;
                        return best;
                    }
                } else {
                    bool better = best.inst == kInvalid || t < best.t;
                    if (!better && t == best.t) {
                        // equal t: the smaller (instance, caller's triangle index) wins
                        if (ci < best.inst) better = true;
                        else if (ci == best.inst && best.slot != kInvalid && sv.triIds[slot] < sv.triIds[best.slot]) better = true;
                    }
                    if (better) {
                        best.t = t; best.slot = slot; best.inst = ci; best.beta = be; best.gamma = ga;
                    }
                }
            }
        }
        if (!pop()) break;
    }
    if (best.inst == kInvalid) best.t = INFINITY;
    return best;
}


// ---------------------------------------------------------------------------------------------
// Scenes with analytic spheres among several instances: top-level traversal that reproduces the
// reference's recursion exactly (BVHNode::traceRay, shape/bvh/BVHNode.h:306-397): both child
// boxes tested at the parent, children visited in the order given by the sign of
// d[childOrder axis], the second child skipped when the first child's hit is not farther than
// the second child's entry distance, no pruning by hits found in other subtrees, every instance
// of a visited leaf tested (InstancedModelList::traceRay, shape/list/InstancedModelList.cpp:99-138).
// Hit distances are compared in double like the reference's RayHitInfo::t.
// The frame stack is per-lane private memory (top-level trees are small and shallow); mesh
// trees below it are walked by trace_ray<false,...> with the LDS stack.
// ---------------------------------------------------------------------------------------------
constexpr int kTopFrames = kTopFramesHost;

// AABB::getIntersection (shape/AABB.h:123-131): clamped entry distance, or -1 on a miss.
__device__ __forceinline__ float box_entry(float lox, float loy, float loz, float hix, float hiy, float hiz, F3 o, F3 inv)
{
    float t0;
    return slab_test(lox, loy, loz, hix, hiy, hiz, o, inv, t0) ? t0 : -1.0f;
}

template <bool COUNT>
__device__ __forceinline__ TraceResult trace_ray_reftop_closest(const SceneView& sv, F3 wo, F3 wd, int (*stack)[64], const int lane,
                                                                uint32_t& nBoxes, uint32_t& nTris)
{
    TraceResult best;
    best.t = INFINITY; best.slot = kInvalid; best.inst = kInvalid; best.beta = 0.0f; best.gamma = 0.0f;
    double bestT = INFINITY;
    const F3 winv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);

    int frameRef[kTopFrames];     // pending second child, or kSentinel for a merge frame
    float frameTs[kTopFrames];    // entry distance of the pending child
    double frameT[kTopFrames];    // merge frame: result of the first child
    int fp = 0;
    double acc = INFINITY;        // result (t) of the subtree that has just completed
    int cur = sv.refTopRoot;

    for (;;) {
        // ---- descend until a subtree completes ------------------------------------------------
        for (;;) {
            if (cur >= 0) {
                const float4* np = sv.refNodes + ((size_t)(uint32_t)cur << 2);
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                if (COUNT) nBoxes += 2;
                const float tsA = box_entry(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, wo, winv);
                const float tsB = box_entry(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, wo, winv);
                const int ca = __float_as_int(q3.x), cb = __float_as_int(q3.y);
                const uint32_t axis = (uint32_t)__float_as_int(q3.z);
                const float order = axis == 0 ? wd.x : (axis == 1 ? wd.y : wd.z);
                int first, second;
                float tsF, tsS;
                if (order > 0) { first = ca; tsF = tsA; second = cb; tsS = tsB; }
                else           { first = cb; tsF = tsB; second = ca; tsS = tsA; }
                // order == 0: child 1 first as well, but without the early exit (flag in frameT)
                if (tsF >= 0) {
                    if (fp >= kTopFrames) return best; // cannot happen: depth is checked at scene creation
                    frameRef[fp] = second;
                    frameTs[fp] = tsS;
                    frameT[fp] = (order == 0) ? 1.0 : 0.0; // 1 = no early exit
                    ++fp;
                    acc = INFINITY;
                    cur = first;
                } else if (tsS >= 0) {
                    acc = INFINITY;
                    cur = second;
                } else {
                    acc = INFINITY;
                    break;
                }
            } else {
                // leaf: InstancedModelList::traceRay over [first, first+count)
                const uint32_t leaf = (uint32_t)(~cur);
                const uint32_t first = leaf >> 8, count = (leaf & 255u) + 1u;
                double leafT = INFINITY;
                TraceResult leafBest = best;
                bool leafHit = false;
                for (uint32_t k = 0; k < count; ++k) {
                    const uint32_t ii = sv.instOrder[first + k];
                    const DevInstance& in = sv.inst[ii];
                    if (in.shape < 0) {
                        const F3 o = xf_point_inv(in.inv, wo), d = xf_vector_inv(in.inv, wd);
                        double ts;
                        if (COUNT) nTris += 1;
                        if (sphere_test(o, d, ts) && (!leafHit || leafT > ts)) {
                            leafHit = true; leafT = ts;
                            leafBest.t = (float)ts; leafBest.slot = kInvalid; leafBest.inst = ii; leafBest.beta = 0.0f; leafBest.gamma = 0.0f;
                        }
                    } else {
                        const TraceResult r = trace_ray<false, false, COUNT>(sv, wo, wd, INFINITY, stack, lane, nBoxes, nTris, ii);
                        if (r.inst != kInvalid && (!leafHit || leafT > (double)r.t)) {
                            leafHit = true; leafT = (double)r.t;
                            leafBest = r;
                        }
                    }
                }
                acc = leafHit ? leafT : INFINITY;
                if (leafHit && (best.inst == kInvalid || bestT > leafT)) { bestT = leafT; best = leafBest; }
                break;
            }
        }
        // ---- unwind -------------------------------------------------------------------------------
        bool descend = false;
        while (fp > 0) {
            --fp;
            const int ref = frameRef[fp];
            if (ref == kSentinel) {
                // merge: result(N) = first child's hit unless the second child's is strictly closer
                const double tf = frameT[fp];
                if (!(tf > acc)) acc = tf;
                continue;
            }
            const float ts = frameTs[fp];
            const bool noEarlyExit = frameT[fp] != 0.0;
            const bool hitFirst = acc < INFINITY;
            if (ts >= 0 && (!hitFirst || noEarlyExit || acc > (double)ts)) {
                if (hitFirst) { frameRef[fp] = kSentinel; frameT[fp] = acc; ++fp; }
                acc = INFINITY;
                cur = ref;
                descend = true;
                break;
            }
            // second child not visited: N completes with the first child's result
        }
        if (!descend) break;
    }
    if (best.inst == kInvalid) best.t = INFINITY;
    return best;
}

// BVHNode::testVisibility (shape/bvh/BVHNode.h:399-446) + InstancedModelList::testVisibility
// (shape/list/InstancedModelList.cpp:205-233) on the reference-shaped top-level tree: a child is
// entered when its box is hit with entry <= maxT; the answer (anything hit within maxT) does not
// depend on the visiting order.
__device__ __forceinline__ bool trace_ray_reftop_any(const SceneView& sv, F3 wo, F3 wd, float tmax, int (*stack)[64], const int lane)
{
    const F3 winv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
    int pending[kTopFrames];
    int pp = 0;
    int cur = sv.refTopRoot;
    uint32_t nb = 0, nt = 0;
    for (;;) {
        if (cur >= 0) {
            const float4* np = sv.refNodes + ((size_t)(uint32_t)cur << 2);
            const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
            float t0a, t0b;
            const bool ha = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, wo, winv, t0a) && t0a <= tmax;
            const bool hb = slab_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, wo, winv, t0b) && t0b <= tmax;
            const int ca = __float_as_int(q3.x), cb = __float_as_int(q3.y);
            if (ha && hb) {
                if (pp >= kTopFrames) return false;
                pending[pp++] = cb;
                cur = ca;
                continue;
            }
            if (ha) { cur = ca; continue; }
            if (hb) { cur = cb; continue; }
        } else {
            const uint32_t leaf = (uint32_t)(~cur);
            const uint32_t first = leaf >> 8, count = (leaf & 255u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t ii = sv.instOrder[first + k];
                const DevInstance& in = sv.inst[ii];
                if (in.shape < 0) {
                    const F3 o = xf_point_inv(in.inv, wo), d = xf_vector_inv(in.inv, wd);
                    double ts;
                    if (sphere_test(o, d, ts) && ts <= (double)tmax) return true;
                } else {
                    const TraceResult r = trace_ray<false, true, false>(sv, wo, wd, tmax, stack, lane, nb, nt, ii);
                    if (r.inst != kInvalid) return true;
                }
            }
        }
        if (pp == 0) return false;
        cur = pending[--pp];
    }
}

} // namespace rt
