// rt_types.h -- layouts shared by the host builder, the C-ABI layer and the gfx950 kernels.
//
// HBM layout of a scene (all arrays hipMalloc'ed once per scene, read-only while tracing):
//
//   nodes  : NodeQ64[] every tree (one per mesh + the top-level tree over instances)
//            concatenated, each tree in depth-first order.  A node is one INNER node of a four-way
//            tree and carries the boxes of ALL its children (the reference also tests the child boxes at
//            the parent, shape/bvh/BVHNode.h:316-320), quantised to 8 bits per plane on the node's own
//            grid, so a traversal step is one 64-byte fetch (4 x dwordx4 from one aligned block).
//            Leaves have no record of their own: a child reference < 0 encodes (first primitive, count).
//            Node128 is the same node with fp32 boxes (the builder's intermediate form), Node64 the builder's
//            binary form (and the form of the reference-order top-level tree of scenes with spheres).
//   tris   : TriRec[]  48 bytes per triangle, in leaf order per mesh: vertex a, edges
//            e1 = a-b, e2 = a-c (the same fp32 subtractions Triangle::intersect performs per
//            test, math/Triangle.h:38-39) and the three cofactors of rows (e1,e2) shared by the
//            pDet and p0Det determinants.  Three aligned dwordx4 loads per test.
//   triIds : uint32[]  leaf slot -> the caller's triangle index (the reference's
//            permutation[triangleIndex], shape/TriangleMesh.cpp:33-36, 239-268).
//   inst   : DevInstance[] inverse transform rows in fp64 (math/Transformation.h:78-93), tree root,
//            triangle base, material, plus the fp32 data the shade stage needs.
#pragma once
#include <stdint.h>

namespace rt {

constexpr int kStackDepth = 96;      // per-lane traversal stack entries: the first kLdsStack in LDS, the rest in private memory
                                     // (a four-way node leaves up to 3 pending children per level)
constexpr int kLdsStack = 16;        // 16 x 64 lanes x 4 B = 4 KB of LDS per wave (does not limit occupancy); deeper entries are rare
constexpr int kMaxTreeDepth = 30;    // builder guarantee per tree (top tree + mesh tree share the stack)
constexpr int kMaxLeafHard = 16;     // leaf size field is 4 bits
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

struct Node64 {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t c0, c1;      // >= 0: inner node index (relative to the tree's first node); < 0: leaf ~c = first << 4 | (count-1)
    uint32_t axis;       // split axis (diagnostics)
    uint32_t pad;
};
static_assert(sizeof(Node64) == 64, "Node64 must be one 64-byte block");

struct TriRec {
    float ax, ay, az, c0;
    float e1x, e1y, e1z, c1;
    float e2x, e2y, e2z, c2;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

// The trees the kernels walk are four-way: a Node128 is one inner node holding the boxes of its (up to four) children, made by
// merging every inner node of the builder's binary tree with its two children (the children's children become the node's
// children; a child that is a leaf stays one).  Half as many dependent fetches per ray as the binary form, and about a
// third of the bookkeeping (stealing, step scheduling) per box tested.  Two 64-byte lines, eight dwordx4 loads.
constexpr int32_t kEmptyChild = 0x7FFFFFFE;   // unused child slot of a Node128 (its box is never tested)
struct Node128 {
    float box[4][6];     // child k: lo.x, hi.x, lo.y, hi.y, lo.z, hi.z (plane pairs side by side for the packed fp32 instructions)
    int32_t c[4];        // >= 0: inner node index (relative to the tree's first node); < 0: leaf ~c = first << 4 | (count-1)
    uint32_t pad[4];
};
static_assert(sizeof(Node128) == 128, "Node128 must be two 64-byte blocks");

// What the kernels fetch: the same four-way node with its child boxes quantised to 8 bits per plane on a grid laid over the
// node's own box (origin = the box's low corner, one step = 1/255 of its extent per axis, rounded outwards, so a decoded
// child box always contains the exact one).  One 64-byte line, four dwordx4 loads instead of seven: the traversal kernels
// keep a compute unit's vector-memory path busy 75-90 % of the time, and the tree takes half the cache.
// Plane words: byte 0 / 1 = low / high plane of the first child of the pair, byte 2 / 3 = of the second.
struct NodeQ64 {
    float origin[3];
    float sx;            // grid step per axis
    float sy, sz;
    uint32_t x01, x23;   // x planes of children 0, 1 / 2, 3
    uint32_t y01, y23, z01, z23;
    int32_t c[4];        // as Node128::c
};
static_assert(sizeof(NodeQ64) == 64, "NodeQ64 must be one 64-byte block");

inline int32_t encode_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 4) | (count - 1)); }
// top-level leaves (instances): up to 256 instances per leaf, first < 2^23
constexpr uint32_t kMaxTopLeaf = 256;
constexpr int kTopFramesHost = 48;   // per-lane frame stack of the reference-order top-level walk
inline int32_t encode_top_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 8) | (count - 1)); }

struct DevInstance {
    double inv[12];      // rows 0..2 of the world->object 4x4 (row 3 must be 0 0 0 1)
    uint32_t root;       // first node of the mesh tree (index into nodes), unused for spheres
    uint32_t triBase;    // first TriRec of the mesh
    int32_t shape;       // mesh index or -1 (sphere)
    uint32_t material;
    float wlo[3], whi[3]; // world AABB (InstancedModelList::getAABB)
    uint32_t meshVertBase; // first vertex-normal triple record of the mesh (shade stage)
    uint32_t keyBase;    // triangles of all instances before this one (tie-break key of the cooperative instanced traversal)
};

} // namespace rt
