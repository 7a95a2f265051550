// rt_photon_inl.h -- the photon gather (device side), shared by the query kernel and the integrator.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

#include "rt_shade_inl.h"

namespace rt {

constexpr int kGatherK = 20;   // std::vector<const Photon*> photons(20), material/DiffuseMaterial.cpp:61

// The k nearest photons around `q` whose normal does not face away from `dir` (k = 20), as the radiance estimate
// of DiffuseMaterial.cpp:62-75 needs them: count, the reference's final search radius, energies summed nearest first.
__device__ inline void gather_photons(const SceneView& sv, F3 q, F3 dir, uint32_t& count, float& maxDist, RGBf& sum)
{
    float hd[kGatherK];       // max-heap on squared distance
    uint32_t hi_[kGatherK];
    int found = 0;
    float r2 = 1E9f * 1E9f;   // getElementsNearestTo(hitpoint, 20, 1E9, ...)
    // The walk follows KDTreeNode::getElementsNearestTo (photonmapping/KDTree.h:90-164): the child on the query's side
    // first, then the node's own photon, then the other child if the split plane is within the search radius -- near
    // photons fill the list first and the radius is small before the far sides are looked at.  A stack entry is the
    // far range of a node; popping it first tests that node's photon (its near subtree is finished by then).
    constexpr uint32_t kNoParent = 1u << 30, kRightSide = 1u << 31;
    struct Entry { uint32_t lo, hi; float plane2, d2; uint32_t dk; };   // d2: squared distance of the node's photon; dk: depth | flags
    Entry stack[40];
    int sp = 0;
    if (sv.nPhotons) { stack[0] = Entry{0u, sv.nPhotons, 0.0f, 0.0f, kNoParent}; sp = 1; }
    const float qa[3] = {q.x, q.y, q.z};
    const float dirLen = norm3(dir);
    // a photon at squared distance d2: the filter of DiffuseMaterial.cpp:66-68, then KDTreeNode::insert (KDTree.h:225-245)
    auto offer = [&](uint32_t node, float d2) {
        if (!(d2 < r2)) return;
        const float4 n = sv.kdNormal[node];
        if (!(dot3(dir, f3(n.x, n.y, n.z)) >= 0)) return;
        if (found < kGatherK) {
            int c = found++;
            while (c > 0 && hd[(c - 1) / 2] < d2) { hd[c] = hd[(c - 1) / 2]; hi_[c] = hi_[(c - 1) / 2]; c = (c - 1) / 2; }
            hd[c] = d2; hi_[c] = node;
            if (found == kGatherK) r2 = hd[0];
        } else {
            int c = 0;   // replace the farthest, sift down
            for (;;) {
                int l = 2 * c + 1, rr = l + 1, big = -1;
                float bigv = d2;
                if (l < kGatherK && hd[l] > bigv) { big = l; bigv = hd[l]; }
                if (rr < kGatherK && hd[rr] > bigv) { big = rr; }
                if (big < 0) break;
                hd[c] = hd[big]; hi_[c] = hi_[big]; c = big;
            }
            hd[c] = d2; hi_[c] = node;
            r2 = hd[0];
        }
    };
    while (sp > 0) {
        const Entry e = stack[--sp];
        if (!(e.dk & kNoParent)) offer((e.dk & kRightSide) ? e.lo - 1u : e.hi, e.d2);   // the node whose far range this is
        if (e.hi <= e.lo || e.plane2 > r2) continue;   // distanceToSplitPlane <= searchRadius, KDTree.h:153
        uint32_t lo = e.lo, hi = e.hi, depth = e.dk & 0xFFu;
        while (hi > lo && sp < 40) {                   // down the query's side, leaving the far ranges behind
            const uint32_t mid = lo + (hi - lo) / 2u;
            const float4 p = sv.kdPos[2u * mid], cone = sv.kdPos[2u * mid + 1u];
            // no normal in this range's cone can have a non-negative dot product with dir: nothing here passes the filter
            if (dot3(dir, f3(cone.x, cone.y, cone.z)) < cone.w * dirLen) break;
            const F3 diff = f3(p.x, p.y, p.z) - q;
            const float d2 = dot3(diff, diff);
            if (hi - lo == 1u) { offer(mid, d2); break; }   // a leaf: nothing below it
            const uint32_t axis = (depth + 1u) % 3u;   // KDTreeBuilder::buildNode: the axis after the parent's, x before the root
            const float pa = axis == 0u ? p.x : (axis == 1u ? p.y : p.z);
            const float delta = qa[axis] - pa;
            const bool right = pa < qa[axis];          // targetContainingChildIdx, KDTree.h:122
            depth += 1u;
            if (right) { stack[sp++] = Entry{lo, mid, delta * delta, d2, depth}; lo = mid + 1u; }
            else { stack[sp++] = Entry{mid + 1u, hi, delta * delta, d2, depth | kRightSide}; hi = mid; }
        }
    }
    count = (uint32_t)found;
    maxDist = found == kGatherK ? sqrtf(hd[0]) : (found > 0 ? FLT_MAX : 1E9f);   // KDTreeNode::insert returns FLT_MAX until the list is full
    // value += photons[i]->energy, nearest first
    for (int a = 1; a < found; ++a) {
        const float kd = hd[a]; const uint32_t ki = hi_[a];
        int b = a - 1;
        while (b >= 0 && hd[b] > kd) { hd[b + 1] = hd[b]; hi_[b + 1] = hi_[b]; --b; }
        hd[b + 1] = kd; hi_[b + 1] = ki;
    }
    sum = rgb(0, 0, 0);
    for (int a = 0; a < found; ++a) {
        const float4 e = sv.kdEnergy[hi_[a]];
        sum.r += e.x; sum.g += e.y; sum.b += e.z;
    }
}

} // namespace rt
