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
    struct Range { uint32_t lo, hi; float plane2; uint32_t depth; };
    Range stack[40];
    int sp = 0;
    if (sv.nPhotons) { stack[0] = Range{0u, sv.nPhotons, 0.0f, 0u}; sp = 1; }
    const float qa[3] = {q.x, q.y, q.z};
    while (sp > 0) {
        const Range rg = stack[--sp];
        if (rg.plane2 > r2) continue;   // distanceToSplitPlane <= searchRadius, photonmapping/KDTree.h:153
        const uint32_t mid = rg.lo + (rg.hi - rg.lo) / 2u;
        const float4 p = sv.kdPos[mid];
        const uint32_t axis = (rg.depth + 1u) % 3u;   // KDTreeBuilder::buildNode: the axis after the parent's, x before the root
        const float pa = axis == 0u ? p.x : (axis == 1u ? p.y : p.z);
        const float delta = qa[axis] - pa;
        const bool right = pa < qa[axis];             // targetContainingChildIdx, KDTree.h:122
        const uint32_t nlo = right ? mid + 1u : rg.lo, nhi = right ? rg.hi : mid;
        const uint32_t flo = right ? rg.lo : mid + 1u, fhi = right ? mid : rg.hi;
        if (fhi > flo && sp < 39) stack[sp++] = Range{flo, fhi, delta * delta, rg.depth + 1u};
        if (nhi > nlo && sp < 39) stack[sp++] = Range{nlo, nhi, 0.0f, rg.depth + 1u};
        const F3 diff = f3(p.x, p.y, p.z) - q;
        const float d2 = dot3(diff, diff);
        if (d2 < r2) {
            const float4 n = sv.kdNormal[mid];
            if (dot3(dir, f3(n.x, n.y, n.z)) >= 0) {
                if (found < kGatherK) {
                    int c = found++;
                    while (c > 0 && hd[(c - 1) / 2] < d2) { hd[c] = hd[(c - 1) / 2]; hi_[c] = hi_[(c - 1) / 2]; c = (c - 1) / 2; }
                    hd[c] = d2; hi_[c] = mid;
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
                    hd[c] = d2; hi_[c] = mid;
                    r2 = hd[0];
                }
            }
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
