// rt_photon_inl.h -- the photon gather (device side), shared by the query kernel and the integrator.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

#include "rt_shade_inl.h"

namespace rt {

constexpr int kGatherK = 20;   // std::vector<const Photon*> photons(20), material/DiffuseMaterial.cpp:61

// The k nearest photons around `q` whose normal does not face away from `dir` (k = 20), as the radiance estimate
// of DiffuseMaterial.cpp:62-75 needs them: count, the reference's final search radius, energies summed nearest first.
//
// `budget` bounds the tree nodes one call may look at: a call that runs out returns false with nothing written and
// the caller hands the query to gather_photons_wave.  A few queries need 100x to 10 000x the nodes of the others (on
// faces whose neighbourhood's photons all fail the filter the 20 accepted photons are far away, and every photon
// within that distance has to be looked at); left in their lanes they would hold their waves for seconds.
constexpr uint32_t kGatherUnbounded = 0xFFFFFFFFu;

__device__ inline bool gather_photons(const SceneView& sv, F3 q, F3 dir, uint32_t budget, uint32_t& count, float& maxDist, RGBf& sum)
{
    float hd[kGatherK];       // max-heap on squared distance
    uint32_t hi_[kGatherK];
    int found = 0;
    float r2 = 1E9f * 1E9f;   // getElementsNearestTo(hitpoint, 20, 1E9, ...)
    // The walk follows KDTreeNode::getElementsNearestTo (photonmapping/KDTree.h:90-164): the child on the query's side
    // first, then the node's own photon, then the other child if the split plane is within the search radius -- near
    // photons fill the list first and the radius is small before the far sides are looked at.  A stack entry is the
    // far range of a node; popping it first tests that node's photon (its near subtree is finished by then).
    // A stack entry is 16 bytes (the stacks of a wave's lanes are what the kernel moves most through memory): lo and hi
    // are below 2^31 and carry depth % 3 in their top bits, the two squared distances are not negative and carry the
    // flags "no parent to test" and "the range lies right of its parent" in their sign bits.  The tree is at most 31
    // levels deep, so 32 entries are enough.
    constexpr uint32_t kTop = 0x80000000u;
    constexpr int kStack = 32;
    uint4 stack[kStack];
    int sp = 0;
    if (sv.nPhotons) { stack[0] = make_uint4(0u, sv.nPhotons, kTop, 0u); sp = 1; }   // depth 0, no parent
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
    uint32_t visited = 0;
    while (sp > 0) {
        if (visited > budget) return false;
        const uint4 e = stack[--sp];
        uint32_t lo = e.x & ~kTop, hi = e.y & ~kTop;
        uint32_t d3 = (e.x >> 31) | ((e.y >> 31) << 1);   // depth % 3
        const float plane2 = __uint_as_float(e.z & ~kTop);
        if (!(e.z & kTop)) offer((e.w & kTop) ? lo - 1u : hi, __uint_as_float(e.w & ~kTop));   // the node whose far range this is
        if (hi <= lo || plane2 > r2) continue;   // distanceToSplitPlane <= searchRadius, KDTree.h:153
        while (hi > lo && sp < kStack) {               // down the query's side, leaving the far ranges behind
            const uint32_t mid = lo + (hi - lo) / 2u;
            ++visited;
            const float4 p = sv.kdPos[2u * mid], cone = sv.kdPos[2u * mid + 1u];
            // no normal in this range's cone can have a non-negative dot product with dir: nothing here passes the filter
            if (dot3(dir, f3(cone.x, cone.y, cone.z)) < cone.w * dirLen) break;
            const F3 diff = f3(p.x, p.y, p.z) - q;
            const float d2 = dot3(diff, diff);
            if (hi - lo == 1u) { offer(mid, d2); break; }   // a leaf: nothing below it
            const uint32_t axis = d3 == 2u ? 0u : d3 + 1u;   // KDTreeBuilder::buildNode: the axis after the parent's, x before the root
            const float pa = axis == 0u ? p.x : (axis == 1u ? p.y : p.z);
            const float delta = qa[axis] - pa;
            const bool right = pa < qa[axis];          // targetContainingChildIdx, KDTree.h:122
            d3 = axis;                                 // (depth + 1) % 3
            const uint32_t dLo = (d3 & 1u) << 31, dHi = (d3 >> 1) << 31;
            const uint32_t planeBits = __float_as_uint(delta * delta), d2Bits = __float_as_uint(d2);
            if (right) { stack[sp++] = make_uint4(lo | dLo, mid | dHi, planeBits, d2Bits); lo = mid + 1u; }
            else { stack[sp++] = make_uint4((mid + 1u) | dLo, hi | dHi, planeBits, d2Bits | kTop); hi = mid; }
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
    return true;
}

// ---- one query per wave --------------------------------------------------------------------------------
// The same search by the 64 lanes of a wave together, for the queries gather_photons() gave up on.  Pending ranges
// sit on a stack in LDS; every iteration up to 64 of them are popped, one per lane, each lane looks at its range's
// median photon and pushes the two halves (far halves below near halves, so the near side of the search still runs
// ahead and the radius shrinks early).  The result list is kept sorted in the registers of lanes 0..19 and updated
// by the whole wave for one candidate at a time.  Ranges are cut off by the same two tests (split plane against the
// current radius, cone of normals against the filter), so the photons found are the k nearest that pass the
// filter -- the set gather_photons() finds, ties in distance aside.
#ifndef RT_WAVE_STACK
#define RT_WAVE_STACK 448
#endif
constexpr int kWaveStack = RT_WAVE_STACK;      // entries the stack may hold while 64 lanes pop together
constexpr int kWaveStackSlack = 64;    // room for the one-lane descent that runs when the stack is that full

struct GatherWaveLds {
    uint4 stack[kWaveStack + kWaveStackSlack];   // lo, hi, lower bound of the squared distance (float bits), depth
    uint32_t listIdx[kGatherK];
};

__device__ inline void gather_photons_wave(const SceneView& sv, F3 q, F3 dir, GatherWaveLds& L, const int lane, uint32_t& count, float& maxDist, RGBf& sum)
{
    float myD = INFINITY;   // lane k < kGatherK: squared distance and index of the k-th nearest accepted photon so far
    uint32_t myI = 0;
    int found = 0;
    float r2 = 1E9f * 1E9f;
    int sp = 0;
    if (sv.nPhotons) {
        if (lane == 0) L.stack[0] = make_uint4(0u, sv.nPhotons, 0u, 0u);
        sp = 1;
    }
    __syncthreads();
    const float qa[3] = {q.x, q.y, q.z};
    const float dirLen = norm3(dir);
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (sp > 0) {
        int n = sp < 64 ? sp : 64;
        if (n > kWaveStack - sp) n = kWaveStack - sp;   // every lane may push one entry more than it pops
        if (n < 1) n = 1;                               // a full stack: one lane descends, a level per iteration
        const bool active = lane < n;
        const uint4 e = L.stack[active ? sp - 1 - lane : 0];
        sp -= n;
        __syncthreads();
        bool hasNear = false, hasFar = false, cand = false;
        uint4 nearE = make_uint4(0, 0, 0, 0), farE = nearE;
        float cd2 = 0.0f;
        uint32_t ci = 0;
        if (active && e.y > e.x && __uint_as_float(e.z) <= r2) {
            const uint32_t lo = e.x, hi = e.y, depth = e.w;
            const uint32_t mid = lo + (hi - lo) / 2u;
            const float4 p = sv.kdPos[2u * mid], cone = sv.kdPos[2u * mid + 1u];
            if (!(dot3(dir, f3(cone.x, cone.y, cone.z)) < cone.w * dirLen)) {
                const F3 diff = f3(p.x, p.y, p.z) - q;
                const float d2 = dot3(diff, diff);
                if (d2 < r2) {
                    const float4 nn = sv.kdNormal[mid];
                    if (dot3(dir, f3(nn.x, nn.y, nn.z)) >= 0) { cand = true; cd2 = d2; ci = mid; }
                }
                if (hi - lo > 1u) {
                    const uint32_t axis = (depth + 1u) % 3u;
                    const float pa = axis == 0u ? p.x : (axis == 1u ? p.y : p.z);
                    const float delta = qa[axis] - pa;
                    const bool right = pa < qa[axis];
                    const float bound = __uint_as_float(e.z), farBound = fmaxf(bound, delta * delta);
                    const uint32_t nlo = right ? mid + 1u : lo, nhi = right ? hi : mid;
                    const uint32_t flo = right ? lo : mid + 1u, fhi = right ? mid : hi;
                    hasNear = nhi > nlo;
                    hasFar = fhi > flo && farBound <= r2;
                    nearE = make_uint4(nlo, nhi, e.z, depth + 1u);
                    farE = make_uint4(flo, fhi, __float_as_uint(farBound), depth + 1u);
                }
            }
        }
        // candidates enter the list one at a time, nearest-first order kept by position
        unsigned long long cm = __ballot(cand);
        while (cm != 0ull) {
            const int src = __ffsll((long long)cm) - 1;
            cm &= cm - 1ull;
            const float d = __shfl(cd2, src, 64);
            const uint32_t i = (uint32_t)__shfl((int)ci, src, 64);
            if (d < r2) {
                const int pos = __popcll(__ballot(lane < kGatherK && myD <= d));
                const float upD = __shfl_up(myD, 1, 64);
                const uint32_t upI = (uint32_t)__shfl_up((int)myI, 1, 64);
                if (lane < kGatherK) {
                    if (lane == pos) { myD = d; myI = i; }
                    else if (lane > pos) { myD = upD; myI = upI; }
                }
                if (found < kGatherK) ++found;
                if (found == kGatherK) r2 = __shfl(myD, kGatherK - 1, 64);
            }
        }
        const unsigned long long fm = __ballot(hasFar), nm = __ballot(hasNear);
        const int nf = __popcll(fm), nn = __popcll(nm);
        if (hasFar) L.stack[sp + __popcll(fm & lt)] = farE;
        if (hasNear) L.stack[sp + nf + nn - 1 - __popcll(nm & lt)] = nearE;   // lane 0 held the top entry: its near half goes on top
        sp += nf + nn;
        __syncthreads();
    }
    if (lane < kGatherK) L.listIdx[lane] = myI;
    __syncthreads();
    count = (uint32_t)found;
    const float far2 = __shfl(myD, kGatherK - 1, 64);
    maxDist = found == kGatherK ? sqrtf(far2) : (found > 0 ? FLT_MAX : 1E9f);
    sum = rgb(0, 0, 0);
    for (int a = 0; a < found; ++a) {   // value += photons[i]->energy, nearest first (every lane computes the same sum)
        const float4 en = sv.kdEnergy[L.listIdx[a]];
        sum.r += en.x; sum.g += en.y; sum.b += en.z;
    }
    __syncthreads();
}

} // namespace rt
