// rt_steal_inl.h -- wave-cooperative traversal with work stealing through the LDS stacks.
//
// A ray's traversal is a chain of dependent fetches, and a few rays (grazing the surface) need
// 10-25x the average number of steps: with one lane per ray the slowest ray of a launch sets
// the launch time while its 63 neighbour lanes -- and most of the chip -- idle.  Here a lane is a
// *worker*: whenever some lanes of the wave have run out of work they take pending subtrees from
// the top (newest = nearest) end of busy lanes' stacks, which live in LDS as
// stack[entry][lane] and are therefore readable by every lane of the wave.  Several workers can
// then be inside one ray's tree at once; they share the ray's best hit through an LDS word per
// ray:
//
//   best[ray] = (float bits of t) << 32 | caller's triangle index      (64-bit atomic min in LDS)
//
// Non-negative floats order like their bit patterns, so the atomic min keeps the smallest t and,
// among equal t, the smallest triangle index -- exactly the tree-independent result definition of
// trace_ray(), whatever the schedule.  The upper half is also every worker's pruning distance.
// The winning triangle's leaf slot and barycentric coordinates are written by the lane whose value
// is the minimum when it re-reads the word after its atomic (exactly one lane per ray and step:
// the atomics of a step have all been served by then); a later, smaller hit overwrites them in
// program order.  The LDS block is 5 KB per wave.
//
// trace_wave_steal() serves single-mesh scenes (triangle soups), trace_wave_steal_inst() scenes of
// several mesh instances; scenes with analytic spheres use the per-lane walks of rt_trace_inl.h.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_trace_inl.h"

namespace rt {

#ifndef RT_STEAL_STACK
#define RT_STEAL_STACK 14
#endif
constexpr int kStealStack = RT_STEAL_STACK;   // LDS stack entries per lane; with 14 the block is 5 KB: 8 waves per SIMD use all of a CU's 160 KB

template <bool COUNT>
struct StealCounters {
    uint32_t cnt[2][64];   // per-ray work counters
};
template <>
struct StealCounters<false> {};

template <bool COUNT>
struct StealLds : StealCounters<COUNT> {
    int stack[kStealStack][64];
    unsigned long long best[64];
    uint32_t paySlot[64];
    float payBeta[64], payGamma[64];   // barycentric coordinates of the hit in paySlot (closest-hit queries)
    int giveList[64];
};

// t must be >= +0 or +inf: the bit pattern of -0 (what Triangle::intersect returns for a ray that starts exactly in the
// triangle's plane: -0 < 0 is false, so the reference accepts it, math/Triangle.h:60-66) would order above +inf.  Callers
// pass t + 0.0f, which is t for every other value and +0 for -0.
__device__ __forceinline__ unsigned long long pack_hit(float t, uint32_t id)
{
    return ((unsigned long long)__float_as_uint(t) << 32) | id;
}

__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// set bits of `m` below this lane's position
__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <bool ANY, bool COUNT, bool WIDE>
__device__ __forceinline__ TraceResult trace_wave_steal_walk(const SceneView& sv, bool hasRay, F3 wo, F3 wd, float tmax,
                                                             StealLds<COUNT>& L, const int lane, uint32_t& nBoxes, uint32_t& nTris,
                                                             const int minIdle, const int leafPct)
{
    const DevInstance& in = sv.inst[0];
    const uint32_t nodeBase = in.root, triBase = in.triBase;
    F3 o, d;
    xf_ray_inv(in.inv, sv.identityXf != 0, wo, wd, o, d);
    // rays with a NaN component find nothing (see trace_ray)
    const bool valid = hasRay && ((wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z) == (wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z));
    F3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);

    L.best[lane] = pack_hit(ANY ? tmax : INFINITY, kInvalid);
    L.paySlot[lane] = kInvalid;
    if constexpr (COUNT) { L.cnt[0][lane] = 0; L.cnt[1][lane] = 0; }

    // A lane's state is `cur` alone: >= 0 an inner node; kIdle (the most negative number) no work; any other negative number
    // ~(first << 4 | count - 1), a leaf with `count` triangles left to test.  Every vote of the loop below is then ONE
    // compare on a register (a vote on a condition put together from several costs two more vector instructions each: the
    // compiler turns the combined lane mask into 0 / 1 per lane and compares that again).
    constexpr int kIdle = (int)0x80000000;
    int owner = lane;          // the ray this worker is traversing for
    int cur = valid ? 0 : kIdle;
    int sp = 0, sb = 0;        // this lane's stack holds entries [sb, sp) (sb stays 0 here; the instanced walk drops a bottom sentinel); thieves take from the top
    float tlim = ANY ? tmax : INFINITY;
    float rtmax = tmax;        // any-hit: the limit of the ray this worker is traversing for
    uint32_t wb = 0, wt = 0;
    int spill[kStackDepth - kStealStack];
#ifdef RT_DEV_STATS
    float dsOuter = 0, dsSteal = 0, dsNodeIt = 0, dsLeafIt = 0, dsBusy = 0, dsStolen = 0, dsHi56 = 0, dsHi48 = 0, dsLo32 = 0, dsLo16 = 0;
    const uint32_t dsT0 = (uint32_t)wall_clock64();
#endif

    // The stack's first kStealStack entries live in LDS, deeper ones -- rarely reached -- in private
    // memory.  A step runs its straight-line form (unconditional LDS write and read, no branches)
    // unless some lane of the wave is that deep.
    auto push_deep = [&](int v) {
        if (sp < kStealStack) L.stack[sp][lane] = v;
        else spill[sp - kStealStack] = v;
        ++sp;
    };
    auto pop_deep = [&]() {
        --sp;
        cur = L.stack[sp < kStealStack ? sp : 0][lane];          // always an LDS read ...
        asm volatile("" : "+v"(cur));                            // (keeps the two loads apart: merged, they become one slow generic-address load)
        if (sp >= kStealStack) cur = spill[sp - kStealStack];    // ... replaced by the private entry when that deep
    };

    // Every iteration the wave runs one step of one kind: a node step for the lanes standing at an
    // inner node, or a leaf step (one triangle test) for the lanes standing at a leaf.  Lanes of
    // the other kind wait, so the kind is chosen per iteration: node steps as long as few lanes
    // wait at leaves.
    for (;;) {
        const unsigned long long busyMask = wave_ballot(cur != kIdle);
        if (busyMask == 0ull) break;
        const unsigned long long idleMask = ~busyMask;
#ifdef RT_DEV_STATS
        if (lane == 0) {
            dsOuter += 1; dsBusy += __popcll(busyMask);
            if (__popcll(busyMask) >= 56) dsHi56 += 1;
            if (__popcll(busyMask) >= 48) dsHi48 += 1;
            if (__popcll(busyMask) < 32) dsLo32 += 1;
            if (__popcll(busyMask) < 16) dsLo16 += 1;
        }
#endif
        // ---- steal phase -------------------------------------------------------------------------
        // Idle lanes take the top entries of busy lanes' stacks -- the nearest pending subtrees, the work those rays need
        // next, so thief and victim close in on the hit together (taking the bottom, farthest entry instead needed fewer
        // rounds, but a fifth of the boxes stolen that way were cut off later by a nearer hit).  A lane gives its top entry,
        // and its second one too when idle lanes are left over.
        if (__popcll(idleMask) >= minIdle) {
            const bool canGive = (uint32_t)(sp - 1) < (uint32_t)kStealStack;   // 0 < sp <= kStealStack (sb stays 0 in this walk; an idle lane's sp is 0)
            const unsigned long long giveMask = wave_ballot(canGive);
            if (giveMask != 0ull) {
                const int rg = lanes_below(giveMask), ri = lanes_below(idleMask);
                const int nGive = __popcll(giveMask), nIdle = __popcll(idleMask);
                const bool canGive2 = (uint32_t)(sp - 2) < (uint32_t)(kStealStack - 1);   // 2 <= sp <= kStealStack
                const unsigned long long give2Mask = wave_ballot(canGive2);
                const int rg2 = nGive + lanes_below(give2Mask), nGiveAll = nGive + __popcll(give2Mask);
                if (canGive) L.giveList[rg] = lane;
                if (canGive2 && rg2 < 64) L.giveList[rg2] = lane | 64;   // (bit 6: the second entry from the top)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool thief = cur == kIdle && ri < nGiveAll;
#ifdef RT_DEV_STATS
                if (lane == 0) { dsSteal += 1; dsStolen += nGiveAll < nIdle ? nGiveAll : nIdle; }
#endif
                const int ge = L.giveList[thief ? ri : 0];
                const int v = ge & 63;
                const int ventry = __shfl(sp, v, 64) - 1 - (ge >> 6);
                const int vowner = __shfl(owner, v, 64);
                const float vox = __shfl(o.x, v, 64), voy = __shfl(o.y, v, 64), voz = __shfl(o.z, v, 64);
                const float vdx = __shfl(d.x, v, 64), vdy = __shfl(d.y, v, 64), vdz = __shfl(d.z, v, 64);
                const float vix = __shfl(inv.x, v, 64), viy = __shfl(inv.y, v, 64), viz = __shfl(inv.z, v, 64);
                const float vtmax = __shfl(rtmax, v, 64);
                if (thief) {
                    cur = L.stack[ventry][v];
                    owner = vowner;
                    o = f3(vox, voy, voz); d = f3(vdx, vdy, vdz); inv = f3(vix, viy, viz);
                    rtmax = vtmax;
                    sp = 0; sb = 0;
                    tlim = __uint_as_float((uint32_t)(L.best[owner] >> 32));
                }
                sp -= (canGive && rg < nIdle ? 1 : 0) + (canGive2 && rg2 < nIdle ? 1 : 0);
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long nowBusy = wave_ballot(cur != kIdle);
        const unsigned long long nodeMask = wave_ballot(cur >= 0);
        const int nBusy = __popcll(nowBusy), nNode = __popcll(nodeMask);
        const bool nodeStep = nNode > 0 && (nBusy - nNode) * 128 < leafPct * nBusy;
        const bool deep = wave_ballot(sp >= kStealStack - 2) != 0ull;   // a node step may push three entries
        bool done = false;
        if (nodeStep) {
            // ---- inner node ------------------------------------------------------------------------
            if (cur >= 0) {
                // another worker of the same ray may have found a closer hit since this lane's last leaf
                // (read before the node is asked for, so that the LDS round trip runs under the memory latency)
                tlim = __uint_as_float(((const uint32_t*)&L.best[owner])[1]);
                __builtin_amdgcn_sched_barrier(0);   // keeps the LDS read in front of the loads below
                int c[4];
                uint32_t tested;
                const int nh = node4_hits<!ANY, WIDE>(node_ptr(sv, nodeBase + (uint32_t)cur), o, inv, tlim, c, tested);
                if (COUNT) wb += tested;
#ifdef RT_DEV_STATS
                dsNodeIt += 1.0f / (float)__popcll(wave_ballot(true));
#endif
                if (!deep) {
                    // the nearest child is entered, the others go on the stack farthest first: c[j] at sp + nh - 1 - j
                    const int popped = L.stack[sp > 0 ? sp - 1 : 0][lane];
                    if (nh > 3) L.stack[sp + nh - 4][lane] = c[3];
                    if (nh > 2) L.stack[sp + nh - 3][lane] = c[2];
                    if (nh > 1) L.stack[sp + nh - 2][lane] = c[1];
                    done = nh == 0 && sp == sb;
                    cur = nh > 0 ? c[0] : popped;
                    sp += nh > 0 ? nh - 1 : (done ? 0 : -1);
                } else {
                    if (nh > 3) push_deep(c[3]);
                    if (nh > 2) push_deep(c[2]);
                    if (nh > 1) push_deep(c[1]);
                    if (nh > 0) cur = c[0];
                    else if (sp == sb) done = true;
                    else pop_deep();
                }
            }
        } else {
            // ---- leaf step: the triangles waiting at the wave's leaves, dealt out over all 64 lanes -----------
            // A lane at a leaf offers up to four of its triangles; offers are numbered across the wave (prefix
            // counts from three ballots) and written to a list, and lane k tests offer k with the offering
            // lane's ray.  Results meet in best[], so who tests what does not matter.
            const bool atLeaf = (uint32_t)cur > (uint32_t)kIdle;   // negative, but not kIdle
            const uint32_t leaf = (uint32_t)(~cur);
            const int left = (int)(leaf & 15u) + 1;
            const int want = atLeaf ? (left < 4 ? left : 4) : 0;
            const int wantBits = atLeaf ? want - 1 : 0;   // offers = 1 + bit 0 + 2 x bit 1
            const unsigned long long m0 = wave_ballot(atLeaf);
            const unsigned long long m1 = wave_ballot((wantBits & 1) != 0);
            const unsigned long long m2 = wave_ballot((wantBits & 2) != 0);
            const int pre = lanes_below(m0) + lanes_below(m1) + 2 * lanes_below(m2);
            const int total = __popcll(m0) + __popcll(m1) + 2 * __popcll(m2);
            int served = 64 - pre;
            served = served < want ? served : want;
            served = served > 0 ? served : 0;
            if (served > 0) L.giveList[pre] = lane;
            if (served > 1) L.giveList[pre + 1] = lane | (1 << 6);
            if (served > 2) L.giveList[pre + 2] = lane | (2 << 6);
            if (served > 3) L.giveList[pre + 3] = lane | (3 << 6);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool helper = lane < total;
            const int e = L.giveList[helper ? lane : 0];
            const int src = e & 63;
            const int hcur = __shfl(cur, src, 64);
            const int howner = __shfl(owner, src, 64);
            const float hox = __shfl(o.x, src, 64), hoy = __shfl(o.y, src, 64), hoz = __shfl(o.z, src, 64);
            const float hdx = __shfl(d.x, src, 64), hdy = __shfl(d.y, src, 64), hdz = __shfl(d.z, src, 64);
            const float hlim = __shfl(ANY ? rtmax : tlim, src, 64);
            if (helper) {
                const uint32_t slot = triBase + ((uint32_t)(~hcur) >> 4) + (uint32_t)(e >> 6);
                const float4* tp = tri_ptr(sv, slot);
                const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
                float t, tb, tg;
#ifdef RT_DEV_STATS
                dsLeafIt += 1.0f / (float)__popcll(wave_ballot(true));
#endif
                if (tri_test_flat(r0, r1, r2, f3(hox, hoy, hoz), f3(hdx, hdy, hdz), t, tb, tg) && t <= hlim) {
                    if (ANY) {
                        atomicMin(&L.best[howner], 0ull);
                    } else {
                        const unsigned long long mine = pack_hit(t + 0.0f, sv.triIds[slot]);   // -0 (a ray that starts in the triangle's plane) as +0
                        const unsigned long long old = atomicMin(&L.best[howner], mine);
                        if (mine < old && L.best[howner] == mine) { L.paySlot[howner] = slot; L.payBeta[howner] = tb; L.payGamma[howner] = tg; }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (served > 0) {
                if (COUNT) wt += (uint32_t)served;
                const unsigned long long b = L.best[owner];
                tlim = __uint_as_float((uint32_t)(b >> 32));
                if (left > served && !(ANY && b == 0ull)) {
                    cur = ~(int)(leaf + 15u * (uint32_t)served);        // first + served, count - served
                } else if ((ANY && b == 0ull) || sp == sb) {
                    done = true;
                } else if (!deep) {
                    --sp;
                    cur = L.stack[sp][lane];
                } else {
                    pop_deep();
                }
            }
        }
        if (done) {
            cur = kIdle;
            sp = 0; sb = 0;
            if constexpr (COUNT) { atomicAdd(&L.cnt[0][owner], wb); atomicAdd(&L.cnt[1][owner], wt); wb = 0; wt = 0; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    TraceResult r;
    const unsigned long long b = L.best[lane];
    r.prim = (uint32_t)b;
    if (ANY) {
        r.t = tmax; r.slot = kInvalid; r.beta = 0.0f; r.gamma = 0.0f;
        r.inst = (b == 0ull && valid) ? 0u : kInvalid;
    } else {
        const bool hit = L.paySlot[lane] != kInvalid;
        r.t = hit ? __uint_as_float((uint32_t)(b >> 32)) : INFINITY;
        r.slot = L.paySlot[lane];
        r.inst = hit ? 0u : kInvalid;
        r.beta = hit ? L.payBeta[lane] : 0.0f;
        r.gamma = hit ? L.payGamma[lane] : 0.0f;
    }
    if constexpr (COUNT) { nBoxes = L.cnt[0][lane]; nTris = L.cnt[1][lane]; }
#ifdef RT_DEV_STATS
    // developer build: lanes 0..7 report wave-level schedule statistics in place of their tri counts
    if (COUNT) {
        __shared__ float ds[12];
        if (lane < 12) ds[lane] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        atomicAdd(&ds[0], dsOuter); atomicAdd(&ds[1], dsSteal); atomicAdd(&ds[2], dsNodeIt);
        atomicAdd(&ds[3], dsLeafIt); atomicAdd(&ds[4], dsBusy); atomicAdd(&ds[5], dsStolen);
        atomicAdd(&ds[8], dsHi56); atomicAdd(&ds[9], dsHi48); atomicAdd(&ds[10], dsLo32); atomicAdd(&ds[11], dsLo16);
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) nTris = (uint32_t)(ds[lane] + 0.5f);
        if (lane == 6) nTris = dsT0;
        if (lane == 7) nTris = (uint32_t)wall_clock64();
        if (lane >= 8 && lane < 12) nTris = (uint32_t)(ds[lane] + 0.5f);   // iterations with >= 56, >= 48, < 32, < 16 busy lanes
    }
#endif
    return r;
}

// The walk with the plain box decode, or -- when some ray of the wave starts farther from the tree than the plain decode's
// rounding allows (plane_offsets, far_from_tree) -- the same walk with the widened decode.
template <bool ANY, bool COUNT>
__device__ __forceinline__ TraceResult trace_wave_steal(const SceneView& sv, bool hasRay, F3 wo, F3 wd, float tmax,
                                                        StealLds<COUNT>& L, const int lane, uint32_t& nBoxes, uint32_t& nTris,
                                                        const int minIdle = 1, const int leafPct = 48)
{
    const DevInstance& in = sv.inst[0];
    F3 o, d;
    xf_ray_inv(in.inv, sv.identityXf != 0, wo, wd, o, d);
    const bool wide = wave_ballot(hasRay && far_from_tree(node_ptr(sv, in.root), o)) != 0ull;
    if (wide) return trace_wave_steal_walk<ANY, COUNT, true>(sv, hasRay, wo, wd, tmax, L, lane, nBoxes, nTris, minIdle, leafPct);
    return trace_wave_steal_walk<ANY, COUNT, false>(sv, hasRay, wo, wd, tmax, L, lane, nBoxes, nTris, minIdle, leafPct);
}

// ---- instanced scenes ------------------------------------------------------------------------
//
// The same scheme over a top-level tree with one mesh instance per leaf (scenes without analytic
// spheres; those keep the reference-order walk).  A worker is either at the top level (world-space
// ray, top-level nodes) or inside one instance (object-space ray, that mesh's nodes); its stack
// holds both kinds of entries, separated by a sentinel that marks the return to the top level.
// A stolen entry is of the kind its position says: below the victim's sentinel it is a top-level
// subtree (the thief takes the world ray), above it a subtree of the victim's current instance
// (the thief takes the victim's object-space ray and instance).  A sentinel at the very bottom
// of a stack has nothing to return to and is dropped.  Thieves take the top entry (see trace_wave_steal_walk) unless it is
// the sentinel.
//
// best[ray] = t bits << 32 | key, key = keyBase[instance] + caller's triangle index with keyBase
// the running sum of the instances' triangle counts: keys order like (instance, triangle index),
// so the atomic min implements the result definition of trace_ray() here too.  Scene creation
// falls back to the per-lane kernel when the keys do not fit 32 bits.

template <bool COUNT>
struct StealLdsInst : StealLds<COUNT> {
    uint32_t payInst[64];
};

template <bool ANY, bool COUNT>
__device__ __forceinline__ TraceResult trace_wave_steal_inst(const SceneView& sv, bool hasRay, F3 wo0, F3 wd0, float tmax,
                                                             StealLdsInst<COUNT>& L, const int lane, uint32_t& nBoxes, uint32_t& nTris,
                                                             const int minIdle = 1, const int leafPct = 48)
{
    // rays with a NaN component find nothing (see trace_ray)
    const bool valid = hasRay && ((wo0.x + wo0.y + wo0.z) + (wd0.x + wd0.y + wd0.z) == (wo0.x + wo0.y + wo0.z) + (wd0.x + wd0.y + wd0.z));

    L.best[lane] = pack_hit(ANY ? tmax : INFINITY, kInvalid);
    L.paySlot[lane] = kInvalid;
    L.payInst[lane] = kInvalid;
    if constexpr (COUNT) { L.cnt[0][lane] = 0; L.cnt[1][lane] = 0; }

    F3 wo = wo0, wd = wd0;                                   // world ray of the ray this worker works for
    F3 o = wo, d = wd, inv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);   // ray of the current level
    uint32_t nodeBase = sv.topRoot, triBase = 0, curInst = 0, keyBase = 0;
    bool inMesh = false;
    int sentPos = -1;          // stack index of the sentinel, -1: none (nothing to return to)
    constexpr int kIdle = (int)0x80000000;   // a lane without work (see trace_wave_steal_walk: votes on `cur` are single compares)
    int owner = lane;
    int cur = valid ? 0 : kIdle;
    int sp = 0, sb = 0;
    float tlim = ANY ? tmax : INFINITY;
    float rtmax = tmax;
    uint32_t wb = 0, wt = 0;
    int spill[kStackDepth - kStealStack];

    auto push = [&](int v) {
        if (sp < kStealStack) L.stack[sp][lane] = v;
        else spill[sp - kStealStack] = v;
        ++sp;
    };
    // Pops the next pending subtree, returning to the top level when the sentinel surfaces; true = nothing left.
    auto pop = [&]() -> bool {
        for (;;) {
            if (sp == sb) return true;
            --sp;
            cur = L.stack[sp < kStealStack ? sp : 0][lane];
            asm volatile("" : "+v"(cur));
            if (sp >= kStealStack) cur = spill[sp - kStealStack];
            if (cur != kSentinel) return false;
            inMesh = false;
            sentPos = -1;
            o = wo; d = wd; inv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
            nodeBase = sv.topRoot;
        }
    };

    for (;;) {
        const unsigned long long busyMask = wave_ballot(cur != kIdle);
        if (busyMask == 0ull) break;
        const unsigned long long idleMask = ~busyMask;
        // ---- steal phase -------------------------------------------------------------------------
        if (__popcll(idleMask) >= minIdle) {
            if (sentPos >= 0 && sb == sentPos) { ++sb; sentPos = -1; }   // a bottom sentinel has nothing to return to
            // (the top entry, but not the sentinel itself -- the victim pops that next and returns to the top level)
            const bool canGive = sb < sp && sp <= kStealStack && sentPos != sp - 1;   // (an idle lane's stack is empty)
            const unsigned long long giveMask = wave_ballot(canGive);
            if (giveMask != 0ull) {
                const int rg = lanes_below(giveMask), ri = lanes_below(idleMask);
                const int nGive = __popcll(giveMask), nIdle = __popcll(idleMask);
                // (a second entry from the top when idle lanes are left over, as in trace_wave_steal_walk -- unless it is the sentinel)
                const bool canGive2 = canGive && sp - sb >= 2 && sentPos != sp - 2;
                const unsigned long long give2Mask = wave_ballot(canGive2);
                const int rg2 = nGive + lanes_below(give2Mask), nGiveAll = nGive + __popcll(give2Mask);
                if (canGive) L.giveList[rg] = lane;
                if (canGive2 && rg2 < 64) L.giveList[rg2] = lane | 64;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool thief = cur == kIdle && ri < nGiveAll;
                const int ge = L.giveList[thief ? ri : 0];
                const int v = ge & 63;
                const int vsb = __shfl(sp, v, 64) - 1 - (ge >> 6);
                const int vowner = __shfl(owner, v, 64);
                const int vsent = __shfl(sentPos, v, 64);
                const int vinMesh = __shfl((int)inMesh, v, 64);
                const uint32_t vnodeBase = (uint32_t)__shfl((int)nodeBase, v, 64), vtriBase = (uint32_t)__shfl((int)triBase, v, 64);
                const uint32_t vcurInst = (uint32_t)__shfl((int)curInst, v, 64), vkeyBase = (uint32_t)__shfl((int)keyBase, v, 64);
                const float vox = __shfl(o.x, v, 64), voy = __shfl(o.y, v, 64), voz = __shfl(o.z, v, 64);
                const float vdx = __shfl(d.x, v, 64), vdy = __shfl(d.y, v, 64), vdz = __shfl(d.z, v, 64);
                const float vix = __shfl(inv.x, v, 64), viy = __shfl(inv.y, v, 64), viz = __shfl(inv.z, v, 64);
                const float wox = __shfl(wo.x, v, 64), woy = __shfl(wo.y, v, 64), woz = __shfl(wo.z, v, 64);
                const float wdx = __shfl(wd.x, v, 64), wdy = __shfl(wd.y, v, 64), wdz = __shfl(wd.z, v, 64);
                const float vtmax = __shfl(rtmax, v, 64);
                if (thief) {
                    cur = L.stack[vsb][v];
                    owner = vowner;
                    wo = f3(wox, woy, woz); wd = f3(wdx, wdy, wdz);
                    const bool meshLevel = vinMesh != 0 && (vsent < 0 || vsb > vsent);
                    if (meshLevel) {
                        o = f3(vox, voy, voz); d = f3(vdx, vdy, vdz); inv = f3(vix, viy, viz);
                        nodeBase = vnodeBase; triBase = vtriBase; curInst = vcurInst; keyBase = vkeyBase;
                        inMesh = true;
                    } else {
                        o = wo; d = wd; inv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
                        nodeBase = sv.topRoot;
                        inMesh = false;
                    }
                    sentPos = -1;
                    rtmax = vtmax;
                    sp = 0; sb = 0;
                    tlim = __uint_as_float((uint32_t)(L.best[owner] >> 32));
                }
                sp -= (canGive && rg < nIdle ? 1 : 0) + (canGive2 && rg2 < nIdle ? 1 : 0);
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long nowBusy = wave_ballot(cur != kIdle);
        const unsigned long long nodeMask = wave_ballot(cur >= 0);
        const int nBusy = __popcll(nowBusy), nNode = __popcll(nodeMask);
        const bool nodeStep = nNode > 0 && (nBusy - nNode) * 128 < leafPct * nBusy;
        bool done = false;
        if (nodeStep) {
            // ---- inner node of the top-level tree or of a mesh tree ------------------------------------
            if (cur >= 0) {
                tlim = __uint_as_float(((const uint32_t*)&L.best[owner])[1]);
                int c[4];
                uint32_t tested;
                const int nh = node4_hits<!ANY, true>(node_ptr(sv, nodeBase + (uint32_t)cur), o, inv, tlim, c, tested);   // (object-space rays of scaled instances: always the widened decode)
                if (COUNT) wb += tested;
                if (nh > 3) push(c[3]);
                if (nh > 2) push(c[2]);
                if (nh > 1) push(c[1]);
                if (nh > 0) cur = c[0];
                else done = pop();
            }
        } else {
            // ---- leaf step: mesh triangles dealt out over all 64 lanes (see trace_wave_steal), then top-level leaves
            const bool atLeaf = (uint32_t)cur > (uint32_t)kIdle;   // negative, but not kIdle
            const bool triLeaf = atLeaf && inMesh;
            const uint32_t leaf = (uint32_t)(~cur);
            const int left = (int)(leaf & 15u) + 1;
            const int want = triLeaf ? (left < 4 ? left : 4) : 0;
            const unsigned long long m0 = wave_ballot(triLeaf);
            if (m0 != 0ull) {
                const unsigned long long m1 = wave_ballot(triLeaf && ((want - 1) & 1) != 0);
                const unsigned long long m2 = wave_ballot(triLeaf && ((want - 1) & 2) != 0);
                const int pre = lanes_below(m0) + lanes_below(m1) + 2 * lanes_below(m2);
                const int total = __popcll(m0) + __popcll(m1) + 2 * __popcll(m2);
                int served = 64 - pre;
                served = served < want ? served : want;
                served = served > 0 ? served : 0;
                if (served > 0) L.giveList[pre] = lane;
                if (served > 1) L.giveList[pre + 1] = lane | (1 << 6);
                if (served > 2) L.giveList[pre + 2] = lane | (2 << 6);
                if (served > 3) L.giveList[pre + 3] = lane | (3 << 6);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool helper = lane < total;
                const int e = L.giveList[helper ? lane : 0];
                const int src = e & 63;
                const int hcur = __shfl(cur, src, 64);
                const int howner = __shfl(owner, src, 64);
                const uint32_t htriBase = (uint32_t)__shfl((int)triBase, src, 64), hkeyBase = (uint32_t)__shfl((int)keyBase, src, 64);
                const uint32_t hinst = (uint32_t)__shfl((int)curInst, src, 64);
                const float hox = __shfl(o.x, src, 64), hoy = __shfl(o.y, src, 64), hoz = __shfl(o.z, src, 64);
                const float hdx = __shfl(d.x, src, 64), hdy = __shfl(d.y, src, 64), hdz = __shfl(d.z, src, 64);
                const float hlim = __shfl(ANY ? rtmax : tlim, src, 64);
                if (helper) {
                    const uint32_t slot = htriBase + ((uint32_t)(~hcur) >> 4) + (uint32_t)(e >> 6);
                    const float4* tp = tri_ptr(sv, slot);
                    const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
                    float t, tb, tg;
                    if (tri_test_flat(r0, r1, r2, f3(hox, hoy, hoz), f3(hdx, hdy, hdz), t, tb, tg) && t <= hlim) {
                        if (ANY) {
                            atomicMin(&L.best[howner], 0ull);
                        } else {
                            const unsigned long long mine = pack_hit(t + 0.0f, hkeyBase + sv.triIds[slot]);   // -0 as +0, see pack_hit
                            const unsigned long long old = atomicMin(&L.best[howner], mine);
                            if (mine < old && L.best[howner] == mine) { L.paySlot[howner] = slot; L.payInst[howner] = hinst; L.payBeta[howner] = tb; L.payGamma[howner] = tg; }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (served > 0) {
                    if (COUNT) wt += (uint32_t)served;
                    const unsigned long long b = L.best[owner];
                    tlim = __uint_as_float((uint32_t)(b >> 32));
                    if (ANY && b == 0ull) done = true;
                    else if (left > served) cur = ~(int)(leaf + 15u * (uint32_t)served);
                    else done = pop();
                }
            }
            if (atLeaf && !triLeaf) {
                // ---- top-level leaf: enter the instance (InstancedModelList::traceRay, shape/list/InstancedModelList.cpp:99-138)
                curInst = sv.instOrder[leaf >> 4];
                const DevInstance& in = sv.inst[curInst];
                o = xf_point_inv(in.inv, wo);
                d = xf_vector_inv(in.inv, wd);
                inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                nodeBase = in.root;
                triBase = in.triBase;
                keyBase = in.keyBase;
                inMesh = true;
                if (sp > sb) { sentPos = sp; push(kSentinel); }
                else sentPos = -1;
                cur = 0;
            }
        }
        if (done) {
            cur = kIdle;
            sp = 0; sb = 0;
            sentPos = -1;
            if constexpr (COUNT) { atomicAdd(&L.cnt[0][owner], wb); atomicAdd(&L.cnt[1][owner], wt); wb = 0; wt = 0; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    TraceResult r;
    const unsigned long long b = L.best[lane];
    r.beta = 0.0f; r.gamma = 0.0f;
    if (ANY) {
        r.t = tmax; r.slot = kInvalid; r.prim = kInvalid;
        r.inst = (b == 0ull && valid) ? 0u : kInvalid;
    } else {
        r.slot = L.paySlot[lane];
        const bool hit = r.slot != kInvalid;
        r.t = hit ? __uint_as_float((uint32_t)(b >> 32)) : INFINITY;
        r.inst = hit ? L.payInst[lane] : kInvalid;
        r.prim = hit ? sv.triIds[r.slot] : kInvalid;
        if (hit) { r.beta = L.payBeta[lane]; r.gamma = L.payGamma[lane]; }
    }
    if constexpr (COUNT) { nBoxes = L.cnt[0][lane]; nTris = L.cnt[1][lane]; }
    return r;
}

} // namespace rt
