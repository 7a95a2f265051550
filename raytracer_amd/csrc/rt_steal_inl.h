// rt_steal_inl.h -- wave-cooperative traversal with work stealing through the LDS stacks.
//
// A ray's traversal is a chain of dependent fetches, and a few rays (grazing the surface) need
// 10-25x the average number of steps: with one lane per ray the slowest ray of a launch sets
// the launch time while its 63 neighbour lanes -- and most of the chip -- idle.  Here a lane is a
// *worker*: whenever some lanes of the wave have run out of work they take pending subtrees from
// the bottom (oldest = largest) end of busy lanes' stacks, which live in LDS as
// stack[entry][lane] and are therefore readable by every lane of the wave.  Several workers can
// then be inside one ray's tree at once; they share the ray's best hit through an LDS word per
// ray:
//
//   best[ray] = (float bits of t) << 32 | caller's triangle index      (64-bit atomic min in LDS)
//
// Non-negative floats order like their bit patterns, so the atomic min keeps the smallest t and,
// among equal t, the smallest triangle index -- exactly the tree-independent result definition of
// trace_ray(), whatever the schedule.  The upper half is also every worker's pruning distance.
// The payload (leaf slot, beta, gamma) is written by the lane whose value is the minimum when it
// re-reads the word after its atomic; a later, smaller hit overwrites it in program order.
//
// Single-mesh scenes only (the benchmark and render workloads are triangle soups); instanced
// scenes use trace_ray().
#pragma once
#include <hip/hip_runtime.h>

#include "rt_trace_inl.h"

namespace rt {

struct StealLds {
    int stack[kLdsStack][64];
    unsigned long long best[64];
    uint32_t paySlot[64];
    float payBeta[64];
    float payGamma[64];
    int giveList[64];
    uint32_t cnt[2][64];
};


__device__ __forceinline__ unsigned long long pack_hit(float t, uint32_t id)
{
    return ((unsigned long long)__float_as_uint(t) << 32) | id;
}

template <bool ANY, bool COUNT>
__device__ __forceinline__ TraceResult trace_wave_steal(const SceneView& sv, bool hasRay, F3 wo, F3 wd, float tmax,
                                                        StealLds& L, const int lane, uint32_t& nBoxes, uint32_t& nTris,
                                                        const int minIdle = 1, const int leafPct = 48)
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
    L.payBeta[lane] = 0.0f;
    L.payGamma[lane] = 0.0f;
    if (COUNT) { L.cnt[0][lane] = 0; L.cnt[1][lane] = 0; }

    int owner = lane;          // the ray this worker is traversing for
    int cur = 0;
    int sp = 0, sb = 0;        // this lane's stack holds entries [sb, sp); thieves take from sb
    bool busy = valid;
    float tlim = ANY ? tmax : INFINITY;
    float rtmax = tmax;        // any-hit: the limit of the ray this worker is traversing for
    uint32_t wb = 0, wt = 0;
    int spill[kStackDepth - kLdsStack];
    const unsigned long long ltMask = (1ull << lane) - 1ull;
#ifdef RT_DEV_STATS
    float dsOuter = 0, dsSteal = 0, dsNodeIt = 0, dsLeafIt = 0, dsBusy = 0, dsStolen = 0;
    const uint32_t dsT0 = (uint32_t)wall_clock64();
#endif

    // Every iteration the wave runs one step of one kind: a node step for the lanes standing at an
    // inner node, or a leaf step for the lanes standing at a leaf.  Lanes of the other kind wait, so
    // the kind is chosen per iteration: node steps as long as few lanes wait at leaves.
    for (;;) {
        const unsigned long long busyMask = __ballot(busy);
        if (busyMask == 0ull) break;
        const unsigned long long idleMask = ~busyMask;
#ifdef RT_DEV_STATS
        if (lane == 0) { dsOuter += 1; dsBusy += __popcll(busyMask); }
#endif
        // ---- steal phase -------------------------------------------------------------------------
        if (__popcll(idleMask) >= minIdle) {
            const bool canGive = busy && sb < sp && sb < kLdsStack;
            const unsigned long long giveMask = __ballot(canGive);
            if (giveMask != 0ull) {
                const int rg = __popcll(giveMask & ltMask), ri = __popcll(idleMask & ltMask);
                const int nGive = __popcll(giveMask), nIdle = __popcll(idleMask);
                if (canGive) L.giveList[rg] = lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool thief = !busy && ri < nGive;
#ifdef RT_DEV_STATS
                if (lane == 0) { dsSteal += 1; dsStolen += nGive < nIdle ? nGive : nIdle; }
#endif
                const int v = L.giveList[thief ? ri : 0];
                const int vsb = __shfl(sb, v, 64);
                const int vowner = __shfl(owner, v, 64);
                const float vox = __shfl(o.x, v, 64), voy = __shfl(o.y, v, 64), voz = __shfl(o.z, v, 64);
                const float vdx = __shfl(d.x, v, 64), vdy = __shfl(d.y, v, 64), vdz = __shfl(d.z, v, 64);
                const float vix = __shfl(inv.x, v, 64), viy = __shfl(inv.y, v, 64), viz = __shfl(inv.z, v, 64);
                const float vtmax = __shfl(rtmax, v, 64);
                if (thief) {
                    cur = L.stack[vsb][v];
                    owner = vowner;
                    o = f3(vox, voy, voz); d = f3(vdx, vdy, vdz); inv = f3(vix, viy, viz);
                    rtmax = vtmax;
                    sp = 0; sb = 0;
                    busy = true;
                    tlim = __uint_as_float((uint32_t)(L.best[owner] >> 32));
                }
                if (canGive && rg < nIdle) ++sb;
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long nowBusy = __ballot(busy);
        const unsigned long long nodeMask = __ballot(busy && cur >= 0);
        const int nBusy = __popcll(nowBusy), nNode = __popcll(nodeMask);
        const bool nodeStep = nNode > 0 && (nBusy - nNode) * 128 < leafPct * nBusy;
        bool done = false;
        if (nodeStep) {
            // ---- inner node ------------------------------------------------------------------------
            if (busy && cur >= 0) {
                const float4* np = node_ptr(sv, nodeBase + (uint32_t)cur);
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                if (COUNT) wb += 2;
#ifdef RT_DEV_STATS
                dsNodeIt += 1.0f / (float)__popcll(__ballot(true));
#endif
                float t0a, t0b;
                const bool ha = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, tlim, t0a);
                const bool hb = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, tlim, t0b);
                const int ca = __float_as_int(q3.x), cb = __float_as_int(q3.y);
                if (ha && hb) {
                    const bool aFirst = t0a <= t0b;
                    cur = aFirst ? ca : cb;
                    const int far = aFirst ? cb : ca;
                    if (sp < kLdsStack) L.stack[sp][lane] = far;
                    else spill[sp - kLdsStack] = far;
                    ++sp;
                } else if (ha) {
                    cur = ca;
                } else if (hb) {
                    cur = cb;
                } else if (sp == sb) {
                    done = true;
                } else {
                    --sp;
                    cur = L.stack[sp < kLdsStack ? sp : 0][lane];
                    asm volatile("" : "+v"(cur));
                    if (sp >= kLdsStack) cur = spill[sp - kLdsStack];
                }
            }
        } else if (busy && cur < 0) {
            // ---- triangle leaf -----------------------------------------------------------------------
            const uint32_t leaf = (uint32_t)(~cur);
            const uint32_t first = leaf >> 4, count = (leaf & 15u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t slot = triBase + first + k;
                const float4* tp = tri_ptr(sv, slot);
                const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
                float t, be, ga;
                if (COUNT) wt += 1;
#ifdef RT_DEV_STATS
                dsLeafIt += 1.0f / (float)__popcll(__ballot(true));
#endif
                if (tri_test(r0, r1, r2, o, d, t, be, ga)) {
                    if (ANY) {
                        if (t <= rtmax) { atomicMin(&L.best[owner], 0ull); }
                    } else if (t <= tlim) {
                        const unsigned long long mine = pack_hit(t, sv.triIds[slot]);
                        const unsigned long long old = atomicMin(&L.best[owner], mine);
                        if (mine < old && L.best[owner] == mine) {
                            L.paySlot[owner] = slot; L.payBeta[owner] = be; L.payGamma[owner] = ga;
                        }
                    }
                }
            }
            tlim = __uint_as_float((uint32_t)(L.best[owner] >> 32));
            if ((ANY && tlim == 0.0f && L.best[owner] == 0ull) || sp == sb) {
                done = true;
            } else {
                --sp;
                cur = L.stack[sp < kLdsStack ? sp : 0][lane];
                asm volatile("" : "+v"(cur));
                if (sp >= kLdsStack) cur = spill[sp - kLdsStack];
            }
        }
        if (done) {
            busy = false;
            sp = 0; sb = 0;
            if (COUNT) { atomicAdd(&L.cnt[0][owner], wb); atomicAdd(&L.cnt[1][owner], wt); wb = 0; wt = 0; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    TraceResult r;
    const unsigned long long b = L.best[lane];
    if (ANY) {
        r.t = tmax; r.slot = kInvalid; r.beta = 0.0f; r.gamma = 0.0f;
        r.inst = (b == 0ull && valid) ? 0u : kInvalid;
    } else {
        const bool hit = L.paySlot[lane] != kInvalid;
        r.t = hit ? __uint_as_float((uint32_t)(b >> 32)) : INFINITY;
        r.slot = L.paySlot[lane];
        r.inst = hit ? 0u : kInvalid;
        r.beta = L.payBeta[lane];
        r.gamma = L.payGamma[lane];
    }
    if (COUNT) { nBoxes = L.cnt[0][lane]; nTris = L.cnt[1][lane]; }
#ifdef RT_DEV_STATS
    // developer build: lanes 0..5 report wave-level schedule statistics in place of their tri counts
    if (COUNT) {
        __shared__ float ds[8];
        if (lane < 8) ds[lane] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        atomicAdd(&ds[0], dsOuter); atomicAdd(&ds[1], dsSteal); atomicAdd(&ds[2], dsNodeIt);
        atomicAdd(&ds[3], dsLeafIt); atomicAdd(&ds[4], dsBusy); atomicAdd(&ds[5], dsStolen);
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) nTris = (uint32_t)(ds[lane] + 0.5f);
        if (lane == 6) nTris = dsT0;
        if (lane == 7) nTris = (uint32_t)wall_clock64();
    }
#endif
    return r;
}

// ---- persistent waves over a ray queue --------------------------------------------------------
//
// trace_wave_steal() gives every lane one ray and lets idle lanes steal; the wave still lives as
// long as its heaviest ray, and at the end of a launch the chip drains while a few long-lived
// waves finish.  trace_wave_queue() keeps a fixed number of resident waves alive for the whole
// launch instead: a lane that runs out of work first takes the next ray of the launch (chunks of
// 64 ray indices are reserved from a few global counters), and only when the launch has no rays
// left do idle lanes steal subtrees from their neighbours -- so the last rays of the launch are
// traversed by all 64 lanes of their wave together.
//
// A ray in flight occupies one of 64 per-wave slots (best word, winning leaf slot, ray index and
// the number of lanes currently working for it).  The lane that drops a slot's count to zero
// hands the ray's result to `finish`.  The rays of a reserved chunk are loaded into registers
// (lane k holds ray k of the chunk) as soon as the chunk is reserved and reach the lanes that take
// them through shuffles, so that an iteration never waits for more than its one node or triangle
// fetch.

struct RayQueue {
    uint32_t* counters;    // nCounters words, 64 words apart, zero before the launch
    uint32_t nRays;
    uint32_t firstQueued;  // rays below this index are assigned statically: wave w starts with [64 w, 64 w + 64)
    uint32_t rangeSize;    // ray indices handed out per counter (a multiple of 64)
    uint32_t nCounters;
    int32_t fetchMin;      // idle lanes a wave needs before it takes new rays
    uint32_t xcdStride;    // waves per XCD (grid / 8), 0 = no remapping: workgroup w runs on XCD w % 8, and its
                           // first chunk is chosen so that every XCD starts on one contiguous run of rays
};

template <bool COUNT>
struct QueueLds {
    int stack[kLdsStack][64];
    unsigned long long best[64];
    uint32_t paySlot[64];
    float payBeta[64];
    float payGamma[64];
    uint32_t rayIdx[64];
    int refs[64];
    int list[64];
    uint32_t cnt[COUNT ? 2 : 1][COUNT ? 64 : 1];
};

// Reserves the next chunk of ray indices [next, end); false when the launch has none left.
__device__ __forceinline__ bool queue_reserve(const RayQueue& q, uint32_t& kc, uint32_t& tries, const int lane,
                                              uint32_t& next, uint32_t& end)
{
    while (tries < q.nCounters) {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&q.counters[kc * 64u], 64u);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        if (v < q.rangeSize) {
            const uint32_t base = q.firstQueued + kc * q.rangeSize + v;
            if (base < q.nRays) {
                next = base;
                end = base + 64u < q.nRays ? base + 64u : q.nRays;
                return true;
            }
        }
        // stay on this XCD's counters (k, k + 8, ...) as long as they last, then move to the next XCD's
        kc = kc + 8u < q.nCounters ? kc + 8u : ((kc & 7u) + 1u) & 7u;
        ++tries;
    }
    return false;
}

// fetch(ray, o, d, tmax) loads world-space ray `ray`; finish(ray, result, nBoxes, nTris) stores its
// result (ANY: result.inst != kInvalid means occluded).  Every lane of the wave must call.
template <bool ANY, bool COUNT, class Fetch, class Finish>
__device__ __forceinline__ void trace_wave_queue(const SceneView& sv, const RayQueue& q, QueueLds<COUNT>& L, const int lane,
                                                 Fetch fetch, Finish finish, const int minIdle, const int leafPct
#ifdef RT_DEV_STATS
                                                 , uint32_t* devOut = nullptr
#endif
                                                 )
{
    const DevInstance& in = sv.inst[0];
    const uint32_t nodeBase = in.root, triBase = in.triBase;

    L.refs[lane] = 0;
    const uint32_t chunk0 = q.xcdStride ? (blockIdx.x & 7u) * q.xcdStride + (blockIdx.x >> 3) : blockIdx.x;
    uint32_t poolNext = chunk0 * 64u < q.firstQueued ? chunk0 * 64u : q.firstQueued;
    uint32_t poolEnd = poolNext + 64u < q.firstQueued ? poolNext + 64u : q.firstQueued;
    uint32_t chunkBase = poolNext;
    // counters are spread over the XCDs the same way: XCD x uses counters x, x + 8, ...
    uint32_t kc = blockIdx.x % q.nCounters, tries = 0;
    bool queueDone = q.firstQueued >= q.nRays;

    // lane k's copy of ray chunkBase + k (world space), loaded when the chunk is reserved
    F3 po = f3(0, 0, 0), pd = f3(0, 0, 1);
    float ptmax = 0.0f;
    if (poolNext + (uint32_t)lane < poolEnd) fetch(poolNext + (uint32_t)lane, po, pd, ptmax);

    F3 o = f3(0, 0, 0), d = f3(0, 0, 1), inv = f3(0, 0, 1);
    int owner = lane;
    int cur = 0;
    int sp = 0, sb = 0;
    bool busy = false;
    float tlim = INFINITY, rtmax = INFINITY;
    uint32_t wb = 0, wt = 0;
    int spill[kStackDepth - kLdsStack];
    const unsigned long long ltMask = (1ull << lane) - 1ull;
#ifdef RT_DEV_STATS
    float dsOuter = 0, dsSteal = 0, dsNodeIt = 0, dsLeafIt = 0, dsBusy = 0, dsStolen = 0;
    const uint32_t dsT0 = (uint32_t)wall_clock64();
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (;;) {
        const unsigned long long busyMask = __ballot(busy);
        const int nIdle0 = 64 - __popcll(busyMask);
        if (poolNext == poolEnd && !queueDone) {
            // the next chunk is reserved (and its rays requested) as soon as the last one is used up
            if (queue_reserve(q, kc, tries, lane, poolNext, poolEnd)) {
                chunkBase = poolNext;
                if (poolNext + (uint32_t)lane < poolEnd) fetch(poolNext + (uint32_t)lane, po, pd, ptmax);
            } else {
                queueDone = true;
            }
        }
        if (busyMask == 0ull && poolNext == poolEnd) break;
#ifdef RT_DEV_STATS
        if (lane == 0) { dsOuter += 1; dsBusy += __popcll(busyMask); }
#endif
        if (poolNext != poolEnd) {
            // ---- new rays for idle lanes ---------------------------------------------------------
            if (nIdle0 >= q.fetchMin || busyMask == 0ull) {
                const unsigned long long idleMask = ~busyMask;
                const unsigned long long freeMask = __ballot(L.refs[lane] == 0);
                if (L.refs[lane] == 0) L.list[__popcll(freeMask & ltMask)] = lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int ri = __popcll(idleMask & ltMask);
                const uint32_t avail = poolEnd - poolNext;
                const bool take = !busy && (uint32_t)ri < avail;
                const int src = take ? (int)(poolNext - chunkBase) + ri : 0;
                F3 wo, wd;
                wo.x = __shfl(po.x, src, 64); wo.y = __shfl(po.y, src, 64); wo.z = __shfl(po.z, src, 64);
                wd.x = __shfl(pd.x, src, 64); wd.y = __shfl(pd.y, src, 64); wd.z = __shfl(pd.z, src, 64);
                const float tmax = ANY ? __shfl(ptmax, src, 64) : INFINITY;
                if (take) {
                    const uint32_t ray = poolNext + (uint32_t)ri;
                    const int slot = L.list[ri];
                    // rays with a NaN component find nothing (see trace_ray)
                    const bool valid = (wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z) == (wo.x + wo.y + wo.z) + (wd.x + wd.y + wd.z);
                    if (valid) {
                        xf_ray_inv(in.inv, sv.identityXf != 0, wo, wd, o, d);
                        inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        owner = slot;
                        L.best[slot] = pack_hit(ANY ? tmax : INFINITY, kInvalid);
                        L.paySlot[slot] = kInvalid;
                        L.rayIdx[slot] = ray;
                        L.refs[slot] = 1;
                        if (COUNT) { L.cnt[0][slot] = 0; L.cnt[1][slot] = 0; }
                        cur = 0; sp = 0; sb = 0;
                        tlim = ANY ? tmax : INFINITY;
                        rtmax = tmax;
                        busy = true;
                    } else {
                        TraceResult r;
                        r.t = ANY ? tmax : INFINITY; r.slot = kInvalid; r.inst = kInvalid; r.beta = 0.0f; r.gamma = 0.0f; r.prim = kInvalid;
                        finish(ray, r, 0u, 0u);
                    }
                }
                const uint32_t nTake = (uint32_t)nIdle0 < avail ? (uint32_t)nIdle0 : avail;
                poolNext += nTake;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else if (nIdle0 >= minIdle) {
            // ---- steal phase (the launch has no rays left for this wave) ---------------------------
            const unsigned long long idleMask = ~busyMask;
            const bool canGive = busy && sb < sp && sb < kLdsStack;
            const unsigned long long giveMask = __ballot(canGive);
            if (giveMask != 0ull) {
                const int rg = __popcll(giveMask & ltMask), ri = __popcll(idleMask & ltMask);
                const int nGive = __popcll(giveMask), nIdle = __popcll(idleMask);
                if (canGive) L.list[rg] = lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool thief = !busy && ri < nGive;
#ifdef RT_DEV_STATS
                if (lane == 0) { dsSteal += 1; dsStolen += nGive < nIdle ? nGive : nIdle; }
#endif
                const int v = L.list[thief ? ri : 0];
                const int vsb = __shfl(sb, v, 64);
                const int vowner = __shfl(owner, v, 64);
                const float vox = __shfl(o.x, v, 64), voy = __shfl(o.y, v, 64), voz = __shfl(o.z, v, 64);
                const float vdx = __shfl(d.x, v, 64), vdy = __shfl(d.y, v, 64), vdz = __shfl(d.z, v, 64);
                const float vix = __shfl(inv.x, v, 64), viy = __shfl(inv.y, v, 64), viz = __shfl(inv.z, v, 64);
                const float vtmax = __shfl(rtmax, v, 64);
                if (thief) {
                    cur = L.stack[vsb][v];
                    owner = vowner;
                    o = f3(vox, voy, voz); d = f3(vdx, vdy, vdz); inv = f3(vix, viy, viz);
                    rtmax = vtmax;
                    sp = 0; sb = 0;
                    busy = true;
                    tlim = __uint_as_float((uint32_t)(L.best[owner] >> 32));
                    atomicAdd(&L.refs[owner], 1);
                }
                if (canGive && rg < nIdle) ++sb;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long nowBusy = __ballot(busy);
        const unsigned long long nodeMask = __ballot(busy && cur >= 0);
        const int nBusy = __popcll(nowBusy), nNode = __popcll(nodeMask);
        const bool nodeStep = nNode > 0 && (nBusy - nNode) * 128 < leafPct * nBusy;
        bool done = false;
        if (nodeStep) {
            // ---- inner node ------------------------------------------------------------------------
            if (busy && cur >= 0) {
                const float4* np = node_ptr(sv, nodeBase + (uint32_t)cur);
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                if (COUNT) wb += 2;
#ifdef RT_DEV_STATS
                dsNodeIt += 1.0f / (float)__popcll(__ballot(true));
#endif
                float t0a, t0b;
                const bool ha = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, tlim, t0a);
                const bool hb = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, tlim, t0b);
                const int ca = __float_as_int(q3.x), cb = __float_as_int(q3.y);
                if (ha && hb) {
                    const bool aFirst = t0a <= t0b;
                    cur = aFirst ? ca : cb;
                    const int far = aFirst ? cb : ca;
                    if (sp < kLdsStack) L.stack[sp][lane] = far;
                    else spill[sp - kLdsStack] = far;
                    ++sp;
                } else if (ha) {
                    cur = ca;
                } else if (hb) {
                    cur = cb;
                } else if (sp == sb) {
                    done = true;
                } else {
                    --sp;
                    cur = L.stack[sp < kLdsStack ? sp : 0][lane];
                    asm volatile("" : "+v"(cur));
                    if (sp >= kLdsStack) cur = spill[sp - kLdsStack];
                }
            }
        } else if (busy && cur < 0) {
            // ---- triangle leaf -----------------------------------------------------------------------
            const uint32_t leaf = (uint32_t)(~cur);
            const uint32_t first = leaf >> 4, count = (leaf & 15u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t slot = triBase + first + k;
                const float4* tp = tri_ptr(sv, slot);
                const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
                float t, be, ga;
                if (COUNT) wt += 1;
#ifdef RT_DEV_STATS
                dsLeafIt += 1.0f / (float)__popcll(__ballot(true));
#endif
                if (tri_test(r0, r1, r2, o, d, t, be, ga)) {
                    if (ANY) {
                        if (t <= rtmax) { atomicMin(&L.best[owner], 0ull); }
                    } else if (t <= tlim) {
                        const unsigned long long mine = pack_hit(t, sv.triIds[slot]);
                        const unsigned long long old = atomicMin(&L.best[owner], mine);
                        if (mine < old && L.best[owner] == mine) { L.paySlot[owner] = slot; L.payBeta[owner] = be; L.payGamma[owner] = ga; }
                    }
                }
            }
            tlim = __uint_as_float((uint32_t)(L.best[owner] >> 32));
            if ((ANY && tlim == 0.0f && L.best[owner] == 0ull) || sp == sb) {
                done = true;
            } else {
                --sp;
                cur = L.stack[sp < kLdsStack ? sp : 0][lane];
                asm volatile("" : "+v"(cur));
                if (sp >= kLdsStack) cur = spill[sp - kLdsStack];
            }
        }
        if (done) {
            busy = false;
            sp = 0; sb = 0;
            if (COUNT) { atomicAdd(&L.cnt[0][owner], wb); atomicAdd(&L.cnt[1][owner], wt); wb = 0; wt = 0; }
            if (atomicSub(&L.refs[owner], 1) == 1) {
                // last lane working for this ray: assemble and store its result
                const unsigned long long b = L.best[owner];
                TraceResult r;
                r.slot = kInvalid; r.beta = 0.0f; r.gamma = 0.0f; r.prim = (uint32_t)b;
                if (ANY) {
                    r.t = rtmax;
                    r.inst = b == 0ull ? 0u : kInvalid;
                } else {
                    const uint32_t ps = L.paySlot[owner];
                    r.t = INFINITY; r.inst = kInvalid;
                    if (ps != kInvalid) {
                        r.t = __uint_as_float((uint32_t)(b >> 32));
                        r.slot = ps;
                        r.inst = 0u;
                        r.beta = L.payBeta[owner];
                        r.gamma = L.payGamma[owner];
                    }
                }
                finish(L.rayIdx[owner], r, COUNT ? L.cnt[0][owner] : 0u, COUNT ? L.cnt[1][owner] : 0u);
            }
        }
    }
#ifdef RT_DEV_STATS
    if (COUNT && devOut) {
        // developer build: the wave's schedule statistics, 8 words per wave behind the ray counts
        __shared__ float ds[8];
        if (lane < 8) ds[lane] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        atomicAdd(&ds[0], dsOuter); atomicAdd(&ds[1], dsSteal); atomicAdd(&ds[2], dsNodeIt);
        atomicAdd(&ds[3], dsLeafIt); atomicAdd(&ds[4], dsBusy); atomicAdd(&ds[5], dsStolen);
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) devOut[blockIdx.x * 8u + lane] = (uint32_t)(ds[lane] + 0.5f);
        if (lane == 6) devOut[blockIdx.x * 8u + lane] = dsT0;
        if (lane == 7) devOut[blockIdx.x * 8u + lane] = (uint32_t)wall_clock64();
    }
#endif
}

} // namespace rt
