// rt_wave_inl.h -- what the kernels of the integrator and of the photon tracer share: closest / any hit for the
// rays of one wave (cooperative traversal where the scene allows it) and the shading normal of a hit.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_shade_inl.h"
#include "rt_steal_inl.h"

namespace rt {

// How a kernel finds hits: the wave-cooperative walk over one mesh, the wave-cooperative walk over mesh instances
// (what a file of many objects gives), or the per-lane walks (scenes with analytic spheres, whose top-level tree is the
// reference's own, and scenes whose tie-break keys do not fit 32 bits).  A template parameter, so that every kernel
// variant only carries the registers and the LDS of its own walk.
enum TraceMode : int { kTraceSingle = 0, kTraceInstances = 1, kTraceLanes = 2 };
inline int trace_mode(const SceneView& sv) { return sv.singleMesh ? kTraceSingle : (sv.coopInstanced ? kTraceInstances : kTraceLanes); }

// Per-wave LDS of a traversal kernel.
template <int MODE>
struct TraceLds {
    struct type { int stack[kLdsStack][64]; };   // per-lane stacks
};
template <>
struct TraceLds<kTraceSingle> {
    using type = StealLds<false>;
};
template <>
struct TraceLds<kTraceInstances> {
    using type = StealLdsInst<false>;
};

// Closest hit for the wave's rays (every lane calls; lanes without a ray pass has = false).
template <int MODE>
__device__ __forceinline__ TraceResult wave_closest(const SceneView& sv, bool has, F3 o, F3 d, typename TraceLds<MODE>::type& lds, int lane)
{
    uint32_t nb = 0, nt = 0;
    if constexpr (MODE == kTraceSingle) {
        return trace_wave_steal<false, false>(sv, has, o, d, INFINITY, lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct);
    } else if constexpr (MODE == kTraceInstances) {
        return trace_wave_steal_inst<false, false>(sv, has, o, d, INFINITY, lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct);
    } else {
        TraceResult r;
        r.t = INFINITY; r.slot = kInvalid; r.inst = kInvalid; r.beta = 0.0f; r.gamma = 0.0f;
        if (!has) return r;
        if (sv.refTop) return trace_ray_reftop_closest<false>(sv, o, d, lds.stack, lane, nb, nt);
        return trace_ray<true, false, false>(sv, o, d, INFINITY, lds.stack, lane, nb, nt);
    }
}
template <int MODE>
__device__ __forceinline__ bool wave_any(const SceneView& sv, bool has, F3 o, F3 d, float tmax, typename TraceLds<MODE>::type& lds, int lane)
{
    uint32_t nb = 0, nt = 0;
    if constexpr (MODE == kTraceSingle) {
        return trace_wave_steal<true, false>(sv, has, o, d, tmax, lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct).inst != kInvalid;
    } else if constexpr (MODE == kTraceInstances) {
        return trace_wave_steal_inst<true, false>(sv, has, o, d, tmax, lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct).inst != kInvalid;
    } else {
        if (!has) return false;
        if (sv.refTop) return trace_ray_reftop_any(sv, o, d, tmax, lds.stack, lane);
        return trace_ray<true, true, false>(sv, o, d, tmax, lds.stack, lane, nb, nt).inst != kInvalid;
    }
}

// What the shade stage needs of the triangle at a leaf slot: vertex normals and material id (SceneView::shadeRecs).
struct ShadeRec {
    float4 q0, q1, q2;
    __device__ __forceinline__ uint32_t material() const { return __float_as_uint(q2.y); }
};
// slot == kInvalid (a sphere, a miss): nothing is read, the material id is kInvalid
__device__ __forceinline__ ShadeRec load_shade_rec(const SceneView& sv, uint32_t slot)
{
    ShadeRec r;
    r.q0 = r.q1 = r.q2 = make_float4(0.0f, __uint_as_float(kInvalid), 0.0f, 0.0f);
    if (slot != kInvalid) {
        const float4* p = sv.shadeRecs + (size_t)slot * 3;
        r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2];
    }
    return r;
}

// World-space shading normal of a hit: interpolated vertex normals (shape/TriangleMesh.cpp:114-120)
// or the sphere's object-space hit point (shape/Sphere.cpp:52-53), then transformNormal + normalize
// (shape/list/InstancedModelList.cpp:129-130, math/Transformation.h:95-99).  `rec` is the record of `slot` (unused for spheres).
__device__ __forceinline__ F3 hit_normal(const DevInstance& in, F3 wo, F3 wd, float t, uint32_t slot, float beta, float gamma, const ShadeRec& rec)
{
    F3 objN;
    if (slot != kInvalid) {
        const float alfa = 1.0f - beta - gamma;
        const F3 na = f3(rec.q0.x, rec.q0.y, rec.q0.z), nb = f3(rec.q0.w, rec.q1.x, rec.q1.y), nc = f3(rec.q1.z, rec.q1.w, rec.q2.x);
        objN = ((alfa * na) + (beta * nb)) + (gamma * nc);
    } else {
        const F3 oo = xf_point_inv(in.inv, wo), od = xf_vector_inv(in.inv, wd);
        objN = oo + (t * od);
    }
    const double nx = objN.x, ny = objN.y, nz = objN.z;
    const double* m = in.inv;
    const F3 wn = f3((float)(((nx * m[0] + ny * m[4]) + nz * m[8]) + 0.0),
                     (float)(((nx * m[1] + ny * m[5]) + nz * m[9]) + 0.0),
                     (float)(((nx * m[2] + ny * m[6]) + nz * m[10]) + 0.0));
    return normalized3(wn);
}

} // namespace rt
