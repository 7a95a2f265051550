// rt_trace.hip -- batch closest-hit / any-hit kernels and the primary-ray generator (gfx950).
//
// Launch shape: one wave64 per workgroup (traversal has no intra-workgroup cooperation, so the
// smallest workgroup gives the dispatcher the finest grain and frees the 8 KB LDS stack of a
// finished wave at once), one ray per lane, grid = ceil(n / 64).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "rt_rng_inl.h"
#include "rt_steal_inl.h"
#include "rt_trace_inl.h"

// waves per SIMD of the one-mesh batch kernels (see RT_TRACE_WAVES in rt_integrator.hip; bounce rays 6420 -> 6540 Mrays/s)
#ifndef RT_BATCH_WAVES
#define RT_BATCH_WAVES 8
#endif

namespace rt {

// Launch position -> ray index.  A wave's 64 positions are spread over 8 runs of 8 consecutive
// rays taken from 8 evenly spaced places of the batch: expensive rays cluster in ray order (the
// silhouette rows of an image), and a wave made of one such run would outlive all others; mixed
// like this every wave gets its share.  8 consecutive rays still load as one 96-byte run.
// nb8 = number of 8-ray blocks that take part (a multiple of 8); positions behind them map to themselves.
__device__ __forceinline__ uint64_t interleave8(uint64_t p, uint64_t nb8)
{
    if (p >= nb8 * 8) return p;
    const uint64_t c = p >> 6, g = (p >> 3) & 7u, w = p & 7u;
    return ((c + g * (nb8 >> 3)) << 3) + w;
}
static inline uint64_t interleave_blocks(uint64_t n) { return (n >> 3) & ~7ull; }

template <bool INSTANCED, bool COUNT>
__global__ __launch_bounds__(64) void trace_closest_kernel(SceneView sv, const float* __restrict__ ro,
                                                           const float* __restrict__ rd, uint64_t n,
                                                           rtgpu_hit* __restrict__ out, uint32_t* __restrict__ counts)
{
    __shared__ int stack[kLdsStack][64];
    const int lane = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    F3 o = f3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]);
    F3 d = f3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    uint32_t nb = 0, nt = 0;
    TraceResult r;
    if (INSTANCED && sv.refTop) r = trace_ray_reftop_closest<COUNT>(sv, o, d, stack, lane, nb, nt);
    else r = trace_ray<INSTANCED, false, COUNT>(sv, o, d, INFINITY, stack, lane, nb, nt);
    rtgpu_hit h;
    h.t = r.t;
    h.prim = r.slot != kInvalid ? sv.triIds[r.slot] : kInvalid;
    h.inst = r.inst;
    h.beta = r.beta;
    h.gamma = r.gamma;
    out[i] = h;
    if (COUNT) {
        counts[2 * i] = nb;
        counts[2 * i + 1] = nt;
    }
}

// Single-mesh scenes: wave-cooperative traversal with work stealing (rt_steal_inl.h).
template <bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_BATCH_WAVES))) void trace_closest_steal_kernel(SceneView sv, const float* __restrict__ ro,
                                                                 const float* __restrict__ rd, uint64_t n, uint64_t nb8,
                                                                 rtgpu_hit* __restrict__ out, uint32_t* __restrict__ counts)
{
    __shared__ StealLds<COUNT> lds;
    const int lane = threadIdx.x;
    const uint64_t i = interleave8((uint64_t)blockIdx.x * 64 + lane, nb8);
    const bool has = i < n;
    const uint64_t j = has ? i : 0;
    const F3 o = f3(ro[3 * j], ro[3 * j + 1], ro[3 * j + 2]);
    const F3 d = f3(rd[3 * j], rd[3 * j + 1], rd[3 * j + 2]);
    uint32_t nb = 0, nt = 0;
    const TraceResult r = trace_wave_steal<false, COUNT>(sv, has, o, d, INFINITY, lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct);
    if (!has) return;
    rtgpu_hit h;
    h.t = r.t;
    h.prim = r.prim;
    h.inst = r.inst;
    h.beta = r.beta;
    h.gamma = r.gamma;
    out[i] = h;
    if (COUNT) {
        counts[2 * i] = nb;
        counts[2 * i + 1] = nt;
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_BATCH_WAVES))) void trace_any_steal_kernel(SceneView sv, const float* __restrict__ ro,
                                                             const float* __restrict__ rd, const float* __restrict__ tmax,
                                                             uint64_t n, uint64_t nb8, uint8_t* __restrict__ occ)
{
    __shared__ StealLds<false> lds;
    const int lane = threadIdx.x;
    const uint64_t i = interleave8((uint64_t)blockIdx.x * 64 + lane, nb8);
    const bool has = i < n;
    const uint64_t j = has ? i : 0;
    const F3 o = f3(ro[3 * j], ro[3 * j + 1], ro[3 * j + 2]);
    const F3 d = f3(rd[3 * j], rd[3 * j + 1], rd[3 * j + 2]);
    uint32_t nb = 0, nt = 0;
    const TraceResult r = trace_wave_steal<true, false>(sv, has, o, d, tmax[j], lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct);
    if (has) occ[i] = r.inst != kInvalid ? 1 : 0;
}

// Instanced scenes without spheres: the cooperative traversal over top-level and mesh trees.
template <bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void trace_closest_inst_steal_kernel(SceneView sv, const float* __restrict__ ro,
                                                                      const float* __restrict__ rd, uint64_t n, uint64_t nb8,
                                                                      rtgpu_hit* __restrict__ out, uint32_t* __restrict__ counts)
{
    __shared__ StealLdsInst<COUNT> lds;
    const int lane = threadIdx.x;
    const uint64_t i = interleave8((uint64_t)blockIdx.x * 64 + lane, nb8);
    const bool has = i < n;
    const uint64_t j = has ? i : 0;
    const F3 o = f3(ro[3 * j], ro[3 * j + 1], ro[3 * j + 2]);
    const F3 d = f3(rd[3 * j], rd[3 * j + 1], rd[3 * j + 2]);
    uint32_t nb = 0, nt = 0;
    const TraceResult r = trace_wave_steal_inst<false, COUNT>(sv, has, o, d, INFINITY, lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct);
    if (!has) return;
    rtgpu_hit h;
    h.t = r.t;
    h.prim = r.prim;
    h.inst = r.inst;
    h.beta = r.beta;
    h.gamma = r.gamma;
    out[i] = h;
    if (COUNT) {
        counts[2 * i] = nb;
        counts[2 * i + 1] = nt;
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void trace_any_inst_steal_kernel(SceneView sv, const float* __restrict__ ro,
                                                                  const float* __restrict__ rd, const float* __restrict__ tmax,
                                                                  uint64_t n, uint64_t nb8, uint8_t* __restrict__ occ)
{
    __shared__ StealLdsInst<false> lds;
    const int lane = threadIdx.x;
    const uint64_t i = interleave8((uint64_t)blockIdx.x * 64 + lane, nb8);
    const bool has = i < n;
    const uint64_t j = has ? i : 0;
    const F3 o = f3(ro[3 * j], ro[3 * j + 1], ro[3 * j + 2]);
    const F3 d = f3(rd[3 * j], rd[3 * j + 1], rd[3 * j + 2]);
    uint32_t nb = 0, nt = 0;
    const TraceResult r = trace_wave_steal_inst<true, false>(sv, has, o, d, tmax[j], lds, lane, nb, nt, sv.stealMinIdle, sv.stealLeafPct);
    if (has) occ[i] = r.inst != kInvalid ? 1 : 0;
}

template <bool INSTANCED>
__global__ __launch_bounds__(64) void trace_any_kernel(SceneView sv, const float* __restrict__ ro,
                                                       const float* __restrict__ rd, const float* __restrict__ tmax,
                                                       uint64_t n, uint8_t* __restrict__ occ)
{
    __shared__ int stack[kLdsStack][64];
    const int lane = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * 64 + lane;
    if (i >= n) return;
    F3 o = f3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]);
    F3 d = f3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    uint32_t nb = 0, nt = 0;
    bool hit;
    if (INSTANCED && sv.refTop) hit = trace_ray_reftop_any(sv, o, d, tmax[i], stack, lane);
    else hit = trace_ray<INSTANCED, true, false>(sv, o, d, tmax[i], stack, lane, nb, nt).inst != kInvalid;
    occ[i] = hit ? 1 : 0;
}

// PerspectiveCamera::generateRay (camera/PerspectiveCamera.cpp:26-52).
__global__ __launch_bounds__(256) void generate_primary_kernel(rtgpu_camera cam, rtgpu_render_settings st, int sampleIndex,
                                                               float* __restrict__ ro, float* __restrict__ rd)
{
    const int tw = st.x_end - st.x_start, th = st.y_end - st.y_start;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= (uint32_t)(tw * th)) return;
    const int x = st.x_start + (int)(k % (uint32_t)tw), y = st.y_start + (int)(k / (uint32_t)tw);
    F3 o, d;
    camera_ray(cam, st, x, y, sampleIndex, o, d);
    ro[3 * k] = o.x; ro[3 * k + 1] = o.y; ro[3 * k + 2] = o.z;
    rd[3 * k] = d.x; rd[3 * k + 1] = d.y; rd[3 * k + 2] = d.z;
}

static int env_int(const char* name, int def)
{
    const char* e = getenv(name);
    return e ? atoi(e) : def;
}

hipError_t launch_trace_closest(const SceneView& sv, const float* dO, const float* dD, uint64_t n,
                                rtgpu_hit* dHits, uint32_t* dCounts, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    // RTGPU_TRACE_VARIANT=-1 forces the per-lane walk (one lane per ray, no cooperation): the comparison point of the
    // cooperative kernels and a second implementation of the same result definition for the determinism checks
    static const int variant = env_int("RTGPU_TRACE_VARIANT", 0);
    const uint64_t blocks = (n + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block(64);
    // Work counters are those of a ray traversed on its own (near child first, pruned by its own
    // hits): a figure of the ray and the tree.  The cooperative kernel tests more boxes than that
    // -- stolen subtrees are entered before the owner's nearer hit is known -- and its own count
    // (developer statistics) is only reported on request.
#ifdef RT_DEV_STATS
    static const bool countCooperative = env_int("RTGPU_COUNT_COOPERATIVE", 1) != 0;
#else
    static const bool countCooperative = env_int("RTGPU_COUNT_COOPERATIVE", 0) != 0;
#endif
    if (sv.singleMesh && variant != -1 && (!dCounts || countCooperative)) {
        const uint64_t nb8 = interleave_blocks(n);
        if (dCounts) hipLaunchKernelGGL((trace_closest_steal_kernel<true>), grid, block, 0, stream, sv, dO, dD, n, nb8, dHits, dCounts);
        else hipLaunchKernelGGL((trace_closest_steal_kernel<false>), grid, block, 0, stream, sv, dO, dD, n, nb8, dHits, dCounts);
        return hipGetLastError();
    }
    if (sv.coopInstanced && variant != -1 && !dCounts) {
        hipLaunchKernelGGL((trace_closest_inst_steal_kernel<false>), grid, block, 0, stream, sv, dO, dD, n, interleave_blocks(n), dHits, dCounts);
        return hipGetLastError();
    }
    if (sv.singleMesh) {
        if (dCounts) hipLaunchKernelGGL((trace_closest_kernel<false, true>), grid, block, 0, stream, sv, dO, dD, n, dHits, dCounts);
        else hipLaunchKernelGGL((trace_closest_kernel<false, false>), grid, block, 0, stream, sv, dO, dD, n, dHits, dCounts);
    } else {
        if (dCounts) hipLaunchKernelGGL((trace_closest_kernel<true, true>), grid, block, 0, stream, sv, dO, dD, n, dHits, dCounts);
        else hipLaunchKernelGGL((trace_closest_kernel<true, false>), grid, block, 0, stream, sv, dO, dD, n, dHits, dCounts);
    }
    return hipGetLastError();
}

hipError_t launch_trace_any(const SceneView& sv, const float* dO, const float* dD, const float* dTmax,
                            uint64_t n, uint8_t* dOcc, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block(64);
    static const int variant = env_int("RTGPU_TRACE_VARIANT", 0);
    if (sv.singleMesh && variant != -1) hipLaunchKernelGGL(trace_any_steal_kernel, grid, block, 0, stream, sv, dO, dD, dTmax, n, interleave_blocks(n), dOcc);
    else if (sv.coopInstanced && variant != -1) hipLaunchKernelGGL(trace_any_inst_steal_kernel, grid, block, 0, stream, sv, dO, dD, dTmax, n, interleave_blocks(n), dOcc);
    else if (sv.singleMesh) hipLaunchKernelGGL((trace_any_kernel<false>), grid, block, 0, stream, sv, dO, dD, dTmax, n, dOcc);
    else hipLaunchKernelGGL((trace_any_kernel<true>), grid, block, 0, stream, sv, dO, dD, dTmax, n, dOcc);
    return hipGetLastError();
}

hipError_t launch_generate_primary(const rtgpu_camera& cam, const rtgpu_render_settings& st, int32_t sampleIndex,
                                   float* dO, float* dD, hipStream_t stream)
{
    const int64_t n = (int64_t)(st.x_end - st.x_start) * (st.y_end - st.y_start);
    if (n <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(generate_primary_kernel, grid, block, 0, stream, cam, st, (int)sampleIndex, dO, dD);
    return hipGetLastError();
}

} // namespace rt
