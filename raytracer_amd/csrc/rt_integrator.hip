// rt_integrator.hip -- the path-tracing integrator as a wavefront pipeline (gfx950).
//
// Restates RenderTileTask::execute / samplePath / calculatePathEnergy (renderer/Renderer.cpp:14-270)
// over queues instead of per-pixel recursion:
//
//   primary : one lane per (pixel, geometry sample i): camera ray + closest hit + the area-light
//             pre-test and the miss/environment case (Renderer.cpp:150-225)
//   shade   : one lane per live path vertex: material closure (Diffuse / Glossy / Emissive /
//             ConstMix), next-event estimation, Russian roulette; emits one extension ray or one
//             shadow ray, or ends the path.  The first shade pass of a batch runs one lane per
//             path (pixel, i, material sample j) and takes vertex 0 from the sample's primary hit
//   shadow  : any-hit query; an unoccluded path receives its pending contribution
//   extend  : closest-hit query for the extension rays (one launch carries both: trace_stage_kernel)
//   resolve : per pixel, sums the path values in the reference's order and writes the frame buffer
//
// Queues hold path indices and are filled with wave-aggregated appends (one atomic per wave:
// ballot + popcount + prefix).  Every path ends with at most one contribution (next-event
// estimation is terminal in the reference's estimator), which is stored per path and summed by
// `resolve` in a fixed order: the image does not depend on queue order, batch size or GPU count.
//
// The reference evaluates a finished path backwards (calculatePathEnergy); here the product of
// the per-vertex factors is carried forwards as a throughput.  The factors are the same; only
// the order of the float multiplications differs.
//
// Resampling (materialAA): the reference samples the path once, then re-samples it materialAA-1
// times from the first vertex whose material has variance.  Here every (i, j) is its own path
// from vertex 0 that shares the sample's primary hit; vertices before the first variance vertex
// draw their random numbers with j = 0, so all j make identical decisions there -- the same
// joint distribution as sampling that prefix once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_fastdiv.h"
#include "rt_scene.h"
#include "rt_shade_inl.h"
#include "rt_steal_inl.h"
#include "rt_photon_inl.h"
#include "rt_wave_inl.h"

// the first pass fits six waves per SIMD without a spill (78 VGPRs: shade stage 11.9 -> 11.6 ms per 1080p frame; seven spill
// nine registers, 11.7 ms); the later passes need 96 registers (six waves: 19 spills, 13.1 ms)
#ifndef RT_SHADE_WAVES_FIRST
#define RT_SHADE_WAVES_FIRST 6
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 5   // waves per SIMD of the shade kernels without glass / add / photon-map code
#endif
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256
#endif

// Waves per SIMD of the one-mesh trace stage.  Eight = 64 VGPRs (three spilled registers) and the whole 160 KB of LDS;
// since the kernel lost a tenth of its instructions it waits more than it issues, and the eighth wave pays
// (1080p trace stage 23.7 -> 23.15 ms, 49.8 M-triangle scene 6.82 -> 6.47 ms); with the earlier code it cost 9 %.
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 8
#endif
namespace rt {

namespace {

constexpr uint32_t ST_K_MASK = 63u;
constexpr uint32_t ST_PASSED_VAR = 1u << 6;
constexpr uint32_t ST_GLOSSY_PENDING = 1u << 7;
constexpr uint32_t ST_GLOSSY_ROUGH = 1u << 8;
constexpr uint32_t ST_RETRACE = 1u << 9;
constexpr uint32_t ST_GLASS_PENDING = 1u << 10;   // a glass vertex waits for the hit of its own ray
constexpr uint32_t ST_GLASS_INTERNAL = 1u << 11;  // ... and its transport direction points against the (flipped) normal
constexpr uint32_t ST_GLASS_STICKY = 1u << 12;    // ... and a visibility query along its other direction is in flight
constexpr uint32_t ST_DIFFUSE_SHIFT = 13;         // 4 bits: earlier vertices of the path with specularity < 0.8 (photon-map mode)
constexpr uint32_t ST_DIFFUSE_MASK = 15u << ST_DIFFUSE_SHIFT;
constexpr uint32_t ST_PREV_DIFFUSE = 1u << 17;    // caustics mode: the previous vertex was a diffuse bounce (it left a callback)
constexpr uint32_t ST_PREV_ROUGH = 1u << 18;      // caustics mode: the previous vertex was a bounce with specularity < 0.8

// An entry of the extension queue: the path in the low 25 bits, above them the vertex the next shade pass will sample
// for it (what that pass sorts its paths by before it has loaded anything of them).
constexpr uint32_t kQueuePathBits = 25;
constexpr uint32_t kQueuePathMask = (1u << kQueuePathBits) - 1u;
static_assert((1u << kQueuePathBits) == (uint32_t)RTGPU_MAX_PATHS_PER_PIXEL, "the path limit the header states is the queue entry's");
static_assert(kQueuePathBits + 7 <= 32 && 64 + 1 <= (1 << 7), "a queue entry holds the path and the next vertex (at most max_path_length = 64)");

// The extension and the shadow queue are cut into kQueueSegs segments, each with a length of its own: workgroup b of a
// shade pass appends to segment b % kQueueSegs, and the workgroups that read a queue take segment b % kQueueSegs, block
// b / kQueueSegs of it.  Atomics on one address are served one after the other (6-11 ns each on this chip), and a pass
// makes one per workgroup and queue: with a single length per queue that chain was as long as the pass itself.  A
// segment holds what the workgroups of one residue can append, segCap = 256 * ceil(workgroups / kQueueSegs) entries at
// most; a path leaves at most one extension entry behind, so a segment never grows from pass to pass.
#ifndef RT_QUEUE_SEGMENTS
#define RT_QUEUE_SEGMENTS 8
#endif
constexpr uint32_t kQueueSegs = RT_QUEUE_SEGMENTS;
constexpr uint32_t kCtrLine = 32;   // words (128 bytes) between two lengths that are added to at the same time
// one pass's counters: words 2 / 3 / 4 = photon queue length, prefix rays repeated per material sample, photon tail queue
// length; word kCtrExt + s * kCtrLine = length of extension segment s, kCtrShadow + s * kCtrLine of shadow segment s
constexpr uint32_t kCtrExt = kCtrLine, kCtrShadow = kCtrLine * (1 + kQueueSegs);
constexpr uint32_t kCtrStride = kCtrLine * (1 + 2 * kQueueSegs);

#ifdef RT_SHADE_STATS
// developer build: lanes and waves that reach marked places of the later shade passes (slot 2k = lanes, 2k + 1 = waves)
__device__ unsigned long long g_shade_stats[64];
__device__ unsigned long long g_shade_clocks[64][16];   // [workgroup mod 64][2k = cycles, 2k + 1 = waves]: spread so that the counting itself does not queue up
#ifdef RT_SHADE_CLOCKS_ONLY
#define SHADE_STAT(k, pred) do { } while (0)
#else
#define SHADE_STAT(k, pred) do { if (!FIRST) { const unsigned long long m_ = __ballot(pred); if (m_ != 0ull && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m_) - 1)) { atomicAdd(&g_shade_stats[2 * (k)], (unsigned long long)__popcll(m_)); atomicAdd(&g_shade_stats[2 * (k) + 1], 1ull); } } } while (0)
#endif
// ... and the shader-clock cycles a wave spends between two marks
#ifdef RT_SHADE_CLOCKS_FIRST   // the marks count in the first pass instead of in the later ones
#define SHADE_CLOCKED(first) (first)
#else
#define SHADE_CLOCKED(first) (!(first))
#endif
#define SHADE_T0(k) const unsigned long long clk0_##k = SHADE_CLOCKED(FIRST) ? clock64() : 0ull
#define SHADE_T1(k) do { if (SHADE_CLOCKED(FIRST)) shade_clock_add(k, clock64() - clk0_##k); } while (0)
__device__ __forceinline__ void shade_clock_add(int k, unsigned long long dt)
{
    // the lanes of a wave that are here read the clock in the same instruction: one of them reports
    const unsigned long long m = __ballot(true);
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1)) {
        atomicAdd(&g_shade_clocks[blockIdx.x & 63u][2 * k], dt);
        atomicAdd(&g_shade_clocks[blockIdx.x & 63u][2 * k + 1], 1ull);
    }
}
#else
#define SHADE_STAT(k, pred) do { } while (0)
#define SHADE_T0(k) do { } while (0)
#define SHADE_T1(k) do { } while (0)
#endif

// Path state lives in 16-byte records so that every access is one full-width load or store per
// lane (paths are reached through queues, i.e. at scattered indices: 4-byte fields in separate
// arrays would each cost a memory transaction of their own).
// One of the four 16-byte records every pass touches.  They share a 64-byte block per path (block[path][4]): a path
// that is reached through an index queue -- at a scattered address -- then costs one cache line, not four.
struct PathField {
    float4* base;
    __device__ __forceinline__ float4& operator[](size_t p) const { return base[p * 4]; }
};

struct Pool {
    PathField rayA;       // per path: origin of the pending ray / of the ray that produced the current hit; w = shadow-ray limit
    PathField rayB;       // per path: direction; w = state bits
    PathField hit;        // per path: t, leaf slot (bits), beta, gamma of the current vertex
    uint32_t* hitInst;    // per path: instance of the current vertex (scenes that are not a single mesh)
    PathField thr;        // per path: throughput
    float4* aux;          // per path: pending contribution (shadow ray) or the glossy vertex's hit point
    float4* result;       // per path
    // scenes with glass: normal used by the glass vertex (xyz) and the vertex's own t (w); the visibility
    // query along the direction the vertex did not take (origin + limit, direction + light index / answer)
    float4* glassN;
    float4* glassA;
    float4* glassB;
    // scenes with add materials: emission collected so far; emission of a vertex whose divisor is decided later
    float4* acc;
    float4* accPend;
    // per sample (pixel, i): the primary ray and its hit
    float4* sA;           // origin, t
    float4* sB;           // direction
    float4* sC;           // leaf slot (bits), instance (bits), beta, gamma
    uint8_t* sLive;       // 1: paths are spawned from this sample
    uint8_t* sHasVar;     // 1: the sample's path reached a vertex with variance (divide by materialAA)
    // queues
    uint32_t* qA;
    uint32_t* qB;
    uint32_t* qShadow;
    uint32_t* qPhoton;    // paths whose diffuse vertex reads the photon map (scenes with a photon map)
    uint32_t* qPhotonTail; // those of them whose gather ran out of its first budget
    // caustics mode: what the last diffuse bounce of a path would be worth as the end of the path --
    // hit point, incoming direction, throughput x colour, value collected before it
    float4* cHit;
    float4* cDir;
    float4* cThrColor;
    float4* cBase;
    uint32_t* counters;   // [pass][kCtrStride]
    uint32_t segCap;      // entries between the starts of two segments of qA / qB / qShadow
};

__device__ __forceinline__ float4 f4(float x, float y, float z, float w)
{
    float4 r;
    r.x = x; r.y = y; r.z = z; r.w = w;
    return r;
}

struct Batch {
    uint32_t pixelStart;  // first pixel of the batch, counted in the call's pixel order
    uint32_t nPixels;
    // Pixel order of the call.  tileSize == 0: the settings' tile row by row, results go to the frame buffer.
    // tileSize = T > 0: the settings' tile is cut into T x T squares, square k (row by row) belongs to rank k % nRanks,
    // a rank walks its squares in order and each square row by row; results go to a compact buffer in that order
    // (squares that stick out of the tile keep their full T x T slots, the outside pixels are written as zeros).
    uint32_t tileSize, nRanks, rank, tilesX, nTiles;
    uint32_t sqrtM;       // (int)sqrt((double)material_aa): the stratification grid of a sample's first variance vertex
    FastDiv divM, divG, divSqrtM;   // by material_aa, geometry_aa, sqrtM
    FastDiv divTw;        // by the tile's width
    FastDiv divTT, divT, divTilesX;   // tileSize > 0: by tileSize^2, tileSize, tilesX
};


// Position `idx` of the call's pixel order -> pixel (x, y).  Returns the pixel's index in the frame buffer (the key
// of its random numbers, so an image does not depend on how it was cut); `valid` is false for the padding of a square
// that sticks out of the tile and for squares behind the last one.
__device__ __forceinline__ uint32_t tile_pixel(const rtgpu_render_settings& st, const Batch& b, uint32_t idx, int& x, int& y, bool& valid)
{
    if (b.tileSize == 0u) {
        const uint32_t tw = (uint32_t)(st.x_end - st.x_start);
        const uint32_t row = fast_div(idx, b.divTw);
        x = st.x_start + (int)(idx - row * tw);
        y = st.y_start + (int)row;
        valid = true;
    } else {
        const uint32_t T = b.tileSize, tt = T * T;
        const uint32_t j = fast_div(idx, b.divTT), w = idx - j * tt;
        const uint32_t k = j * b.nRanks + b.rank;
        const uint32_t ky = fast_div(k, b.divTilesX), wy = fast_div(w, b.divT);
        x = st.x_start + (int)((k - ky * b.tilesX) * T + (w - wy * T));
        y = st.y_start + (int)(ky * T + wy);
        valid = k < b.nTiles && x < st.x_end && y < st.y_end;
    }
    return (uint32_t)(y * st.width + x);
}

// Appends `value` to the extension queue (predE) and / or the shadow queue (predS) with one atomic
// per queue and workgroup: the waves' counts are combined through LDS first.  Atomics on one
// address are served one after the other (about 11 ns each on this chip), so one atomic per wave
// would bound a shade pass by the number of its waves.  Every thread of the workgroup must call.
template <int BLOCK>
__device__ __forceinline__ void queue_push2(uint32_t* qE, uint32_t* cE, bool predE, uint32_t* qS, uint32_t* cS, bool predS, uint32_t value, uint32_t valueE)
{
    constexpr int WAVES = BLOCK / 64;
    __shared__ uint32_t cnt[2][WAVES];
    __shared__ uint32_t base[2];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const unsigned long long mE = __ballot(predE), mS = __ballot(predS);
    if (lane == 0) { cnt[0][wave] = (uint32_t)__popcll(mE); cnt[1][wave] = (uint32_t)__popcll(mS); }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t total = 0;
        for (int w = 0; w < WAVES; ++w) { const uint32_t c = cnt[threadIdx.x][w]; cnt[threadIdx.x][w] = total; total += c; }
        base[threadIdx.x] = total ? atomicAdd(threadIdx.x == 0 ? cE : cS, total) : 0u;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (predE) qE[base[0] + cnt[0][wave] + (uint32_t)__popcll(mE & lt)] = valueE;
    if (predS) qS[base[1] + cnt[1][wave] + (uint32_t)__popcll(mS & lt)] = value;
}

// The same for one queue.
template <int BLOCK>
__device__ __forceinline__ void queue_push1(uint32_t* q, uint32_t* c, bool pred, uint32_t value)
{
    constexpr int WAVES = BLOCK / 64;
    __shared__ uint32_t cnt[WAVES];
    __shared__ uint32_t base;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const unsigned long long m = __ballot(pred);
    if (lane == 0) cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < WAVES; ++w) { const uint32_t k = cnt[w]; cnt[w] = total; total += k; }
        base = total ? atomicAdd(c, total) : 0u;
    }
    __syncthreads();
    if (pred) q[base + cnt[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// ---- primary ---------------------------------------------------------------------------------
#ifndef RT_PRIMARY_WAVES
#define RT_PRIMARY_WAVES 7   // waves per SIMD of the one-mesh primary kernel (72 VGPRs; left alone it takes 84 = 5 waves: primary stage 2.24 -> 2.14 ms per 1080p frame)
#endif
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == kTraceSingle ? RT_PRIMARY_WAVES : 1))) void primary_kernel(SceneView sv, Pool pool, rtgpu_camera cam, rtgpu_render_settings st, Batch batch)
{
    __shared__ typename TraceLds<MODE>::type lds;
    const int lane = threadIdx.x;
    const uint32_t G = (uint32_t)st.geometry_aa, M = (uint32_t)st.material_aa;
    const uint32_t s0 = blockIdx.x * 64u + lane;
    const bool inBatch = s0 < batch.nPixels * G;
    const uint32_t s = inBatch ? s0 : 0u;
    const uint32_t pl = fast_div(s, batch.divG), i = s - pl * G;
    int x, y;
    bool valid;
    tile_pixel(st, batch, batch.pixelStart + pl, x, y, valid);
    const bool has = inBatch && valid;
    F3 o, d;
    camera_ray(cam, st, x, y, (int)i, o, d);
    const TraceResult r = wave_closest<MODE>(sv, has, o, d, lds, lane);
    if (!inBatch) return;
    if (!valid) {   // padding of the compact buffer: a black sample without paths
        pool.sLive[s] = 0;
        pool.sHasVar[s] = 0;
        pool.result[(size_t)s * M] = f4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const bool hit = r.inst != kInvalid;

    RGBf value = rgb(0, 0, 0);
    bool live = hit;
    // an area light closer than the hit (renderer/Renderer.cpp:184-204)
    {
        float bestT = hit ? r.t : INFINITY;
        int light = -1;
        for (uint32_t li = 0; li < sv.nAreaLights; ++li) {
            float lt;
            if (area_light_hit(sv.areaLights[li], o, d, lt) && bestT > lt) { light = (int)li; bestT = lt; }
        }
        if (light >= 0) {
            value = rgb_scale_f(area_light_irradiance(sv.areaLights[light]), 1.0f / (float)2); // .divide(2)
            live = false;
        } else if (!hit) {
            if (sv.hasEnv) value = env_radiance(sv, d); // Renderer.cpp:217-225
            live = false;
        }
    }
    pool.sLive[s] = live ? 1 : 0;
    pool.sHasVar[s] = 0;
    if (!live) {
        pool.result[(size_t)s * M] = f4(value.r, value.g, value.b, 0.0f);
        return;
    }
    pool.sA[s] = f4(o.x, o.y, o.z, r.t);
    pool.sB[s] = f4(d.x, d.y, d.z, 0.0f);
    pool.sC[s] = f4(__uint_as_float(r.slot), __uint_as_float(r.inst), r.beta, r.gamma);
}

// ---- shade -----------------------------------------------------------------------------------
// FIRST: the batch's first pass -- lane q is path q, vertex 0 is its sample's primary hit, and
// nothing has been written for the path yet.  Otherwise lane q shades path qIn[q].
// RICH: the scene has glass or add materials, normal maps or a photon map (their code and registers are left out otherwise).
// A path's result is stored when the path ends here (black unless it collected something); a path
// that goes on through a shadow ray gets black here and its contribution from the shadow stage.
constexpr int kShadeBlock = RT_SHADE_BLOCK;

// Which lanes shade together.  Whether a vertex runs next-event estimation or bounces, and towards which kind of light,
// is drawn per path, so in queue order neighbouring lanes part at every vertex (17 to 33 of 64 lanes were active per
// instruction).  Those draws only depend on the path's indices, so they are known before anything is loaded: a
// workgroup sorts its 256 paths by them (bounce / point or directional light / area light) and lane t then shades the
// t-th path of that order.  A path's result does not depend on the lane that computes it; the guess of the draws' keys
// below (the material-sample index counts, the vertex after a pending one) only has to be right most of the time.
struct ShadeItem {
    uint32_t p, pixel, i, j;
    bool active;
};
template <int BLOCK>
__device__ __forceinline__ ShadeItem shade_regroup(const SceneView& sv, uint32_t seed, uint32_t M, ShadeItem mine, uint32_t vertex)
{
    constexpr int WAVES = BLOCK / 64;
    constexpr uint32_t CLASSES = 4;   // bounce, light without area, area light, no path
    __shared__ uint32_t wcnt[CLASSES][WAVES];
    __shared__ uint32_t ex[4][BLOCK];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    uint32_t cls = 3u;
    if (mine.active) {
        const uint64_t prefix = rand_prefix(mine.pixel, mine.i * M + mine.j);
        cls = 0u;
        if (rand_unit_from(prefix, seed, vertex, DIM_NEE_CHOICE) > 0.5f) {
            const float choice = rand_unit_from(prefix, seed, vertex, DIM_LIGHT_TYPE);
            cls = (!(choice < sv.neePointP) && choice < sv.neePointP + sv.neeAreaP) ? 2u : 1u;
        }
    }
    unsigned long long same = 0ull;
    for (uint32_t c = 0; c < CLASSES; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (lane == 0) wcnt[c][wave] = (uint32_t)__popcll(m);
        if (cls == c) same = m;
    }
    __syncthreads();
    // position in the order (class, wave, lane)
    uint32_t pos = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)), nPaths = 0;
    for (uint32_t c = 0; c < CLASSES; ++c)
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t n = wcnt[c][w];
            if (c < cls || (c == cls && w < wave)) pos += n;
            if (c < 3u) nPaths += n;
        }
    ex[0][pos] = mine.p; ex[1][pos] = mine.pixel; ex[2][pos] = mine.i; ex[3][pos] = mine.j;
    __syncthreads();
    ShadeItem out;
    out.p = ex[0][threadIdx.x]; out.pixel = ex[1][threadIdx.x]; out.i = ex[2][threadIdx.x]; out.j = ex[3][threadIdx.x];
    out.active = threadIdx.x < nPaths;
    return out;
}

struct ShadeArgs {
    SceneView sv;
    Pool pool;
    rtgpu_render_settings st;
    Batch batch;
    const uint32_t* qIn;
    uint32_t nPaths;
    const uint32_t* cntIn;
    uint32_t* qExt;
    uint32_t* cnt;
};

template <bool FIRST, bool RICH>
// later passes, five waves per SIMD: the register allocator meets 96 VGPRs without spilling a vector register (it took
// 98-111 of its own accord, i.e. four waves), and the passes wait on loads, so the fifth wave pays (six or eight spill and
// lose, and so does five for the RICH variants); the first pass: six (RT_SHADE_WAVES_FIRST)
__global__ __launch_bounds__(kShadeBlock) __attribute__((amdgpu_waves_per_eu(RICH ? 4 : (FIRST ? RT_SHADE_WAVES_FIRST : RT_SHADE_WAVES), RICH ? 4 : (FIRST ? RT_SHADE_WAVES_FIRST : RT_SHADE_WAVES)))) void shade_kernel(ShadeArgs)
{
    // The arguments are read where they are used, through the kernel-argument segment's address, instead of through the
    // parameter: as a parameter every field the kernel uses is loaded into scalar registers at its entry and stays there,
    // far more than a wave has, and the compiler then keeps them in lanes of a vector register and fetches them back
    // one v_readlane at a time (a fifth of the kernel's vector instructions).
    const ShadeArgs& A = *(const ShadeArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const SceneView& sv = A.sv;
    const Pool& pool = A.pool;
    const rtgpu_render_settings& st = A.st;
    const Batch& batch = A.batch;
    const uint32_t* __restrict__ qIn = A.qIn;
    const uint32_t nPaths = A.nPaths;
    const uint32_t* __restrict__ cntIn = A.cntIn;
    uint32_t* __restrict__ qExt = A.qExt;
    uint32_t* __restrict__ cnt = A.cnt;
    // FIRST: every path of the batch; later passes: the extension queue the previous pass filled, whose length is read
    // here (the grid is sized from an upper bound the host knows, surplus workgroups find nothing to do)
    const uint32_t seg = blockIdx.x % kQueueSegs;   // the segment this workgroup appends to and (later passes) reads
    const uint32_t inBlock = FIRST ? blockIdx.x : blockIdx.x / kQueueSegs;
    const uint32_t nIn = FIRST ? nPaths : cntIn[kCtrExt + seg * kCtrLine];
    if (inBlock * (uint32_t)kShadeBlock >= nIn) return;   // a surplus workgroup (the grid covers an upper bound of the queue): nothing to regroup or push
    SHADE_T0(0);
    SHADE_T0(1);
    const uint32_t q = inBlock * (uint32_t)kShadeBlock + threadIdx.x;
    bool pushExt = false, pushShadow = false, pushPhoton = false, repeated = false;
    uint32_t nextVertex = 0;   // the vertex the next pass samples for a path that goes on: the one after a vertex that waits for its own ray
    const uint32_t G = (uint32_t)st.geometry_aa, M = (uint32_t)st.material_aa;
    ShadeItem it;
    it.p = 0; it.pixel = 0; it.i = 0; it.j = 0;
    it.active = q < nIn;
    if (FIRST && it.active) it.active = pool.sLive[fast_div(q, batch.divM)] != 0;
    uint32_t vertexGuess = 0;
    if (it.active) {
        const uint32_t entry = FIRST ? q : qIn[seg * pool.segCap + q];
        it.p = entry & kQueuePathMask;
        const uint32_t s0 = fast_div(it.p, batch.divM), pl = fast_div(s0, batch.divG);
        it.j = it.p - s0 * M;
        it.i = s0 - pl * G;
        int px, py;
        bool pvalid;
        it.pixel = tile_pixel(st, batch, batch.pixelStart + pl, px, py, pvalid);
        vertexGuess = entry >> kQueuePathBits;   // the vertex the pass will sample (see the queue push below)
    }
    if (sv.shadeRegroup) it = shade_regroup<kShadeBlock>(sv, st.seed, M, it, vertexGuess);   // uniform branch
    const bool active = it.active;
    const uint32_t p = it.p;
    SHADE_T1(1);
    SHADE_STAT(0, true);
    SHADE_STAT(1, active);
    // A path's block is one 64-byte line at a scattered address.  Four lanes fetch the four records of one block (the
    // memory pipeline then looks one line up per four lanes, not one per lane and record) and the records reach the
    // lanes that shade the paths through the wave's tile in LDS.
    float4 blkA = f4(0, 0, 0, 0), blkB = blkA, blkH = blkA, blkT = blkA;
    __shared__ float4 blockTile[kShadeBlock / 64][64][4];
    float4 (*tile)[4] = blockTile[threadIdx.x >> 6];
    const int laneC = (int)(threadIdx.x & 63u), pieceC = laneC & 3;
    if constexpr (!FIRST) {
        float4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = (laneC >> 2) + 16 * r;
            const uint32_t ps = (uint32_t)__shfl((int)p, src, 64);
            const bool as = __shfl((int)active, src, 64) != 0;
            v[r] = f4(0, 0, 0, 0);
            if (as) v[r] = pool.rayA.base[(size_t)ps * 4 + pieceC];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = (laneC >> 2) + 16 * r;
            tile[src][(pieceC + src) & 3] = v[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        blkA = tile[laneC][(0 + laneC) & 3];
        blkB = tile[laneC][(1 + laneC) & 3];
        blkH = tile[laneC][(2 + laneC) & 3];
        blkT = tile[laneC][(3 + laneC) & 3];
    }
    // the shading records of the hits (48 bytes each) the same way, three lanes of four per record
    ShadeRec recC;
    recC.q0 = recC.q1 = recC.q2 = f4(0.0f, __uint_as_float(kInvalid), 0.0f, 0.0f);
    if constexpr (!FIRST) {
        const uint32_t slotC = active ? __float_as_uint(blkH.y) : kInvalid;
        float4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = (laneC >> 2) + 16 * r;
            const uint32_t ss = (uint32_t)__shfl((int)slotC, src, 64);
            v[r] = f4(0.0f, __uint_as_float(kInvalid), 0.0f, 0.0f);
            if (ss != kInvalid && pieceC < 3) v[r] = sv.shadeRecs[(size_t)ss * 3 + pieceC];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // every lane has read its block out of the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = (laneC >> 2) + 16 * r;
            tile[src][(pieceC + src) & 3] = v[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        recC.q0 = tile[laneC][(0 + laneC) & 3];
        recC.q1 = tile[laneC][(1 + laneC) & 3];
        recC.q2 = tile[laneC][(2 + laneC) & 3];
    }
    // what the path leaves behind in its block: records 0 .. 3 = ray origin + limit, direction + state, hit, throughput
    float4 outA = f4(0, 0, 0, 0), outB = outA, outT = outA;
    uint32_t pieceMask = 0;
    if (active) {
        SHADE_T0(2);
        const int maxLen = st.max_path_length;
        const uint32_t s = fast_div(p, batch.divM), j = it.j, i = it.i;
        const uint32_t pixel = it.pixel;
        const float4 ra = FIRST ? pool.sA[s] : blkA;
        const float4 rb = FIRST ? pool.sB[s] : blkB;
        float4 hc;
        uint32_t inst;
        if (FIRST) {
            const float4 c = pool.sC[s];
            hc = f4(ra.w, c.x, c.z, c.w);
            inst = __float_as_uint(c.y);
        } else {
            hc = blkH;
            inst = sv.singleMesh ? (__float_as_uint(hc.y) != kInvalid ? 0u : kInvalid) : pool.hitInst[p];
        }
        uint32_t state = FIRST ? 0u : __float_as_uint(rb.w);
        int k = (int)(state & ST_K_MASK);
        F3 o = f3(ra.x, ra.y, ra.z);
        F3 d = f3(rb.x, rb.y, rb.z);
        RGBf thr = rgb(1.0f, 1.0f, 1.0f);
        if (!FIRST) thr = rgb(blkT.x, blkT.y, blkT.z);
        const float t = hc.x;
        const bool hit = inst != kInvalid;
        // what the path leaves behind for the next stage
        float outTmax = 0.0f;
        F3 outAux = f3(0, 0, 0);
        float outAuxW = 0.0f;
        bool writeThr = false, writeAux = false;
        bool passed = (state & ST_PASSED_VAR) != 0;
        bool done = false;
        // A path that ends here either counts -- its value is `acc + result` -- or was cut by Russian
        // roulette / the length limit, and then the whole path is black (renderer/Renderer.cpp:237-240).
        RGBf result = rgb(0, 0, 0);
        bool completed = false;
        // emission collected from add materials (material/AddMaterial.cpp:12-17) along the path
        RGBf acc = rgb(0, 0, 0), pend = rgb(0, 0, 0);
        bool writeAcc = FIRST, writePend = false;
        if (!FIRST && (RICH && sv.hasAdd)) { const float4 aq = pool.acc[p]; acc = rgb(aq.x, aq.y, aq.z); }

        if (state & ST_GLOSSY_PENDING) {
            SHADE_STAT(2, true);
            // GlossyMaterial::sampleTransport after its own traceRay (material/GlossyMaterial.cpp:126-178),
            // then the Russian roulette of samplePath for that vertex (renderer/Renderer.cpp:52)
            const bool rough = (state & ST_GLOSSY_ROUGH) != 0;
            int lightIdx = -1;
            if (rough) {
                for (uint32_t li = 0; li < sv.nAreaLights; ++li) {
                    float lt;
                    if (area_light_hit(sv.areaLights[li], o, d, lt) && (!hit || t > lt)) { lightIdx = (int)li; break; }
                }
            } else {
                double bestT = hit ? (double)t : 1E99;
                for (uint32_t li = 0; li < sv.nAreaLights; ++li) {
                    float lt;
                    if (area_light_hit(sv.areaLights[li], o, d, lt) && bestT > (double)lt) { lightIdx = (int)li; bestT = lt; }
                }
            }
            if ((RICH && sv.hasAdd)) { const float4 pq = pool.accPend[p]; pend = rgb(pq.x, pq.y, pq.z); }
            if (lightIdx >= 0) {
                done = true;
                completed = k + 1 != maxLen;
                result = pend;
                if (!rough) // mirror: the light's irradiance / 2 (GlossyMaterial.cpp:198-203)
                    result = rgb_add(pend, rgb_mul(thr, rgb_scale_f(area_light_irradiance(sv.areaLights[lightIdx]), 1.0f / (float)2)));
            } else {
                const uint32_t rs = i * M + (passed ? j : 0u);
                const bool terminated = (k + 1 == maxLen) || rand_unit(st.seed, pixel, rs, (uint32_t)k, DIM_RR) < 0.2f;
                if (terminated) {
                    done = true;
                } else {
                    const float f = (rough ? 2.0f : 1.0f) * (1.0f / (1.0f - 0.2f));
                    thr = rgb_scale_f(thr, f);
                    if ((RICH && sv.hasAdd)) { acc = rgb_add(acc, rgb_div_f(pend, 1.0f - 0.2f)); writeAcc = true; }
                    state &= ~(ST_GLOSSY_PENDING | ST_GLOSSY_ROUGH);
                    if (!hit) {
                        // ctx.nextHit is empty: samplePath traces again from hitpoint + dir * 1e-4 (Renderer.cpp:57-61)
                        const float4 hq = pool.aux[p];
                        const F3 hp = f3(hq.x, hq.y, hq.z);
                        o = hp + (d * 0.0001f);
                        state |= ST_RETRACE;
                        writeThr = true;
                        pushExt = true;
                        done = true; // nothing more for this vertex now
                    } else {
                        k += 1;
                        state = (state & ~ST_K_MASK) | (uint32_t)k;
                    }
                }
            }
        } else if (RICH && (state & ST_GLASS_PENDING)) {
            // GlassMaterial::sampleTransport after its traceRay (material/GlassMaterial.cpp:109-132), the
            // vertex's bsdf (:135-178) and the Russian roulette of samplePath (renderer/Renderer.cpp:52)
            const float4 hq = pool.aux[p];
            const F3 hp = f3(hq.x, hq.y, hq.z);
            const DevMaterial& gm = sv.materials[__float_as_uint(hq.w)];
            const float4 gn = pool.glassN[p];
            if ((RICH && sv.hasAdd)) { const float4 pq = pool.accPend[p]; pend = rgb(pq.x, pq.y, pq.z); }
            double bestT = hit ? (double)t : 1E99;
            int lightIdx = -1;
            for (uint32_t li = 0; li < sv.nAreaLights; ++li) {
                float lt;
                if (area_light_hit(sv.areaLights[li], o, d, lt) && bestT > (double)lt) { lightIdx = (int)li; bestT = lt; }
            }
            if (lightIdx < 0 && (state & ST_GLASS_STICKY)) {
                // an earlier re-sampling pass of this vertex went the other way and met a light there: the
                // vertex stays emissive with that light (the metadata's lightHit is never cleared, :115-122)
                const uint32_t answer = __float_as_uint(pool.glassB[p].w);
                if (answer != kInvalid) lightIdx = (int)answer;
            }
            const float metaT = (float)bestT;
            RGBf atten = rgb(1.0f, 1.0f, 1.0f);
            const bool internal = (state & ST_GLASS_INTERNAL) != 0;
            if (internal) atten = rgb_pow_d(rgb(gm.color[0], gm.color[1], gm.color[2]), (double)gm.roughness * (double)metaT);
            state &= ~(ST_GLASS_PENDING | ST_GLASS_INTERNAL | ST_GLASS_STICKY);
            if (lightIdx >= 0) {
                done = true;
                completed = k + 1 != maxLen;
                const DevAreaLight& L = sv.areaLights[lightIdx];
                const RGBf lightRadiance = rgb(L.radiance[0], L.radiance[1], L.radiance[2]);
                const F3 lampPoint = d * metaT;   // as the reference has it: not offset by the hit point
                const F3 objectToLamp = normalized3(lampPoint - hp);
                const float angle = fmaxf(0.0f, dot3(f3(gn.x, gn.y, gn.z), objectToLamp));
                RGBf val = rgb_scale_d(lightRadiance, (double)angle);
                if (internal) val = rgb_mul(val, atten);
                result = rgb_add(pend, rgb_mul(thr, val));
            } else {
                const uint32_t rs = i * M + (passed ? j : 0u);
                const bool terminated = (k + 1 == maxLen) || rand_unit(st.seed, pixel, rs, (uint32_t)k, DIM_RR) < 0.1f;
                if (terminated) {
                    done = true;
                } else {
                    thr = rgb_div_f(rgb_mul(thr, atten), 1.0f - 0.1f);
                    if ((RICH && sv.hasAdd)) { acc = rgb_add(acc, rgb_div_f(pend, 1.0f - 0.1f)); writeAcc = true; }
                    if (!hit) {
                        // the reference traces the same ray once more (Renderer.cpp:57-61) and it misses again
                        done = true;
                        completed = true;
                        if (sv.hasEnv) result = rgb_mul(thr, env_radiance(sv, d));
                    } else {
                        k += 1;
                        state = (state & ~ST_K_MASK) | (uint32_t)k;
                    }
                }
            }
        } else if (state & ST_RETRACE) {
            state &= ~ST_RETRACE;
            if (!hit) {
                done = true;
                completed = true;
                if (sv.hasEnv) result = rgb_mul(thr, env_radiance(sv, d));
            } else {
                k += 1;
                state = (state & ~ST_K_MASK) | (uint32_t)k;
            }
        } else if (k > 0 && !hit) {
            SHADE_STAT(3, true);
            // the extension ray left the scene: the path ends at the previous vertex and collects
            // the environment along its transport direction (renderer/Renderer.cpp:73-76, 94-98)
            done = true;
            completed = true;
            if (sv.hasEnv) result = rgb_mul(thr, env_radiance(sv, d));
        }

        SHADE_T1(2);
        if (!done) {
            SHADE_T0(3);
            SHADE_STAT(4, true);
            // ---- vertex k: IMaterial::sampleTransport + the vertex's factor of calculatePathEnergy ----
            // Everything the vertex needs from the scene is asked for here, side by side: the triangle's shading record
            // (vertex normals and material index) and the instance record.  (Fetched where they are first used, behind the branches on the
            // material type, they queue up one behind the other, and with four waves per SIMD that wait is what the pass
            // spends its time on.)  In a one-mesh scene the instance record is the same for every lane and comes through
            // the scalar cache.
            const DevInstance& in = sv.inst[sv.singleMesh ? 0u : inst];
            const uint32_t slot = __float_as_uint(hc.y);
            const float beta = hc.z, gamma = hc.w;
            const ShadeRec rec = FIRST ? load_shade_rec(sv, slot) : recC;
            const uint32_t tm = rec.material();
            F3 normal = sv.singleMesh ? hit_normal(sv.inst[0], o, d, t, slot, beta, gamma, rec) : hit_normal(in, o, d, t, slot, beta, gamma, rec);
            uint32_t mi = in.material;
            if (slot != kInvalid && tm != kInvalid) mi = tm;
            // PhotonMapMode::full: once `pmDepth` earlier vertices were diffuse, a diffuse closure reads the map
            // (material/DiffuseMaterial.cpp:40-54) and has no variance (:228-247)
            const uint32_t diffuseCount = (state & ST_DIFFUSE_MASK) >> ST_DIFFUSE_SHIFT;
            const bool readsMap = RICH && sv.pmMode == RTGPU_PM_FULL && (int)diffuseCount >= sv.pmDepth;
            const bool curVar = mat_has_variance(sv, mi, normal, d, readsMap);
            const bool isFirstVar = curVar && !passed;
            const int sampleCount = isFirstVar ? (int)M : 1;
            const int sampleI = isFirstVar ? (int)j : 0;
            const int sampleGrid = isFirstVar ? (int)batch.sqrtM : 1;   // (int)sqrt((double)sampleCount)
            const int sampleY = isFirstVar ? (int)fast_div(j, batch.divSqrtM) : 0, sampleX = sampleI - sampleY * sampleGrid;   // the sample's cell of that grid
            passed = passed || curVar;
            if (passed) state |= ST_PASSED_VAR;
            if (curVar && j == 0) pool.sHasVar[s] = 1;
            const uint32_t rs = i * M + (passed ? j : 0u);
            const uint32_t vk = (uint32_t)k;
            const uint64_t rp = rand_prefix(pixel, rs);   // first hashing round of every random number of this vertex
            // The walk from the hit's material to the closure that samples this vertex:
            // MixMaterial::sampleTransport picks a child (material/MixMaterial.cpp:9-26), AddMaterial goes on
            // with its second child and adds its emissive first child to the vertex (material/AddMaterial.cpp:5-17).
            // walkBits names the walk, two bits per step: 1 / 2 = mix chose its second / first child, 3 = add.
            const uint32_t rootMi = mi;
            uint32_t walkBits = 0;
            RGBf addE = rgb(0, 0, 0);
            auto surface_color = [&](const DevMaterial& m) {
                RGBf c = rgb(m.color[0], m.color[1], m.color[2]);
                if (m.texture != 0u && (m.type == RTGPU_MAT_DIFFUSE || m.type == RTGPU_MAT_EMISSIVE)) {
                    float tu, tv;
                    hit_texcoord(sv, slot, beta, gamma, tu, tv);
                    c = m.type == RTGPU_MAT_DIFFUSE ? albedo_lookup(sv, m.texture - 1u, tu, tv) : emission_lookup(sv, m.texture - 1u, tu, tv);
                }
                return c;
            };
            for (uint32_t level = 0; level < 4; ++level) {
                const DevMaterial& mm = sv.materials[mi];
                if (mat_is_mix(mm)) {
                    const bool choseFirst = rand_unit_from(rp, st.seed, vk, DIM_MIX0 + level) > mat_mix_factor(mm, normal, d);
                    walkBits |= (choseFirst ? 2u : 1u) << (2u * level);
                    mi = choseFirst ? mm.first : mm.second;
                } else if (RICH && mm.type == RTGPU_MAT_ADD) {
                    const DevMaterial& em = sv.materials[mm.first];
                    addE = rgb_add(addE, rgb_scale_d(surface_color(em), em.intensity));
                    walkBits |= 3u << (2u * level);
                    mi = mm.second;
                } else {
                    break;
                }
            }
            const DevMaterial& mat = sv.materials[mi];
            const RGBf color = surface_color(mat);
            const F3 hitpoint = o + (d * t); // RayHitInfo::getHitpoint, math/RayHitInfo.h:48-51
            const bool lastVertex = (k + 1 == maxLen); // such a path is "terminated" and contributes nothing
            // caustics mode (PhotonMapMode::caustics): the callback a diffuse bounce leaves behind fires when this
            // vertex is specular (specularity > 0.8: glass, transparent, glossy with roughness < 0.2) and ends the
            // path at that bounce with the caustic map's estimate (material/DiffuseMaterial.cpp:104-135,
            // renderer/Renderer.cpp:40-47); glass behind a rough bounce is black (material/GlassMaterial.cpp:31-38)
            const bool caustics = RICH && sv.pmMode == RTGPU_PM_CAUSTICS;
            const uint32_t prevFlags = state & (ST_PREV_DIFFUSE | ST_PREV_ROUGH);
            state &= ~(ST_PREV_DIFFUSE | ST_PREV_ROUGH);
            float specularity = 0.0f;
            if (mat.type == RTGPU_MAT_GLOSSY) specularity = 1.0f - mat.roughness;
            else if (mat.type == RTGPU_MAT_GLASS || mat.type == RTGPU_MAT_TRANSPARENT) specularity = 1.0f;

            SHADE_T1(3);
            if (caustics && (prevFlags & ST_PREV_DIFFUSE) && specularity > 0.8) {
                done = true;
                pushPhoton = true;
            } else if (caustics && mat.type == RTGPU_MAT_GLASS && k != 0 && (prevFlags & ST_PREV_ROUGH)) {
                done = true;
                completed = !lastVertex;
                result = rgb_mul(thr, addE);
            } else if (mat.type == RTGPU_MAT_EMISSIVE) {
                // EmissiveMaterial (material/EmissiveMaterial.cpp:5-22)
                done = true;
                completed = !lastVertex;
                result = rgb_mul(thr, rgb_add(addE, rgb_scale_d(color, mat.intensity)));
            } else if (lastVertex) {
                done = true;
                SHADE_STAT(5, true);
            } else if (mat.type == RTGPU_MAT_TRANSPARENT) {
                // TransparentMaterial (material/TransparentMaterial.cpp:3-16): straight on, bsdf = identity,
                // Russian roulette 0.01 (renderer/Renderer.cpp:52, 104-107)
                done = true;
                if (!(rand_unit_from(rp, st.seed, vk, DIM_RR) < 0.01f)) {
                    if ((RICH && sv.hasAdd)) { acc = rgb_add(acc, rgb_div_f(rgb_mul(thr, addE), 1.0f - 0.01f)); writeAcc = true; }
                    thr = rgb_div_f(thr, 1.0f - 0.01f);
                    o = hitpoint + (d * 0.0001f);
                    writeThr = true;
                    state = (state & ~ST_K_MASK) | (uint32_t)(k + 1);
                    pushExt = true;
                }
            } else {
                // the normal a diffuse / glossy closure shades with (DiffuseMaterial.cpp:33-37, GlossyMaterial.cpp:81-85);
                // Glossy's test of the sampled direction against the surface keeps the hit's own normal (:114)
                const F3 surfaceNormal = normal;
                if (RICH && mat.normalMap != 0u && (mat.type == RTGPU_MAT_DIFFUSE || mat.type == RTGPU_MAT_GLOSSY))
                    normal = sample_normal_map(sv, mat.normalMap - 1u, inst, slot, beta, gamma, normal);
                if (RICH && mat.type == RTGPU_MAT_GLASS) {
                    // GlassMaterial::sampleTransport up to its traceRay (material/GlassMaterial.cpp:21-107)
                    const F3 objToIncoming = -d;
                    const F3 reflDir = (-objToIncoming) + ((float)(2.0 * (double)dot3(normal, objToIncoming)) * normal);
                    double nDotWo = dot3(normal, objToIncoming);
                    double incomingAngle = nDotWo;
                    const bool internalIn = nDotWo < 0;
                    double curRelIor = mat.intensity / 1.0;
                    if (internalIn) curRelIor = 1.0 / curRelIor;
                    const double cosTransmissionSqr = 1.0 - ((1.0 - (incomingAngle * incomingAngle)) / (curRelIor * curRelIor));
                    const bool tir = cosTransmissionSqr < 0;
                    float reflW = 1.0f;
                    F3 transDir = reflDir, nUsed = normal;
                    if (!tir) {
                        if (internalIn) { nDotWo = -nDotWo; incomingAngle = -incomingAngle; nUsed = -normal; } // the node's normal is flipped for good (:73-78)
                        const double cosRefractAngle = sqrt(cosTransmissionSqr);
                        transDir = (d / (float)curRelIor) - ((float)(cosRefractAngle - (incomingAngle / curRelIor)) * nUsed);
                        const double rs_ = ((curRelIor * nDotWo) - cosRefractAngle) / ((curRelIor * nDotWo) + cosRefractAngle);
                        const double rp_ = (nDotWo - (curRelIor * cosRefractAngle)) / (nDotWo + (curRelIor * cosRefractAngle));
                        const double kr = ((rs_ * rs_) + (rp_ * rp_)) / 2.0;
                        const double reflectionWeight = kr, transmissionWeight = 1.0 - kr;
                        reflW = (float)(reflectionWeight / (reflectionWeight + transmissionWeight));
                    }
                    const bool reflect = rand_unit_from(rp, st.seed, vk, DIM_GLASS) < reflW;
                    // The reference keeps this state in the node's metadata, and the node that is re-sampled
                    // materialAA times (the sample's first variance vertex) keeps its metadata from pass to pass
                    // (:26-29, renderer/Renderer.cpp:246).  What an earlier pass left behind follows from its
                    // random numbers: did it reach this closure, and which of the two directions did it take.
                    bool firstVisit = true, otherBefore = false;
                    if (isFirstVar) {
                        for (uint32_t jj = 0; jj < j; ++jj) {
                            const uint32_t rsj = i * M + jj;
                            uint32_t m2 = rootMi, bits2 = 0;
                            for (uint32_t level = 0; level < 4; ++level) {
                                const DevMaterial& mm = sv.materials[m2];
                                if (mat_is_mix(mm)) {
                                    const bool choseFirst = rand_unit(st.seed, pixel, rsj, vk, DIM_MIX0 + level) > mat_mix_factor(mm, normal, d);
                                    bits2 |= (choseFirst ? 2u : 1u) << (2u * level);
                                    m2 = choseFirst ? mm.first : mm.second;
                                } else if (mm.type == RTGPU_MAT_ADD) {
                                    bits2 |= 3u << (2u * level);
                                    m2 = mm.second;
                                } else {
                                    break;
                                }
                            }
                            if (bits2 != walkBits) continue;
                            firstVisit = false;
                            if ((rand_unit(st.seed, pixel, rsj, vk, DIM_GLASS) < reflW) != reflect) otherBefore = true;
                        }
                    }
                    done = true;
                    if (tir && firstVisit) {
                        // total internal reflection, first visit (:62-70): no look for lights, the vertex is an
                        // ordinary bounce and absorbs over its own incoming distance (meta->t is NaN, :173)
                        if (!(rand_unit_from(rp, st.seed, vk, DIM_RR) < 0.1f)) {
                            RGBf atten = rgb(1.0f, 1.0f, 1.0f);
                            if (dot3(normal, reflDir) < 0) atten = rgb_pow_d(color, (double)mat.roughness * (double)t);
                            if ((RICH && sv.hasAdd)) { acc = rgb_add(acc, rgb_div_f(rgb_mul(thr, addE), 1.0f - 0.1f)); writeAcc = true; }
                            thr = rgb_div_f(rgb_mul(thr, atten), 1.0f - 0.1f);
                            o = hitpoint + (reflDir * 0.0001f);
                            d = reflDir;
                            writeThr = true;
                            state = (state & ~ST_K_MASK) | (uint32_t)(k + 1);
                            pushExt = true;
                        }
                    } else {
                        const F3 dir = reflect ? reflDir : transDir;
                        if ((RICH && sv.hasAdd)) { pend = rgb_mul(thr, addE); writePend = true; }
                        state = (state & ~ST_K_MASK) | (uint32_t)k | ST_GLASS_PENDING | (dot3(nUsed, dir) < 0 ? ST_GLASS_INTERNAL : 0u);
                        if (!firstVisit && otherBefore) {
                            // an earlier pass looked for lights along the other direction (:109-122)
                            const F3 other = reflect ? transDir : reflDir;
                            const F3 o2 = hitpoint + (other * 0.0001f);
                            double bestT = 1E99;
                            int lightIdx = -1;
                            for (uint32_t li = 0; li < sv.nAreaLights; ++li) {
                                float lt;
                                if (area_light_hit(sv.areaLights[li], o2, other, lt) && bestT > (double)lt) { lightIdx = (int)li; bestT = lt; }
                            }
                            if (lightIdx >= 0) {
                                pool.glassA[p] = f4(o2.x, o2.y, o2.z, (float)bestT);
                                pool.glassB[p] = f4(other.x, other.y, other.z, __uint_as_float((uint32_t)lightIdx));
                                state |= ST_GLASS_STICKY;
                                pushShadow = true;
                            }
                        }
                        pool.glassN[p] = f4(nUsed.x, nUsed.y, nUsed.z, t);
                        o = hitpoint + (dir * 0.0001f);
                        d = dir;
                        outAux = hitpoint;
                        outAuxW = __uint_as_float(mi);
                        writeAux = true;
                        writeThr = true;
                        pushExt = true;
                    }
                } else if (RICH && mat.type == RTGPU_MAT_DIFFUSE && readsMap) {
                    // the vertex ends the path with the map's radiance estimate (material/DiffuseMaterial.cpp:56-80,
                    // 154-166); the gather stage adds it to what the path has collected
                    done = true;
                    completed = true;
                    result = rgb_mul(thr, addE);
                    o = hitpoint;
                    outAux = f3(color.r, color.g, color.b);
                    outAuxW = __uint_as_float(mi);
                    writeAux = true;
                    writeThr = true;
                    // before any variance vertex all material samples j are the same path and only j = 0 is read
                    // (resolve_kernel): the others skip the search
                    pushPhoton = passed || j == 0u;
                } else {
                    // DiffuseMaterial::sampleTransport / bsdf (material/DiffuseMaterial.cpp:84-103, 183-195) or
                    // GlossyMaterial::sampleTransport / bsdf (material/GlossyMaterial.cpp:68-207).  What the two have in
                    // common -- the choice of next-event estimation, the light sample, the stratified square of a bounce
                    // -- is written once, so that a wave holding vertices of both kinds runs it once for all of them.
                    const bool glossy = mat.type != RTGPU_MAT_DIFFUSE;
                    const float roughness = mat.roughness;
                    const bool neeEnabled = !glossy || roughness > 0;
                    const bool useNEE = neeEnabled && rand_unit_from(rp, st.seed, vk, DIM_NEE_CHOICE) > 0.5f;
                    if (useNEE) {
                        SHADE_T0(4);
                        SHADE_STAT(6, !glossy);
                        SHADE_STAT(8, glossy);
                        const NeeSample ns = nee_sample(sv, hitpoint, sampleX, sampleY, sampleGrid, st.seed, rp, vk);
                        done = true;
                        completed = true;
                        result = rgb_mul(thr, addE);   // what the vertex is worth when the light is hidden
                        if (ns.valid) {
                            RGBf contribution;
                            bool black;
                            if (!glossy) {
                                const double angle = fmaxf(0.0f, dot3(normal, ns.dir));
                                const RGBf direct = rgb_scale_d(ns.radiance, angle);
                                const RGBf value = rgb_scale_f(rgb_scale_d(direct, mat.intensityOverPi), 2.0f);
                                contribution = rgb_mul(color, value);
                                black = value.r == 0.0f && value.g == 0.0f && value.b == 0.0f;
                            } else {
                                const F3 micro = normalized3((-d) + ns.dir);
                                const double angle = fmaxf(0.0f, dot3(micro, ns.dir));
                                const float f = ggx_brdf(roughness, -d, ns.dir, normal, micro);
                                const RGBf neeRadiance = rgb_scale_d(rgb_scale_d(ns.radiance, angle), f);
                                contribution = rgb_mul(rgb_scale_f(neeRadiance, 2.0f), color);
                                black = neeRadiance.r == 0.0f && neeRadiance.g == 0.0f && neeRadiance.b == 0.0f;
                            }
                            const RGBf c = rgb_add(acc, rgb_mul(thr, rgb_add(addE, contribution)));
                            o = hitpoint + (ns.dir * 0.0001f);
                            d = ns.dir;
                            outTmax = ns.tmax;
                            outAux = f3(c.r, c.g, c.b);
                            writeAux = true;
                            // a sample worth exactly nothing (the light is behind the surface or faces away) leaves the path's value
                            // what it is when the light is hidden: no query (NaN or infinite values are not zero and are traced)
                            pushShadow = !(sv.skipBlackShadow && black);
                        }
                        SHADE_T1(4);
                    } else {
                        SHADE_T0(5);
                        SHADE_STAT(7, !glossy);
                        SHADE_STAT(9, glossy);
                        // the diffuse bounce stratifies over the grid's edge, the glossy one over the sample count
                        // (material/DiffuseMaterial.cpp:98, material/GlossyMaterial.cpp:107)
                        float sx, sy;
                        stratified_square(glossy ? sampleCount : sampleGrid, (float)sampleI,
                                          rand_unit_from(rp, st.seed, vk, DIM_SQUARE_X), rand_unit_from(rp, st.seed, vk, DIM_SQUARE_Y), sx, sy);
                        if (!glossy) {
                            const Basis3 basis = basis_from(normal);
                            const F3 l = cosine_hemisphere(sx, sy);
                            const F3 dir = ((basis.u * l.x) + (basis.v * l.y)) + (basis.w * l.z);
                            const bool terminated = rand_unit_from(rp, st.seed, vk, DIM_RR) < 0.1f;
                            if (terminated) {
                                done = true;
                            } else {
                                const double angle = fmaxf(0.0f, dot3(normal, dir));
                                const RGBf unit = rgb(1.0f, 1.0f, 1.0f);
                                const RGBf value = rgb_scale_f(rgb_scale_d(rgb_scale_d(unit, angle), mat.intensity), 2.0f);
                                const RGBf thrBefore = thr, accBefore = acc;
                                const F3 rayDirIn = d;
                                if ((RICH && sv.hasAdd)) { acc = rgb_add(acc, rgb_div_f(rgb_mul(thr, addE), 1.0f - 0.1f)); writeAcc = true; }
                                thr = rgb_div_f(rgb_mul(thr, rgb_mul(color, value)), 1.0f - 0.1f);
                                o = hitpoint + (dir * 0.0001f);
                                d = dir;
                                writeThr = true;
                                state = (state & ~ST_K_MASK) | (uint32_t)(k + 1);
                                if (RICH && diffuseCount < 15u) state += 1u << ST_DIFFUSE_SHIFT;   // specularity 0
                                if (caustics) {
                                    state |= ST_PREV_DIFFUSE | ST_PREV_ROUGH;
                                    const RGBf tc = rgb_mul(thrBefore, color), cb = rgb_add(accBefore, rgb_mul(thrBefore, addE));
                                    pool.cHit[p] = f4(hitpoint.x, hitpoint.y, hitpoint.z, 0.0f);
                                    pool.cDir[p] = f4(rayDirIn.x, rayDirIn.y, rayDirIn.z, 0.0f);
                                    pool.cThrColor[p] = f4(tc.r, tc.g, tc.b, 0.0f);
                                    pool.cBase[p] = f4(cb.r, cb.g, cb.b, 0.0f);
                                }
                                pushExt = true;
                                done = true;
                            }
                        } else {
                            const F3 micro = normalized3(vndf_sample(normal, -d, roughness, sx, sy));
                            // getPerfectReflectionDir (GlossyMaterial.cpp:22-25): the double factor narrows to float
                            const F3 sampleDir = d + ((float)(2.0 * (double)dot3(micro, -d)) * micro);
                            done = true;
                            completed = true;   // a direction below the surface ends the path as an emissive black vertex (GlossyMaterial.cpp:117-122)
                            result = rgb_mul(thr, addE);
                            if (!(dot3(sampleDir, surfaceNormal) < 0)) {
                                if ((RICH && sv.hasAdd)) { pend = rgb_mul(thr, addE); writePend = true; }
                                thr = rgb_mul(thr, color);
                                o = hitpoint + (sampleDir * 0.001f);
                                d = sampleDir;
                                writeThr = true;
                                outAux = hitpoint;
                                writeAux = true;
                                state = (state & ~ST_K_MASK) | (uint32_t)k | ST_GLOSSY_PENDING | (neeEnabled ? ST_GLOSSY_ROUGH : 0u);
                                if (RICH && (1.0f - roughness) < 0.8 && diffuseCount < 15u) state += 1u << ST_DIFFUSE_SHIFT;   // specularity = 1 - roughness
                                if (caustics && (1.0f - roughness) < 0.8) state |= ST_PREV_ROUGH;
                                pushExt = true;
                            }
                        }
                        SHADE_T1(5);
                    }
                }
            }
        }
        SHADE_T0(6);
        nextVertex = (state & ST_K_MASK) + ((state & (ST_GLOSSY_PENDING | ST_GLASS_PENDING | ST_RETRACE)) != 0u ? 1u : 0u);
        repeated = j != 0 && (state & ST_PASSED_VAR) == 0;
        if (pushExt || pushShadow || pushPhoton) {
            SHADE_STAT(10, true);
            outA = f4(o.x, o.y, o.z, outTmax);
            outB = f4(d.x, d.y, d.z, __uint_as_float(state));
            // the first pass writes whole blocks, the hit record too, so that no cache line leaves the pass partly written
            pieceMask = FIRST ? 7u : 3u;
            if (writeThr) { outT = f4(thr.r, thr.g, thr.b, 0.0f); pieceMask |= 8u; }
            // a path that ends with its shadow ray has no use for its throughput any more: what the ray is worth when it
            // gets through waits in that record, inside the path's block, instead of in a cache line of its own
            if (writeAux) {
                if (pushShadow && !pushExt && !pushPhoton) { outT = f4(outAux.x, outAux.y, outAux.z, outAuxW); pieceMask |= 8u; }
                else pool.aux[p] = f4(outAux.x, outAux.y, outAux.z, outAuxW);
            }
            if ((RICH && sv.hasAdd) && pushExt) {
                if (writeAcc) pool.acc[p] = f4(acc.r, acc.g, acc.b, 0.0f);
                if (writePend) pool.accPend[p] = f4(pend.r, pend.g, pend.b, 0.0f);
            }
        }
        if (!pushExt) {
            const RGBf total = rgb_add(acc, result);
            pool.result[p] = completed ? f4(total.r, total.g, total.b, 0.0f) : f4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        SHADE_T1(6);
    }
    // the blocks go out the way they came in: every lane leaves its records in the tile, four lanes write one block
    tile[laneC][(0 + laneC) & 3] = outA;
    tile[laneC][(1 + laneC) & 3] = outB;
    tile[laneC][(2 + laneC) & 3] = f4(0.0f, 0.0f, 0.0f, 0.0f);
    tile[laneC][(3 + laneC) & 3] = outT;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int src = (laneC >> 2) + 16 * r;
        const uint32_t ps = (uint32_t)__shfl((int)p, src, 64);
        const uint32_t m = (uint32_t)__shfl((int)pieceMask, src, 64);
        if ((m >> pieceC) & 1u) pool.rayA.base[(size_t)ps * 4 + pieceC] = tile[src][(pieceC + src) & 3];
    }
    SHADE_T0(7);
    queue_push2<kShadeBlock>(qExt + seg * pool.segCap, cnt + kCtrExt + seg * kCtrLine, pushExt, pool.qShadow + seg * pool.segCap, cnt + kCtrShadow + seg * kCtrLine,
                             pushShadow, p, p | (nextVertex << kQueuePathBits));
    if (RICH) {
        if (sv.pmMode != RTGPU_PM_NONE) queue_push1<kShadeBlock>(pool.qPhoton, cnt + 2, pushPhoton, p);   // uniform branch
    }
    // Extension rays of vertices before the first variance vertex are identical for every material
    // sample j (the reference traces them once per geometry sample): count the repeats so that
    // the ray statistics only contain distinct rays.
    const unsigned long long rep = __ballot(pushExt && repeated);
    if (rep != 0ull && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)rep) - 1)) atomicAdd(cnt + 3, (uint32_t)__popcll(rep));
    SHADE_T1(7);
    SHADE_T1(0);
}

// ---- shadow / connect and extend ------------------------------------------------------------------
// Both run after a shade pass and are independent of each other, so one launch carries both: the
// first workgroups trace the extension rays (closest hit), the rest the shadow rays (any hit).
template <int MODE>
__device__ __forceinline__ void shadow_block(const SceneView& sv, const Pool& pool, const uint32_t* __restrict__ queue, uint32_t block, uint32_t n,
                                             typename TraceLds<MODE>::type& lds, int lane)
{
    const uint32_t q = block * 64u + lane;
    const bool has = q < n;
    const uint32_t p = queue[has ? q : 0u];
    float4 ra = pool.rayA[p], rb = pool.rayB[p];
    // a glass vertex asks whether the light along the direction it did not take is visible
    const bool sticky = sv.hasGlass && (__float_as_uint(rb.w) & ST_GLASS_STICKY) != 0;
    if (sticky) { ra = pool.glassA[p]; rb = pool.glassB[p]; }
    const bool occluded = wave_any<MODE>(sv, has, f3(ra.x, ra.y, ra.z), f3(rb.x, rb.y, rb.z), ra.w, lds, lane);
    if (!has) return;
    if (sticky) {
        if (occluded) pool.glassB[p] = f4(rb.x, rb.y, rb.z, __uint_as_float(kInvalid));
    } else if (!occluded) {
        pool.result[p] = pool.thr[p];   // the pending contribution (see shade_kernel)
    }
}

template <int MODE>
__device__ __forceinline__ void extend_block(const SceneView& sv, const Pool& pool, const uint32_t* __restrict__ queue, uint32_t block,
                                             uint32_t n, typename TraceLds<MODE>::type& lds, int lane)
{
    const uint32_t q = block * 64u + lane;
    const bool has = q < n;
    const uint32_t p = queue[has ? q : 0u] & kQueuePathMask;
    const float4 ra = pool.rayA[p], rb = pool.rayB[p];
    const TraceResult r = wave_closest<MODE>(sv, has, f3(ra.x, ra.y, ra.z), f3(rb.x, rb.y, rb.z), lds, lane);
    if (!has) return;
    pool.hit[p] = f4(r.t, __uint_as_float(r.slot), r.beta, r.gamma);
    if (MODE != kTraceSingle) pool.hitInst[p] = r.inst;
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == kTraceInstances ? 6 : (MODE == kTraceSingle ? RT_TRACE_WAVES : 1)))) void trace_stage_kernel(SceneView sv, Pool pool, const uint32_t* __restrict__ extQueue, const uint32_t* __restrict__ cnt)
{
    __shared__ typename TraceLds<MODE>::type lds;
    const int lane = threadIdx.x;
    // workgroup b: block b / kQueueSegs of segment b % kQueueSegs, the segment's extension entries first.  Lengths as
    // the shade pass left them; the grid covers an upper bound of the longest segment's two queues together
    const uint32_t seg = blockIdx.x % kQueueSegs, block = blockIdx.x / kQueueSegs;
    const uint32_t nExt = cnt[kCtrExt + seg * kCtrLine], nShadow = cnt[kCtrShadow + seg * kCtrLine];
    const uint32_t extBlocks = (nExt + 63u) / 64u, shadowBlocks = (nShadow + 63u) / 64u;
    if (block < extBlocks) extend_block<MODE>(sv, pool, extQueue + seg * pool.segCap, block, nExt, lds, lane);
    else if (block - extBlocks < shadowBlocks) shadow_block<MODE>(sv, pool, pool.qShadow + seg * pool.segCap, block - extBlocks, nShadow, lds, lane);
}

// The same stage with every ray traversed on its own by one lane (nearer child first, pruned by its own hits) and the
// boxes / triangles it tests added up: work[0..2] = extension rays, boxes, triangles; work[3..5] the same for shadow
// rays.  The figures the roofline model of the stage is defined on; results are those of the stage above.
__device__ __forceinline__ void wave_add64(unsigned long long* dst, uint32_t v, int lane)
{
    unsigned long long sum = v;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if (lane == 0 && sum) atomicAdd(dst, sum);
}
__device__ __forceinline__ TraceResult solo_closest(const SceneView& sv, F3 o, F3 d, int (*stack)[64], int lane, uint32_t& nb, uint32_t& nt)
{
    if (sv.singleMesh) return trace_ray<false, false, true>(sv, o, d, INFINITY, stack, lane, nb, nt);
    if (sv.refTop) return trace_ray_reftop_closest<true>(sv, o, d, stack, lane, nb, nt);
    return trace_ray<true, false, true>(sv, o, d, INFINITY, stack, lane, nb, nt);
}
__global__ __launch_bounds__(64) void trace_stage_count_kernel(SceneView sv, Pool pool, const uint32_t* __restrict__ extQueue,
                                                               const uint32_t* __restrict__ cnt, unsigned long long* __restrict__ work)
{
    __shared__ int stack[kLdsStack][64];
    const int lane = threadIdx.x;
    const uint32_t seg = blockIdx.x % kQueueSegs, block = blockIdx.x / kQueueSegs;
    const uint32_t nExt = cnt[kCtrExt + seg * kCtrLine], nShadow = cnt[kCtrShadow + seg * kCtrLine];
    const uint32_t extBlocks = (nExt + 63u) / 64u, shadowBlocks = (nShadow + 63u) / 64u;
    uint32_t nb = 0, nt = 0;
    if (block < extBlocks) {
        const uint32_t q = block * 64u + lane;
        const bool has = q < nExt;
        if (has) {
            const uint32_t p = extQueue[seg * pool.segCap + q] & kQueuePathMask;
            const float4 ra = pool.rayA[p], rb = pool.rayB[p];
            const TraceResult r = solo_closest(sv, f3(ra.x, ra.y, ra.z), f3(rb.x, rb.y, rb.z), stack, lane, nb, nt);
            pool.hit[p] = f4(r.t, __uint_as_float(r.slot), r.beta, r.gamma);
            if (!sv.singleMesh) pool.hitInst[p] = r.inst;
        }
        wave_add64(work + 0, has ? 1u : 0u, lane);
        wave_add64(work + 1, nb, lane);
        wave_add64(work + 2, nt, lane);
    } else if (block - extBlocks < shadowBlocks) {
        const uint32_t q = (block - extBlocks) * 64u + lane;
        const bool has = q < nShadow;
        if (has) {
            const uint32_t p = pool.qShadow[seg * pool.segCap + q];
            float4 ra = pool.rayA[p], rb = pool.rayB[p];
            const bool sticky = sv.hasGlass && (__float_as_uint(rb.w) & ST_GLASS_STICKY) != 0;
            if (sticky) { ra = pool.glassA[p]; rb = pool.glassB[p]; }
            const F3 o = f3(ra.x, ra.y, ra.z), d = f3(rb.x, rb.y, rb.z);
            bool occluded;
            if (sv.singleMesh) occluded = trace_ray<false, true, true>(sv, o, d, ra.w, stack, lane, nb, nt).inst != kInvalid;
            else if (sv.refTop) occluded = trace_ray_reftop_any(sv, o, d, ra.w, stack, lane);   // (its boxes are not counted)
            else occluded = trace_ray<true, true, true>(sv, o, d, ra.w, stack, lane, nb, nt).inst != kInvalid;
            if (sticky) {
                if (occluded) pool.glassB[p] = f4(rb.x, rb.y, rb.z, __uint_as_float(kInvalid));
            } else if (!occluded) {
                pool.result[p] = pool.thr[p];   // the pending contribution (see shade_kernel)
            }
        }
        wave_add64(work + 3, has ? 1u : 0u, lane);
        wave_add64(work + 4, nb, lane);
        wave_add64(work + 5, nt, lane);
    }
}

// ---- photon gather ----------------------------------------------------------------------------------
// One lane per path whose diffuse vertex reads the photon map: photonLighting = sum of the 20 nearest photons'
// energies * diffuseIntensity / (pi * maxDist^2) / pi, bsdf = diffuseColor * photonLighting
// (material/DiffuseMaterial.cpp:62-75, 154-166), added to what the shade stage stored for the path.
// The query of a queued path and what its result adds to the path's value.
__device__ __forceinline__ void gather_path_query(const SceneView& sv, const Pool& pool, uint32_t p, F3& q, F3& dir)
{
    const bool caustics = sv.pmMode == RTGPU_PM_CAUSTICS;
    const float4 a = caustics ? pool.cHit[p] : pool.rayA[p], b = caustics ? pool.cDir[p] : pool.rayB[p];
    q = f3(a.x, a.y, a.z);
    dir = f3(-b.x, -b.y, -b.z);
}
__device__ __forceinline__ void gather_path_apply(const SceneView& sv, const Pool& pool, uint32_t p, RGBf value, float maxDist)
{
    if (sv.pmMode == RTGPU_PM_CAUSTICS) {
        // the diffuse bounce before a specular vertex: value / (pi * maxDist^2) / pi, bsdf = colour * that * 2
        // (material/DiffuseMaterial.cpp:113-128, 154-164)
        const float4 tc = pool.cThrColor[p], cb = pool.cBase[p];
        const RGBf photonLighting = rgb_div_d(rgb_div_d(value, RT_PI * (maxDist * maxDist)), RT_PI);
        const RGBf out = rgb_scale_f(rgb_mul(rgb(tc.x, tc.y, tc.z), photonLighting), (float)2);
        pool.result[p] = f4(cb.x + out.r, cb.y + out.g, cb.z + out.b, 0.0f);
        return;
    }
    const float4 au = pool.aux[p], tq = pool.thr[p], base = pool.result[p];
    const DevMaterial& m = sv.materials[__float_as_uint(au.w)];
    const RGBf photonLighting = rgb_div_d(rgb_div_d(rgb_scale_d(value, m.intensity), (RT_PI * maxDist) * maxDist), RT_PI);
    const RGBf out = rgb_mul(rgb(tq.x, tq.y, tq.z), rgb_mul(rgb(au.x, au.y, au.z), photonLighting));
    pool.result[p] = f4(base.x + out.r, base.y + out.g, base.z + out.b, 0.0f);
}

// First attempt, one lane per queued path; the paths that run out of budget go to the tail queue (counters[4]).
__global__ __launch_bounds__(64) void photon_gather_stage_kernel(SceneView sv, Pool pool, uint32_t* __restrict__ cnt)
{
    const uint32_t n = cnt[2];   // the grid covers an upper bound
    const uint32_t qi = blockIdx.x * 64u + threadIdx.x;
    const bool has = qi < n;
    const uint32_t p = pool.qPhoton[has ? qi : 0u];
    F3 q, dir;
    gather_path_query(sv, pool, p, q, dir);
    uint32_t count;
    float maxDist;
    RGBf value;
    const bool finished = has && gather_photons(sv, q, dir, sv.gatherBudget, count, maxDist, value);
    if (finished) gather_path_apply(sv, pool, p, value, maxDist);
    queue_push1<64>(pool.qPhotonTail, cnt + 4, has && !finished, p);
}

// The tail queue, one wave per path.  Launched with a fixed grid right behind the first attempt: the number of
// queued paths is read on the device.
__global__ __launch_bounds__(64) void photon_gather_tail_kernel(SceneView sv, Pool pool, const uint32_t* __restrict__ cnt)
{
    __shared__ GatherWaveLds lds;
    const uint32_t n = cnt[4];
    for (uint32_t qi = blockIdx.x; qi < n; qi += gridDim.x) {
        const uint32_t p = pool.qPhotonTail[qi];
        F3 q, dir;
        gather_path_query(sv, pool, p, q, dir);
        uint32_t count;
        float maxDist;
        RGBf value;
        gather_photons_wave(sv, q, dir, lds, (int)threadIdx.x, count, maxDist, value);
        if (threadIdx.x == 0) gather_path_apply(sv, pool, p, value, maxDist);
    }
}

// ---- resolve ---------------------------------------------------------------------------------------
// pixelValue accumulation of renderer/Renderer.cpp:236-263: per geometry sample the material
// samples are added in order and divided by materialAA (only when a variance vertex exists),
// the sample values are added in order and divided by geometryAA.
__global__ __launch_bounds__(256) void resolve_kernel(Pool pool, rtgpu_render_settings st, Batch batch, float* __restrict__ fb)
{
    const uint32_t pl = blockIdx.x * 256u + threadIdx.x;
    if (pl >= batch.nPixels) return;
    const uint32_t G = (uint32_t)st.geometry_aa, M = (uint32_t)st.material_aa;
    int x, y;
    bool valid;
    uint32_t pixel = tile_pixel(st, batch, batch.pixelStart + pl, x, y, valid);
    if (batch.tileSize != 0u) pixel = batch.pixelStart + pl;   // compact buffer: the call's pixel order
    RGBf px = rgb(0, 0, 0);
    for (uint32_t i = 0; i < G; ++i) {
        const uint32_t s = pl * G + i;
        const float4* r = pool.result + (size_t)s * M;
        RGBf m = rgb(r[0].x, r[0].y, r[0].z);
        if (pool.sLive[s] && pool.sHasVar[s]) {
            for (uint32_t j = 1; j < M; ++j) { const float4 v = r[j]; m.r += v.x; m.g += v.y; m.b += v.z; }
            m = rgb_scale_f(m, 1.0f / (float)M);
        }
        px.r += m.r; px.g += m.g; px.b += m.b;
    }
    px = rgb_scale_f(px, 1.0f / (float)G);
    fb[3 * (size_t)pixel] = px.r; fb[3 * (size_t)pixel + 1] = px.g; fb[3 * (size_t)pixel + 2] = px.b;
}

// The same sums with M = 2, 4, 8 or 16 lanes per pixel: lane j of a pixel's group loads result j of a sample (the group's
// loads are one run of M * 16 bytes instead of one 16-byte piece per lane and kilobyte), and the sample's sum is formed in
// the same order -- ((r0 + r1) + r2) + ... -- by M - 1 additions whose second operand comes from the lane to the left
// (DPP row shift): after step s lane j holds the left-to-right sum of results j - s .. j, so lane M - 1 ends with the sum
// resolve_kernel forms.  Bit-identical pixels; one thread per pixel left the kernel waiting on its loads (96 % of its cycles).
template <int M>
__global__ __launch_bounds__(256) void resolve_lanes_kernel(Pool pool, rtgpu_render_settings st, Batch batch, float* __restrict__ fb)
{
    constexpr uint32_t kPerBlock = 256u / (uint32_t)M;
    const uint32_t j = threadIdx.x % (uint32_t)M;
    const uint32_t pl = blockIdx.x * kPerBlock + threadIdx.x / (uint32_t)M;
    const bool has = pl < batch.nPixels;
    const uint32_t plc = has ? pl : 0u;
    const uint32_t G = (uint32_t)st.geometry_aa;
    RGBf px = rgb(0, 0, 0);
    // four samples at a time: their flags are read first, then their results, so that four loads per lane are in flight
    for (uint32_t i0 = 0; i0 < G; i0 += 4u) {
        bool all[4];
        float4 v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t s = plc * G + (i0 + u < G ? i0 + u : G - 1u);
            all[u] = pool.sLive[s] && pool.sHasVar[s];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t s = plc * G + i0 + u;
            // (a sample without a variance vertex only has result 0: the other lanes read nothing)
            v[u] = f4(0.0f, 0.0f, 0.0f, 0.0f);
            if (has && i0 + u < G && (all[u] || j == 0u)) v[u] = pool.result[(size_t)s * M + j];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (i0 + u >= G) break;   // uniform
            float ar = v[u].x, ag = v[u].y, ab = v[u].z;
#pragma unroll
            for (int step = 1; step < M; ++step) {
                // row_shr:1 = 0x111; lanes without a left neighbour in their row get 0 (they are lane 0 of a group: not read)
                ar = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ar), 0x111, 0xF, 0xF, true)) + v[u].x;
                ag = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ag), 0x111, 0xF, 0xF, true)) + v[u].y;
                ab = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ab), 0x111, 0xF, 0xF, true)) + v[u].z;
            }
            // lane M - 1 now holds ((r0 + r1) + ...) + r(M-1); a sample without variance is its result 0 alone, which lane 0 holds
            RGBf m = rgb(ar, ag, ab);
            if (all[u]) m = rgb_scale_f(m, 1.0f / (float)M);
            else {
                // bring lane 0's value to lane M - 1 (M - 1 more shifts would do; one ds_bpermute is cheaper)
                const int src = (int)(threadIdx.x & 63u) - (M - 1);
                m.r = __shfl(v[u].x, src, 64); m.g = __shfl(v[u].y, src, 64); m.b = __shfl(v[u].z, src, 64);
            }
            px.r += m.r; px.g += m.g; px.b += m.b;
        }
    }
    if (!has || j != (uint32_t)(M - 1)) return;
    int x, y;
    bool valid;
    uint32_t pixel = tile_pixel(st, batch, batch.pixelStart + pl, x, y, valid);
    if (batch.tileSize != 0u) pixel = batch.pixelStart + pl;   // compact buffer: the call's pixel order
    px = rgb_scale_f(px, 1.0f / (float)G);
    fb[3 * (size_t)pixel] = px.r; fb[3 * (size_t)pixel + 1] = px.g; fb[3 * (size_t)pixel + 2] = px.b;
}

// Primary rays of a batch traversed on their own, only to count their work: work[6..8] = rays, boxes, triangles.
__global__ __launch_bounds__(64) void primary_count_kernel(SceneView sv, rtgpu_camera cam, rtgpu_render_settings st, Batch batch,
                                                           unsigned long long* __restrict__ work)
{
    __shared__ int stack[kLdsStack][64];
    const int lane = threadIdx.x;
    const uint32_t G = (uint32_t)st.geometry_aa;
    const uint32_t s0 = blockIdx.x * 64u + lane;
    const bool inBatch = s0 < batch.nPixels * G;
    const uint32_t s = inBatch ? s0 : 0u;
    int x, y;
    bool valid;
    tile_pixel(st, batch, batch.pixelStart + s / G, x, y, valid);
    uint32_t nb = 0, nt = 0;
    const bool has = inBatch && valid;
    if (has) {
        F3 o, d;
        camera_ray(cam, st, x, y, (int)(s % G), o, d);
        (void)solo_closest(sv, o, d, stack, lane, nb, nt);
    }
    wave_add64(work + 6, has ? 1u : 0u, lane);
    wave_add64(work + 7, nb, lane);
    wave_add64(work + 8, nt, lane);
}

constexpr int kMaxSteps = 2 * 64 + 4;       // shade passes of one batch: two per path vertex (max_path_length <= 64) and the drain
constexpr size_t kCarveAlign = 256;

// What a batch's workspace has to hold.
struct PoolDims {
    uint64_t P = 0, S = 0;   // paths and samples resident per batch
    bool single = false, glass = false, add = false, photons = false, caustics = false;
    bool operator==(const PoolDims& o) const
    {
        return P == o.P && S == o.S && single == o.single && glass == o.glass && add == o.add && photons == o.photons && caustics == o.caustics;
    }
};

PoolDims pool_dims(const SceneView& sv, uint64_t P, uint64_t S)
{
    PoolDims d;
    d.P = P; d.S = S;
    d.single = sv.singleMesh != 0;
    d.glass = sv.hasGlass != 0;
    d.add = sv.hasAdd != 0;
    d.photons = sv.pmMode != RTGPU_PM_NONE;
    d.caustics = sv.pmMode == RTGPU_PM_CAUSTICS;
    return d;
}

// Lays the arrays of a Pool out in one block of device memory starting at `base` (null: only the size is computed).
// Returns the bytes needed.  The per-step counters come last: counters[step][kCtrStride].
// What the shade workgroups of one residue (mod kQueueSegs) of a batch of nPaths paths can append to their segment.
inline uint32_t queue_segment_capacity(uint64_t nPaths)
{
    const uint64_t blocks = (nPaths + kShadeBlock - 1) / kShadeBlock;
    return (uint32_t)((blocks + kQueueSegs - 1) / kQueueSegs * kShadeBlock);
}

size_t pool_carve(Pool& pool, char* base, const PoolDims& d)
{
    size_t off = 0;
    auto take = [&](size_t bytes) -> char* {
        char* p = base ? base + off : nullptr;
        off += (std::max<size_t>(bytes, 1) + kCarveAlign - 1) / kCarveAlign * kCarveAlign;
        return p;
    };
    const size_t P = (size_t)d.P, S = (size_t)d.S;
    // (the shade kernel addresses the blocks' records as rayA.base[4 * path + 0 .. 3])
    float4* const blocks = (float4*)take(P * 64);
    pool.rayA.base = blocks; pool.rayB.base = blocks ? blocks + 1 : nullptr; pool.hit.base = blocks ? blocks + 2 : nullptr;
    pool.thr.base = blocks ? blocks + 3 : nullptr;
    pool.hitInst = (uint32_t*)take(d.single ? 4 : P * 4);
    pool.aux = (float4*)take(P * 16); pool.result = (float4*)take(P * 16);
    pool.glassN = (float4*)take(d.glass ? P * 16 : 16); pool.glassA = (float4*)take(d.glass ? P * 16 : 16); pool.glassB = (float4*)take(d.glass ? P * 16 : 16);
    pool.acc = (float4*)take(d.add ? P * 16 : 16); pool.accPend = (float4*)take(d.add ? P * 16 : 16);
    pool.sA = (float4*)take(S * 16); pool.sB = (float4*)take(S * 16); pool.sC = (float4*)take(S * 16);
    pool.sLive = (uint8_t*)take(S); pool.sHasVar = (uint8_t*)take(S);
    const size_t Q = (size_t)kQueueSegs * queue_segment_capacity(d.P) * 4;   // (the segments' roundings: up to 256 entries each more than P)
    pool.qA = (uint32_t*)take(Q); pool.qB = (uint32_t*)take(Q); pool.qShadow = (uint32_t*)take(Q);
    pool.qPhoton = (uint32_t*)take(d.photons ? P * 4 : 4);
    pool.qPhotonTail = (uint32_t*)take(d.photons ? P * 4 : 4);
    const size_t nC = d.caustics ? P * 16 : 16;
    pool.cHit = (float4*)take(nC); pool.cDir = (float4*)take(nC); pool.cThrColor = (float4*)take(nC); pool.cBase = (float4*)take(nC);
    pool.counters = (uint32_t*)take((size_t)kMaxSteps * kCtrStride * sizeof(uint32_t));
    return off;
}

// Events around the launches of a profiling render: spans[k] = (stage, first event, second event).
enum Stage : int { kStagePrimary = 0, kStageShade = 1, kStageTrace = 2, kStageGather = 3, kStageResolve = 4, kStageCount = 5 };
struct StageTimer {
    std::vector<hipEvent_t> events;
    size_t used = 0;
    struct Span { int stage; size_t a, b; };
    std::vector<Span> spans;
    hipError_t mark(hipStream_t s, size_t& idx)
    {
        if (used == events.size()) {
            hipEvent_t ev = nullptr;
            const hipError_t e = hipEventCreate(&ev);
            if (e != hipSuccess) return e;
            events.push_back(ev);
        }
        idx = used++;
        return hipEventRecord(events[idx], s);
    }
    ~StageTimer() { for (hipEvent_t ev : events) (void)hipEventDestroy(ev); }
};

// One batch in flight: a stream, its workspace, and (unless the caller must not be waited on) the pinned mirror of
// the per-step counters with the events that say when a step's counters have arrived.
struct LaneRun {
    hipStream_t stream = nullptr;
    Pool pool{};
    uint32_t* hostCounters = nullptr;      // kMaxSteps * kCtrStride words, pinned; null: never read back
    hipEvent_t stepEv[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long* work = nullptr;    // device, 9 counters: the counting render
    StageTimer* timer = nullptr;           // the profiling render: events around every launch
    uint64_t extension = 0, shadow = 0;
};

template <typename K, typename... A>
inline void launch(K kernel, uint32_t blocks, uint32_t threads, hipStream_t s, A... args)
{
    hipLaunchKernelGGL(kernel, dim3(std::max(blocks, 1u)), dim3(threads), 0, s, args...);
}

// Enqueues one batch: primary, then shade / trace passes until no path is left, then resolve.
//
// The host never waits for the pass it has just launched.  Queue lengths live in device memory (one set of counters
// per pass) and every kernel reads its own; a grid only needs an upper bound of its queue, and the queue of pass s is
// at most as long as that of pass s - 1, which the host reads two passes late (by then it has long arrived).  Without a
// pinned mirror (callers that may not synchronise) every pass runs with the bound of the first and the full number
// of passes is enqueued.
hipError_t run_batch(const SceneView& sv, LaneRun& ln, const rtgpu_camera& cam, const rtgpu_render_settings& st, const Batch& batch, float* dOut,
                     std::string& err)
{
    const hipStream_t ls = ln.stream;
    Pool& pool = ln.pool;
    const uint32_t G = (uint32_t)st.geometry_aa, M = (uint32_t)st.material_aa;
    const uint32_t nS = batch.nPixels * G;
    const uint32_t nP = nS * M;
    const int mode = trace_mode(sv);
    const bool rich = sv.hasGlass || sv.hasAdd || sv.hasNormalMap || sv.pmMode != RTGPU_PM_NONE;
    static const uint32_t tailCap = std::getenv("RTGPU_GATHER_TAIL_BLOCKS") ? (uint32_t)std::atoi(std::getenv("RTGPU_GATHER_TAIL_BLOCKS")) : 16384u;
    const int maxSteps = std::min(kMaxSteps, 2 * st.max_path_length + 2);
    hipError_t e = hipSuccess;
#define RB_TRY(call) do { e = (call); if (e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e); return e; } } while (0)
    size_t spanA = 0, spanB = 0;
#define TB() do { if (ln.timer) RB_TRY(ln.timer->mark(ls, spanA)); } while (0)
#define TE(stage) do { if (ln.timer) { RB_TRY(ln.timer->mark(ls, spanB)); ln.timer->spans.push_back(StageTimer::Span{stage, spanA, spanB}); } } while (0)
    RB_TRY(hipMemsetAsync(pool.counters, 0, (size_t)maxSteps * kCtrStride * sizeof(uint32_t), ls));
    TB();
    if (mode == kTraceSingle) launch(primary_kernel<kTraceSingle>, (nS + 63) / 64, 64, ls, sv, pool, cam, st, batch);
    else if (mode == kTraceInstances) launch(primary_kernel<kTraceInstances>, (nS + 63) / 64, 64, ls, sv, pool, cam, st, batch);
    else launch(primary_kernel<kTraceLanes>, (nS + 63) / 64, 64, ls, sv, pool, cam, st, batch);
    TE(kStagePrimary);
    if (ln.work) launch(primary_count_kernel, (nS + 63) / 64, 64, ls, sv, cam, st, batch, ln.work);
    pool.segCap = queue_segment_capacity(nP);
    uint32_t bound = pool.segCap;  // upper bound of the number of paths of the longest queue segment the next pass shades
    uint32_t* qIn = pool.qA;
    uint32_t* qOut = pool.qB;
    int steps = 0;
    for (int step = 0; step < maxSteps; ++step) {
        if (ln.hostCounters && step >= 2) {
            RB_TRY(hipEventSynchronize(ln.stepEv[(step - 2) & 3]));
            const uint32_t* h = ln.hostCounters + (size_t)(step - 2) * kCtrStride;   // the paths pass step - 1 shaded, by segment
            bound = 0;
            for (uint32_t sg = 0; sg < kQueueSegs; ++sg) bound = std::max(bound, h[kCtrExt + sg * kCtrLine]);
            if (bound == 0) break;
        }
        uint32_t* cnt = pool.counters + (size_t)step * kCtrStride;
        const uint32_t* cntIn = step ? cnt - kCtrStride : cnt;
        const uint32_t shadeBlocks = step == 0 ? (nP + kShadeBlock - 1) / kShadeBlock : kQueueSegs * ((bound + kShadeBlock - 1) / kShadeBlock);
        TB();
        const ShadeArgs sa{sv, pool, st, batch, (const uint32_t*)qIn, nP, cntIn, qOut, cnt};
        if (rich) {
            if (step == 0) launch(shade_kernel<true, true>, shadeBlocks, kShadeBlock, ls, sa);
            else launch(shade_kernel<false, true>, shadeBlocks, kShadeBlock, ls, sa);
        } else {
            if (step == 0) launch(shade_kernel<true, false>, shadeBlocks, kShadeBlock, ls, sa);
            else launch(shade_kernel<false, false>, shadeBlocks, kShadeBlock, ls, sa);
        }
        TE(kStageShade);
        if (ln.hostCounters) {
            RB_TRY(hipMemcpyAsync(ln.hostCounters + (size_t)step * kCtrStride, cnt, kCtrStride * sizeof(uint32_t), hipMemcpyDeviceToHost, ls));
            RB_TRY(hipEventRecord(ln.stepEv[step & 3], ls));
        }
        if (sv.pmMode != RTGPU_PM_NONE) {
            TB();
            const uint32_t shaded = (uint32_t)std::min<uint64_t>((uint64_t)bound * kQueueSegs, nP);   // (the photon queue is one list)
            launch(photon_gather_stage_kernel, (shaded + 63) / 64, 64, ls, sv, pool, cnt);
            launch(photon_gather_tail_kernel, std::min(shaded, tailCap), 64, ls, sv, pool, (const uint32_t*)cnt);
            TE(kStageGather);
        }
        // A shaded path leaves at most one ray behind -- an extension ray or a shadow ray -- so both queues of a segment
        // together are at most `bound` long (two block counts rounded up: + 2); only a glass vertex can leave both (its own
        // ray and the look along the direction it did not take), then each can be `bound` long.  Surplus workgroups find
        // nothing to do and leave, but their launches are not free.
        const uint32_t traceBlocks = kQueueSegs * (sv.hasGlass ? 2 * ((bound + 63) / 64) : (bound + 63) / 64 + 2);
        TB();
        if (ln.work) launch(trace_stage_count_kernel, traceBlocks, 64, ls, sv, pool, (const uint32_t*)qOut, (const uint32_t*)cnt, ln.work);
        else if (mode == kTraceSingle) launch(trace_stage_kernel<kTraceSingle>, traceBlocks, 64, ls, sv, pool, (const uint32_t*)qOut, (const uint32_t*)cnt);
        else if (mode == kTraceInstances) launch(trace_stage_kernel<kTraceInstances>, traceBlocks, 64, ls, sv, pool, (const uint32_t*)qOut, (const uint32_t*)cnt);
        else launch(trace_stage_kernel<kTraceLanes>, traceBlocks, 64, ls, sv, pool, (const uint32_t*)qOut, (const uint32_t*)cnt);
        TE(kStageTrace);
        std::swap(qIn, qOut);
        steps = step + 1;
    }
    TB();
    const char* const resolveEnv = std::getenv("RTGPU_RESOLVE_LANES");   // 0: one thread per pixel for every M
    const bool resolveLanes = !(resolveEnv && std::atoi(resolveEnv) == 0);
    switch (resolveLanes ? M : 0u) {
    case 2: launch(resolve_lanes_kernel<2>, (batch.nPixels + 127) / 128, 256, ls, pool, st, batch, dOut); break;
    case 4: launch(resolve_lanes_kernel<4>, (batch.nPixels + 63) / 64, 256, ls, pool, st, batch, dOut); break;
    case 8: launch(resolve_lanes_kernel<8>, (batch.nPixels + 31) / 32, 256, ls, pool, st, batch, dOut); break;
    case 16: launch(resolve_lanes_kernel<16>, (batch.nPixels + 15) / 16, 256, ls, pool, st, batch, dOut); break;
    default: launch(resolve_kernel, (batch.nPixels + 255) / 256, 256, ls, pool, st, batch, dOut); break;
    }
    TE(kStageResolve);
    RB_TRY(hipGetLastError());
    if (ln.hostCounters && steps > 0) {
        // ray statistics: every pass's counters have arrived once the last pass's have
        RB_TRY(hipEventSynchronize(ln.stepEv[(steps - 1) & 3]));
        for (int s2 = 0; s2 < steps; ++s2) {
            const uint32_t* h = ln.hostCounters + (size_t)s2 * kCtrStride;
            for (uint32_t sg = 0; sg < kQueueSegs; ++sg) {
                ln.extension += h[kCtrExt + sg * kCtrLine];
                ln.shadow += h[kCtrShadow + sg * kCtrLine];
            }
            ln.extension -= h[3];
        }
    }
#undef TB
#undef TE
#undef RB_TRY
    return hipSuccess;
}

} // namespace

void area_light_derived(DevAreaLight& l)
{
    // AreaLight::getNormal (light/AreaLight.cpp:43-48) and Triangle::getSurfaceArea (math/Triangle.h:7-18), fp32
    auto sub = [](const float* a, const float* b, float* r) { for (int k = 0; k < 3; ++k) r[k] = a[k] - b[k]; };
    auto dot = [](const float* a, const float* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); };
    float ab[3], ac[3];
    sub(l.b, l.a, ab);
    sub(l.c, l.a, ac);
    float n[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
    const float z = dot(n, n);
    if (z > 0) { const float len = std::sqrt(z); for (float& c : n) c /= len; }
    for (int k = 0; k < 3; ++k) l.normal[k] = n[k];
    const float v1Length = std::sqrt(dot(ab, ab));
    float v1n[3] = {ab[0] / v1Length, ab[1] / v1Length, ab[2] / v1Length};
    const float pr = dot(ac, v1n);
    float h[3] = {ac[0] - v1n[0] * pr, ac[1] - v1n[1] * pr, ac[2] - v1n[2] * pr};
    const float height = std::sqrt(dot(h, h));
    l.area = v1Length * height / 2.0f;
    // RGB::scale(double) / RGB::divide(double) (film/RGB.h:44-47, 60-63): promote, multiply, narrow, per channel
    const double pi = 3.14159265358979323;   // math/Constants.h
    for (int k = 0; k < 3; ++k) {
        const float e = (float)(l.color[k] * l.intensity);
        l.irradiance[k] = (float)(e * (1.0 / (double)l.area));
        l.radiance[k] = (float)(l.irradiance[k] * (1.0 / pi));
    }
}

// ---- host side ---------------------------------------------------------------------------------------
// Streams, events, pinned counters and workspaces of the batches in flight are kept with the scene between calls
// (a frame rendered as many tiles, or frame after frame, would otherwise pay about 25 allocations of up to 2 GB per
// call); they grow when a call needs more and go when the scene does.
struct RenderCache {
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        hipEvent_t stepEv[4] = {nullptr, nullptr, nullptr, nullptr};
        uint32_t* hostCounters = nullptr;
        void* mem = nullptr;
        size_t bytes = 0;
        PoolDims dims;
        Pool pool{};
    };
    std::vector<Lane> lanes;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* work = nullptr;   // 9 counters of the counting render
};

void destroy_render_cache(RenderCache* c)
{
    if (!c) return;
    for (RenderCache::Lane& l : c->lanes) {
        if (l.stream) (void)hipStreamDestroy(l.stream);
        if (l.done) (void)hipEventDestroy(l.done);
        for (hipEvent_t ev : l.stepEv) if (ev) (void)hipEventDestroy(ev);
        if (l.hostCounters) (void)hipHostFree(l.hostCounters);
        if (l.mem) (void)hipFree(l.mem);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->work) (void)hipFree(c->work);
    delete c;
}

namespace {

// Pixels the call covers, in its pixel order, and the Batch fields that describe that order.
uint64_t call_pixels(const rtgpu_render_settings& st, const TileOrder* order, Batch& proto)
{
    const uint32_t tw = (uint32_t)(st.x_end - st.x_start), th = (uint32_t)(st.y_end - st.y_start);
    proto = Batch{};
    proto.sqrtM = (uint32_t)(int)std::sqrt((double)st.material_aa);
    proto.divM = fast_div_make((uint32_t)st.material_aa);
    proto.divG = fast_div_make((uint32_t)st.geometry_aa);
    proto.divSqrtM = fast_div_make(proto.sqrtM);
    proto.divTw = fast_div_make(tw);
    if (!order || order->tileSize == 0) return (uint64_t)tw * th;
    const uint32_t T = order->tileSize;
    proto.tileSize = T;
    proto.nRanks = order->nRanks;
    proto.rank = order->rank;
    proto.tilesX = (tw + T - 1) / T;
    proto.nTiles = proto.tilesX * ((th + T - 1) / T);
    proto.divTT = fast_div_make(T * T);
    proto.divT = fast_div_make(T);
    proto.divTilesX = fast_div_make(proto.tilesX);
    const uint32_t mine = proto.nTiles > order->rank ? (proto.nTiles - order->rank + order->nRanks - 1) / order->nRanks : 0u;
    return (uint64_t)mine * T * T;
}

hipError_t ensure_lane(RenderCache::Lane& l, const PoolDims& dims, std::string& err)
{
    hipError_t e = hipSuccess;
#define EL_TRY(call) do { e = (call); if (e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e); return e; } } while (0)
    if (!l.stream) EL_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    if (!l.done) EL_TRY(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
    for (hipEvent_t& ev : l.stepEv) if (!ev) EL_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (!l.hostCounters) EL_TRY(hipHostMalloc((void**)&l.hostCounters, (size_t)kMaxSteps * kCtrStride * sizeof(uint32_t)));
    if (!(l.dims == dims) || !l.mem) {
        Pool probe{};
        const size_t need = pool_carve(probe, nullptr, dims);
        if (need > l.bytes) {
            if (l.mem) { EL_TRY(hipStreamSynchronize(l.stream)); (void)hipFree(l.mem); l.mem = nullptr; l.bytes = 0; }
            e = hipMalloc(&l.mem, need);
            if (e != hipSuccess) { err = std::string("render workspace: ") + hipGetErrorString(e); return e; }
            l.bytes = need;
        }
        pool_carve(l.pool, (char*)l.mem, dims);
        l.dims = dims;
    }
#undef EL_TRY
    return hipSuccess;
}

} // namespace

#ifdef RT_SHADE_STATS
struct ShadeStatsPrinter {
    ~ShadeStatsPrinter()
    {
        unsigned long long h[64];
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shade_stats), sizeof(h)) != hipSuccess) return;
        static const char* names[] = {"entered", "active after regroup", "glossy pending", "miss after bounce", "vertex evaluated", "last vertex",
                                      "diffuse nee", "diffuse bounce", "glossy nee", "glossy bounce", "state written"};
        for (int k = 0; k < 11; ++k)
            std::fprintf(stderr, "shade stat %-22s lanes %14llu waves %12llu lanes/wave %.1f\n", names[k], h[2 * k], h[2 * k + 1], h[2 * k + 1] ? (double)h[2 * k] / (double)h[2 * k + 1] : 0.0);
        static const char* regions[] = {"whole kernel", "until regrouped", "loads, pending vertex", "vertex set-up", "light sample", "bounce", "state written",
                                        "queues"};
        static unsigned long long c[64][16];
        if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_shade_clocks), sizeof(c)) != hipSuccess) return;
        unsigned long long tot[16] = {};
        for (int b = 0; b < 64; ++b)
            for (int k = 0; k < 16; ++k) tot[k] += c[b][k];
        for (int k = 0; k < 8; ++k)
            std::fprintf(stderr, "shade clock %-22s waves %12llu cycles/wave %10.0f share %.3f\n", regions[k], tot[2 * k + 1],
                         tot[2 * k + 1] ? (double)tot[2 * k] / (double)tot[2 * k + 1] : 0.0, tot[0] ? (double)tot[2 * k] / (double)tot[0] : 0.0);
        std::memset(c, 0, sizeof(c));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shade_clocks), c, sizeof(c));
        std::memset(h, 0, sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shade_stats), h, sizeof(h));
    }
};
#endif

hipError_t render_wavefront(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, const TileOrder* order,
                            float* dOut, rtgpu_render_stats* stats, uint64_t* work, hipStream_t stream, std::string& err, rtgpu_render_profile* profile)
{
#ifdef RT_SHADE_STATS
    ShadeStatsPrinter shadeStatsPrinter;
#endif
    const SceneView& sv = scene.view;
    std::lock_guard<std::mutex> cacheLock(scene.cacheMutex);
    Batch proto;
    const uint64_t nPixelsTotal = call_pixels(st, order, proto);
    if (stats) *stats = rtgpu_render_stats{};
    if (work) std::memset(work, 0, 9 * sizeof(uint64_t));
    if (nPixelsTotal == 0) return hipSuccess;
    if (nPixelsTotal >= (1ull << 32)) { err = "more than 2^32 pixels in one call"; return hipErrorInvalidValue; }
    const uint32_t G = (uint32_t)st.geometry_aa, M = (uint32_t)st.material_aa;
    const uint64_t pathsPerPixel = (uint64_t)G * M;

    // paths resident per batch (about 110 bytes of state each): as many as a queue entry's 25 bits of path allow (on the 4K
    // frame 2^24 measured 1.5 % slower, 2^23 3 %), but a call is cut into at least two batches, so that two are in flight
    int lanes = 2;
    if (const char* e = std::getenv("RTGPU_RENDER_LANES")) lanes = std::max(1, std::min(8, std::atoi(e)));
    uint64_t capPaths = 1ull << kQueuePathBits;
    bool capGiven = false;
    if (const char* e = std::getenv("RTGPU_BATCH_PATHS")) {
        const long long v = std::atoll(e);
        if (v > 0) { capPaths = std::min<uint64_t>((uint64_t)v, 1ull << kQueuePathBits); capGiven = true; }
    }
    uint64_t pixelsPerBatch = std::max<uint64_t>(1, capPaths / pathsPerPixel);
    pixelsPerBatch = std::min<uint64_t>(pixelsPerBatch, nPixelsTotal);
    if (!capGiven && !profile && nPixelsTotal * pathsPerPixel >= (1ull << 22) && (nPixelsTotal + pixelsPerBatch - 1) / pixelsPerBatch < (uint64_t)lanes)
        pixelsPerBatch = (nPixelsTotal + (uint64_t)lanes - 1) / (uint64_t)lanes;
    const uint64_t P = pixelsPerBatch * pathsPerPixel;
    const uint64_t S = pixelsPerBatch * G;
    // a batch holds whole pixels and a queue entry 25 bits of path: more paths per pixel than that would alias with the vertex bits
    if (P > (1ull << kQueuePathBits)) { err = "geometry_aa * material_aa exceeds 2^25 paths per pixel"; return hipErrorInvalidValue; }

    // Batches are independent (disjoint pixels), so `lanes` of them are in flight at once, each on its own stream
    // with its own workspace and host thread: the thin late-bounce launches of one batch overlap with the other
    // batch's kernels.
    const uint64_t nBatches = (nPixelsTotal + pixelsPerBatch - 1) / pixelsPerBatch;
    lanes = (int)std::min<uint64_t>((uint64_t)lanes, nBatches);
    if (profile) lanes = 1;   // one batch at a time: the launches of two streams would overlap and stretch each other's spans
    StageTimer timer;

    if (!scene.renderCache) scene.renderCache = new RenderCache();
    RenderCache& rc = *scene.renderCache;
    hipError_t e = hipSuccess;
#define RT_TRY(call) do { e = (call); if (e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e); return e; } } while (0)
    if (!rc.ev0) RT_TRY(hipEventCreate(&rc.ev0));
    if (!rc.ev1) RT_TRY(hipEventCreate(&rc.ev1));
    if ((int)rc.lanes.size() < lanes) rc.lanes.resize((size_t)lanes);
    const PoolDims dims = pool_dims(sv, P, S);
    for (int li = 0; li < lanes; ++li)
        if ((e = ensure_lane(rc.lanes[(size_t)li], dims, err)) != hipSuccess) return e;
    if (work) {
        if (!rc.work) RT_TRY(hipMalloc((void**)&rc.work, 9 * sizeof(unsigned long long)));
        RT_TRY(hipMemsetAsync(rc.work, 0, 9 * sizeof(unsigned long long), stream));
    }
    RT_TRY(hipEventRecord(rc.ev0, stream));
    for (int li = 0; li < lanes; ++li) RT_TRY(hipStreamWaitEvent(rc.lanes[(size_t)li].stream, rc.ev0, 0));

    std::vector<LaneRun> runs((size_t)lanes);
    std::vector<hipError_t> laneErr((size_t)lanes, hipSuccess);
    std::vector<std::string> laneMsg((size_t)lanes);
    auto run_lane = [&](int li) {
        RenderCache::Lane& cl = rc.lanes[(size_t)li];
        LaneRun& ln = runs[(size_t)li];
        hipError_t le = hipSetDevice(scene.device);
        if (le != hipSuccess) { laneErr[(size_t)li] = le; laneMsg[(size_t)li] = "hipSetDevice failed"; return; }
        ln.stream = cl.stream;
        ln.pool = cl.pool;
        ln.hostCounters = cl.hostCounters;
        for (int k = 0; k < 4; ++k) ln.stepEv[k] = cl.stepEv[k];
        ln.work = work ? rc.work : nullptr;
        ln.timer = profile ? &timer : nullptr;
        for (uint64_t bi = (uint64_t)li; bi < nBatches; bi += (uint64_t)lanes) {
            const uint64_t start = bi * pixelsPerBatch;
            Batch batch = proto;
            batch.pixelStart = (uint32_t)start;
            batch.nPixels = (uint32_t)std::min<uint64_t>(pixelsPerBatch, nPixelsTotal - start);
            le = run_batch(sv, ln, cam, st, batch, dOut, laneMsg[(size_t)li]);
            if (le != hipSuccess) { laneErr[(size_t)li] = le; return; }
        }
        le = hipEventRecord(cl.done, cl.stream);
        if (le != hipSuccess) { laneErr[(size_t)li] = le; laneMsg[(size_t)li] = "hipEventRecord failed"; }
    };
    {
        std::vector<std::thread> threads;
        for (int li = 1; li < lanes; ++li) threads.emplace_back(run_lane, li);
        run_lane(0);
        for (std::thread& t : threads) t.join();
    }
    for (int li = 0; li < lanes; ++li)
        if (laneErr[(size_t)li] != hipSuccess) {
            // leave nothing in flight on the cached streams before reporting
            for (int k = 0; k < lanes; ++k) (void)hipStreamSynchronize(rc.lanes[(size_t)k].stream);
            err = laneMsg[(size_t)li];
            return laneErr[(size_t)li];
        }
    for (int li = 0; li < lanes; ++li) RT_TRY(hipStreamWaitEvent(stream, rc.lanes[(size_t)li].done, 0));
    RT_TRY(hipEventRecord(rc.ev1, stream));
    RT_TRY(hipStreamSynchronize(stream));
    if (stats) {
        float ms = 0;
        RT_TRY(hipEventElapsedTime(&ms, rc.ev0, rc.ev1));
        for (const LaneRun& l : runs) { stats->extension_rays += l.extension; stats->shadow_rays += l.shadow; }
        // one primary ray per geometry sample of every pixel of the call (the padding of squares that stick out is not traced)
        uint64_t pixels = nPixelsTotal;
        if (proto.tileSize) {
            pixels = 0;
            const uint32_t T = proto.tileSize, tw = (uint32_t)(st.x_end - st.x_start), th = (uint32_t)(st.y_end - st.y_start);
            for (uint32_t k = proto.rank; k < proto.nTiles; k += proto.nRanks) {
                const uint32_t x0 = (k % proto.tilesX) * T, y0 = (k / proto.tilesX) * T;
                pixels += (uint64_t)std::min(T, tw - x0) * std::min(T, th - y0);
            }
        }
        stats->primary_rays = pixels * G;
        stats->device_ms = ms;
    }
    if (profile) {
        *profile = rtgpu_render_profile{};
        double* ms[kStageCount] = {&profile->primary_ms, &profile->shade_ms, &profile->trace_ms, &profile->gather_ms, &profile->resolve_ms};
        uint32_t* n[kStageCount] = {&profile->primary_launches, &profile->shade_launches, &profile->trace_launches, &profile->gather_launches, &profile->resolve_launches};
        for (const StageTimer::Span& sp : timer.spans) {
            float t = 0;
            RT_TRY(hipEventElapsedTime(&t, timer.events[sp.a], timer.events[sp.b]));
            *ms[sp.stage] += t;
            *n[sp.stage] += 1;
        }
        float total = 0;
        RT_TRY(hipEventElapsedTime(&total, rc.ev0, rc.ev1));
        profile->total_ms = total;
    }
    if (work) {
        unsigned long long h[9];
        RT_TRY(hipMemcpy(h, rc.work, sizeof(h), hipMemcpyDeviceToHost));
        for (int k = 0; k < 9; ++k) work[k] = h[k];
    }
#undef RT_TRY
    return hipSuccess;
}

// ---- the render that allocates nothing -----------------------------------------------------------------
// For callers that cannot allocate device memory while they run (serving loops, harnesses that forbid hipMalloc, code
// inside a stream capture): the integrator's workspace is a fixed array inside the library's code object instead of a
// hipMalloc'ed block, batches are sized to fit it, everything is enqueued on the caller's stream, and nothing is read back
// (every pass is launched over the batch's full path count and finds its queue length on the device).
namespace {

static __device__ __align__(256) unsigned char scratch[RTGPU_STATIC_WORKSPACE_BYTES];

constexpr int kMaxDevices = 64;
void* g_scratchPtr[kMaxDevices] = {};
std::mutex g_scratchMutex;

// The array's address on `device` (every GPU holds its own copy at its own address); looked up once per device with that
// device current.  In host code the array's name is only a handle for this look-up, never an address to hand to a kernel.
hipError_t static_workspace(int device, void** out)
{
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!g_scratchPtr[device]) {
        int before = 0;
        hipError_t e = hipGetDevice(&before);
        if (e != hipSuccess) return e;
        e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        void* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        (void)hipSetDevice(before);   // the caller's device stays what it was
        if (e != hipSuccess) return e;
        g_scratchPtr[device] = ptr;
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

} // namespace

void static_workspace_prepare(int device)
{
    void* p = nullptr;
    (void)static_workspace(device, &p);   // a failure is reported by the render that needs the array
}

int render_static(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, float* dRgb, hipStream_t stream)
{
    const SceneView& sv = scene.view;
    Batch proto;
    const uint64_t nPixelsTotal = call_pixels(st, nullptr, proto);
    if (nPixelsTotal == 0) return RTGPU_OK;
    const uint32_t G = (uint32_t)st.geometry_aa, M = (uint32_t)st.material_aa;
    const uint64_t pathsPerPixel = (uint64_t)G * M;
    // the largest batch whose workspace fits the array
    auto bytes_for = [&](uint64_t pixels) { Pool probe{}; return pool_carve(probe, nullptr, pool_dims(sv, pixels * pathsPerPixel, pixels * G)); };
    if (pathsPerPixel > (1ull << kQueuePathBits) || bytes_for(1) > RTGPU_STATIC_WORKSPACE_BYTES)
        return set_error(RTGPU_ERR_LIMIT, "geometry_aa * material_aa = " + std::to_string(pathsPerPixel) + " paths per pixel do not fit the " +
                                              std::to_string(RTGPU_STATIC_WORKSPACE_BYTES >> 20) + " MiB static workspace");
    uint64_t lo = 1, hi = std::min<uint64_t>(nPixelsTotal, (1ull << kQueuePathBits) / pathsPerPixel);
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) / 2;
        if (bytes_for(mid) <= RTGPU_STATIC_WORKSPACE_BYTES) lo = mid;
        else hi = mid - 1;
    }
    const uint64_t pixelsPerBatch = lo;
    void* base = nullptr;
    hipError_t e = static_workspace(scene.device, &base);
    if (e != hipSuccess) return set_error(RTGPU_ERR_HIP, std::string("hipGetSymbolAddress(static workspace): ") + hipGetErrorString(e));
    LaneRun ln;
    ln.stream = stream;
    pool_carve(ln.pool, (char*)base, pool_dims(sv, pixelsPerBatch * pathsPerPixel, pixelsPerBatch * G));
    std::string err;
    for (uint64_t start = 0; start < nPixelsTotal; start += pixelsPerBatch) {
        Batch batch = proto;
        batch.pixelStart = (uint32_t)start;
        batch.nPixels = (uint32_t)std::min<uint64_t>(pixelsPerBatch, nPixelsTotal - start);
        e = run_batch(sv, ln, cam, st, batch, dRgb, err);   // batches follow each other on the one stream, so they can share the workspace
        if (e != hipSuccess) return set_error(RTGPU_ERR_HIP, err.empty() ? hipGetErrorString(e) : err);
    }
    return RTGPU_OK;
}

} // namespace rt
