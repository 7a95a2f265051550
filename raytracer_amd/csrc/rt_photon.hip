// rt_photon.hip -- the photon map on the GPU (gfx950): photon tracing, the search structure and the gather.
//
// Restates PhotonMapBuilder::buildPhotonMap (photonmapping/PhotonMapBuilder.cpp:15-41): PhotonTracer::tracePhotons
// (photonmapping/PhotonTracer.cpp:8-172) with IMaterial::interactPhoton of every closure, and the query
// KDTree::getElementsNearestTo as DiffuseMaterial uses it (photonmapping/KDTree.h:90-164,
// material/DiffuseMaterial.cpp:62-75).
//
//   trace  : one lane per emitted ray, up to 12 vertices; every lane of a wave walks its photon through the
//            scene with the wave-cooperative closest-hit query, so lanes whose photons left the scene still
//            help the others.  A vertex that stores a photon writes slot [ray][vertex]; the host keeps the
//            filled slots in (light, ray, vertex) order, so the map does not depend on scheduling.
//   tree   : the reference sorts each slice along the next axis (y, z, x, ...) and takes the element at count / 2
//            (a slice of two keeps its first element, with the parent's axis, and hangs the second to the right:
//            photonmapping/KDTreeBuilder.h:178-231).  Here a median tree of the same kind -- not node for node the
//            reference's -- is kept without pointers: photons are permuted so that every range [lo, hi) has its axis
//            median at lo + (hi - lo) / 2 (std::nth_element at every level, two-element ranges included; host threads
//            for the upper levels), and far sides are pruned by the distance to the split plane where the reference
//            uses the distance to the node's box.
//   gather : one lane per query point, depth-first near side, node, far side (the reference's visiting pattern) with a stack
//            of far ranges in private memory and a bounded max-heap of the k best; ranges whose cone of photon
//            normals cannot pass the query's filter are skipped.  A query that needs more than a budget of nodes is
//            answered again by a whole wave (gather_photons_wave).  The k nearest photons that pass the filter are
//            the same set whatever the tree and the order of the search; which photon wins a distance tie at the
//            k-th place, and the last bits of an energy sum whose terms tie in distance, depend on the tree: parity
//            with the reference there is unpinned (no fixture of the reference covers photon-mapped output).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "rt_scene.h"
#include "rt_photon_inl.h"
#include "rt_wave_inl.h"

namespace rt {

namespace {

constexpr uint32_t kPhotonPixel = 0x40000000u;  // RNG stream of photons: (seed, kPhotonPixel | light, ray, vertex, dim)
constexpr uint32_t kVertexEmit = 254;
constexpr int kMaxPhotonDepth = 12;              // PhotonDriver::trace, photonmapping/PhotonTracer.cpp:17
constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;

struct EmitBatch {       // one task of PhotonTracer::tracePhotons
    uint32_t light;      // index into the point or area lights
    uint32_t area;       // 0: point light, 1: area light
    uint32_t count;      // rays of the task (the stratification level of an area light)
    uint32_t firstRay;   // ordinal of the task's first ray among the light's rays (RNG key)
};

// Rand::sampleStdNormalDist stand-in: Box-Muller on two uniform draws
__device__ __forceinline__ float std_normal(float u1, float u2) { return (float)(sqrt(-2.0 * log(1.0 - (double)u1)) * cos(2.0 * RT_PI * (double)u2)); }

// AreaLight::generateStratifiedJitteredRandomPoint, light/AreaLight.cpp:26-41
__device__ __forceinline__ F3 area_point(const DevAreaLight& L, int level, int i, float r0, float r1)
{
    const int n = (int)sqrt((double)level);
    const F3 a = f3(L.a[0], L.a[1], L.a[2]), b = f3(L.b[0], L.b[1], L.b[2]), c = f3(L.c[0], L.c[1], L.c[2]);
    const F3 v1 = (c - a) / (float)n;
    const F3 v2 = (b - a) / (float)n;
    const float u = 1.0f - sqrtf(r0);
    const float v = (1.0f - u) * r1;
    const int y = i / n;
    const int x = i % n;
    return (a + ((u + x) * v1)) + ((v + y) * v2);
}

// IMaterial::interactPhoton of the closures (Mix: MixMaterial.cpp:47-57, Add: AddMaterial.cpp:19-23, Diffuse:
// DiffuseMaterial.cpp:206-226, Glossy: GlossyMaterial.cpp:209-224, Glass: GlassMaterial.cpp:180-252, Transparent:
// TransparentMaterial.cpp:18-21, Emissive: EmissiveMaterial.cpp:24-27)
__device__ void interact_photon(const SceneView& sv, uint32_t mi, uint32_t inst, uint32_t slot, float beta, float gamma, F3 normal, F3 rd, float t,
                                RGBf incoming, uint32_t seed, uint32_t pixel, uint32_t ray, uint32_t vertex, F3& dir, RGBf& energy, float& diffuseness)
{
    for (uint32_t level = 0; level < 4; ++level) {
        const DevMaterial& mm = sv.materials[mi];
        if (mat_is_mix(mm)) mi = rand_unit(seed, pixel, ray, vertex, DIM_MIX0 + level) > mat_mix_factor(mm, normal, rd) ? mm.first : mm.second;
        else if (mm.type == RTGPU_MAT_ADD) mi = mm.first;
        else break;
    }
    const DevMaterial& m = sv.materials[mi];
    F3 shadeN = normal;
    if (m.normalMap != 0u && (m.type == RTGPU_MAT_DIFFUSE || m.type == RTGPU_MAT_GLOSSY)) shadeN = sample_normal_map(sv, m.normalMap - 1u, inst, slot, beta, gamma, normal);
    if (m.type == RTGPU_MAT_DIFFUSE) {
        RGBf color = rgb(m.color[0], m.color[1], m.color[2]);
        if (m.texture != 0u) {
            float tu, tv;
            hit_texcoord(sv, slot, beta, gamma, tu, tv);
            color = emission_lookup(sv, m.texture - 1u, tu, tv);   // the abs(fmod) wrap of DiffuseMaterial.cpp:211-212
        }
        const Basis3 basis = basis_from(shadeN);
        const F3 l = cosine_hemisphere(rand_unit(seed, pixel, ray, vertex, DIM_SQUARE_X), rand_unit(seed, pixel, ray, vertex, DIM_SQUARE_Y));
        dir = ((basis.u * l.x) + (basis.v * l.y)) + (basis.w * l.z);
        energy = rgb_mul(color, incoming);
        diffuseness = 1.0f;
    } else if (m.type == RTGPU_MAT_GLOSSY) {
        const F3 micro = normalized3(vndf_sample(shadeN, -rd, m.roughness, rand_unit(seed, pixel, ray, vertex, DIM_SQUARE_X), rand_unit(seed, pixel, ray, vertex, DIM_SQUARE_Y)));
        dir = rd + ((float)(2.0 * (double)dot3(micro, -rd)) * micro);
        energy = incoming;
        diffuseness = m.roughness;
    } else if (m.type == RTGPU_MAT_GLASS) {
        F3 n = normal;
        const bool attenuate = dot3(n, rd) < 0;
        const F3 objToIncoming = -rd;
        const F3 reflDir = (-objToIncoming) + ((float)(2.0 * (double)dot3(n, objToIncoming)) * n);
        double nDotWo = dot3(n, objToIncoming);
        double incomingAngle = nDotWo;
        const bool internal = nDotWo < 0;
        double curRelIor = m.intensity / 1.0;
        if (internal) curRelIor = 1.0 / curRelIor;
        const double cosTransmissionSqr = 1.0 - ((1.0 - (incomingAngle * incomingAngle)) / (curRelIor * curRelIor));
        if (cosTransmissionSqr < 0) {
            dir = reflDir;
        } else {
            if (internal) { nDotWo = -nDotWo; incomingAngle = -incomingAngle; n = -n; }
            const double cosRefractAngle = sqrt(cosTransmissionSqr);
            const F3 transDir = (rd / (float)curRelIor) - ((float)(cosRefractAngle - (incomingAngle / curRelIor)) * n);
            const double rs = ((curRelIor * nDotWo) - cosRefractAngle) / ((curRelIor * nDotWo) + cosRefractAngle);
            const double rp = (nDotWo - (curRelIor * cosRefractAngle)) / (nDotWo + (curRelIor * cosRefractAngle));
            const double kr = ((rs * rs) + (rp * rp)) / 2.0;
            const double reflectionWeight = kr / (kr + (1.0 - kr));
            dir = (double)rand_unit(seed, pixel, ray, vertex, DIM_GLASS) < reflectionWeight ? reflDir : transDir;
        }
        energy = incoming;
        if (attenuate) energy = rgb_mul(incoming, rgb_pow_d(rgb(m.color[0], m.color[1], m.color[2]), (double)m.roughness * (double)t));
        diffuseness = 0.0f;
    } else if (m.type == RTGPU_MAT_TRANSPARENT) {
        dir = rd; energy = incoming; diffuseness = 0.0f;
    } else {
        dir = -rd; energy = rgb(0, 0, 0); diffuseness = 0.0f;
    }
}

// One lane per ray of the launch; batchStart[b] is the launch-local index of batch b's first ray.
template <int MODE>
__global__ __launch_bounds__(64) void photon_trace_kernel(SceneView sv, const EmitBatch* __restrict__ batches, const uint32_t* __restrict__ batchStart,
                                                          uint32_t nBatches, uint32_t nRays, int mode, double perPoint, float perArea, uint32_t seed,
                                                          rtgpu_photon* __restrict__ slots)
{
    __shared__ typename TraceLds<MODE>::type lds;
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x * 64u + lane;
    bool alive = r < nRays;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1);
    RGBf energy = rgb(0, 0, 0);
    uint32_t pixel = 0, ray = 0;
    if (alive) {
        // the batch of this ray: last batch whose start is <= r
        uint32_t lo = 0, hi = nBatches;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (batchStart[mid] <= r) lo = mid; else hi = mid; }
        const EmitBatch b = batches[lo];
        const uint32_t k = r - batchStart[lo];
        ray = b.firstRay + k;
        if (!b.area) { // PointLightPhotonTracingTask::execute, photonmapping/PhotonTracer.cpp:84-96
            const DevPointLight& L = sv.pointLights[b.light];
            pixel = kPhotonPixel | b.light;
            energy = rgb_scale_d(rgb(L.color[0], L.color[1], L.color[2]), L.intensity / perPoint);
            const float x = std_normal(rand_unit(seed, pixel, ray, kVertexEmit, 0), rand_unit(seed, pixel, ray, kVertexEmit, 1));
            const float y = std_normal(rand_unit(seed, pixel, ray, kVertexEmit, 2), rand_unit(seed, pixel, ray, kVertexEmit, 3));
            const float z = std_normal(rand_unit(seed, pixel, ray, kVertexEmit, 4), rand_unit(seed, pixel, ray, kVertexEmit, 5));
            const float len = sqrtf(x * x + y * y + z * z);   // sampleUniformSphere(1.0), math/Sampler.h:42-49
            o = f3(L.pos[0], L.pos[1], L.pos[2]);
            d = f3(1.0f * x / len, 1.0f * y / len, 1.0f * z / len);
        } else {       // AreaLightPhotonTracingTask::execute, :121-137
            const DevAreaLight& L = sv.areaLights[b.light];
            pixel = kPhotonPixel | (sv.nPointLights + b.light);
            energy = rgb_div_f(rgb_scale_d(rgb(L.color[0], L.color[1], L.color[2]), L.intensity), perArea);
            const F3 a = f3(L.a[0], L.a[1], L.a[2]), bb = f3(L.b[0], L.b[1], L.b[2]), c = f3(L.c[0], L.c[1], L.c[2]);
            const Basis3 basis = basis_from(cross3(bb - a, c - a));
            const F3 pos = area_point(L, (int)b.count, (int)k, rand_unit(seed, pixel, ray, kVertexEmit, DIM_AREA_U), rand_unit(seed, pixel, ray, kVertexEmit, DIM_AREA_V));
            const F3 l = cosine_hemisphere(rand_unit(seed, pixel, ray, kVertexEmit, DIM_SQUARE_X), rand_unit(seed, pixel, ray, kVertexEmit, DIM_SQUARE_Y));
            d = normalized3(((basis.u * l.x) + (basis.v * l.y)) + (basis.w * l.z));
            o = pos + (d * 0.0001f);
        }
    }
    bool hasPassedSpecular = false;
    for (int v = 0; v < kMaxPhotonDepth; ++v) {   // PhotonDriver::trace, :12-58
        if (__ballot(alive) == 0ull) break;
        const TraceResult hr = wave_closest<MODE>(sv, alive, o, d, lds, lane);
        if (!alive) continue;
        if (hr.inst == kInvalid) { alive = false; continue; }
        const DevInstance& in = sv.inst[hr.inst];
        uint32_t mi = in.material;
        const ShadeRec rec = load_shade_rec(sv, hr.slot);
        if (hr.slot != kInvalid && rec.material() != kInvalid) mi = rec.material();
        const F3 hitpoint = o + (d * hr.t);
        const F3 normal = hit_normal(in, o, d, hr.t, hr.slot, hr.beta, hr.gamma, rec);
        F3 newDir;
        RGBf newEnergy;
        float diffuseness;
        interact_photon(sv, mi, hr.inst, hr.slot, hr.beta, hr.gamma, normal, d, hr.t, energy, seed, pixel, ray, (uint32_t)v, newDir, newEnergy, diffuseness);
        newDir = normalized3(newDir);
        const bool isDiffuseTransport = diffuseness >= 0.2;
        const bool isCaustic = isDiffuseTransport && hasPassedSpecular;
        if ((mode == RTGPU_PM_CAUSTICS && isCaustic) || (mode == RTGPU_PM_FULL && isDiffuseTransport)) {
            rtgpu_photon ph;
            ph.pos[0] = hitpoint.x; ph.pos[1] = hitpoint.y; ph.pos[2] = hitpoint.z;
            ph.normal[0] = normal.x; ph.normal[1] = normal.y; ph.normal[2] = normal.z;
            ph.energy[0] = energy.r; ph.energy[1] = energy.g; ph.energy[2] = energy.b;
            ph.is_caustic = isCaustic ? 1u : 0u;
            slots[(size_t)r * kMaxPhotonDepth + v] = ph;
        }
        hasPassedSpecular = hasPassedSpecular || !isDiffuseTransport;
        o = hitpoint + (newDir * 0.00005f);
        d = newDir;
        energy = newEnergy;
    }
}

__global__ void fill_empty_kernel(rtgpu_photon* slots, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i].is_caustic = kEmptySlot;
}

} // namespace

namespace {

__device__ __forceinline__ void store_gather(rtgpu_photon_gather* __restrict__ out, uint32_t i, uint32_t count, float maxDist, RGBf sum)
{
    rtgpu_photon_gather g;
    g.count = count; g.max_dist = maxDist; g.energy[0] = sum.r; g.energy[1] = sum.g; g.energy[2] = sum.b;
    out[i] = g;
}

// First attempt, one lane per query, with a bounded number of tree nodes; queries that run out are listed in
// tail[1..], their number in tail[0] (one atomic per wave), and gather_tail_kernel answers each with a whole wave.
__global__ __launch_bounds__(64) void gather_kernel(SceneView sv, const float* __restrict__ points, const float* __restrict__ dirs, uint32_t n,
                                                   rtgpu_photon_gather* __restrict__ out, uint32_t* __restrict__ tail)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    bool unfinished = false;
    if (i < n) {
        uint32_t count;
        float maxDist;
        RGBf sum;
        if (gather_photons(sv, f3(points[3 * i], points[3 * i + 1], points[3 * i + 2]), f3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), sv.gatherBudget, count, maxDist, sum))
            store_gather(out, i, count, maxDist, sum);
        else
            unfinished = true;
    }
    const unsigned long long m = __ballot(unfinished);
    if (m == 0ull) return;
    const int lane = (int)threadIdx.x, leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(tail, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (unfinished) tail[1u + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

__global__ __launch_bounds__(64) void gather_tail_kernel(SceneView sv, const float* __restrict__ points, const float* __restrict__ dirs,
                                                        rtgpu_photon_gather* __restrict__ out, const uint32_t* __restrict__ tail)
{
    __shared__ GatherWaveLds lds;
    const uint32_t n = tail[0];
    for (uint32_t qi = blockIdx.x; qi < n; qi += gridDim.x) {
        const uint32_t i = tail[1u + qi];
        uint32_t count;
        float maxDist;
        RGBf sum;
        gather_photons_wave(sv, f3(points[3 * i], points[3 * i + 1], points[3 * i + 2]), f3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), lds, (int)threadIdx.x, count, maxDist, sum);
        if (threadIdx.x == 0) store_gather(out, i, count, maxDist, sum);
    }
}

// Position and index of a photon, 16 bytes: what the ordering moves around (the photons themselves stay where they are).
struct KdKey { float pos[3]; uint32_t index; };

// Permutes keys[lo, hi) so that every range has its median along axis (depth + 1) % 3 in the middle.
void kd_order(KdKey* keys, size_t lo, size_t hi, uint32_t depth, int spawnLevels)
{
    while (hi - lo > 1) {
        const size_t mid = lo + (hi - lo) / 2;
        const int axis = (int)((depth + 1u) % 3u);
        std::nth_element(keys + lo, keys + mid, keys + hi, [axis](const KdKey& a, const KdKey& b) { return a.pos[axis] < b.pos[axis]; });
        if (spawnLevels > 0 && hi - lo > 100000) {
            std::thread left([=] { kd_order(keys, lo, mid, depth + 1, spawnLevels - 1); });
            kd_order(keys, mid + 1, hi, depth + 1, spawnLevels - 1);
            left.join();
            return;
        }
        kd_order(keys, lo, mid, depth + 1, 0);
        lo = mid + 1;
        depth += 1;
    }
}

// f(begin, end) over [0, n) on `threads` host threads.
template <typename F>
void host_parallel(size_t n, int threads, const F& f)
{
    if (threads <= 1 || n < 65536) { f((size_t)0, n); return; }
    std::vector<std::thread> pool;
    const size_t chunk = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        const size_t b = (size_t)t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        pool.emplace_back([&f, b, e] { f(b, e); });
    }
    for (std::thread& t : pool) t.join();
}

// Bounding cone of the surface normals of every range (stored at the range's middle element): axis and the
// value the gather compares dot(dir, axis) / |dir| with -- below it no normal of the range can pass the filter
// dir . normal >= 0.  Built bottom-up and conservatively: half angle = the largest of the node's own deviation and
// each child cone's axis deviation plus its half angle, a little inflated; ranges whose cone reaches 90 degrees
// are never culled (limit -2).
struct Cone { double sx, sy, sz; float ax, ay, az; double alpha; };
Cone kd_cones(const float4* normals, float4* out, size_t lo, size_t hi, int spawnLevels)
{
    const size_t mid = lo + (hi - lo) / 2;
    const float pn[3] = {normals[mid].x, normals[mid].y, normals[mid].z};
    Cone l{}, r{};
    const bool hasL = mid > lo, hasR = hi > mid + 1;
    if (spawnLevels > 0 && hi - lo > 100000) {
        std::thread t([&] { l = kd_cones(normals, out, lo, mid, spawnLevels - 1); });
        if (hasR) r = kd_cones(normals, out, mid + 1, hi, spawnLevels - 1);
        t.join();
    } else {
        if (hasL) l = kd_cones(normals, out, lo, mid, 0);
        if (hasR) r = kd_cones(normals, out, mid + 1, hi, 0);
    }
    Cone c{};
    c.sx = pn[0] + (hasL ? l.sx : 0.0) + (hasR ? r.sx : 0.0);
    c.sy = pn[1] + (hasL ? l.sy : 0.0) + (hasR ? r.sy : 0.0);
    c.sz = pn[2] + (hasL ? l.sz : 0.0) + (hasR ? r.sz : 0.0);
    const double len = std::sqrt(c.sx * c.sx + c.sy * c.sy + c.sz * c.sz);
    const double kPi = 3.14159265358979323;
    c.alpha = kPi;
    if (len > 1e-6 && std::isfinite(len)) {
        const double ax = c.sx / len, ay = c.sy / len, az = c.sz / len;
        c.ax = (float)ax; c.ay = (float)ay; c.az = (float)az;
        auto angle = [&](double x, double y, double z) {
            const double n = std::sqrt(x * x + y * y + z * z);
            if (!(n > 0) || !std::isfinite(n)) return kPi;   // a degenerate normal: its dot product with anything is 0 or NaN, keep the range
            return std::acos(std::max(-1.0, std::min(1.0, (x * c.ax + y * c.ay + z * c.az) / n)));
        };
        double a = angle(pn[0], pn[1], pn[2]);
        if (hasL) a = std::max(a, angle(l.ax, l.ay, l.az) + l.alpha);
        if (hasR) a = std::max(a, angle(r.ax, r.ay, r.az) + r.alpha);
        c.alpha = std::min(kPi, a + 1e-3);
    }
    const float limit = c.alpha < 0.5 * kPi - 1e-3 ? (float)(-std::sin(c.alpha) - 1e-4) : -2.0f;
    out[2 * mid + 1] = make_float4(c.ax, c.ay, c.az, limit);
    return c;
}

// RTGPU_PM_STATS=1: phase times of the photon map build on stderr (developer aid)
struct PhaseClock {
    const bool on = std::getenv("RTGPU_PM_STATS") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "photon map: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

struct Dev {
    void* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
};

#define PH_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return set_error(e_ == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

} // namespace

// Uploads the search structure for scene.photons and switches the scene to (mode, depth).
int install_photon_map(rtgpu_scene& scene, int mode, int depth)
{
    SceneView& v = scene.view;
    PH_HIP(hipSetDevice(scene.device));
    // copies of the scene on other devices (multi-GPU renders) hold the old map: they are made again when next needed
    {
        std::lock_guard<std::mutex> multiLock(scene.multiMutex);
        destroy_multi(scene.multi);
        scene.multi = nullptr;
    }
    for (const void* p : {(const void*)v.kdPos, (const void*)v.kdNormal, (const void*)v.kdEnergy})
        if (p) (void)hipFree(const_cast<void*>(p));
    scene.set_array(&SceneView::kdPos, (const float4*)nullptr, 0);
    scene.set_array(&SceneView::kdNormal, (const float4*)nullptr, 0);
    scene.set_array(&SceneView::kdEnergy, (const float4*)nullptr, 0);
    v.nPhotons = 0;
    v.pmMode = RTGPU_PM_NONE;
    v.pmDepth = 0;
    const std::vector<rtgpu_photon>& ph = scene.photons;
    if (mode == RTGPU_PM_NONE) return RTGPU_OK;
    if (ph.size() >= (1ull << 31)) return set_error(RTGPU_ERR_LIMIT, "more than 2^31 photons");
    const size_t n = ph.size();
    if (n) {
        PhaseClock clock;
        const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::unique_ptr<KdKey[]> keys(new KdKey[n]);
        host_parallel(n, threads, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) keys[i] = KdKey{{ph[i].pos[0], ph[i].pos[1], ph[i].pos[2]}, (uint32_t)i};
        });
        kd_order(keys.get(), 0, n, 0, 4);
        clock.lap("median ordering");
        // pos: position and normal cone of the range rooted at the element.  Not value-initialised: the pages are first
        // touched by the threads that fill them.
        std::unique_ptr<float4[]> posMem(new float4[2 * n]), nrmMem(new float4[n]), enMem(new float4[n]);
        float4 *pos = posMem.get(), *nrm = nrmMem.get(), *en = enMem.get();
        host_parallel(n, threads, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const rtgpu_photon& p = ph[keys[i].index];
                pos[2 * i] = make_float4(p.pos[0], p.pos[1], p.pos[2], 0.0f);
                nrm[i] = make_float4(p.normal[0], p.normal[1], p.normal[2], 0.0f);
                en[i] = make_float4(p.energy[0], p.energy[1], p.energy[2], 0.0f);
            }
        });
        clock.lap("search-order arrays");
        kd_cones(nrm, pos, 0, n, 4);
        clock.lap("normal cones");
        void *dp = nullptr, *dn = nullptr, *de = nullptr;
        PH_HIP(hipMalloc(&dp, 2 * n * sizeof(float4)));
        scene.set_array(&SceneView::kdPos, (const float4*)dp, 2 * n * sizeof(float4));
        PH_HIP(hipMalloc(&dn, n * sizeof(float4)));
        scene.set_array(&SceneView::kdNormal, (const float4*)dn, n * sizeof(float4));
        PH_HIP(hipMalloc(&de, n * sizeof(float4)));
        scene.set_array(&SceneView::kdEnergy, (const float4*)de, n * sizeof(float4));
        PH_HIP(hipMemcpy(dp, pos, 2 * n * sizeof(float4), hipMemcpyHostToDevice));
        PH_HIP(hipMemcpy(dn, nrm, n * sizeof(float4), hipMemcpyHostToDevice));
        PH_HIP(hipMemcpy(de, en, n * sizeof(float4), hipMemcpyHostToDevice));
        clock.lap("upload");
    }
    v.nPhotons = (uint32_t)n;
    v.pmMode = mode;
    v.pmDepth = depth;
    return RTGPU_OK;
}

int trace_photons(rtgpu_scene& scene, int mode, uint64_t perPoint, uint64_t perArea, uint32_t seed)
{
    const SceneView& sv = scene.view;
    scene.photons.clear();
    if (mode == RTGPU_PM_NONE) return RTGPU_OK;
    PH_HIP(hipSetDevice(scene.device));
    // the tasks of PhotonTracer::tracePhotons (:139-172): createBatches (utility/Batcher.h) cuts photonsPerPointLight
    // rays -- for area lights too -- into batches of 10 000 and issues a last partial batch twice
    std::vector<EmitBatch> batches;
    const uint64_t batchSize = 10000;
    if (perPoint >= (1ull << 31)) return set_error(RTGPU_ERR_LIMIT, "more than 2^31 rays per light");
    auto add = [&](uint32_t light, uint32_t area) {
        uint64_t ray = 0;
        for (uint64_t i = 0; i < perPoint; i += batchSize) {
            const uint64_t e = std::min(i + batchSize, perPoint);
            batches.push_back(EmitBatch{light, area, (uint32_t)(e - i), (uint32_t)ray});
            ray += e - i;
        }
        const uint64_t rem = perPoint % batchSize;
        if (rem > 0) batches.push_back(EmitBatch{light, area, (uint32_t)rem, (uint32_t)ray});
    };
    for (uint32_t l = 0; l < sv.nPointLights; ++l) add(l, 0u);
    for (uint32_t l = 0; l < sv.nAreaLights; ++l) add(l, 1u);
    if (batches.empty()) return RTGPU_OK;
    // launches of at most 100 batches (1 M rays, 480 MB of slots)
    const size_t perLaunch = 100;
    Dev dSlots, dBatches, dStarts;
    const size_t slotCap = perLaunch * batchSize * kMaxPhotonDepth;
    PH_HIP(hipMalloc(&dSlots.p, slotCap * sizeof(rtgpu_photon)));
    PH_HIP(hipMalloc(&dBatches.p, perLaunch * sizeof(EmitBatch)));
    PH_HIP(hipMalloc(&dStarts.p, perLaunch * sizeof(uint32_t)));
    std::vector<rtgpu_photon> host;
    PhaseClock clock;
    for (size_t b0 = 0; b0 < batches.size(); b0 += perLaunch) {
        const size_t nb = std::min(perLaunch, batches.size() - b0);
        std::vector<uint32_t> starts(nb);
        uint32_t nRays = 0;
        for (size_t b = 0; b < nb; ++b) { starts[b] = nRays; nRays += batches[b0 + b].count; }
        const size_t nSlots = (size_t)nRays * kMaxPhotonDepth;
        PH_HIP(hipMemcpy(dBatches.p, batches.data() + b0, nb * sizeof(EmitBatch), hipMemcpyHostToDevice));
        PH_HIP(hipMemcpy(dStarts.p, starts.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, nullptr, (rtgpu_photon*)dSlots.p, nSlots);
        const dim3 grid((nRays + 63) / 64), block(64);
        const int traceMode = trace_mode(sv);
        if (traceMode == kTraceSingle)
            hipLaunchKernelGGL((photon_trace_kernel<kTraceSingle>), grid, block, 0, nullptr, sv, (const EmitBatch*)dBatches.p,
                               (const uint32_t*)dStarts.p, (uint32_t)nb, nRays, mode, (double)perPoint, (float)perArea, seed, (rtgpu_photon*)dSlots.p);
        else if (traceMode == kTraceInstances)
            hipLaunchKernelGGL((photon_trace_kernel<kTraceInstances>), grid, block, 0, nullptr, sv, (const EmitBatch*)dBatches.p,
                               (const uint32_t*)dStarts.p, (uint32_t)nb, nRays, mode, (double)perPoint, (float)perArea, seed, (rtgpu_photon*)dSlots.p);
        else
            hipLaunchKernelGGL((photon_trace_kernel<kTraceLanes>), grid, block, 0, nullptr, sv, (const EmitBatch*)dBatches.p,
                               (const uint32_t*)dStarts.p, (uint32_t)nb, nRays, mode, (double)perPoint, (float)perArea, seed, (rtgpu_photon*)dSlots.p);
        PH_HIP(hipGetLastError());
        host.resize(nSlots);
        PH_HIP(hipMemcpy(host.data(), dSlots.p, nSlots * sizeof(rtgpu_photon), hipMemcpyDeviceToHost));
        clock.lap("trace + slots to host");
        {   // the filled slots, in slot order, appended to the scene's photons
            const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
            const size_t chunk = (nSlots + (size_t)threads - 1) / (size_t)threads;
            std::vector<size_t> count((size_t)threads + 1, 0);
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; ++t)
                pool.emplace_back([&, t] {
                    const size_t b = std::min(nSlots, (size_t)t * chunk), e = std::min(nSlots, b + chunk);
                    size_t c = 0;
                    for (size_t i = b; i < e; ++i) c += host[i].is_caustic != kEmptySlot;
                    count[(size_t)t + 1] = c;
                });
            for (std::thread& t : pool) t.join();
            pool.clear();
            for (int t = 0; t < threads; ++t) count[(size_t)t + 1] += count[(size_t)t];
            const size_t base = scene.photons.size();
            if (b0 == 0 && batches.size() > perLaunch) {
                // the first launch tells how full the slots get: room for the launches to come (one growth instead of several)
                const double perBatch = (double)count[(size_t)threads] / (double)nb;
                scene.photons.reserve((size_t)(perBatch * 1.1 * (double)batches.size()) + 1024);
            }
            scene.photons.resize(base + count[(size_t)threads]);
            for (int t = 0; t < threads; ++t)
                pool.emplace_back([&, t] {
                    const size_t b = std::min(nSlots, (size_t)t * chunk), e = std::min(nSlots, b + chunk);
                    rtgpu_photon* out = scene.photons.data() + base + count[(size_t)t];
                    for (size_t i = b; i < e; ++i)
                        if (host[i].is_caustic != kEmptySlot) *out++ = host[i];
                });
            for (std::thread& t : pool) t.join();
        }
        clock.lap("keep filled slots");
    }
    return RTGPU_OK;
}

int gather_photons_host(const rtgpu_scene& scene, const float* points, const float* dirs, uint64_t n, rtgpu_photon_gather* out)
{
    PH_HIP(hipSetDevice(scene.device));
    const uint64_t cap = std::min<uint64_t>(n, 1ull << 22);
    Dev dP, dD, dO, dTail;
    PH_HIP(hipMalloc(&dTail.p, (cap + 1) * sizeof(uint32_t)));
    PH_HIP(hipMalloc(&dP.p, cap * 12));
    PH_HIP(hipMalloc(&dD.p, cap * 12));
    PH_HIP(hipMalloc(&dO.p, cap * sizeof(rtgpu_photon_gather)));
    for (uint64_t b = 0; b < n; b += cap) {
        const uint64_t m = std::min(cap, n - b);
        PH_HIP(hipMemcpy(dP.p, points + 3 * b, m * 12, hipMemcpyHostToDevice));
        PH_HIP(hipMemcpy(dD.p, dirs + 3 * b, m * 12, hipMemcpyHostToDevice));
        PH_HIP(hipMemsetAsync(dTail.p, 0, sizeof(uint32_t), nullptr));
        const unsigned blocks = (unsigned)((m + 63) / 64);
        hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(64), 0, nullptr, scene.view, (const float*)dP.p, (const float*)dD.p, (uint32_t)m,
                           (rtgpu_photon_gather*)dO.p, (uint32_t*)dTail.p);
        hipLaunchKernelGGL(gather_tail_kernel, dim3((unsigned)std::min<uint64_t>(m, 16384)), dim3(64), 0, nullptr, scene.view, (const float*)dP.p, (const float*)dD.p,
                           (rtgpu_photon_gather*)dO.p, (const uint32_t*)dTail.p);
        PH_HIP(hipGetLastError());
        PH_HIP(hipMemcpy(out + b, dO.p, m * sizeof(rtgpu_photon_gather), hipMemcpyDeviceToHost));
    }
    return RTGPU_OK;
}

} // namespace rt

extern "C" {

int rtgpu_scene_build_photon_map(rtgpu_scene* scene, int32_t mode, int32_t depth, uint64_t rays_per_point_light, uint64_t rays_per_area_light,
                                 uint32_t seed, uint64_t* n_photons)
{
    if (!scene) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (mode < RTGPU_PM_NONE || mode > RTGPU_PM_FULL) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "unknown photon map mode");
    if (depth < 0 || depth > 15) return rt::set_error(RTGPU_ERR_LIMIT, "photon map depth must be 0..15");
    if (mode != RTGPU_PM_NONE && (rays_per_point_light == 0 || rays_per_area_light == 0))
        return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "Invalid PM raycount settings!");   // exe/Raytracer.cpp:237-241
    return rt::guarded([&]() -> int {
        if (int rc = rt::trace_photons(*scene, mode, rays_per_point_light, rays_per_area_light, seed)) return rc;
        if (n_photons) *n_photons = scene->photons.size();
        return rt::install_photon_map(*scene, mode, depth);
    });
}

int rtgpu_scene_get_photons(const rtgpu_scene* scene, rtgpu_photon* out, uint64_t capacity, uint64_t* n)
{
    if (!scene) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    const uint64_t m = std::min<uint64_t>(capacity, scene->photons.size());
    if (m && !out) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "null photon buffer");
    if (m) std::memcpy(out, scene->photons.data(), m * sizeof(rtgpu_photon));
    if (n) *n = scene->photons.size();
    return RTGPU_OK;
}

int rtgpu_scene_set_photons(rtgpu_scene* scene, const rtgpu_photon* photons, uint64_t n, int32_t mode, int32_t depth)
{
    if (!scene) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (mode < RTGPU_PM_NONE || mode > RTGPU_PM_FULL) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "unknown photon map mode");
    if (depth < 0 || depth > 15) return rt::set_error(RTGPU_ERR_LIMIT, "photon map depth must be 0..15");
    if (n && !photons) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "null photon buffer");
    return rt::guarded([&]() -> int {
        scene->photons.assign(photons, photons + n);
        return rt::install_photon_map(*scene, mode, depth);
    });
}

int rtgpu_scene_gather_photons(const rtgpu_scene* scene, const float* points, const float* directions, uint64_t n, rtgpu_photon_gather* out)
{
    if (!scene) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (n == 0) return RTGPU_OK;
    if (!points || !directions || !out) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "null query/result buffer");
    return rt::guarded([&]() -> int { return rt::gather_photons_host(*scene, points, directions, n, out); });
}

} // extern "C"
