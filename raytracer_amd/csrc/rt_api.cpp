// rt_api.cpp -- implementation of the C ABI declared in include/rtgpu.h (host side, HIP runtime).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/rtgpu.h"
#include "bvh_build.h"
#include "rt_device.h"
#include "rt_scene.h"
#include "top_tree_ref.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg)
{
    g_error = msg;
    return code;
}

#define RT_HIP(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(e_ == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP,                 \
                        std::string(#call) + ": " + hipGetErrorString(e_));                                  \
    } while (0)

template <typename F>
void parallel_for(size_t n, int threads, const F& f)
{
    if (threads <= 1 || n < 4096) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t b = (size_t)t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        th.emplace_back([&f, b, e] { f(b, e); });
    }
    for (auto& t : th) t.join();
}

// Transformation::transform(Point), math/Transformation.h:71-76
void xf_point(const double* M, const float p[3], float out[3])
{
    double v[4];
    for (int i = 0; i < 4; ++i) v[i] = ((M[i * 4 + 0] * p[0] + M[i * 4 + 1] * p[1]) + M[i * 4 + 2] * p[2]) + M[i * 4 + 3] * 1.0;
    for (int i = 0; i < 3; ++i) out[i] = (float)(v[i] / v[3]);
}
// Transformation::transform(Vector3), math/Transformation.h:83-87
void xf_vector(const double* M, const float p[3], float out[3])
{
    for (int i = 0; i < 3; ++i) out[i] = (float)(((M[i * 4 + 0] * p[0] + M[i * 4 + 1] * p[1]) + M[i * 4 + 2] * p[2]) + M[i * 4 + 3] * 0.0);
}

bool affine(const double* m) { return m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0; }

// Copies a host array to the scene's device and hangs it on the SceneView member (registered with its size, so that the
// scene can be copied to another device later).
template <typename M, typename T>
int upload(rtgpu_scene& sc, const M* rt::SceneView::*member, const std::vector<T>& v, uint64_t& bytes)
{
    if (v.empty()) return RTGPU_OK;
    void* p = nullptr;
    RT_HIP(hipMalloc(&p, v.size() * sizeof(T)));
    sc.set_array(member, (const M*)p, v.size() * sizeof(T));
    bytes += v.size() * sizeof(T);
    RT_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RTGPU_OK;
}

int validate(const rtgpu_scene_desc* d)
{
    if (!d) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene description is null");
    if (d->n_instances == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene has no instances");
    if (d->n_instances >= (1u << 27)) return fail(RTGPU_ERR_LIMIT, "more than 2^27 instances");
    if ((d->n_meshes && !d->meshes) || !d->instances) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null mesh/instance array");
    if (d->n_materials == 0 || !d->materials) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene has no materials");
    if ((d->n_point_lights && !d->point_lights) || (d->n_area_lights && !d->area_lights) ||
        (d->n_directional_lights && !d->directional_lights))
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "null light array");
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        const rtgpu_mesh_desc& md = d->meshes[m];
        if (md.n_triangles == 0 || md.n_vertices == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "mesh is empty"); // std::runtime_error("mesh is empty"), shape/TriangleMesh.cpp:63-65
        if (md.n_triangles >= (1u << 27)) return fail(RTGPU_ERR_LIMIT, "mesh has more than 2^27 triangles");
        if (!md.vertices || !md.indices || !md.normals) return fail(RTGPU_ERR_INVALID_ARGUMENT, "mesh arrays must not be null");
        for (size_t i = 0; i < (size_t)md.n_triangles * 3; ++i)
            if (md.indices[i] >= md.n_vertices) return fail(RTGPU_ERR_INVALID_ARGUMENT, "triangle index out of range");
        if (md.tri_material)
            for (uint32_t i = 0; i < md.n_triangles; ++i)
                if (md.tri_material[i] >= d->n_materials) return fail(RTGPU_ERR_INVALID_ARGUMENT, "triangle material id out of range");
        for (size_t i = 0; i < (size_t)md.n_vertices * 3; ++i)
            if (std::isnan(md.vertices[i])) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NaN vertex"); // assert(!hasNaN), shape/AABB.cpp:15-16
    }
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        const rtgpu_instance_desc& id = d->instances[i];
        if (id.shape != RTGPU_SHAPE_SPHERE && (id.shape < 0 || (uint32_t)id.shape >= d->n_meshes))
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "instance shape index out of range");
        if (id.material >= d->n_materials) return fail(RTGPU_ERR_INVALID_ARGUMENT, "instance material id out of range");
        if (!affine(id.transform) || !affine(id.inverse))
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "instance transforms must be affine (last row 0 0 0 1)");
    }
    if (d->has_environment < 0 || d->has_environment > 3) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown environment type");
    if (d->has_environment == 2) {
        // Texture's constructor checks the buffer size (material/Texture.h:18-21)
        if (!d->env_map || d->env_map_width == 0 || d->env_map_height == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "environment map without pixels");
        if ((uint64_t)d->env_map_width * d->env_map_height >= (1ull << 28)) return fail(RTGPU_ERR_LIMIT, "environment map larger than 2^28 pixels");
    }
    if (d->n_textures && !d->textures) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null texture array");
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const rtgpu_texture& t = d->textures[i];
        // Texture's constructor checks the buffer size (material/Texture.h:18-21); an empty image cannot be sampled
        if (!t.rgba || t.width == 0 || t.height == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "texture without pixels");
        if ((uint64_t)t.width * t.height >= (1ull << 30)) return fail(RTGPU_ERR_LIMIT, "texture larger than 2^30 pixels");
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rtgpu_material& m = d->materials[i];
        if (m.type < RTGPU_MAT_DIFFUSE || m.type > RTGPU_MAT_ADD) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown material type");
        if (m.texture > d->n_textures || m.normal_map > d->n_textures) return fail(RTGPU_ERR_INVALID_ARGUMENT, "material texture index out of range");
        const bool inner = m.type == RTGPU_MAT_CONSTMIX || m.type == RTGPU_MAT_FRESNELMIX || m.type == RTGPU_MAT_ADD;
        if (inner && (m.first >= d->n_materials || m.second >= d->n_materials))
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "mix / add material child out of range");
        if (m.type == RTGPU_MAT_ADD && d->materials[m.first].type != RTGPU_MAT_EMISSIVE)
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "the first child of an add material must be emissive");
    }
    // The shade stage resolves a material in at most 4 steps (mix and add materials above the leaf), and
    // a glass material may have at most two mix levels above it (the reference keeps its per-vertex
    // state in a tree of depth 3, utility/DynAllocTree.h:118-131).  A cycle exceeds both.
    struct Walk {
        const rtgpu_scene_desc* d;
        const char* error = nullptr;
        void visit(uint32_t mi, int steps, int mixes)
        {
            if (error) return;
            const rtgpu_material& m = d->materials[mi];
            const bool mix = m.type == RTGPU_MAT_CONSTMIX || m.type == RTGPU_MAT_FRESNELMIX;
            if (mix || m.type == RTGPU_MAT_ADD) {
                if (steps == 4) { error = "mix / add materials nest deeper than 4"; return; }
                visit(m.first, steps + 1, mixes + (mix ? 1 : 0));
                visit(m.second, steps + 1, mixes + (mix ? 1 : 0));
            } else if (m.type == RTGPU_MAT_GLASS && mixes > 2) {
                error = "glass material below more than two mix levels";
            }
        }
    };
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        Walk w{d};
        w.visit(i, 0, 0);
        if (w.error) return fail(RTGPU_ERR_LIMIT, w.error);
    }
    return RTGPU_OK;
}

// IMaterial::hasVariance where it does not depend on the hit: 0 no, 1 yes, 2 hit-dependent
// (DiffuseMaterial.cpp:228-247, GlossyMaterial.cpp:226-229, GlassMaterial.cpp:254-257, AddMaterial.cpp:25-28,
// MixMaterial.cpp:59-65 with the constant factor of MixMaterial.h:19-26 narrowed to float)
uint32_t material_variance(const rtgpu_scene_desc& d, uint32_t mi, int depth, bool diffuseReadsMap = false)
{
    const rtgpu_material& m = d.materials[mi];
    if (depth > 4) return 0;
    switch (m.type) {
    case RTGPU_MAT_DIFFUSE: return diffuseReadsMap ? 0 : 1;
    case RTGPU_MAT_GLOSSY: return m.roughness > 0.0f ? 1 : 0;
    case RTGPU_MAT_GLASS: return 1;
    case RTGPU_MAT_ADD: {
        const uint32_t a = material_variance(d, m.first, depth + 1, diffuseReadsMap), b = material_variance(d, m.second, depth + 1, diffuseReadsMap);
        return (a == 1 || b == 1) ? 1 : ((a == 2 || b == 2) ? 2 : 0);
    }
    case RTGPU_MAT_CONSTMIX: {
        const float f = (float)m.mix_factor;
        if (f == 0.0f) return material_variance(d, m.first, depth + 1, diffuseReadsMap);
        if (f == 1.0f) return material_variance(d, m.second, depth + 1, diffuseReadsMap);
        return 1;
    }
    case RTGPU_MAT_FRESNELMIX: {
        const uint32_t a = material_variance(d, m.first, depth + 1, diffuseReadsMap), b = material_variance(d, m.second, depth + 1, diffuseReadsMap);
        return (a == 1 && b == 1) ? 1 : 2;
    }
    default: return 0;
    }
}

int ceil_log2(uint32_t n)
{
    int l = 0;
    while ((1u << l) < n) ++l;
    return l;
}

} // namespace

namespace rt {
int set_error(int code, const std::string& msg) { return fail(code, msg); }
} // namespace rt

// ---- the host-pointer entry points --------------------------------------------------------------------
// Rays arrive in and results go back to ordinary (pageable) host arrays, so a batch crosses PCIe twice.  The batch is cut
// into chunks; each chunk is copied into a pinned buffer by a host thread, sent to the device, traced and fetched back on
// that thread's stream, and copied out of its pinned buffer -- several threads do this side by side, so the host copies,
// the two DMA directions and the kernels of different chunks overlap.  Pinned and device buffers and the streams are made
// once per scene and kept.
namespace rt {

struct HostPipe {
    static constexpr uint64_t kChunk = 1ull << 17;      // rays per chunk
    static constexpr size_t kInBytes = 28, kOutBytes = 20;   // per ray: origin + direction + limit in, hit record out
    struct Slot {
        char* pinIn = nullptr;
        char* pinOut = nullptr;
        char* devIn = nullptr;
        char* devOut = nullptr;
        hipStream_t stream = nullptr;
    };
    std::vector<Slot> slots;
    // Callers whose arrays are page-locked already (rtgpu_host_alloc, hipHostMalloc, hipHostRegister) need no staging:
    // the DMA engines read and write their arrays.  These slots hold device buffers only.
    static constexpr uint64_t kDirectChunk = 1ull << 17;
    static constexpr size_t kDirectSlots = 4;
    std::vector<Slot> direct;
};

void destroy_host_pipe(HostPipe* p)
{
    if (!p) return;
    for (HostPipe::Slot& s : p->direct) {
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.devIn) (void)hipFree(s.devIn);
        if (s.devOut) (void)hipFree(s.devOut);
    }
    for (HostPipe::Slot& s : p->slots) {
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.pinIn) (void)hipHostFree(s.pinIn);
        if (s.pinOut) (void)hipHostFree(s.pinOut);
        if (s.devIn) (void)hipFree(s.devIn);
        if (s.devOut) (void)hipFree(s.devOut);
    }
    delete p;
}

} // namespace rt

namespace {

hipError_t ensure_host_pipe(const rtgpu_scene& scene, size_t nSlots)
{
    using rt::HostPipe;
    if (!scene.hostPipe) scene.hostPipe = new HostPipe();
    HostPipe& hp = *scene.hostPipe;
    while (hp.slots.size() < nSlots) {
        HostPipe::Slot s;
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.pinIn, HostPipe::kChunk * HostPipe::kInBytes);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.pinOut, HostPipe::kChunk * HostPipe::kOutBytes);
        if (e == hipSuccess) e = hipMalloc((void**)&s.devIn, HostPipe::kChunk * HostPipe::kInBytes);
        if (e == hipSuccess) e = hipMalloc((void**)&s.devOut, HostPipe::kChunk * HostPipe::kOutBytes);
        hp.slots.push_back(s);   // kept even when incomplete: destroy_host_pipe frees what there is
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Runs `n` rays through the chunk pipeline.  fill(slot, first, m) copies the chunk's inputs into slot.pinIn and returns
// the bytes to send; launch(slot, m) enqueues the kernel on slot.stream; drain(slot, first, m, outBytes) copies
// slot.pinOut to the caller's array.
template <typename Fill, typename Launch, typename Drain>
int host_pipeline(const rtgpu_scene& scene, uint64_t n, size_t outBytesPerRay, const Fill& fill, const Launch& launch, const Drain& drain)
{
    using rt::HostPipe;
    std::lock_guard<std::mutex> cacheLock(scene.cacheMutex);
    const uint64_t nChunks = (n + HostPipe::kChunk - 1) / HostPipe::kChunk;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nThreads = (size_t)std::min<uint64_t>(nChunks, std::min<unsigned>(8u, std::max(1u, hw / 2)));
    RT_HIP(ensure_host_pipe(scene, 2 * nThreads));   // two slots per thread: one chunk in flight while the next is filled
    std::vector<hipError_t> errs(nThreads, hipSuccess);
    auto worker = [&](size_t t) {
        hipError_t e = hipSetDevice(scene.device);
        HostPipe::Slot* slots = &scene.hostPipe->slots[2 * t];
        uint64_t pendingFirst[2] = {0, 0}, pendingCount[2] = {0, 0};
        auto finish = [&](int k) {   // waits for the chunk in slot k and hands its results to the caller
            if (pendingCount[k] == 0 || e != hipSuccess) return;
            e = hipStreamSynchronize(slots[k].stream);
            if (e == hipSuccess) drain(slots[k], pendingFirst[k], pendingCount[k]);
            pendingCount[k] = 0;
        };
        int k = 0;
        for (uint64_t c = t; c < nChunks && e == hipSuccess; c += nThreads, k ^= 1) {
            finish(k);
            if (e != hipSuccess) break;
            HostPipe::Slot& slot = slots[k];
            const uint64_t first = c * HostPipe::kChunk, m = std::min(HostPipe::kChunk, n - first);
            const size_t inBytes = fill(slot, first, m);
            e = hipMemcpyAsync(slot.devIn, slot.pinIn, inBytes, hipMemcpyHostToDevice, slot.stream);
            if (e == hipSuccess) e = launch(slot, m);
            if (e == hipSuccess) e = hipMemcpyAsync(slot.pinOut, slot.devOut, m * outBytesPerRay, hipMemcpyDeviceToHost, slot.stream);
            pendingFirst[k] = first;
            pendingCount[k] = m;
        }
        finish(k);
        finish(k ^ 1);
        if (e != hipSuccess) { (void)hipStreamSynchronize(slots[0].stream); (void)hipStreamSynchronize(slots[1].stream); }
        errs[t] = e;
    };
    std::vector<std::thread> threads;
    for (size_t t = 1; t < nThreads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (std::thread& t : threads) t.join();
    for (hipError_t e : errs) RT_HIP(e);
    return RTGPU_OK;
}

// True when [p, p + bytes) is host memory the HIP runtime has page-locked.
bool host_range_pinned(const void* p, size_t bytes)
{
    const char* ends[2] = {(const char*)p, (const char*)p + (bytes ? bytes - 1 : 0)};
    for (const char* q : ends) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) {
            (void)hipGetLastError();   // an ordinary host pointer is reported as an error by some runtimes: not ours to keep
            return false;
        }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// The pipeline for page-locked caller arrays: no staging copies and no helper threads; chunks go round a few streams,
// each with its own device buffers (stream order keeps a buffer's next upload behind its last download).
template <typename Launch>
int direct_pipeline(const rtgpu_scene& scene, uint64_t n, const float* origins, const float* dirs, const float* tmax, void* out,
                    size_t outBytesPerRay, const Launch& launch)
{
    using rt::HostPipe;
    std::lock_guard<std::mutex> cacheLock(scene.cacheMutex);
    if (!scene.hostPipe) scene.hostPipe = new HostPipe();
    HostPipe& hp = *scene.hostPipe;
    while (hp.direct.size() < HostPipe::kDirectSlots) {
        HostPipe::Slot s;
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void**)&s.devIn, HostPipe::kDirectChunk * HostPipe::kInBytes);
        if (e == hipSuccess) e = hipMalloc((void**)&s.devOut, HostPipe::kDirectChunk * HostPipe::kOutBytes);
        hp.direct.push_back(s);   // kept even when incomplete: destroy_host_pipe frees what there is
        RT_HIP(e);
    }
    // chunks of 2^17 rays measured best for 2 073 600 rays (2^16: 1.59 ms, 2^17: 1.48 ms, 2^18: 1.59 ms, 2^19: 1.74 ms): small
    // enough for the two DMA directions to overlap, large enough to keep the runtime calls per ray few
    const uint64_t chunk = HostPipe::kDirectChunk;
    hipError_t e = hipSuccess;
    uint64_t c = 0;
    for (uint64_t first = 0; first < n && e == hipSuccess; first += chunk, ++c) {
        HostPipe::Slot& s = hp.direct[c % HostPipe::kDirectSlots];
        const uint64_t m = std::min(chunk, n - first);
        e = hipMemcpyAsync(s.devIn, origins + 3 * first, m * 12, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s.devIn + m * 12, dirs + 3 * first, m * 12, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess && tmax) e = hipMemcpyAsync(s.devIn + m * 24, tmax + first, m * 4, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = launch(s, m);
        if (e == hipSuccess) e = hipMemcpyAsync((char*)out + first * outBytesPerRay, s.devOut, m * outBytesPerRay, hipMemcpyDeviceToHost, s.stream);
    }
    for (HostPipe::Slot& s : hp.direct) {
        const hipError_t e2 = hipStreamSynchronize(s.stream);
        if (e == hipSuccess) e = e2;
    }
    RT_HIP(e);
    return RTGPU_OK;
}

} // namespace

void rtgpu_scene::release()
{
    (void)hipSetDevice(device);
    rt::destroy_multi(multi);
    multi = nullptr;
    rt::destroy_render_cache(renderCache);
    renderCache = nullptr;
    rt::destroy_host_pipe(hostPipe);
    hostPipe = nullptr;
    arrays.clear();
    const void* ptrs[] = {view.nodes, view.refNodes, view.tris, view.triIds, view.inst, view.instOrder, view.shadeRecs,
                          view.materials, view.pointLights, view.areaLights, view.dirLights, view.vertUV, view.textures, view.texels, view.instFwd, view.envMap, view.kdPos, view.kdNormal, view.kdEnergy};
    for (const void* p : ptrs)
        if (p) (void)hipFree(const_cast<void*>(p));
    view = rt::SceneView{};
}

extern "C" {

const char* rtgpu_last_error(void) { return g_error.c_str(); }

int rtgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rtgpu_set_device(int device)
{
    RT_HIP(hipSetDevice(device));
    return RTGPU_OK;
}

static int scene_create(const rtgpu_scene_desc* desc, rtgpu_scene** out_scene);

int rtgpu_scene_create(const rtgpu_scene_desc* desc, rtgpu_scene** out_scene)
{
    return rt::guarded([&]() -> int { return scene_create(desc, out_scene); });
}

static int scene_create(const rtgpu_scene_desc* desc, rtgpu_scene** out_scene)
{
    using namespace rt;
    if (!out_scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "out_scene is null");
    *out_scene = nullptr;
    if (int rc = validate(desc)) return rc;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return fail(RTGPU_ERR_NO_DEVICE, "no HIP device available");

    const auto t0 = std::chrono::steady_clock::now();
    const int threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const uint32_t nInst = desc->n_instances;
    const bool singleMesh = nInst == 1 && desc->instances[0].shape >= 0;

    // ---- top-level tree over the instances' world boxes (built first: its depth fixes the budget
    // the mesh trees get from the shared per-lane stack) --------------------------------------
    // per-mesh boxes are needed first
    struct MeshBox { float lo[3], hi[3]; };
    std::vector<std::vector<PrimBox>> primBoxes(desc->n_meshes);
    std::vector<MeshBox> meshBox(desc->n_meshes);
    for (uint32_t m = 0; m < desc->n_meshes; ++m) {
        const rtgpu_mesh_desc& md = desc->meshes[m];
        auto& pb = primBoxes[m];
        pb.resize(md.n_triangles);
        parallel_for(md.n_triangles, threads, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                // TriangleMesh::getAABB(i), shape/TriangleMesh.cpp:177-191: min/max of the vertices
                // padded by 1e-4 (added in double, narrowed to float)
                const float* p1 = md.vertices + 3 * (size_t)md.indices[3 * i];
                const float* p2 = md.vertices + 3 * (size_t)md.indices[3 * i + 1];
                const float* p3 = md.vertices + 3 * (size_t)md.indices[3 * i + 2];
                for (int a = 0; a < 3; ++a) {
                    float lo = std::min({p1[a], p2[a], p3[a]}), hi = std::max({p1[a], p2[a], p3[a]});
                    pb[i].lo[a] = (float)(lo - 1E-4);
                    pb[i].hi[a] = (float)(hi + 1E-4);
                }
            }
        });
        MeshBox mb;
        for (int a = 0; a < 3; ++a) { mb.lo[a] = INFINITY; mb.hi[a] = -INFINITY; }
        for (const PrimBox& p : pb)
            for (int a = 0; a < 3; ++a) { mb.lo[a] = std::min(mb.lo[a], p.lo[a]); mb.hi[a] = std::max(mb.hi[a], p.hi[a]); }
        meshBox[m] = mb;
    }
    std::vector<DevInstance> inst(nInst);
    std::vector<PrimBox> instBoxes(nInst);
    for (uint32_t i = 0; i < nInst; ++i) {
        const rtgpu_instance_desc& id = desc->instances[i];
        DevInstance& di = inst[i];
        std::memcpy(di.inv, id.inverse, sizeof(double) * 12);
        di.shape = id.shape;
        di.material = id.material;
        di.root = 0; di.triBase = 0; di.meshVertBase = 0; di.keyBase = 0;
        // InstancedModelList::getAABB (shape/list/InstancedModelList.cpp:21-26) =
        // AABB::getAABBOfTransformed (shape/AABB.cpp:33-46) of the shape's box; a sphere declares
        // +-0.5 (shape/Sphere.cpp:11-14)
        float lo[3], hi[3];
        if (id.shape >= 0) { std::memcpy(lo, meshBox[id.shape].lo, 12); std::memcpy(hi, meshBox[id.shape].hi, 12); }
        else { lo[0] = lo[1] = lo[2] = -0.5f; hi[0] = hi[1] = hi[2] = 0.5f; }
        float wlo[3] = {INFINITY, INFINITY, INFINITY}, whi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int c = 0; c < 8; ++c) {
            float p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]}, q[3];
            xf_point(id.transform, p, q);
            for (int a = 0; a < 3; ++a) { wlo[a] = std::min(wlo[a], q[a]); whi[a] = std::max(whi[a], q[a]); }
        }
        for (int a = 0; a < 3; ++a) {
            if (std::isnan(wlo[a]) || std::isnan(whi[a])) return fail(RTGPU_ERR_INVALID_ARGUMENT, "instance box is NaN");
            di.wlo[a] = instBoxes[i].lo[a] = wlo[a];
            di.whi[a] = instBoxes[i].hi[a] = whi[a];
        }
    }
    bool hasSphere = false;
    for (uint32_t i = 0; i < nInst; ++i) hasSphere = hasSphere || desc->instances[i].shape < 0;
    const bool refTop = hasSphere && nInst > 1;
    BuiltTree top;
    std::vector<Node128> top4;
    uint32_t topDepth4 = 0;
    RefTopTree rtop;
    // The per-lane stack holds at most 3 pending children per four-way level (+ one sentinel between the top-level tree and a
    // mesh tree): what the top-level tree does not need is the mesh trees' budget, counted in binary levels for the builder.
    int meshDepthBudget = std::min((int)kMaxTreeDepth, 2 * (kStackDepth / 3));
    if (refTop) {
        // sort key of InstancedModelList::sortByCentroid (:48-58): transform(shape centroid), with
        // TriangleMesh::getCentroid (shape/TriangleMesh.cpp:57-76, 205-212) for meshes
        std::vector<std::array<float, 3>> meshCen(desc->n_meshes);
        for (uint32_t m = 0; m < desc->n_meshes; ++m) {
            const rtgpu_mesh_desc& md = desc->meshes[m];
            const float cnt = (float)md.n_triangles;
            float c[3] = {0, 0, 0};
            for (uint32_t t = 0; t < md.n_triangles; ++t) {
                const float* p1 = md.vertices + 3 * (size_t)md.indices[3 * (size_t)t];
                const float* p2 = md.vertices + 3 * (size_t)md.indices[3 * (size_t)t + 1];
                const float* p3 = md.vertices + 3 * (size_t)md.indices[3 * (size_t)t + 2];
                for (int a = 0; a < 3; ++a) {
                    const float tc = ((p1[a] + p2[a]) + p3[a]) / 3.0f;
                    c[a] = t == 0 ? tc / cnt : c[a] + tc / cnt;
                }
            }
            meshCen[m] = {c[0], c[1], c[2]};
        }
        std::vector<std::array<float, 3>> keys(nInst);
        for (uint32_t i = 0; i < nInst; ++i) {
            const rtgpu_instance_desc& id = desc->instances[i];
            const float zero[3] = {0, 0, 0};
            float k[3];
            xf_point(id.transform, id.shape >= 0 ? meshCen[id.shape].data() : zero, k);
            keys[i] = {k[0], k[1], k[2]};
        }
        rtop = build_top_tree_reference(instBoxes.data(), reinterpret_cast<const float(*)[3]>(keys.data()), nInst);
        if (rtop.maxDepth > (uint32_t)rt::kTopFramesHost) return fail(RTGPU_ERR_LIMIT, "top-level tree deeper than 48 levels");
        if (rtop.maxLeaf > kMaxTopLeaf) return fail(RTGPU_ERR_LIMIT, "top-level leaf with more than 256 instances");
        if (nInst >= (1u << 23)) return fail(RTGPU_ERR_LIMIT, "more than 2^23 instances in a scene with spheres");
    } else if (nInst > 1) {
        BuildParams tp;
        tp.maxLeaf = 1;
        tp.maxLeafHard = 1;
        tp.threads = threads;
        tp.maxDepth = std::min(24, ceil_log2(nInst) + 4);
        top = build_tree(instBoxes.data(), nInst, tp);
        top4 = collapse_tree4(top.nodes, topDepth4);
        meshDepthBudget = std::min((int)kMaxTreeDepth, 2 * ((kStackDepth - 1 - 3 * (int)topDepth4) / 3)); // one entry for the mesh-tree sentinel
    }

    // ---- mesh trees ------------------------------------------------------------------------
    std::vector<Node128> nodes;
    std::vector<TriRec> tris;
    std::vector<uint32_t> triIds;
    std::vector<float> shadeRecs;   // 12 floats per leaf slot: 9 of the vertex normals, the material id's bits, 2 unused
    std::vector<uint32_t> meshRoot(desc->n_meshes), meshTriBase(desc->n_meshes);
    rtgpu_scene_info info{};
    bool anyUV = false;
    for (uint32_t m = 0; m < desc->n_meshes; ++m)
        if (desc->meshes[m].texcoords) anyUV = true;
    if (desc->n_textures == 0) anyUV = false;   // texture coordinates are only read by texture lookups
    std::vector<float> vertUV;
    for (uint32_t m = 0; m < desc->n_meshes; ++m) {
        const rtgpu_mesh_desc& md = desc->meshes[m];
        BuildParams bp;
        bp.threads = threads;
        bp.maxDepth = meshDepthBudget;
        if (const char* e = std::getenv("RTGPU_BUILD_MAX_LEAF")) bp.maxLeaf = std::max(1, std::min(std::atoi(e), (int)kMaxLeafHard));
        if (const char* e = std::getenv("RTGPU_BUILD_TRAVERSAL_COST")) bp.traversalCost = (float)std::atof(e);
        const int need = ceil_log2((md.n_triangles + kMaxLeafHard - 1) / kMaxLeafHard);
        if (need > meshDepthBudget) return fail(RTGPU_ERR_LIMIT, "scene too deep for the traversal stack");
        BuiltTree bt = build_tree(primBoxes[m].data(), md.n_triangles, bp);
        primBoxes[m].clear();
        primBoxes[m].shrink_to_fit();
        meshRoot[m] = (uint32_t)nodes.size();
        meshTriBase[m] = (uint32_t)tris.size();
        {
            uint32_t depth4 = 0;
            const std::vector<Node128> four = collapse_tree4(bt.nodes, depth4);
            bt.nodes.clear();
            bt.nodes.shrink_to_fit();
            nodes.insert(nodes.end(), four.begin(), four.end());
        }
        const size_t base = tris.size();
        tris.resize(base + md.n_triangles);
        triIds.resize(base + md.n_triangles);
        shadeRecs.resize((base + md.n_triangles) * 12);
        if (anyUV) vertUV.resize((base + md.n_triangles) * 6, 0.0f);
        parallel_for(md.n_triangles, threads, [&](size_t b, size_t e) {
            for (size_t s = b; s < e; ++s) {
                const uint32_t tri = bt.order[s];
                const uint32_t i0 = md.indices[3 * (size_t)tri], i1 = md.indices[3 * (size_t)tri + 1], i2 = md.indices[3 * (size_t)tri + 2];
                const float* a = md.vertices + 3 * (size_t)i0;
                const float* bb = md.vertices + 3 * (size_t)i1;
                const float* c = md.vertices + 3 * (size_t)i2;
                TriRec r;
                r.ax = a[0]; r.ay = a[1]; r.az = a[2];
                r.e1x = a[0] - bb[0]; r.e1y = a[1] - bb[1]; r.e1z = a[2] - bb[2]; // e1 = a - b, math/Triangle.h:38
                r.e2x = a[0] - c[0]; r.e2y = a[1] - c[1]; r.e2z = a[2] - c[2];    // e2 = a - c, math/Triangle.h:39
                // cofactors of rows (e1, e2) in Eigen's 3x3 determinant expansion
                r.c0 = r.e1y * r.e2z - r.e1z * r.e2y;
                r.c1 = r.e1x * r.e2z - r.e1z * r.e2x;
                r.c2 = r.e1x * r.e2y - r.e1y * r.e2x;
                tris[base + s] = r;
                triIds[base + s] = tri;
                float* vn = shadeRecs.data() + (base + s) * 12;
                std::memcpy(vn, md.normals + 3 * (size_t)i0, 12);
                std::memcpy(vn + 3, md.normals + 3 * (size_t)i1, 12);
                std::memcpy(vn + 6, md.normals + 3 * (size_t)i2, 12);
                const uint32_t triMat = md.tri_material ? md.tri_material[tri] : kInvalid;
                std::memcpy(vn + 9, &triMat, 4);
                vn[10] = 0.0f; vn[11] = 0.0f;
                if (anyUV) {
                    float* uv = vertUV.data() + (base + s) * 6;
                    if (md.texcoords) {
                        std::memcpy(uv, md.texcoords + 2 * (size_t)i0, 8);
                        std::memcpy(uv + 2, md.texcoords + 2 * (size_t)i1, 8);
                        std::memcpy(uv + 4, md.texcoords + 2 * (size_t)i2, 8);
                    } else {
                        uv[0] = NAN;   // a mesh without texture coordinates: texture coordinate (0, 0), tangent b - a (shape/TriangleMesh.cpp:141-144)
                    }
                }
            }
        });
        info.n_triangles += md.n_triangles;
        info.n_leaves += bt.leafCount;
        info.max_depth = std::max(info.max_depth, bt.maxDepth);
        info.max_leaf_size = std::max(info.max_leaf_size, bt.maxLeafSize);
    }
    const uint32_t topRoot = (uint32_t)nodes.size();
    std::vector<uint32_t> instOrder(nInst);
    for (uint32_t i = 0; i < nInst; ++i) instOrder[i] = i;
    if (refTop) {
        instOrder = rtop.order;   // its binary nodes go to the device as an array of their own (SceneView::refNodes)
        info.max_depth = rtop.maxDepth + info.max_depth;
    } else if (nInst > 1) {
        nodes.insert(nodes.end(), top4.begin(), top4.end());
        instOrder = top.order;
        info.n_leaves += top.leafCount;
        info.max_depth = top.maxDepth + 1 + info.max_depth;
    }
    if (nodes.size() * sizeof(NodeQ64) >= (1ull << 32)) return fail(RTGPU_ERR_LIMIT, "tree nodes exceed 4 GiB (2^26 four-way nodes)");
    const std::vector<NodeQ64> deviceNodes = quantize_tree4(nodes, threads);
    if (tris.size() * sizeof(TriRec) >= (1ull << 32)) return fail(RTGPU_ERR_LIMIT, "triangle records exceed 4 GiB (89 M triangles)");
    info.n_nodes = nodes.size();
    for (uint32_t i = 0; i < nInst; ++i) {
        if (inst[i].shape >= 0) {
            inst[i].root = meshRoot[inst[i].shape];
            inst[i].triBase = meshTriBase[inst[i].shape];
        }
    }
    // tie-break keys of the cooperative instanced traversal: (instance, triangle) -> keyBase[instance] + triangle
    bool coopInstanced = !refTop && !hasSphere && nInst > 1;
    {
        uint64_t running = 0;
        for (uint32_t i = 0; i < nInst && coopInstanced; ++i) {
            inst[i].keyBase = (uint32_t)running;
            running += desc->meshes[inst[i].shape].n_triangles;
            if (running >= 0xFFFFFFFFull) coopInstanced = false;   // keys would not fit: the per-lane kernel handles the scene
        }
    }
    // ---- materials and lights --------------------------------------------------------------
    std::vector<DevMaterial> mats(desc->n_materials);
    bool anyNormalMap = false;
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const rtgpu_material& m = desc->materials[i];
        DevMaterial& dm = mats[i];
        dm.type = m.type;
        std::memcpy(dm.color, m.color, 12);
        dm.roughness = m.roughness;
        dm.mixFactor = (float)m.mix_factor; // calcMixFactor returns float, material/MixMaterial.h:21-25
        dm.first = m.first;
        dm.second = m.second;
        dm.intensity = m.intensity;
        dm.intensityOverPi = m.intensity / 3.14159265358979323;   // math/Constants.h
        dm.texture = m.texture;
        dm.variance = material_variance(*desc, i, 0);
        dm.normalMap = m.normal_map;
        dm.varianceMap = material_variance(*desc, i, 0, true);
        if (m.normal_map != 0) anyNormalMap = true;
    }
    // object -> world rows of every instance: the tangent of a normal-mapped hit is a world vector
    // (shape/list/InstancedModelList.cpp:131)
    std::vector<double> instFwd;
    if (anyNormalMap) {
        instFwd.resize((size_t)nInst * 12);
        for (uint32_t i = 0; i < nInst; ++i) std::memcpy(instFwd.data() + (size_t)i * 12, desc->instances[i].transform, 12 * sizeof(double));
    }
    std::vector<DevTexture> texs(desc->n_textures);
    std::vector<uint32_t> texels;
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const rtgpu_texture& t = desc->textures[i];
        texs[i].offset = (uint32_t)texels.size();
        texs[i].width = t.width;
        texs[i].height = t.height;
        texs[i].gamma = t.gamma;
        const size_t n = (size_t)t.width * t.height;
        if (texels.size() + n >= (1ull << 32)) return fail(RTGPU_ERR_LIMIT, "textures exceed 2^32 texels");
        texels.resize(texels.size() + n);
        std::memcpy(texels.data() + texs[i].offset, t.rgba, n * 4);
    }
    std::vector<DevPointLight> pls(desc->n_point_lights);
    for (uint32_t i = 0; i < desc->n_point_lights; ++i) {
        std::memcpy(pls[i].pos, desc->point_lights[i].pos, 12);
        std::memcpy(pls[i].color, desc->point_lights[i].color, 12);
        pls[i].intensity = desc->point_lights[i].intensity;
    }
    std::vector<DevAreaLight> als(desc->n_area_lights);
    for (uint32_t i = 0; i < desc->n_area_lights; ++i) {
        const rtgpu_area_light& L = desc->area_lights[i];
        DevAreaLight& d = als[i];
        std::memcpy(d.a, L.a, 12); std::memcpy(d.b, L.b, 12); std::memcpy(d.c, L.c, 12);
        std::memcpy(d.color, L.color, 12);
        d.intensity = L.intensity;
        rt::area_light_derived(d);
    }
    std::vector<DevDirectionalLight> dls(desc->n_directional_lights);
    for (uint32_t i = 0; i < desc->n_directional_lights; ++i) {
        const rtgpu_directional_light& L = desc->directional_lights[i];
        DevDirectionalLight& d = dls[i];
        std::memcpy(d.dir, L.direction, 12);
        d.angle = L.angle;
        std::memcpy(d.color, L.color, 12);
        d.pad = 0.0f;
        d.intensity = L.intensity;
        // OrthonormalBasis(-direction) throws on a zero vector (math/OrthonormalBasis.cpp:13-16)
        if (L.direction[0] == 0.0f && L.direction[1] == 0.0f && L.direction[2] == 0.0f)
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "directional light with a zero direction");
    }

    std::vector<float4> envTexels;
    if (desc->has_environment == 2) {
        envTexels.resize((size_t)desc->env_map_width * desc->env_map_height);
        std::memcpy(envTexels.data(), desc->env_map, envTexels.size() * sizeof(float4));
    }

    const auto t1 = std::chrono::steady_clock::now();
    info.build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();

    auto* sc = new rtgpu_scene();
    sc->device = device;
    SceneView& v = sc->view;
    v = SceneView{};
    uint64_t bytes = 0;
    int rc;
    using SV = SceneView;
    if ((rc = upload(*sc, &SV::nodes, deviceNodes, bytes)) || (rc = upload(*sc, &SV::refNodes, rtop.nodes, bytes)) || (rc = upload(*sc, &SV::tris, tris, bytes)) ||
        (rc = upload(*sc, &SV::triIds, triIds, bytes)) || (rc = upload(*sc, &SV::inst, inst, bytes)) || (rc = upload(*sc, &SV::instOrder, instOrder, bytes)) ||
        (rc = upload(*sc, &SV::shadeRecs, shadeRecs, bytes)) ||
        (rc = upload(*sc, &SV::materials, mats, bytes)) || (rc = upload(*sc, &SV::pointLights, pls, bytes)) ||
        (rc = upload(*sc, &SV::areaLights, als, bytes)) || (rc = upload(*sc, &SV::dirLights, dls, bytes)) ||
        (rc = upload(*sc, &SV::vertUV, vertUV, bytes)) || (rc = upload(*sc, &SV::textures, texs, bytes)) || (rc = upload(*sc, &SV::texels, texels, bytes)) ||
        (rc = upload(*sc, &SV::instFwd, instFwd, bytes)) || (rc = upload(*sc, &SV::envMap, envTexels, bytes))) {
        sc->release();
        delete sc;
        return rc;
    }
    v.nInst = nInst;
    v.topRoot = topRoot;
    v.singleMesh = singleMesh ? 1u : 0u;
    v.refTop = refTop ? 1u : 0u;
    v.coopInstanced = coopInstanced ? 1u : 0u;
    if (singleMesh) {
        static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        bool same = true;
        for (int k = 0; k < 12; ++k) same = same && inst[0].inv[k] == ident[k] && !std::signbit(inst[0].inv[k]);
        v.identityXf = same ? 1u : 0u;
    }
    // a wave steals once 16 of its lanes are out of work: with the 64-byte nodes the kernels are bound by instruction issue and
    // fewer, fuller stealing rounds pay (1 / 4 / 8 / 16: bounce rays 5078 / 5122 / 5072 / 5187 Mrays/s, any-hit 5331 / 5433 / 5437 / 5605)
    // A scene that does not fit the 256 MB Infinity Cache waits for memory, and a round's LDS round trips add to every wait:
    // 24 there (49.8 M triangles, trace stage of the path-traced frame 8 / 16 / 24 / 32: 7.57 / 7.35 / 7.04 / 7.06 ms).
    v.stealMinIdle = std::getenv("RTGPU_STEAL_MIN_IDLE") ? std::atoi(std::getenv("RTGPU_STEAL_MIN_IDLE")) : (bytes > (256ull << 20) ? 24 : 16);
    v.gatherBudget = std::getenv("RTGPU_GATHER_BUDGET") ? (uint32_t)std::strtoul(std::getenv("RTGPU_GATHER_BUDGET"), nullptr, 10) : 192u;
    v.skipBlackShadow = std::getenv("RTGPU_SKIP_BLACK_SHADOW") ? std::atoi(std::getenv("RTGPU_SKIP_BLACK_SHADOW")) : 1;
    v.shadeRegroup = std::getenv("RTGPU_SHADE_REGROUP") ? std::atoi(std::getenv("RTGPU_SHADE_REGROUP")) : 1;
    v.stealLeafPct = std::getenv("RTGPU_STEAL_LEAF_PCT") ? std::atoi(std::getenv("RTGPU_STEAL_LEAF_PCT")) : 48;
    v.refTopRoot = (refTop && rtop.rootIsLeaf) ? encode_top_leaf(0, nInst) : 0;
    v.nMaterials = desc->n_materials;
    v.nPointLights = desc->n_point_lights;
    v.nAreaLights = desc->n_area_lights;
    v.nDirLights = desc->n_directional_lights;
    {
        // the pick probabilities of NextEventEstimation::sample, formed once with the operations of the per-sample code
        const bool hasPoint = v.nPointLights > 0, hasArea = v.nAreaLights > 0, hasDir = v.nDirLights > 0;
        const float total = (hasArea ? 1.0f : 0.0f) + (hasPoint ? 1.0f : 0.0f) + (hasDir ? 1.0f : 0.0f);
        v.neePointP = (hasPoint ? 1.0f : 0.0f) / total;
        v.neeAreaP = (hasArea ? 1.0f : 0.0f) / total;
        v.neeDirP = (hasDir ? 1.0f : 0.0f) / total;
        v.neePointScale = 1.0f / (v.neePointP * (1.0f / (float)v.nPointLights));
        v.neeAreaScale = 1.0f / (v.neeAreaP * (1.0f / (float)v.nAreaLights));
        v.neeDirScale = 1.0f / (v.neeDirP * (1.0f / (float)v.nDirLights));
    }
    v.hasEnv = desc->has_environment;
    v.envW = desc->env_map_width;
    v.envH = desc->env_map_height;
    v.envIntensity = desc->env_intensity;
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        if (desc->materials[i].type == RTGPU_MAT_FRESNELMIX) v.hasFresnelMix = 1;
        if (desc->materials[i].type == RTGPU_MAT_GLASS) v.hasGlass = 1;
        if (desc->materials[i].type == RTGPU_MAT_ADD) v.hasAdd = 1;
        if (desc->materials[i].normal_map != 0) v.hasNormalMap = 1;
    }
    // ColorEnvironment::getRadiance = color * intensity (RGB * double), ColorEnvironment.cpp:6-9
    for (int a = 0; a < 3; ++a) v.envRadiance[a] = (float)((double)desc->env_color[a] * desc->env_intensity);
    info.device_bytes = bytes;
    sc->info = info;
    rt::static_workspace_prepare(device);   // rtgpu_render_device_static finds its workspace's address looked up: its first call allocates nothing either
    *out_scene = sc;
    return RTGPU_OK;
}

void rtgpu_scene_destroy(rtgpu_scene* scene)
{
    if (!scene) return;
    scene->release();
    delete scene;
}

int rtgpu_scene_get_info(const rtgpu_scene* scene, rtgpu_scene_info* out)
{
    if (!scene || !out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null argument");
    *out = scene->info;
    return RTGPU_OK;
}

int rtgpu_trace_closest_device(const rtgpu_scene* scene, const float* d_origins, const float* d_dirs,
                               uint64_t n, rtgpu_hit* d_out_hits, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (n && (!d_origins || !d_dirs || !d_out_hits)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null ray/hit buffer");
    RT_HIP(hipSetDevice(scene->device));
    RT_HIP(rt::launch_trace_closest(scene->view, d_origins, d_dirs, n, d_out_hits, nullptr, (hipStream_t)hip_stream));
    return RTGPU_OK;
}

int rtgpu_trace_closest_count_device(const rtgpu_scene* scene, const float* d_origins, const float* d_dirs,
                                     uint64_t n, uint32_t* d_counts, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (n && (!d_origins || !d_dirs || !d_counts)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null ray/count buffer");
    RT_HIP(hipSetDevice(scene->device));
    rtgpu_hit* tmp = nullptr;
    RT_HIP(hipMalloc((void**)&tmp, std::max<uint64_t>(1, n) * sizeof(rtgpu_hit)));
    hipError_t e = rt::launch_trace_closest(scene->view, d_origins, d_dirs, n, tmp, d_counts, (hipStream_t)hip_stream);
    hipError_t e2 = hipStreamSynchronize((hipStream_t)hip_stream);
    (void)hipFree(tmp);
    RT_HIP(e);
    RT_HIP(e2);
    return RTGPU_OK;
}

int rtgpu_trace_any_device(const rtgpu_scene* scene, const float* d_origins, const float* d_dirs,
                           const float* d_tmax, uint64_t n, uint8_t* d_out_occluded, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (n && (!d_origins || !d_dirs || !d_tmax || !d_out_occluded)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null ray/result buffer");
    RT_HIP(hipSetDevice(scene->device));
    RT_HIP(rt::launch_trace_any(scene->view, d_origins, d_dirs, d_tmax, n, d_out_occluded, (hipStream_t)hip_stream));
    return RTGPU_OK;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
} // namespace

int rtgpu_host_alloc(void** out_ptr, uint64_t bytes)
{
    if (!out_ptr) return fail(RTGPU_ERR_INVALID_ARGUMENT, "out_ptr is null");
    *out_ptr = nullptr;
    if (bytes == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "cannot allocate 0 bytes");
    RT_HIP(hipHostMalloc(out_ptr, (size_t)bytes, hipHostMallocPortable));
    return RTGPU_OK;
}

int rtgpu_host_free(void* ptr)
{
    if (!ptr) return RTGPU_OK;
    RT_HIP(hipHostFree(ptr));
    return RTGPU_OK;
}

int rtgpu_trace_closest(const rtgpu_scene* scene, const float* origins, const float* dirs, uint64_t n, rtgpu_hit* out_hits)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (n == 0) return RTGPU_OK;
    if (!origins || !dirs || !out_hits) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null ray/hit buffer");
    RT_HIP(hipSetDevice(scene->device));
    using rt::HostPipe;
    return rt::guarded([&]() -> int {
        const auto launch = [&](HostPipe::Slot& s, uint64_t m) {
            return rt::launch_trace_closest(scene->view, (const float*)s.devIn, (const float*)(s.devIn + m * 12), m, (rtgpu_hit*)s.devOut, nullptr, s.stream);
        };
        if (host_range_pinned(origins, n * 12) && host_range_pinned(dirs, n * 12) && host_range_pinned(out_hits, n * sizeof(rtgpu_hit)))
            return direct_pipeline(*scene, n, origins, dirs, nullptr, out_hits, sizeof(rtgpu_hit), launch);
        return host_pipeline(*scene, n, sizeof(rtgpu_hit),
            [&](HostPipe::Slot& s, uint64_t first, uint64_t m) -> size_t {
                std::memcpy(s.pinIn, origins + 3 * first, m * 12);
                std::memcpy(s.pinIn + m * 12, dirs + 3 * first, m * 12);
                return (size_t)m * 24;
            },
            launch,
            [&](HostPipe::Slot& s, uint64_t first, uint64_t m) { std::memcpy(out_hits + first, s.pinOut, m * sizeof(rtgpu_hit)); });
    });
}

int rtgpu_trace_any(const rtgpu_scene* scene, const float* origins, const float* dirs, const float* tmax, uint64_t n, uint8_t* out_occluded)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (n == 0) return RTGPU_OK;
    if (!origins || !dirs || !tmax || !out_occluded) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null ray/result buffer");
    RT_HIP(hipSetDevice(scene->device));
    using rt::HostPipe;
    return rt::guarded([&]() -> int {
        const auto launch = [&](HostPipe::Slot& s, uint64_t m) {
            return rt::launch_trace_any(scene->view, (const float*)s.devIn, (const float*)(s.devIn + m * 12), (const float*)(s.devIn + m * 24), m,
                                        (uint8_t*)s.devOut, s.stream);
        };
        if (host_range_pinned(origins, n * 12) && host_range_pinned(dirs, n * 12) && host_range_pinned(tmax, n * 4) && host_range_pinned(out_occluded, n))
            return direct_pipeline(*scene, n, origins, dirs, tmax, out_occluded, 1, launch);
        return host_pipeline(*scene, n, 1,
            [&](HostPipe::Slot& s, uint64_t first, uint64_t m) -> size_t {
                std::memcpy(s.pinIn, origins + 3 * first, m * 12);
                std::memcpy(s.pinIn + m * 12, dirs + 3 * first, m * 12);
                std::memcpy(s.pinIn + m * 24, tmax + first, m * 4);
                return (size_t)m * 28;
            },
            launch,
            [&](HostPipe::Slot& s, uint64_t first, uint64_t m) { std::memcpy(out_occluded + first, s.pinOut, m); });
    });
}

static int check_settings(const rtgpu_render_settings* st)
{
    if (!st) return fail(RTGPU_ERR_INVALID_ARGUMENT, "settings are null");
    if (st->width <= 0 || st->height <= 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "resolution must be positive"); // film/FrameBuffer.cpp:9-14
    if (st->x_start > st->x_end || st->y_start > st->y_end) return fail(RTGPU_ERR_INVALID_ARGUMENT, "Invalid tile bounds"); // film/Tile.cpp:9-12
    if (st->x_start < 0 || st->y_start < 0 || st->x_end > st->width || st->y_end > st->height)
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "tile outside the frame buffer");
    if ((int64_t)st->width * st->height > (int64_t)RTGPU_MAX_PIXELS)
        return fail(RTGPU_ERR_LIMIT, "width * height exceeds 2^30 pixels");
    if (st->geometry_aa < 1 || st->material_aa < 1 || st->max_path_length < 1)
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "sample counts and path length must be >= 1");
    if ((int64_t)st->geometry_aa * st->material_aa > (int64_t)RTGPU_MAX_PATHS_PER_PIXEL)
        return fail(RTGPU_ERR_LIMIT, "geometry_aa * material_aa exceeds 2^25 paths per pixel");
    if (st->max_path_length > 64) return fail(RTGPU_ERR_LIMIT, "max_path_length exceeds 64");
    if (st->jitter_mode != RTGPU_JITTER_RANDOM && st->jitter_mode != RTGPU_JITTER_CENTER)
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown jitter mode");
    if (st->n_gpus < 0 || st->n_gpus > 64) return fail(RTGPU_ERR_INVALID_ARGUMENT, "n_gpus out of range");
    return RTGPU_OK;
}

int rtgpu_generate_primary_device(const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                                  int32_t sample_index, float* d_origins, float* d_dirs, void* hip_stream)
{
    if (!camera) return fail(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_settings(settings)) return rc;
    // thin lens: min(8, aaLevel / 2) lens samples per sensor sample, a divisor (camera/PerspectiveCamera.cpp:28-29)
    if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
    if (!d_origins || !d_dirs) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null ray buffer");
    RT_HIP(rt::launch_generate_primary(*camera, *settings, sample_index, d_origins, d_dirs, (hipStream_t)hip_stream));
    return RTGPU_OK;
}

int rtgpu_render_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                        float* d_rgb_out, rtgpu_render_stats* stats, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (!camera) return fail(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_settings(settings)) return rc;
    // thin lens: min(8, aaLevel / 2) lens samples per sensor sample, a divisor (camera/PerspectiveCamera.cpp:28-29)
    if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
    if (!d_rgb_out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "frame buffer is null");
    if (settings->n_gpus > 1) return fail(RTGPU_ERR_INVALID_ARGUMENT, "n_gpus > 1 needs rtgpu_render or the tile entry points");
    RT_HIP(hipSetDevice(scene->device));
    return rt::guarded([&]() -> int {
        std::string err;
        hipError_t e = rt::render_wavefront(*scene, *camera, *settings, nullptr, d_rgb_out, stats, nullptr, (hipStream_t)hip_stream, err);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP, err.empty() ? hipGetErrorString(e) : err);
        return RTGPU_OK;
    });
}

int rtgpu_render_device_static(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                               float* d_rgb_out, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (!camera) return fail(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_settings(settings)) return rc;
    if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
    if (!d_rgb_out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "frame buffer is null");
    if (settings->n_gpus > 1) return fail(RTGPU_ERR_INVALID_ARGUMENT, "the static-workspace render runs on one device");
    RT_HIP(hipSetDevice(scene->device));
    return rt::guarded([&]() -> int { return rt::render_static(*scene, *camera, *settings, d_rgb_out, (hipStream_t)hip_stream); });
}

int rtgpu_render_profile_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                                float* d_rgb_out, rtgpu_render_profile* profile, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (!camera) return fail(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_settings(settings)) return rc;
    if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
    if (!d_rgb_out || !profile) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null frame buffer / profile record");
    if (settings->n_gpus > 1) return fail(RTGPU_ERR_INVALID_ARGUMENT, "the profiling render runs on one device");
    RT_HIP(hipSetDevice(scene->device));
    return rt::guarded([&]() -> int {
        std::string err;
        hipError_t e = rt::render_wavefront(*scene, *camera, *settings, nullptr, d_rgb_out, nullptr, nullptr, (hipStream_t)hip_stream, err, profile);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP, err.empty() ? hipGetErrorString(e) : err);
        return RTGPU_OK;
    });
}

int rtgpu_render_work_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                             float* d_rgb_out, rtgpu_render_work* work, void* hip_stream)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (!camera) return fail(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_settings(settings)) return rc;
    if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
        return fail(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
    if (!d_rgb_out || !work) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null frame buffer / work record");
    if (settings->n_gpus > 1) return fail(RTGPU_ERR_INVALID_ARGUMENT, "the counting render runs on one device");
    RT_HIP(hipSetDevice(scene->device));
    return rt::guarded([&]() -> int {
        std::string err;
        uint64_t w[9];
        hipError_t e = rt::render_wavefront(*scene, *camera, *settings, nullptr, d_rgb_out, nullptr, w, (hipStream_t)hip_stream, err);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_HIP, err.empty() ? hipGetErrorString(e) : err);
        work->extension_rays = w[0]; work->extension_boxes = w[1]; work->extension_triangles = w[2];
        work->shadow_rays = w[3]; work->shadow_boxes = w[4]; work->shadow_triangles = w[5];
        work->primary_rays = w[6]; work->primary_boxes = w[7]; work->primary_triangles = w[8];
        return RTGPU_OK;
    });
}

int rtgpu_render(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                 float* rgb_out, rtgpu_render_stats* stats, rtgpu_progress_cb progress, void* user)
{
    if (!scene) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene is null");
    if (!camera) return fail(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
    if (int rc = check_settings(settings)) return rc;
    if (!rgb_out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "frame buffer is null");
    RT_HIP(hipSetDevice(scene->device));
    if (settings->n_gpus > 1) {
        if (camera->aperture < 0.0 || (camera->aperture > 0.0 && settings->geometry_aa < 2))
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "a thin-lens camera (aperture > 0) needs geometry_aa >= 2");
        if (progress) progress("Rendering tiles", 0.0f, 0, user);
        if (int rc = rt::render_multi_entry(*scene, *camera, *settings, rgb_out, stats)) return rc;
        if (progress) progress("Rendering tiles", 1.0f, 1, user);
        return RTGPU_OK;
    }
    const size_t bytes = (size_t)settings->width * settings->height * 3 * sizeof(float);
    DevBuf fb;
    RT_HIP(hipMalloc(&fb.p, bytes));
    // pixels outside the tile keep the caller's values (Renderer::render only writes inside the tile)
    RT_HIP(hipMemcpy(fb.p, rgb_out, bytes, hipMemcpyHostToDevice));
    if (progress) progress("Rendering tiles", 0.0f, 0, user);
    if (int rc = rtgpu_render_device(scene, camera, settings, (float*)fb.p, stats, nullptr)) return rc;
    RT_HIP(hipMemcpy(rgb_out, fb.p, bytes, hipMemcpyDeviceToHost));
    if (progress) progress("Rendering tiles", 1.0f, 1, user);
    return RTGPU_OK;
}

// PerspectiveCamera(fov) + setTransform, camera/PerspectiveCamera.cpp:7-19, with
// OrthonormalBasis(a, b), math/OrthonormalBasis.cpp:33-51.
int rtgpu_camera_from_transform(double fov_x_degrees, const double transform[16], rtgpu_camera* out)
{
    if (!transform || !out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "null argument");
    const double PI = 3.14159265358979323;
    out->width = 2.0 * std::tan(0.5 * (fov_x_degrees * PI / 180.0));
    const float zero[3] = {0, 0, 0}, fwd[3] = {0, 0, -1}, upv[3] = {0, 1, 0};
    float origin[3], lookat[3], up[3];
    xf_point(transform, zero, origin);
    xf_vector(transform, fwd, lookat);
    xf_vector(transform, upv, up);
    const float a[3] = {-lookat[0], -lookat[1], -lookat[2]};
    // cross = b x a
    float cr[3] = {up[1] * a[2] - up[2] * a[1], up[2] * a[0] - up[0] * a[2], up[0] * a[1] - up[1] * a[0]};
    const float len = std::sqrt(cr[0] * cr[0] + (cr[1] * cr[1] + cr[2] * cr[2]));
    if (len == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "The vectors are colinear!"); // OrthonormalBasis.cpp:38-41
    const float an2 = a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]);
    float w[3] = {a[0], a[1], a[2]};
    if (an2 > 0) { const float an = std::sqrt(an2); for (float& c : w) c /= an; }
    float u[3] = {cr[0] / len, cr[1] / len, cr[2] / len};
    float v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
    std::memcpy(out->origin, origin, 12);
    std::memcpy(out->u, u, 12);
    std::memcpy(out->v, v, 12);
    std::memcpy(out->w, w, 12);
    out->aperture = 0;
    out->focal_distance = 0;
    return RTGPU_OK;
}

} // extern "C"
