// rtgpu_adapter.hpp -- header-only C++17 adapter that gives the C ABI of rtgpu.h the shape of the
// reference's call sites, so that a host program written like raytracer/src/exe/Raytracer.cpp
// changes one line (`Renderer render;` -> `rtgpu::GpuRenderer render{gpuScene, camera};`).
//
//   reference                                                   this adapter
//   ---------------------------------------------------------   -------------------------------------
//   Renderer::render(scene, FrameBuffer&, perfBuffer, Tile,      rtgpu::GpuRenderer::render(buffer, width, height,
//     RenderSettings, ProgressMonitor, multithreaded)               tile, settings, progress)
//     renderer/Renderer.h:17                                        -> rtgpu_render
//   std::optional<SceneRayHitInfo> Scene::traceRay(const Ray&)   std::optional<rtgpu_hit> rtgpu::GpuScene::traceRay(o, d)
//     scene/renderable/Scene.h:96                                   -> rtgpu_trace_closest
//   HitBundle Scene::traceRays(RayBundle&)  Scene.h:97            rtgpu::GpuScene::traceRays(o, d, n, out)
//   std::optional<...> Scene::testVisibility(const Ray&, float)  bool rtgpu::GpuScene::testVisibility(o, d, maxT)
//     Scene.h:100                                                   -> rtgpu_trace_any
//
// Errors: the C ABI never throws; this adapter converts its status codes into the exception types
// the reference throws (std::invalid_argument for argument errors such as film/Tile.cpp:11,
// std::runtime_error otherwise, e.g. "No camera in scene." renderer/Renderer.cpp:318).
#pragma once

#include <cstdint>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtgpu.h"

namespace rtgpu {

inline void check(int rc)
{
    if (rc == RTGPU_OK) return;
    const std::string msg = rtgpu_last_error();
    if (rc == RTGPU_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// film/Tile.h:19-23
struct Tile {
    int xStart, yStart, xEnd, yEnd;
};

// renderer/Renderer.h:8-12 (defaults of the reference) plus the fields the reference hard-codes
struct RenderSettings {
    int geometryAAModifier = 4;
    int materialAAModifier = 32;
    int maxPathLength = 10; // renderer/Renderer.cpp:130
    uint32_t seed = 0;      // the reference seeds from std::random_device; here the image is reproducible
    int gpus = 1;           // > 1: the tile is cut into squares shared out over that many GPUs of this process and gathered
                            // with RCCL (the reference: one process per tile + --mergetiles, exe/Raytracer.cpp:260-277)
};

// utility/ProgressMonitor.h:8 without the duration argument
using ProgressMonitor = std::function<void(const std::string&, float, bool)>;

class GpuScene {
public:
    explicit GpuScene(const rtgpu_scene_desc& desc) { check(rtgpu_scene_create(&desc, &scene_)); }
    ~GpuScene() { rtgpu_scene_destroy(scene_); }
    GpuScene(const GpuScene&) = delete;
    GpuScene& operator=(const GpuScene&) = delete;

    const rtgpu_scene* handle() const { return scene_; }
    rtgpu_scene* mutableHandle() { return scene_; }   // for rtgpu_scene_build_photon_map / rtgpu_scene_set_photons

    // PhotonMapBuilder::buildPhotonMap + Scene::setPhotonMap (photonmapping/PhotonMapBuilder.cpp:15-41,
    // exe/Raytracer.cpp:296-323); returns the number of photons.  Later renders of this scene use the map.
    uint64_t buildPhotonMap(rtgpu_photon_map_mode mode, int depth, uint64_t raysPerPointLight, uint64_t raysPerAreaLight, uint32_t seed = 0)
    {
        uint64_t n = 0;
        check(rtgpu_scene_build_photon_map(scene_, mode, depth, raysPerPointLight, raysPerAreaLight, seed, &n));
        return n;
    }

    // The photons of the map (PhotonMap::serialize / deserialize as memory transfers, exe/Raytracer.cpp:299-318) and
    // Scene::setPhotonMap + setPhotonMapMode + setPhotonMapDepth with photons from elsewhere.
    std::vector<rtgpu_photon> photons() const
    {
        uint64_t n = 0;
        check(rtgpu_scene_get_photons(scene_, nullptr, 0, &n));
        std::vector<rtgpu_photon> out(n);
        check(rtgpu_scene_get_photons(scene_, out.data(), n, &n));
        return out;
    }
    void setPhotons(const std::vector<rtgpu_photon>& photons, rtgpu_photon_map_mode mode, int depth)
    {
        check(rtgpu_scene_set_photons(scene_, photons.data(), photons.size(), mode, depth));
    }

    // Scene::traceRay (scene/renderable/Scene.cpp:14-17).  One ray per call costs a PCIe round trip and a kernel launch
    // (tens of microseconds) whatever the ray: a caller shaped like the reference's shading code -- one testVisibility
    // per light sample (material/NextEventEstimation.cpp:16,38,60), one traceRay per glossy bounce
    // (material/GlossyMaterial.cpp:124) -- should collect its rays and use traceRays / testVisibilities below, which
    // stream any number of rays through pinned buffers at PCIe speed; the integrator inside rtgpu_render never
    // leaves the device for them at all.
    std::optional<rtgpu_hit> traceRay(const float origin[3], const float direction[3]) const
    {
        rtgpu_hit h;
        check(rtgpu_trace_closest(scene_, origin, direction, 1, &h));
        if (h.inst == RTGPU_INVALID_ID) return std::nullopt;
        return h;
    }
    // Scene::traceRays over any number of rays (the reference's bundle holds 32, math/Ray.h:35-40).  Arrays that all lie in
    // page-locked memory (HostArray below) are read and written by the DMA engines without staging copies.
    void traceRays(const float* origins, const float* directions, uint64_t n, rtgpu_hit* out) const
    {
        check(rtgpu_trace_closest(scene_, origins, directions, n, out));
    }
    // Scene::testVisibility (scene/renderable/Scene.cpp:19-22): true when something is hit within maxT
    bool testVisibility(const float origin[3], const float direction[3], float maxT) const
    {
        uint8_t occluded = 0;
        check(rtgpu_trace_any(scene_, origin, direction, &maxT, 1, &occluded));
        return occluded != 0;
    }

    // The batched form of testVisibility: occluded[i] = 1 when ray i hits something within maxT[i].
    void testVisibilities(const float* origins, const float* directions, const float* maxT, uint64_t n, uint8_t* occluded) const
    {
        check(rtgpu_trace_any(scene_, origins, directions, maxT, n, occluded));
    }

private:
    rtgpu_scene* scene_ = nullptr;
};

// An array in page-locked host memory (rtgpu_host_alloc).  Ray batches kept in such arrays cross PCIe without the
// library's staging copies (traceRays / testVisibilities: about 1.4 instead of 0.85 Grays/s on 2 M rays); to the
// host it is ordinary memory.  No counterpart in the reference, whose rays never leave the process.
template <typename T>
class HostArray {
public:
    HostArray() = default;
    explicit HostArray(uint64_t n) { resize(n); }
    HostArray(const HostArray&) = delete;
    HostArray& operator=(const HostArray&) = delete;
    HostArray(HostArray&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    HostArray& operator=(HostArray&& o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~HostArray() { release(); }
    // contents are not kept
    void resize(uint64_t n)
    {
        release();
        if (n == 0) return;
        void* p = nullptr;
        check(rtgpu_host_alloc(&p, n * sizeof(T)));
        p_ = static_cast<T*>(p);
        n_ = n;
    }
    T* data() { return p_; }
    const T* data() const { return p_; }
    uint64_t size() const { return n_; }
    T& operator[](uint64_t i) { return p_[i]; }
    const T& operator[](uint64_t i) const { return p_[i]; }

private:
    void release()
    {
        if (p_) (void)rtgpu_host_free(p_);
        p_ = nullptr;
        n_ = 0;
    }
    T* p_ = nullptr;
    uint64_t n_ = 0;
};

class GpuRenderer {
public:
    GpuRenderer(const GpuScene& scene, const rtgpu_camera& camera) : scene_(scene), camera_(camera) {}

    // Renderer::render (renderer/Renderer.cpp:273-299).  `pixels` is the FrameBuffer storage
    // (film/FrameBuffer.h:14-22: width*height RGB triples of float, pixel (x,y) at y*width+x).
    rtgpu_render_stats render(float* pixels, int width, int height, const Tile& tile, const RenderSettings& rs,
                              ProgressMonitor progress = nullptr) const
    {
        rtgpu_render_settings st{};
        st.n_gpus = rs.gpus;
        st.width = width;
        st.height = height;
        st.x_start = tile.xStart; st.y_start = tile.yStart; st.x_end = tile.xEnd; st.y_end = tile.yEnd;
        st.geometry_aa = rs.geometryAAModifier;
        st.material_aa = rs.materialAAModifier;
        st.max_path_length = rs.maxPathLength;
        st.seed = rs.seed;
        st.jitter_mode = RTGPU_JITTER_RANDOM;
        rtgpu_render_stats stats{};
        ProgressMonitor* user = progress ? &progress : nullptr;
        check(rtgpu_render(scene_.handle(), &camera_, &st, pixels, &stats, progress ? &GpuRenderer::trampoline : nullptr, user));
        return stats;
    }
    // full-frame overload (renderer/Renderer.cpp:301-304)
    rtgpu_render_stats render(float* pixels, int width, int height, const RenderSettings& rs, ProgressMonitor progress = nullptr) const
    {
        return render(pixels, width, height, Tile{0, 0, width, height}, rs, std::move(progress));
    }

private:
    static void trampoline(const char* job, float fraction, int finished, void* user)
    {
        (*static_cast<ProgressMonitor*>(user))(job, fraction, finished != 0);
    }
    const GpuScene& scene_;
    rtgpu_camera camera_;
};

} // namespace rtgpu
