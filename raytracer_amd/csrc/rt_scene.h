// rt_scene.h -- the opaque scene handle of the C ABI and the host entry of the integrator.
#pragma once
#include <hip/hip_runtime.h>

#include <new>
#include <stdexcept>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/rtgpu.h"
#include "rt_device.h"

namespace rt {
struct RenderCache;                       // streams, events and workspaces the integrator keeps between calls
void destroy_render_cache(RenderCache*);
struct HostPipe;                          // pinned / device staging buffers and streams of the host-pointer entry points
void destroy_host_pipe(HostPipe*);
// One device array of a scene: where its pointer sits inside the SceneView and how many bytes it holds (what a copy of
// the scene onto another device has to move).
struct SceneArray {
    size_t viewOffset;
    size_t bytes;
};
// The order in which a render call walks its pixels when the frame is cut into squares shared out over ranks
// (tileSize 0: the settings' tile row by row).
struct TileOrder {
    uint32_t tileSize, nRanks, rank;
};
} // namespace rt

struct rtgpu_multi;                       // replicas and communicators of a multi-GPU render (rt_multi.hip)

struct rtgpu_scene {
    int device = 0;
    rt::SceneView view{};
    rtgpu_scene_info info{};
    std::vector<rtgpu_photon> photons;   // the photon map's photons in emission order (host copy)
    std::vector<rt::SceneArray> arrays;  // every device array `view` points to
    // Serialises the calls that use the state cached below (renders, host-pointer ray batches): two host threads may call into
    // one scene at once, the second waits.  Device-pointer ray batches keep nothing and do not take it.
    mutable std::mutex cacheMutex;
    mutable rt::RenderCache* renderCache = nullptr;
    mutable rt::HostPipe* hostPipe = nullptr;
    mutable rtgpu_multi* multi = nullptr;
    mutable std::mutex multiMutex;       // a multi-GPU render holds it from set-up to the end of the gather: one such call at a time per scene
    void release();
    // Registers / replaces the array behind a pointer member of `view`.
    template <typename T>
    void set_array(const T* rt::SceneView::*member, const T* ptr, size_t bytes)
    {
        view.*member = ptr;
        const size_t off = (size_t)((const char*)&(view.*member) - (const char*)&view);
        for (rt::SceneArray& a : arrays)
            if (a.viewOffset == off) { a.bytes = bytes; return; }
        arrays.push_back(rt::SceneArray{off, bytes});
    }
};

namespace rt {

// Stores the calling thread's error message (rtgpu_last_error) and returns `code`.
int set_error(int code, const std::string& msg);

// Runs the body of a C entry point: no C++ exception crosses the C boundary (host allocations of scene-sized
// vectors are the ones that can fail).
template <typename F>
int guarded(const F& body)
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return set_error(RTGPU_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception& e) {
        return set_error(RTGPU_ERR_INVALID_ARGUMENT, e.what());
    }
}

// Fills normal and area of an area light (AreaLight::getNormal / getSurfaceArea, light/AreaLight.cpp:43-53).
void area_light_derived(DevAreaLight& l);

// Renderer::render as a wavefront pipeline; writes the tile's pixels of the device frame buffer.
// `order` (may be null) cuts the settings' tile into squares and renders this rank's share into a compact buffer;
// `work` (may be null) receives 9 counters of a counting render: extension / shadow / primary rays and the boxes and
// triangles each class tests when every ray is traversed on its own; `profile` (may be null) makes it a profiling render:
// one batch at a time, HIP events around every launch, device time per stage.
hipError_t render_wavefront(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, const TileOrder* order,
                            float* dOut, rtgpu_render_stats* stats, uint64_t* work, hipStream_t stream, std::string& err,
                            rtgpu_render_profile* profile = nullptr);
void destroy_multi(rtgpu_multi*);
// rtgpu_render_device_static: the whole call enqueued on `stream`, workspace in the library's fixed device array.
int render_static(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, float* dRgb, hipStream_t stream);
// Looks the static workspace's address on `device` up (once per device; the code object's load is what places the array
// there).  Called when a scene is made on a device, so that by the time a render can be asked for nothing is left to set up.
void static_workspace_prepare(int device);
// rtgpu_render over settings.n_gpus devices of this process (rt_multi.hip).
int render_multi_entry(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st, float* rgbOut, rtgpu_render_stats* stats);

} // namespace rt
