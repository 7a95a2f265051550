// rt_scene.h -- the opaque scene handle of the C ABI and the host entry of the integrator.
#pragma once
#include <hip/hip_runtime.h>

#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtgpu.h"
#include "rt_device.h"

struct rtgpu_scene {
    int device = 0;
    rt::SceneView view{};
    rtgpu_scene_info info{};
    std::vector<rtgpu_photon> photons;   // the photon map's photons in emission order (host copy)
    void release();
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
hipError_t render_wavefront(const rtgpu_scene& scene, const rtgpu_camera& cam, const rtgpu_render_settings& st,
                            float* dRgb, rtgpu_render_stats* stats, hipStream_t stream, std::string& err);

} // namespace rt
