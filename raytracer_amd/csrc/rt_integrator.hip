// rt_integrator.hip -- wavefront path-tracing integrator (placeholder until the pipeline lands).
#include <hip/hip_runtime.h>

#include "rt_scene.h"

namespace rt {

void area_light_derived(DevAreaLight& l)
{
    (void)l;
}

hipError_t render_wavefront(const rtgpu_scene&, const rtgpu_camera&, const rtgpu_render_settings&, float*,
                            rtgpu_render_stats*, hipStream_t, std::string& err)
{
    err = "integrator not built yet";
    return hipErrorNotSupported;
}

} // namespace rt
