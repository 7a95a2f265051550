// rt_rng_inl.h -- counter-based RNG, stratified samplers and the camera model (device side).
//
// The reference draws from a thread-local pcg32 seeded by std::random_device
// (math/FastRandom.h:6-67), which cannot be reproduced or seeded.  Every Rand::unit() call site
// is replaced by a pure function of (seed, pixel, sample, path vertex, dimension), so a pixel's
// value does not depend on which GPU, tile or launch computes it.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_trace_inl.h"

namespace rt {

enum RngDim : uint32_t {
    DIM_CAM_X = 0, DIM_CAM_Y = 1, DIM_LENS_X = 2, DIM_LENS_Y = 3, DIM_LENS_Z = 4,   // with vertex = kVertexCamera
    DIM_NEE_CHOICE = 1, DIM_LIGHT_TYPE = 2, DIM_LIGHT_INDEX = 3, DIM_AREA_U = 4, DIM_AREA_V = 5,
    DIM_SQUARE_X = 6, DIM_SQUARE_Y = 7, DIM_RR = 8, DIM_DIR_ALPHA = 9, DIM_DIR_BETA = 10, DIM_GLASS = 11, DIM_MIX0 = 12
};
constexpr uint32_t kVertexCamera = 255;

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Uniform float in [0,1) with 24 random bits, the stand-in for Rand::unit() (math/FastRandom.h:28-31).
// The first hashing round only depends on (pixel, sample): code that draws several numbers for one path vertex forms
// it once (rand_prefix) and finishes each draw with rand_unit_from.
__host__ __device__ __forceinline__ uint64_t rand_prefix(uint32_t pixel, uint32_t sample)
{
    return splitmix64(((uint64_t)pixel << 32) | sample);
}
__host__ __device__ __forceinline__ float rand_unit_from(uint64_t prefix, uint32_t seed, uint32_t vertex, uint32_t dim)
{
    const uint64_t k2 = ((uint64_t)seed << 32) | (uint64_t)((vertex << 8) | dim);
    const uint32_t x = (uint32_t)(splitmix64(prefix ^ k2) >> 32);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__host__ __device__ __forceinline__ float rand_unit(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t vertex, uint32_t dim)
{
    return rand_unit_from(rand_prefix(pixel, sample), seed, vertex, dim);
}

// sampleUniformStratifiedSquare, math/Sampler.h:13-26 (r0, r1 are the two Rand::unit() draws).
__device__ __forceinline__ void stratified_square(int level, float sampleI, float r0, float r1, float& sx, float& sy)
{
    if (level < 4) { sx = r0; sy = r1; return; }
    const int binsPerAxis = (int)sqrt((double)level);
    const float binX = (float)fmod((double)sampleI, (double)binsPerAxis);
    const float binY = (float)floor(fmod((double)sampleI, (double)level) / binsPerAxis);
    sx = (binX / binsPerAxis) + (r0 / binsPerAxis);
    sy = (binY / binsPerAxis) + (r1 / binsPerAxis);
}

// sampleUniformStratifiedCube, math/Sampler.h:28-41 (r = the three Rand::unit() draws).
__device__ __forceinline__ F3 stratified_cube(float level, float sampleI, F3 r)
{
    const float binsPerAxis = powf(level, 1.0f / 3.0f);
    const float oneOverBinsPerAxis = 1.0f / binsPerAxis;
    const float binX = fmodf(sampleI, binsPerAxis);
    const float binY = floorf(fmodf(sampleI, binsPerAxis * binsPerAxis) * oneOverBinsPerAxis);
    const float binZ = floorf(fmodf(sampleI, level) * (oneOverBinsPerAxis * oneOverBinsPerAxis));
    return f3((binX * oneOverBinsPerAxis) + (r.x * oneOverBinsPerAxis),
              (binY * oneOverBinsPerAxis) + (r.y * oneOverBinsPerAxis),
              (binZ * oneOverBinsPerAxis) + (r.z * oneOverBinsPerAxis));
}

// sampleUniformCircle(radius, rand1, rand2, rand3), math/Sampler.h:90-103.
__device__ __forceinline__ void uniform_circle(float radius, F3 r, float& x, float& y)
{
    const float angle = (float)(r.x * 2.0f * 3.14159265358979323846);
    float dist = (r.y * radius) + (r.z * radius);
    if (dist > radius) dist = (2.0f * radius) - dist;
    x = cosf(angle) * dist;
    y = sinf(angle) * dist;
}

// PerspectiveCamera::generateRay, camera/PerspectiveCamera.cpp:26-52: pinhole, or thin lens when
// aperture > 0 (the caller guarantees geometry_aa >= 2 then: the reference divides by aaLevel / 2).
__device__ __forceinline__ void camera_ray(const rtgpu_camera& cam, const rtgpu_render_settings& st, int x, int y,
                                           int sampleIndex, F3& o, F3& d)
{
    const uint32_t pixel = (uint32_t)(y * st.width + x);
    const uint32_t sample = (uint32_t)(sampleIndex * st.material_aa);
    const bool random = st.jitter_mode == RTGPU_JITTER_RANDOM;
    float r0 = 0.5f, r1 = 0.5f;
    if (random) {
        r0 = rand_unit(st.seed, pixel, sample, kVertexCamera, DIM_CAM_X);
        r1 = rand_unit(st.seed, pixel, sample, kVertexCamera, DIM_CAM_Y);
    }
    const int aaLevel = st.geometry_aa;
    const float apertureSamples = cam.aperture > 0 ? (float)min(8, aaLevel / 2) : 1.0f;
    float sx, sy;
    stratified_square((int)(aaLevel / apertureSamples), sampleIndex / apertureSamples, r0, r1, sx, sy);
    sx = (float)x + sx;
    sy = (float)y + sy;
    const double height = st.height * cam.width / st.width;
    const double u = cam.width * (sx / st.width - 0.5);
    const double v = -height * (sy / st.height - 0.5);
    const F3 U = f3(cam.u[0], cam.u[1], cam.u[2]), V = f3(cam.v[0], cam.v[1], cam.v[2]), W = f3(cam.w[0], cam.w[1], cam.w[2]);
    const F3 dir = ((U * (float)u) + (V * (float)v)) - W;
    d = normalized3(dir);
    o = f3(cam.origin[0], cam.origin[1], cam.origin[2]);
    if (cam.aperture == 0) return;
    const F3 focalPlanePoint = o + (d * (float)cam.focal_distance);
    F3 r = f3(0.5f, 0.5f, 0.5f);
    if (random) {
        r = f3(rand_unit(st.seed, pixel, sample, kVertexCamera, DIM_LENS_X), rand_unit(st.seed, pixel, sample, kVertexCamera, DIM_LENS_Y),
               rand_unit(st.seed, pixel, sample, kVertexCamera, DIM_LENS_Z));
    }
    const F3 cube = stratified_cube((float)(int)apertureSamples, (float)(int)fmod((double)sampleIndex, (double)apertureSamples), r);
    float ox, oy;
    uniform_circle((float)cam.aperture, cube, ox, oy);
    F3 lensPoint = o;
    lensPoint = lensPoint + (U * ox);
    lensPoint = lensPoint + (V * oy);
    o = lensPoint;
    d = normalized3(focalPlanePoint - lensPoint);
}

} // namespace rt
