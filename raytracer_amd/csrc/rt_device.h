// rt_device.h -- device-side scene view and the launch functions of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtgpu.h"
#include "rt_types.h"

namespace rt {

struct DevMaterial {
    int32_t type;
    float color[3];
    float roughness;
    float mixFactor;
    uint32_t first, second;
    double intensity;
    uint32_t texture;   // 1 + texture index (albedo / emission map), 0 = none
    uint32_t variance;  // IMaterial::hasVariance: 0 no, 1 yes, 2 depends on the hit (a Fresnel mix below decides)
    uint32_t normalMap; // 1 + texture index of a diffuse / glossy normal map, 0 = none
    uint32_t varianceMap; // `variance` for a vertex where diffuse closures read the photon map (they have no variance there)
    double intensityOverPi; // intensity / PI, the factor of DiffuseMaterial::bsdf (material/DiffuseMaterial.cpp:183-195), computed once
};

struct DevTexture {
    uint32_t offset;    // first texel in SceneView::texels
    uint32_t width, height;
    uint32_t pad;
    double gamma;
};

struct DevPointLight {
    float pos[3];
    float color[3];
    double intensity;
};

struct DevAreaLight {
    float a[3], b[3], c[3];
    float color[3];
    double intensity;
    float normal[3]; // AreaLight::getNormal, light/AreaLight.cpp:43-48
    float area;      // Triangle::getSurfaceArea, math/Triangle.h:7-18
    // the same for every sample of the light, so computed once on the host with the operations the per-sample code would use:
    float irradiance[3];   // color.scale(intensity).divide(area): the "lightIrradiance" of Renderer.cpp:199-200, GlossyMaterial.cpp:200-201
    float radiance[3];     // irradiance.divide(PI), NextEventEstimation.cpp (neeAreaLight)
};

struct DevDirectionalLight {
    float dir[3];   // direction the light travels in
    float angle;
    float color[3];
    float pad;
    double intensity;
};

// Everything the kernels need, passed by value as a kernel argument.
struct SceneView {
    const float4* nodes;        // NodeQ64 records as 4 x float4 (four-way trees: mesh trees and the library's own top-level tree)
    const float4* refNodes;     // refTop: the reference-order top-level tree, binary Node64 records as 4 x float4
    const float4* tris;         // TriRec records as 3 x float4
    const uint32_t* triIds;     // leaf slot -> caller's triangle index (per mesh, offset by triBase)
    const DevInstance* inst;
    const uint32_t* instOrder;  // top-level leaf slot -> instance index
    uint32_t nInst;
    uint32_t topRoot;           // first node of the top-level tree
    uint32_t singleMesh;        // 1: exactly one instance and it is a mesh (no top-level tree walk)
    uint32_t identityXf;        // singleMesh and the instance's world->object transform is exactly the identity
    uint32_t coopInstanced;     // several mesh instances, no sphere, keys fit 32 bits: cooperative instanced traversal
    uint32_t refTop;            // 1: top-level tree, child order and early exit follow the reference exactly (scenes with spheres)
    int32_t refTopRoot;         // refTop: 0 (inner root) or the leaf reference when the whole list is one leaf
    int32_t stealMinIdle;       // idle lanes a wave needs before it runs a work-stealing round
    uint32_t gatherBudget;      // tree nodes a photon gather may look at in its first attempt
    int32_t stealLeafPct;       // a wave runs a leaf step once this share (of 128) of its busy lanes waits at a leaf
    // shade stage
    const float4* shadeRecs;    // per leaf slot, 3 x float4: the three vertex normals (9 floats), the triangle's material id
                                // (bits; kInvalid -> instance material), two unused words -- what a vertex needs of its
                                // triangle in one 48-byte record (one or two cache lines instead of three)
    const DevMaterial* materials;
    const DevPointLight* pointLights;
    const DevAreaLight* areaLights;
    const DevDirectionalLight* dirLights;
    const float* vertUV;        // per leaf slot: 3 texture coordinates (6 floats), null when no texture is used
    const DevTexture* textures;
    const uint32_t* texels;     // RGBA8 texels of all textures
    const double* instFwd;      // per instance: rows 0..2 of the object->world matrix; null unless a normal map is used
    uint32_t nMaterials, nPointLights, nAreaLights, nDirLights;
    // NextEventEstimation::sample's pick probabilities (material/NextEventEstimation.cpp:8-120), the same for every sample:
    // P(light kind), and 1 / (P(kind) * P(light of that kind)) as rgb_div_f would form it
    float neePointP, neeAreaP, neeDirP, neePointScale, neeAreaScale, neeDirScale;
    int32_t skipBlackShadow;   // a next-event sample whose value is exactly zero before the visibility test (light behind the surface, light
                               // facing away) is not traced: seen or hidden, the path's value is the same (RTGPU_SKIP_BLACK_SHADOW=0 traces them)
    int32_t shadeRegroup;   // shade passes sort a workgroup's paths by their drawn choices (RTGPU_SHADE_REGROUP=0 turns it off)
    int32_t hasEnv;
    int32_t hasFresnelMix;      // some material is a Fresnel mix: mix factors (and hasVariance) need the hit's normal
    int32_t hasGlass;           // some material is glass: paths carry the extra records of a glass vertex
    int32_t hasAdd;             // some material is an add material: paths carry a collected-emission record
    int32_t hasNormalMap;
    // photon map (null / 0 without one): photons in search order (every range [lo, hi) has the median of its
    // axis at lo + (hi - lo) / 2), xyz of position / surface normal / energy; kdPos holds two records per photon:
    // the position, and the cone of the normals of the range rooted there (axis, cull limit)
    const float4* kdPos;
    const float4* kdNormal;
    const float4* kdEnergy;
    uint32_t nPhotons;
    int32_t pmMode;             // rtgpu_photon_map_mode
    int32_t pmDepth;
    float envRadiance[3];
    const float4* envMap;       // hasEnv == 2: the environment's texels (RGBA floats, rows top to bottom)
    uint32_t envW, envH;
    double envIntensity;
};

// Closest hit for n rays (Scene::traceRay).  counts may be null.
hipError_t launch_trace_closest(const SceneView& sv, const float* dO, const float* dD, uint64_t n,
                                rtgpu_hit* dHits, uint32_t* dCounts, hipStream_t stream);
// Any hit (Scene::testVisibility).
hipError_t launch_trace_any(const SceneView& sv, const float* dO, const float* dD, const float* dTmax,
                            uint64_t n, uint8_t* dOcc, hipStream_t stream);
// PerspectiveCamera::generateRay over a tile.
hipError_t launch_generate_primary(const rtgpu_camera& cam, const rtgpu_render_settings& st, int32_t sampleIndex,
                                   float* dO, float* dD, hipStream_t stream);

} // namespace rt
