// rt_shade_inl.h -- device functions of the shade / connect stages: samplers, BxDF closures,
// next-event estimation.  Each function restates the reference arithmetic (float/double mix
// included); Rand::unit() draws are replaced by the counter-based RNG of rt_rng_inl.h.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_rng_inl.h"
#include "rt_trace_inl.h"

namespace rt {

#define RT_PI 3.14159265358979323 /* math/Constants.h */

struct RGBf {
    float r, g, b;
};
__device__ __forceinline__ RGBf rgb(float r, float g, float b) { return RGBf{r, g, b}; }
__device__ __forceinline__ RGBf rgb_add(RGBf a, RGBf b) { return RGBf{a.r + b.r, a.g + b.g, a.b + b.b}; } // RGB::operator+, film/RGB.h:93-105
__device__ __forceinline__ RGBf rgb_mul(RGBf a, RGBf b) { return RGBf{a.r * b.r, a.g * b.g, a.b * b.b}; }
__device__ __forceinline__ RGBf rgb_scale_f(RGBf a, float s) { return RGBf{a.r * s, a.g * s, a.b * s}; }                                  // RGB::scale(float), film/RGB.h:49-52
__device__ __forceinline__ RGBf rgb_scale_d(RGBf a, double s) { return RGBf{(float)(a.r * s), (float)(a.g * s), (float)(a.b * s)}; }       // RGB::scale(double), film/RGB.h:44-47
__device__ __forceinline__ RGBf rgb_div_d(RGBf a, double d) { return rgb_scale_d(a, 1.0 / d); }                                           // film/RGB.h:60-63
__device__ __forceinline__ RGBf rgb_div_f(RGBf a, float d) { return rgb_scale_f(a, 1.0f / d); }                                           // film/RGB.h:65-68

struct Basis3 {
    F3 u, v, w;
};
// OrthonormalBasis(a), math/OrthonormalBasis.cpp:10-31
__device__ __forceinline__ Basis3 basis_from(F3 a)
{
    Basis3 b;
    const float length = norm3(a);
    b.w = a / length;
    if (fabsf(b.w.x) > fabsf(b.w.y)) {
        const float inv_length = 1.0f / sqrtf((b.w.x * b.w.x) + (b.w.z * b.w.z));
        b.u = f3(-b.w.z * inv_length, 0.0f, b.w.x * inv_length);
    } else {
        const float inv_length = 1.0f / sqrtf((b.w.y * b.w.y) + (b.w.z * b.w.z));
        b.u = f3(0.0f, b.w.z * inv_length, -b.w.y * inv_length);
    }
    b.v = cross3(b.w, b.u);
    return b;
}

// mapSampleToCosineWeightedHemisphere, math/Sampler.h:51-59 (radius 1, exponent 1)
__device__ __forceinline__ F3 cosine_hemisphere(float r1, float r2)
{
    const float cos_theta = sqrtf(r1); // pow(r1, 1/(1+1))
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    const double phi = 2.0f * RT_PI * r2;
    const double cos_phi = cos(phi);
    const double sin_phi = sin(phi);
    return f3((float)(1.0f * sin_theta * cos_phi), (float)(1.0f * sin_theta * sin_phi), 1.0f * cos_theta);
}

// sampleP22, material/VNDFGGXSampler.cpp:7-43.  Provenance: this is the slope-sampling routine for the GGX
// distribution of visible normals published by E. Heitz and E. d'Eon, "Importance Sampling Microfacet-Based BSDFs
// using the Distribution of Visible Normals" (EGSR 2014, supplemental material); the fitted constants of the
// rational approximation are the paper's, and the arithmetic keeps the reference's order so that samples agree.
__device__ __forceinline__ void sample_p22(double theta_i, double U1, double U2, float& sx, float& sy)
{
    if (theta_i < 0.0001) {
        const double r = sqrt(U1 / (1 - U1));
        const double phi = 6.28318530718 * U2;
        sx = (float)(r * cos(phi));
        sy = (float)(r * sin(phi));
        return;
    }
    const double tan_theta_i = tan(theta_i);
    const double a = 1 / (tan_theta_i);
    const double G1 = 2 / (1 + sqrt(1.0 + 1.0 / (a * a)));
    const double A = 2.0 * U1 / G1 - 1.0;
    const double tmp = 1.0 / (A * A - 1.0);
    const double B = tan_theta_i;
    const double D = sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
    const double slope_x_1 = B * tmp - D;
    const double slope_x_2 = B * tmp + D;
    const double slope_x = (A < 0 || slope_x_2 > 1.0 / tan_theta_i) ? slope_x_1 : slope_x_2;
    double S;
    if (U2 > 0.5) { S = 1.0; U2 = 2.0 * (U2 - 0.5); }
    else          { S = -1.0; U2 = 2.0 * (0.5 - U2); }
    const double z = (U2 * (U2 * (U2 * 0.27385 - 0.73369) + 0.46341)) / (U2 * (U2 * (U2 * 0.093073 + 0.309420) - 1.000000) + 0.597999);
    const double slope_y = S * z * sqrt(1.0 + slope_x * slope_x);
    sx = (float)slope_x;
    sy = (float)slope_y;
}

// VNDFGGXSampler::sample, material/VNDFGGXSampler.cpp:53-90
__device__ __forceinline__ F3 vndf_sample(F3 smoothNormal, F3 worldIncoming, float roughness, double rand1, double rand2)
{
    const Basis3 basis = basis_from(smoothNormal);
    F3 incoming = f3(dot3(basis.u, worldIncoming), dot3(basis.v, worldIncoming), dot3(basis.w, worldIncoming));
    incoming = normalized3(incoming);
    const F3 st = normalized3(f3(incoming.x * roughness, incoming.y * roughness, incoming.z));
    float theta = 0.0f, phi = 0.0f;
    if (st.z < 0.99999) {
        theta = acosf(st.z);
        phi = atan2f(st.y, st.x);
    }
    float slx, sly;
    sample_p22(theta, rand1, rand2, slx, sly);
    float rx = cosf(phi) * slx - sinf(phi) * sly;
    float ry = sinf(phi) * slx + cosf(phi) * sly;
    rx *= roughness;
    ry *= roughness;
    const F3 micro = normalized3(f3(-rx, -ry, 1.0f));
    return f3(basis.u.x * micro.x + (basis.v.x * micro.y + basis.w.x * micro.z),
              basis.u.y * micro.x + (basis.v.y * micro.y + basis.w.y * micro.z),
              basis.u.z * micro.x + (basis.v.z * micro.y + basis.w.z * micro.z));
}

// ggx + brdf, material/GlossyMaterial.cpp:27-66
__device__ __forceinline__ float ggx_g1(float roughness, F3 v, F3 n, F3 m)
{
    const float theta_v = acosf(dot3(v, n));
    if (dot3(v, m) / dot3(v, n) > 0.0f) {
        const float tan_theta = tanf(theta_v);
        return 2.0f / (1.0f + sqrtf(1.0f + (roughness * roughness) * (tan_theta * tan_theta)));
    }
    return 0.0f;
}
__device__ __forceinline__ float ggx_brdf(float roughness, F3 i, F3 o, F3 n, F3 m)
{
    const float g = ggx_g1(roughness, i, n, m) * ggx_g1(roughness, o, n, m);
    if (g == 0.0f) return 0.0f;
    if (!(dot3(m, n) > 0)) return 0.0f;
    const float theta_m = acosf(dot3(m, n));
    const float aSqr = roughness * roughness;
    const float tan_theta_m = tanf(theta_m);
    const float cos_theta_m = cosf(theta_m);
    const float cos_theta_m_sqr = cos_theta_m * cos_theta_m;
    const float x = aSqr + tan_theta_m * tan_theta_m;
    const float d = aSqr / ((float)M_PI * cos_theta_m_sqr * cos_theta_m_sqr * x * x);
    return g * d / (4.0f * fabsf(dot3(i, n)) * fabsf(dot3(o, n)));
}

// Triangle::intersect against an area light (renderer/Renderer.cpp:184-204, material/GlossyMaterial.cpp:126-178)
__device__ __forceinline__ bool area_light_hit(const DevAreaLight& L, F3 o, F3 d, float& t)
{
    const F3 a = f3(L.a[0], L.a[1], L.a[2]), b = f3(L.b[0], L.b[1], L.b[2]), c = f3(L.c[0], L.c[1], L.c[2]);
    const F3 e1 = a - b, e2 = a - c;
    float4 r0, r1, r2;
    r0.x = a.x; r0.y = a.y; r0.z = a.z; r0.w = e1.y * e2.z - e1.z * e2.y;
    r1.x = e1.x; r1.y = e1.y; r1.z = e1.z; r1.w = e1.x * e2.z - e1.z * e2.x;
    r2.x = e2.x; r2.y = e2.y; r2.z = e2.z; r2.w = e1.x * e2.y - e1.y * e2.x;
    float be, ga;
    return tri_test(r0, r1, r2, o, d, t, be, ga);
}

// light.color * light.intensity / area (the "lightIrradiance" of Renderer.cpp:199-200, GlossyMaterial.cpp:200-201);
// scene creation computed it with area_light_constants() below
__device__ __forceinline__ RGBf area_light_irradiance(const DevAreaLight& L) { return rgb(L.irradiance[0], L.irradiance[1], L.irradiance[2]); }

struct NeeSample {
    RGBf radiance;   // value of NextEventEstimation::sample if the shadow ray is unoccluded
    F3 dir;          // lightDirection
    float tmax;      // lampT
    bool valid;      // a light was chosen (otherwise radiance is black and no ray is needed)
};

// sampleUniformSteradianSphere, math/Sampler.h:76-88: (0,0,1) turned by rotateX(alpha) * rotateY(beta)
// (fp64 sines and cosines, math/Transformation.cpp:35-63; the product's third column is
// (sin b, -sin a cos b, cos a cos b)), then into the basis around `center`.
__device__ __forceinline__ F3 sample_steradian_sphere(F3 center, float angle, float r0, float r1)
{
    const float alpha = (r0 - 0.5f) * 2.0f * angle;
    const float beta = (r1 - 0.5f) * 2.0f * angle;
    const double sa = sin((double)alpha), ca = cos((double)alpha), sb = sin((double)beta), cb = cos((double)beta);
    const F3 l = f3((float)sb, (float)(-sa * cb), (float)(ca * cb));
    const Basis3 basis = basis_from(center);
    return f3(basis.u.x * l.x + (basis.v.x * l.y + basis.w.x * l.z),
              basis.u.y * l.x + (basis.v.y * l.y + basis.w.y * l.z),
              basis.u.z * l.x + (basis.v.z * l.y + basis.w.z * l.z));
}

// NextEventEstimation::sample with neePointLight / neeAreaLight / neeDirectionalLight, material/NextEventEstimation.cpp:8-120,
// and AreaLight::generateStratifiedJitteredRandomPoint, light/AreaLight.cpp:26-41.  The visibility
// query itself is left to the shadow stage.
// `nn` = (int)sqrt((double)sampleCount), the stratification grid of the area light, and (sampleX, sampleY) =
// (sampleI % nn, sampleI / nn), the sample's cell (the caller has the divisor per launch).
// `prefix` = rand_prefix(pixel, sample) of the path vertex.
__device__ __forceinline__ NeeSample nee_sample(const SceneView& sv, F3 hitpoint, int sampleX, int sampleY, int nn,
                                                uint32_t seed, uint64_t prefix, uint32_t vertex)
{
    NeeSample out;
    out.radiance = rgb(0, 0, 0);
    out.dir = f3(0, 0, 0);
    out.tmax = 0.0f;
    out.valid = false;
    const bool hasDir = sv.nDirLights > 0;
    const float pointP = sv.neePointP, areaP = sv.neeAreaP;
    const float choice = rand_unit_from(prefix, seed, vertex, DIM_LIGHT_TYPE);
    if (choice < pointP) {
        const uint32_t n = sv.nPointLights;
        uint32_t idx = (uint32_t)(rand_unit_from(prefix, seed, vertex, DIM_LIGHT_INDEX) * (float)n);
        if (idx > n - 1) idx = n - 1;
        const DevPointLight& L = sv.pointLights[idx];
        F3 toLamp = f3(L.pos[0], L.pos[1], L.pos[2]) - hitpoint;
        const float lampT = norm3(toLamp);
        toLamp = normalized3(toLamp);
        const double geometricFactor = 1.0f / (4.0f * RT_PI * (lampT * lampT));
        const RGBf rad = rgb_scale_d(rgb(L.color[0], L.color[1], L.color[2]), L.intensity * geometricFactor);
        out.radiance = rgb_scale_f(rad, sv.neePointScale);   // .divide(pointP * chosenP)
        out.dir = toLamp;
        out.tmax = lampT;
        out.valid = true;
    } else if (choice < pointP + areaP) {
        const uint32_t n = sv.nAreaLights;
        uint32_t idx = (uint32_t)(rand_unit_from(prefix, seed, vertex, DIM_LIGHT_INDEX) * (float)n);
        if (idx > n - 1) idx = n - 1;
        const DevAreaLight& L = sv.areaLights[idx];
        const F3 a = f3(L.a[0], L.a[1], L.a[2]), b = f3(L.b[0], L.b[1], L.b[2]), c = f3(L.c[0], L.c[1], L.c[2]);
        const F3 v1 = (c - a) / (float)nn;
        const F3 v2 = (b - a) / (float)nn;
        const float u = 1.0f - sqrtf(rand_unit_from(prefix, seed, vertex, DIM_AREA_U));
        const float v = (1.0f - u) * rand_unit_from(prefix, seed, vertex, DIM_AREA_V);
        const int y = sampleY, x = sampleX;
        const F3 lampPoint = (a + ((u + x) * v1)) + ((v + y) * v2);
        F3 toLamp = lampPoint - hitpoint;
        const float lampT = norm3(toLamp);
        toLamp = normalized3(toLamp);
        const RGBf lightRadiance = rgb(L.radiance[0], L.radiance[1], L.radiance[2]);
        const F3 nrm = f3(L.normal[0], L.normal[1], L.normal[2]);
        const float lampAngle = fmaxf(0.0f, dot3(nrm, -toLamp));
        const float geometricFactor = lampAngle / (lampT * lampT);
        const RGBf irr = rgb_scale_d(lightRadiance, geometricFactor);
        out.radiance = rgb_scale_f(rgb_scale_d(irr, (double)L.area), sv.neeAreaScale);   // .divide(areaP * chosenP)
        out.dir = toLamp;
        out.tmax = lampT;
        out.valid = true;
    } else if (hasDir) {
        const uint32_t n = sv.nDirLights;
        uint32_t idx = (uint32_t)(rand_unit_from(prefix, seed, vertex, DIM_LIGHT_INDEX) * (float)n);
        if (idx > n - 1) idx = n - 1;
        const DevDirectionalLight& L = sv.dirLights[idx];
        out.dir = sample_steradian_sphere(-f3(L.dir[0], L.dir[1], L.dir[2]), L.angle,
                                          rand_unit_from(prefix, seed, vertex, DIM_DIR_ALPHA), rand_unit_from(prefix, seed, vertex, DIM_DIR_BETA));
        out.radiance = rgb_scale_f(rgb_scale_d(rgb(L.color[0], L.color[1], L.color[2]), L.intensity), sv.neeDirScale);   // .divide(dirP * chosenP)
        out.tmax = INFINITY;
        out.valid = true;
    }
    return out;
}

// RayHitInfo::texCoord of a triangle hit: (alfa * aTexCoord) + (beta * bTexCoord) + (gamma * cTexCoord),
// shape/TriangleMesh.cpp:122-128; (0, 0) for meshes without texture coordinates (:143-146) and, as a
// definition, for spheres (the reference leaves it uninitialised there, shape/Sphere.cpp:55-58).
__device__ __forceinline__ void hit_texcoord(const SceneView& sv, uint32_t slot, float beta, float gamma, float& u, float& v)
{
    u = 0.0f; v = 0.0f;
    if (slot == kInvalid || !sv.vertUV) return;
    const float* t = sv.vertUV + (size_t)slot * 6;
    if (t[0] != t[0]) return;   // scene creation marks the triangles of meshes without texture coordinates with a NaN
    const float alfa = 1.0f - beta - gamma;
    u = ((alfa * t[0]) + (beta * t[2])) + (gamma * t[4]);
    v = ((alfa * t[1]) + (beta * t[3])) + (gamma * t[5]);
}

// Texture::get(x * width, y * height), material/Texture.h:24-33: the texel without filtering, / 255, ^gamma.
// (An index that rounds up to the image size would read past the image in the reference; it is clamped.)
__device__ __forceinline__ RGBf texture_get(const SceneView& sv, uint32_t texture, float x, float y)
{
    const DevTexture& T = sv.textures[texture];
    uint32_t px = (uint32_t)(x * T.width), py = (uint32_t)(y * T.height);
    if (px > T.width - 1) px = T.width - 1;
    if (py > T.height - 1) py = T.height - 1;
    const uint32_t texel = sv.texels[T.offset + py * T.width + px];
    const float r = (float)(texel & 255u) / 255.0f, g = (float)((texel >> 8) & 255u) / 255.0f, b = (float)((texel >> 16) & 255u) / 255.0f;
    return rgb((float)pow((double)r, T.gamma), (float)pow((double)g, T.gamma), (float)pow((double)b, T.gamma));
}
// diffuseColor of DiffuseMaterial::bsdf with an albedoMap, material/DiffuseMaterial.cpp:144-152
__device__ __forceinline__ RGBf albedo_lookup(const SceneView& sv, uint32_t texture, float u, float v)
{
    float x = fmodf(u, 1.0f), y = fmodf(v, 1.0f);
    if (x < 0) x = 1.0f + x;
    if (y < 0) y = 1.0f + y;
    return texture_get(sv, texture, x, y);
}
// emissionColor of EmissiveMaterial::bsdf with an emissionMap, material/EmissiveMaterial.cpp:14-21
__device__ __forceinline__ RGBf emission_lookup(const SceneView& sv, uint32_t texture, float u, float v)
{
    return texture_get(sv, texture, fabsf(fmodf(u, 1.0f)), fabsf(fmodf(v, 1.0f)));
}

// IEnvironmentMaterial::getRadiance along `dir`: ColorEnvironment (material/environment/ColorEnvironment.cpp:6-9) or
// ImageMapEnvironment (ImageMapEnvironment.cpp:15-23: latitude = asin(y), longitude = pi + atan2(z / cos, x / cos) in
// float, texel (int)(width * longitude / 2 pi), (int)(height * (1 - (pi / 2 + latitude) / pi)) of a Texture<float>,
// times the intensity).  Indices the reference would read outside the image with (straight up or down, the seam) are
// clamped to it; a direction along the y axis (0 / 0 in the longitude) reads column 0.
__device__ __forceinline__ RGBf env_radiance(const SceneView& sv, F3 dir)
{
    if (sv.hasEnv == 3) {
        // SkyEnvironment::getRadiance (material/environment/SkyEnvironment.cpp:3-10): a vertical blend of a sky colour and white
        const float t = 0.5f * (dir.y + 1.0f);
        return rgb_add(rgb_scale_f(rgb(0.5f, 0.7f, 1.0f), t), rgb_scale_f(rgb(1.0f, 1.0f, 1.0f), 1.0f - t));
    }
    if (sv.hasEnv != 2) return rgb(sv.envRadiance[0], sv.envRadiance[1], sv.envRadiance[2]);
    const float latitude = asinf(dir.y);
    const float cl = cosf(latitude);
    const double longitude = RT_PI + (double)atan2f(dir.z / cl, dir.x / cl);
    const double x = (double)sv.envW * longitude / (2.0f * RT_PI);
    const double y = (double)sv.envH * (1.0f - (((RT_PI / 2.0f) + (double)latitude) / RT_PI));
    int xi = x == x ? (int)x : 0, yi = y == y ? (int)y : 0;
    xi = xi < 0 ? 0 : (xi > (int)sv.envW - 1 ? (int)sv.envW - 1 : xi);
    yi = yi < 0 ? 0 : (yi > (int)sv.envH - 1 ? (int)sv.envH - 1 : yi);
    const float4 t = sv.envMap[(size_t)yi * sv.envW + (size_t)xi];
    return rgb_scale_d(rgb(t.x / 1.0f, t.y / 1.0f, t.z / 1.0f), sv.envIntensity);   // Texture::get: / divisor, pow(gamma = 1)
}

// sample_normal_map (material/NormalMapSampler.h:7-21) at a triangle hit.  The hit's tangent is row 0 of
// uvMat^-1 * deltaPosMat (shape/TriangleMesh.cpp:130-139; 2x2 inverse = adjugate / determinant), taken to
// world space as a vector (shape/list/InstancedModelList.cpp:131); tangentToObject (math/Transformation.cpp:
// 132-145) builds the frame (tangent, normal x tangent, normal) from the normalised tangent and normal.
// A sphere has no tangent in the reference (shape/Sphere.cpp:56 leaves it unset): zero here.
__device__ __forceinline__ F3 sample_normal_map(const SceneView& sv, uint32_t texture, uint32_t inst, uint32_t slot, float beta, float gamma, F3 normal)
{
    F3 tangent = f3(0.0f, 0.0f, 0.0f);
    float u = 0.0f, v = 0.0f;
    if (slot != kInvalid) {
        hit_texcoord(sv, slot, beta, gamma, u, v);
        const float4 r1 = sv.tris[(size_t)slot * 3 + 1], r2 = sv.tris[(size_t)slot * 3 + 2];
        const F3 d0 = f3(-r1.x, -r1.y, -r1.z), d1 = f3(-r2.x, -r2.y, -r2.z);   // b - a, c - a (the records hold a - b, a - c)
        const float* t = sv.vertUV ? sv.vertUV + (size_t)slot * 6 : nullptr;
        F3 ot = d0;   // without texture coordinates: tangent = b - a, shape/TriangleMesh.cpp:141-144
        if (t && t[0] == t[0]) {
            const float m00 = t[2] - t[0], m01 = t[3] - t[1], m10 = t[4] - t[0], m11 = t[5] - t[1];
            const float det = (m00 * m11) - (m10 * m01);
            const float invdet = 1.0f / det;
            const float i00 = m11 * invdet, i01 = -m01 * invdet;
            ot = f3((i00 * d0.x) + (i01 * d1.x), (i00 * d0.y) + (i01 * d1.y), (i00 * d0.z) + (i01 * d1.z));
        }
        const double* m = sv.instFwd + (size_t)inst * 12;
        const double x = ot.x, y = ot.y, z = ot.z;
        tangent = f3((float)(((m[0] * x + m[1] * y) + m[2] * z) + m[3] * 0.0), (float)(((m[4] * x + m[5] * y) + m[6] * z) + m[7] * 0.0),
                     (float)(((m[8] * x + m[9] * y) + m[10] * z) + m[11] * 0.0));
    }
    float x = fmodf(u, 1.0f), y = fmodf(v, 1.0f);
    if (x < 0) x = 1.0f + x;
    if (y < 0) y = 1.0f + y;
    const RGBf c = texture_get(sv, texture, x, y);
    const double nx = 2.0f * (c.r - 0.5f), ny = 2.0f * (c.g - 0.5f), nz = 2.0f * (c.b - 0.5f);
    const F3 T = normalized3(tangent), N = normalized3(normal);
    const F3 B = cross3(N, T);
    const F3 mapped = f3((float)((((double)T.x * nx + (double)B.x * ny) + (double)N.x * nz) + 0.0),
                         (float)((((double)T.y * nx + (double)B.y * ny) + (double)N.y * nz) + 0.0),
                         (float)((((double)T.z * nx + (double)B.z * ny) + (double)N.z * nz) + 0.0));
    return normalized3(mapped);
}

// FresnelMixMaterial::calcMixFactor, material/FresnelMixMaterial.cpp:3-41: unpolarised Fresnel
// reflectance of the hit for index of refraction `ior` against air, 1 on total internal reflection.
__device__ __forceinline__ float fresnel_mix_factor(F3 normal, F3 incomingDir, float ior)
{
    const F3 objToIncoming = -incomingDir;
    double nDotWo = dot3(normal, objToIncoming);
    const double incomingAngle = nDotWo;
    const bool isInternalRay = nDotWo < 0;
    double curRelIor = ior / 1.0;
    if (isInternalRay) curRelIor = 1.0 / curRelIor;
    const double cosTransmissionSqr = 1.0 - ((1.0 - (incomingAngle * incomingAngle)) / (curRelIor * curRelIor));
    if (cosTransmissionSqr < 0) return 1.0f;
    if (isInternalRay) nDotWo = -nDotWo;
    const double cosRefractAngle = sqrt(cosTransmissionSqr);
    const double rs = ((curRelIor * nDotWo) - cosRefractAngle) / ((curRelIor * nDotWo) + cosRefractAngle);
    const double rp = (nDotWo - (curRelIor * cosRefractAngle)) / (nDotWo + (curRelIor * cosRefractAngle));
    const double kr = ((rs * rs) + (rp * rp)) / 2.0;
    const double reflectionWeight = kr;
    const double transmissionWeight = 1.0 - kr;
    const double totalWeight = reflectionWeight + transmissionWeight;
    return (float)(reflectionWeight / totalWeight);
}

__device__ __forceinline__ bool mat_is_mix(const DevMaterial& M) { return M.type == RTGPU_MAT_CONSTMIX || M.type == RTGPU_MAT_FRESNELMIX; }

// MixMaterial::calcMixFactor of a mix node at this hit (material/MixMaterial.h:19-26, FresnelMixMaterial.cpp)
__device__ __forceinline__ float mat_mix_factor(const DevMaterial& M, F3 normal, F3 incomingDir)
{
    return M.type == RTGPU_MAT_CONSTMIX ? M.mixFactor : fresnel_mix_factor(normal, incomingDir, M.roughness);
}

// IMaterial::hasVariance: Diffuse true (DiffuseMaterial.cpp:228-247 without photon map), Glossy
// roughness > 0 (GlossyMaterial.cpp:226-229), Glass true (GlassMaterial.cpp:254-257), Emissive and
// Transparent false, Add first || second (AddMaterial.cpp:25-28), Mix: unless (factor==0 && !first)
// || (factor==1 && !second) (MixMaterial.cpp:59-65).  DevMaterial::variance holds the answer for
// materials where it does not depend on the hit (0 / 1); 2 = a Fresnel mix decides per hit.  Mix and
// add materials nest at most 4 deep (checked at scene creation), so the recursion is unrolled.
template <int DEPTH>
__device__ __forceinline__ bool mat_variance_at(const SceneView& sv, uint32_t mi, F3 normal, F3 incomingDir, bool diffuseReadsMap)
{
    const DevMaterial& M = sv.materials[mi];
    const uint32_t cls = diffuseReadsMap ? M.varianceMap : M.variance;
    if (cls != 2u) return cls != 0u;
    if constexpr (DEPTH == 0) {
        return false;
    } else {
        if (M.type == RTGPU_MAT_ADD)
            return mat_variance_at<DEPTH - 1>(sv, M.first, normal, incomingDir, diffuseReadsMap) || mat_variance_at<DEPTH - 1>(sv, M.second, normal, incomingDir, diffuseReadsMap);
        const float f = mat_mix_factor(M, normal, incomingDir);
        const bool noVar = (f == 0.0f && !mat_variance_at<DEPTH - 1>(sv, M.first, normal, incomingDir, diffuseReadsMap)) ||
                           (f == 1.0f && !mat_variance_at<DEPTH - 1>(sv, M.second, normal, incomingDir, diffuseReadsMap));
        return !noVar;
    }
}
// diffuseReadsMap: the path is at a vertex where a diffuse closure reads the photon map instead of sampling
// (PhotonMapMode::full, DiffuseMaterial.cpp:228-247) and therefore has no variance
__device__ __forceinline__ bool mat_has_variance(const SceneView& sv, uint32_t mi, F3 normal, F3 incomingDir, bool diffuseReadsMap = false)
{
    return mat_variance_at<4>(sv, mi, normal, incomingDir, diffuseReadsMap);
}

// RGB::pow(double), film/RGB.h:85-88
__device__ __forceinline__ RGBf rgb_pow_d(RGBf a, double power) { return RGBf{(float)pow((double)a.r, power), (float)pow((double)a.g, power), (float)pow((double)a.b, power)}; }

} // namespace rt
