/*
 * rt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a CPU restatement of the reference's algorithm
 * for the hot path (tree build, traversal, triangle/sphere intersection, integrator).  It is
 * the checker for the HIP path and the CPU baseline of the benchmark; nothing in the product
 * (raytracer_amd/, include/) may call, link or load it.
 *
 * Pinning: the reference cannot be compiled in the build container (Eigen3, legacy TBB,
 * Boost and GLFW are neither installed nor vendored), so the oracle is pinned by the
 * known-answer vectors of the reference's own tests only: raytracer/test/BVH.cpp:18-49
 * (sphere traversal), test/Triangle.cpp:7-23, test/Box.cpp:26-54, test/Quaternion.cpp:7-69.
 * Triangle::intersect, the slab test, traversal order and the integrator have no reference
 * test: for those, PARITY IS UNPINNED (restated from the source text only).
 *
 * Scene/camera/settings structs are the ones of include/rtgpu.h so that the oracle and the
 * HIP path are driven by byte-identical inputs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Scene build = DynamicScene::build: per-mesh SAH BVH + top-level SAH BVH over instances. */
orc_scene* orc_scene_create(const rtgpu_scene_desc* desc, int n_threads);
void orc_scene_destroy(orc_scene* s);

/* Photon map (the photonmapping directory): build by tracing (returns the photon count), transfer, and the gather of
 * DiffuseMaterial.cpp:62-75; orc_render uses the scene's map. */
uint64_t orc_build_photon_map(orc_scene* s, int32_t mode, int32_t depth, uint64_t rays_per_point_light, uint64_t rays_per_area_light,
                              uint32_t seed, int n_threads);
uint64_t orc_get_photons(const orc_scene* s, rtgpu_photon* out, uint64_t capacity);
void orc_set_photons(orc_scene* s, const rtgpu_photon* photons, uint64_t n, int32_t mode, int32_t depth);
void orc_gather_photons(const orc_scene* s, const float* points, const float* dirs, uint64_t n, rtgpu_photon_gather* out, int n_threads);

/* Scene::traceRay per ray, reference traversal order and tie-breaks.  counts (optional):
 * 2 uint32 per ray {AABBs tested, triangles tested}. */
void orc_trace_closest(const orc_scene* s, const float* o, const float* d, uint64_t n,
                       rtgpu_hit* out, uint32_t* counts, int n_threads);
/* Scene::testVisibility per ray. */
void orc_trace_any(const orc_scene* s, const float* o, const float* d, const float* tmax,
                   uint64_t n, uint8_t* occluded, int n_threads);

/* Exhaustive closest hit (every instance, every triangle, no tree): the tree-independent
 * answer.  out = hit with the smallest t (ties: smallest instance, then smallest prim);
 * n_tied[i] = number of primitives whose t is bit-equal to the smallest t. */
void orc_brute_closest(const orc_scene* s, const float* o, const float* d, uint64_t n,
                       rtgpu_hit* out, uint32_t* n_tied, int n_threads);
/* 1 if the ray passes the slab test of the instance's own world AABB (AABB::getIntersection >= 0). */
void orc_ray_hits_instance_box(const orc_scene* s, const float* o, const float* d, uint64_t n,
                               uint32_t inst, uint8_t* out);

/* PerspectiveCamera::generateRay for the pixels of the tile, one sample index. */
void orc_generate_primary(const rtgpu_camera* cam, const rtgpu_render_settings* st,
                          int32_t sample_index, float* o, float* d);

/* Renderer::render with std::thread tile fan-out; stats may be NULL. */
void orc_render(const orc_scene* s, const rtgpu_camera* cam, const rtgpu_render_settings* st,
                float* rgb, rtgpu_render_stats* stats, int n_threads);

/* Known-answer helpers (reference unit tests). */
float orc_triangle_area(const float a[3], const float b[3], const float c[3]);
/* FresnelMixMaterial::calcMixFactor, material/FresnelMixMaterial.cpp:3-41 */
float orc_fresnel_mix_factor(const float normal[3], const float dir[3], float ior);
void orc_aabb_of_transformed(const float lo[3], const float hi[3], const double m[16], float out_lo[3], float out_hi[3]);
void orc_aabb_merge(const float lo0[3], const float hi0[3], const float lo1[3], const float hi1[3], float out_lo[3], float out_hi[3]);
double orc_aabb_area(const float lo[3], const float hi[3]);
void orc_aabb_corners(const float lo[3], const float hi[3], float out[24]); /* Box::getCorners, test/Box.cpp:7-16 */
/* KDTree::getElementsNearestTo on a tree given node by node (node i holds point i; children = node indices or -1; root =
 * node 0), as the reference's tests build theirs (test/KDTree.cpp:20-163), and on the tree KDTreeBuilder::build makes.
 * out_results: `count` point indices nearest first, -1 = empty.  Returns the number found. */
uint32_t orc_kd_nearest_explicit(const float* points, const int32_t* child0, const int32_t* child1, const int32_t* axis,
                                 uint32_t n_nodes, const float target[3], uint32_t count, float max_radius,
                                 int64_t* out_results, float* out_max_dist);
uint32_t orc_kd_nearest_built(const float* points, uint32_t n, const float target[3], uint32_t count, float max_radius,
                              int64_t* out_results, float* out_max_dist);
void orc_rotate_quaternion(double i, double j, double k, double r, double out_m[16], double out_inv[16]);
void orc_transform_vector(const double m[16], const float v[3], float out[3]);
void orc_transform_point(const double m[16], const float p[3], float out[3]);
int orc_triangle_intersect(const float o[3], const float d[3], const float a[3], const float b[3],
                           const float c[3], float* beta, float* gamma, float* t);
int orc_sphere_intersect(const float o[3], const float d[3], double* t);
float orc_rand_unit(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t vertex, uint32_t dim);
void orc_camera_from_transform(double fov_x_degrees, const double transform[16], rtgpu_camera* out);
uint64_t orc_scene_node_count(const orc_scene* s);

#ifdef __cplusplus
}
#endif
#endif
