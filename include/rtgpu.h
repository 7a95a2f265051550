/*
 * rtgpu.h -- C ABI of the MI355X (gfx950) ray-tracing hot path.
 *
 * This is the drop-in boundary for the data-parallel hot path of Wouterdek/raytracer:
 * spatial-tree traversal, ray/triangle and ray/sphere intersection and the path-tracing
 * integrator.  The reference has no FFI layer on this path (it is one C++ process); the
 * two C++ seams this ABI replaces are
 *
 *   (1) Renderer::render(scene, FrameBuffer&, perfBuffer, Tile, RenderSettings, ProgressMonitor, bool)
 *       raytracer/src/lib/renderer/Renderer.h:17        -> rtgpu_render / rtgpu_render_device
 *   (2) Scene::traceRay / Scene::traceRays / Scene::testVisibility
 *       raytracer/src/lib/scene/renderable/Scene.h:96-100 -> rtgpu_trace_closest / rtgpu_trace_any
 *
 * and the scene hand-over replaces DynamicScene::build()
 *   raytracer/src/lib/scene/dynamic/DynamicScene.cpp:99-161 -> rtgpu_scene_create.
 *
 * Conventions: plain pointers and sizes, no exceptions cross the ABI, every function
 * returns an rtgpu_status and rtgpu_last_error() holds the message of the last failure on
 * the calling thread.  All calls block until the result is complete unless the name ends
 * in _device (those enqueue on the given HIP stream and return; the caller synchronises).
 * One host thread per scene handle at a time.
 */
#ifndef RTGPU_H
#define RTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGPU_INVALID_ID 0xFFFFFFFFu /* RayHitInfo::triangleIndex of analytic shapes, math/RayHitInfo.h:22-24 */
#define RTGPU_SHAPE_SPHERE (-1)       /* unit sphere at the object-space origin, shape/Sphere.cpp:16-59 */

typedef enum rtgpu_status {
    RTGPU_OK = 0,
    RTGPU_ERR_INVALID_ARGUMENT = 1, /* std::invalid_argument in the reference (film/Tile.cpp:11, film/FrameBuffer.cpp:9-14) */
    RTGPU_ERR_NO_DEVICE = 2,        /* no gfx950 device / HIP runtime unusable */
    RTGPU_ERR_HIP = 3,              /* a HIP call failed; message holds hipGetErrorString */
    RTGPU_ERR_OUT_OF_MEMORY = 4,
    RTGPU_ERR_NO_CAMERA = 5,        /* std::runtime_error("No camera in scene.") renderer/Renderer.cpp:318 */
    RTGPU_ERR_LIMIT = 6,            /* a documented size limit of an entry point was exceeded */
    RTGPU_ERR_IO = 7,               /* std::runtime_error("Failed to load gltf") io/GLTF.cpp:676-678 and the loader's other
                                     * complaints about a file's content */
    RTGPU_ERR_UNSUPPORTED = 8,      /* std::runtime_error("Unsupported ...") of the loader (io/GLTF.cpp:62,82,91,204,233), and
                                     * file features this library has no decoder for (JPEG, HDR maps) */
    RTGPU_ERR_COMM = 9              /* RCCL could not be loaded or one of its calls failed (multi-GPU rendering) */
} rtgpu_status;

/* ---- scene description (flat arrays; replaces the DynamicScene graph) ---------------- */

/* TriangleMeshData, shape/TriangleMesh.h:8-21.  Normals are per vertex and share the vertex
 * indices (what io/GLTF.cpp produces).  tri_material, when non-NULL, is the flattened
 * CompositeMaterial range table (material/CompositeMaterial.cpp:41-84): one material id per
 * triangle in the caller's (original) triangle order. */
typedef struct rtgpu_mesh_desc {
    const float* vertices;        /* n_vertices * 3 */
    const float* normals;         /* n_vertices * 3 */
    const uint32_t* indices;      /* n_triangles * 3 */
    const uint32_t* tri_material; /* n_triangles or NULL */
    uint32_t n_vertices;
    uint32_t n_triangles;
    const float* texcoords;       /* n_vertices * 2 (TriangleMeshData::texCoords, sharing the vertex indices) or NULL:
                                   * hits then carry the texture coordinate (0, 0), shape/TriangleMesh.cpp:143-146 */
} rtgpu_mesh_desc;

/* SceneNode<Model>: transform + (shape, material), scene/renderable/SceneNode.h:5-36,
 * model/Model.h:9-47.  Matrices are row-major 4x4 doubles like Transformation's Matrix4d
 * pair (math/Transformation.h:40-43); the caller supplies the inverse, as the reference does. */
typedef struct rtgpu_instance_desc {
    int32_t shape;     /* index into meshes[], or RTGPU_SHAPE_SPHERE */
    uint32_t material; /* material id (ignored for triangles of a mesh with tri_material) */
    double transform[16];
    double inverse[16];
} rtgpu_instance_desc;

typedef enum rtgpu_material_type {
    RTGPU_MAT_DIFFUSE = 0,  /* material/DiffuseMaterial.cpp */
    RTGPU_MAT_GLOSSY = 1,   /* material/GlossyMaterial.cpp  ("specular") */
    RTGPU_MAT_EMISSIVE = 2, /* material/EmissiveMaterial.cpp */
    RTGPU_MAT_CONSTMIX = 3, /* material/MixMaterial.h:19-26 */
    RTGPU_MAT_FRESNELMIX = 4,  /* material/FresnelMixMaterial.cpp:3-41: mix whose factor is the Fresnel reflectance
                                * of the hit (clear coat); `roughness` carries the index of refraction */
    RTGPU_MAT_TRANSPARENT = 5, /* material/TransparentMaterial.cpp:3-26: the path goes straight on */
    RTGPU_MAT_GLASS = 6,    /* material/GlassMaterial.cpp:21-178 (GlassMaterial.h:15-17): `intensity` = ior (double there),
                             * `roughness` = attenuationStrength (float there), `color` = the absorption colour.
                             * At most two mix levels above it (the reference's metadata tree, utility/DynAllocTree.h) */
    RTGPU_MAT_ADD = 7       /* material/AddMaterial.cpp:5-17: `first` (an emissive material, as io/GLTF.cpp:435-441
                             * builds it) is added to what `second` returns for the same path vertex */
} rtgpu_material_type;

typedef struct rtgpu_material {
    int32_t type;       /* rtgpu_material_type */
    float color[3];     /* diffuseColor / color */
    double intensity;   /* diffuseIntensity / emissive intensity */
    float roughness;    /* glossy: roughness; fresnel mix: IOR (FresnelMixMaterial.h: float IOR = 1.45f) */
    uint32_t texture;   /* 1 + index into textures[] of the diffuse albedoMap (material/DiffuseMaterial.cpp:144-152) or the
                         * emissive emissionMap (material/EmissiveMaterial.cpp:14-21); 0 = none */
    double mix_factor;  /* const mix: 0 = 100 % first, 1 = 100 % second */
    uint32_t first;     /* children of a const / fresnel mix or of an add material (material ids) */
    uint32_t second;
    uint32_t normal_map; /* 1 + index into textures[] of the diffuse / glossy normalMap (material/NormalMapSampler.h:7-21,
                          * DiffuseMaterial.cpp:33-37, GlossyMaterial.cpp:81-85); 0 = none.  Needs mesh texcoords: the
                          * tangent comes from them (shape/TriangleMesh.cpp:130-139) */
    uint32_t reserved;
} rtgpu_material;

typedef struct rtgpu_point_light { /* light/PointLight.h:9-22 (world space) */
    float pos[3];
    float color[3];
    double intensity;
} rtgpu_point_light;

typedef struct rtgpu_area_light { /* light/AreaLight.h (one world-space triangle) */
    float a[3];
    float b[3];
    float c[3];
    float color[3];
    double intensity;
} rtgpu_area_light;

typedef struct rtgpu_directional_light { /* light/DirectionalLight.h:9-24 (world space) */
    float direction[3]; /* direction the light travels in, normalised (applyTransform, DirectionalLight.cpp:10-13) */
    float angle;        /* "radius size in hemisphere": the sampled cone's half angle in radians */
    float color[3];
    float reserved;
    double intensity;   /* irradiance */
} rtgpu_directional_light;

/* TextureUInt8, material/Texture.h:5-61: RGBA8 pixels, rows top to bottom; a lookup takes the texel at
 * (u * width, v * height) without filtering, divides by 255 and raises to `gamma`. */
typedef struct rtgpu_texture {
    const uint8_t* rgba; /* width * height * 4 */
    uint32_t width;
    uint32_t height;
    double gamma;        /* Texture::gamma, 1.0 = none */
} rtgpu_texture;

typedef struct rtgpu_scene_desc {
    const rtgpu_mesh_desc* meshes;
    const rtgpu_instance_desc* instances;
    const rtgpu_material* materials;
    const rtgpu_point_light* point_lights;
    const rtgpu_area_light* area_lights;
    uint32_t n_meshes;
    uint32_t n_instances;
    uint32_t n_materials;
    uint32_t n_point_lights;
    uint32_t n_area_lights;
    int32_t has_environment; /* 0: none; 1: ColorEnvironment (material/environment/ColorEnvironment.cpp:6-9), radiance =
                              * env_color * env_intensity; 2: ImageMapEnvironment (ImageMapEnvironment.cpp:15-23), radiance =
                              * the texel of env_map at the direction's longitude / latitude * env_intensity; 3: SkyEnvironment
                              * (material/environment/SkyEnvironment.cpp:3-10), a fixed blend of sky blue and white by the direction's
                              * height (no parameters; no loader of the reference creates one) */
    float env_color[3];
    double env_intensity;
    const rtgpu_directional_light* directional_lights; /* may be null when n_directional_lights == 0 */
    uint32_t n_directional_lights;
    uint32_t n_textures;
    const rtgpu_texture* textures; /* may be null when n_textures == 0 */
    /* Texture<float> of the ImageMapEnvironment (material/Texture.h, divisor 1, gamma 1): RGBA floats, rows top to bottom
     * (what read_hdr_file returns, io/HDR.cpp:5-18); read only when has_environment == 2 */
    const float* env_map;
    uint32_t env_map_width;
    uint32_t env_map_height;
} rtgpu_scene_desc;

typedef struct rtgpu_scene rtgpu_scene;

/* ---- query results ------------------------------------------------------------------- */

/* The part of SceneRayHitInfo (scene/renderable/SceneRayHitInfo.h:9-21, math/RayHitInfo.h:13-59)
 * that identifies a hit.  prim is the triangle's index in the caller's order (the reference's
 * permutation[triangleIndex]), RTGPU_INVALID_ID for a sphere; inst is the index into
 * instances[]; a miss has inst == prim == RTGPU_INVALID_ID and t == +inf. */
typedef struct rtgpu_hit {
    float t;
    uint32_t prim;
    uint32_t inst;
    float beta;
    float gamma;
} rtgpu_hit;

/* ---- camera and render settings ------------------------------------------------------ */

/* The state of PerspectiveCamera (camera/PerspectiveCamera.h:22-24; private there, so the
 * host owns this POD).  u,v,w = OrthonormalBasis of the camera; width = 2 tan(fov_x / 2). */
typedef struct rtgpu_camera {
    float origin[3];
    float u[3];
    float v[3];
    float w[3];
    double width;
    double aperture;       /* 0 = pinhole; > 0: lens radius of the thin-lens model (camera/PerspectiveCamera.cpp:39-51,
                            * needs geometry_aa >= 2) */
    double focal_distance; /* distance along the pinhole ray at which lens rays converge */
} rtgpu_camera;

typedef enum rtgpu_jitter_mode {
    RTGPU_JITTER_RANDOM = 0, /* sampleUniformStratifiedSquare as the reference draws it (math/Sampler.h:13-26) */
    RTGPU_JITTER_CENTER = 1  /* deterministic centre of the stratum (pixel centre when geometry_aa < 4) */
} rtgpu_jitter_mode;

/* RenderSettings (renderer/Renderer.h:8-12) + the Tile (film/Tile.h) + FrameBuffer size. */
typedef struct rtgpu_render_settings {
    int32_t width, height;                       /* FrameBuffer resolution; width * height <= RTGPU_MAX_PIXELS */
    int32_t x_start, y_start, x_end, y_end;      /* half-open tile, film/Tile.h:19-23 */
    int32_t geometry_aa;                         /* geometryAAModifier (default 4) */
    int32_t material_aa;                         /* materialAAModifier (default 32); geometry_aa * material_aa, the paths of one
                                                  * pixel, may not exceed RTGPU_MAX_PATHS_PER_PIXEL (RTGPU_ERR_LIMIT) */
    int32_t max_path_length;                     /* reference hard-codes 10, renderer/Renderer.cpp:130 */
    uint32_t seed;                               /* key of the counter-based RNG */
    int32_t jitter_mode;                         /* rtgpu_jitter_mode */
    int32_t n_gpus;                              /* rtgpu_render only: 0 or 1 = the scene's device; G > 1 = the frame's tile is cut into
                                                  * squares shared out over G devices of this process and gathered with RCCL (what the
                                                  * reference does with one process per tile and --mergetiles, exe/Raytracer.cpp:260-277);
                                                  * one such call at a time per scene (a second waits); the gather is given up with
                                                  * RTGPU_ERR_COMM after RTGPU_COMM_TIMEOUT_MS (default 60 000) */
} rtgpu_render_settings;

/* A batch of paths holds whole pixels, and a path's index inside its batch travels in 25 bits of a queue entry. */
#define RTGPU_MAX_PATHS_PER_PIXEL (1 << 25)
/* width * height of a frame (RTGPU_ERR_LIMIT beyond): pixel indices, padded to whole squares when a frame is shared out
 * over devices, stay below 2^31, which the kernels' index arithmetic relies on. */
#define RTGPU_MAX_PIXELS (1 << 30)

typedef struct rtgpu_render_stats {
    uint64_t primary_rays;   /* Scene::traceRay calls from renderer/Renderer.cpp:161,170 */
    uint64_t extension_rays; /* renderer/Renderer.cpp:60, material/GlossyMaterial.cpp:124 */
    uint64_t shadow_rays;    /* material/NextEventEstimation.cpp:16,38 */
    double device_ms;        /* HIP-event time of the whole call on the device */
} rtgpu_render_stats;

/* ProgressMonitor, utility/ProgressMonitor.h:8 (job name, fraction done, finished flag). */
typedef void (*rtgpu_progress_cb)(const char* job, float fraction, int finished, void* user);

/* ---- entry points -------------------------------------------------------------------- */

const char* rtgpu_last_error(void);

/* Number of usable gfx950 devices (0 if none; never fails). */
int rtgpu_device_count(void);

/* Select the HIP device later calls on this thread use (default 0). */
int rtgpu_set_device(int device);

/* Replaces DynamicScene::build (scene/dynamic/DynamicScene.cpp:99-161): builds the
 * per-mesh and top-level SAH trees on the host (shape/bvh/BVHBuilder.h:144-361), flattens
 * them (shape/bvh/BVH.h:70-113) and uploads nodes, triangles, instances, materials and
 * lights to the current device. */
int rtgpu_scene_create(const rtgpu_scene_desc* desc, rtgpu_scene** out_scene);
void rtgpu_scene_destroy(rtgpu_scene* scene);

/* Scene::traceRay over a batch (scene/renderable/Scene.cpp:14-17).  origins/dirs: n*3 floats. */
int rtgpu_trace_closest(const rtgpu_scene* scene, const float* origins, const float* dirs,
                        uint64_t n, rtgpu_hit* out_hits);
/* Same with device-resident buffers, enqueued on hip_stream (a hipStream_t, NULL = default). */
int rtgpu_trace_closest_device(const rtgpu_scene* scene, const float* d_origins, const float* d_dirs,
                               uint64_t n, rtgpu_hit* d_out_hits, void* hip_stream);

/* Page-locked host memory for ray and result arrays (no counterpart in the reference, whose rays never leave the
 * process).  rtgpu_trace_closest / rtgpu_trace_any copy ordinary (pageable) arrays through staging buffers of their own;
 * when every array of a call lies in page-locked memory -- from here, hipHostMalloc or hipHostRegister -- the DMA engines
 * read and write the caller's arrays directly, which roughly doubles the rate of a batch that crosses PCIe.  The memory
 * is usable from every device and is ordinary memory to the host.  rtgpu_host_free(NULL) is a no-op. */
int rtgpu_host_alloc(void** out_ptr, uint64_t bytes);
int rtgpu_host_free(void* ptr);

/* Scene::testVisibility over a batch (scene/renderable/Scene.cpp:19-22): occluded[i] = 1 when
 * something is hit with t <= tmax[i]. */
int rtgpu_trace_any(const rtgpu_scene* scene, const float* origins, const float* dirs,
                    const float* tmax, uint64_t n, uint8_t* out_occluded);
int rtgpu_trace_any_device(const rtgpu_scene* scene, const float* d_origins, const float* d_dirs,
                           const float* d_tmax, uint64_t n, uint8_t* d_out_occluded, void* hip_stream);

/* PerspectiveCamera::generateRay for every pixel of the tile and one sample index
 * (camera/PerspectiveCamera.cpp:26-52).  Writes (x_end-x_start)*(y_end-y_start) rays,
 * row-major inside the tile, into device buffers of 3 floats per ray. */
int rtgpu_generate_primary_device(const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                                  int32_t sample_index, float* d_origins, float* d_dirs, void* hip_stream);

/* Renderer::render (renderer/Renderer.cpp:273-299).  rgb_out is the whole FrameBuffer
 * (width*height*3 floats, pixel (x,y) at (y*width+x)*3, film/FrameBuffer.h:14-22); only pixels
 * inside the tile are written.  stats and progress may be NULL. */
int rtgpu_render(const rtgpu_scene* scene, const rtgpu_camera* camera,
                 const rtgpu_render_settings* settings, float* rgb_out,
                 rtgpu_render_stats* stats, rtgpu_progress_cb progress, void* user);
/* Same, writing into a device-resident framebuffer on hip_stream; returns after the work is
 * complete (the integrator loop needs host-side queue sizes). */
int rtgpu_render_device(const rtgpu_scene* scene, const rtgpu_camera* camera,
                        const rtgpu_render_settings* settings, float* d_rgb_out,
                        rtgpu_render_stats* stats, void* hip_stream);

/* ---- one frame over several GPUs ------------------------------------------------------------
 *
 * The reference's scale-out is one process per tile (--xstart/--xend/--ystart/--yend, exe/Raytracer.cpp:65-68) whose
 * TILE files (io/TileFile.cpp:8-27) a last run pastes together (--mergetiles, exe/Raytracer.cpp:260-277,
 * io/TileFile.cpp:29-70).  Here the settings' tile is cut into tile_size x tile_size squares numbered row by row; square k
 * belongs to rank k % world; a rank's squares lie one after the other in its compact buffer (3 floats per pixel, each
 * square row by row, squares that stick out of the tile padded with zeros); the buffers meet on rank 0 through RCCL
 * send / receive pairs and a kernel writes them into FrameBuffer order.  A pixel's random numbers depend on its position in
 * the frame only: the frame is bit-identical for every world size.
 *
 * rtgpu_render with settings->n_gpus > 1 does all of this inside one process.  The entry points below are for hosts that
 * run one process per GPU: rank 0 makes an id, the host program hands it to the other processes, every process makes its
 * communicator on its own device (rtgpu_set_device first), renders its squares and joins the gather. */
#define RTGPU_COMM_ID_BYTES 128
typedef struct rtgpu_comm rtgpu_comm;
int rtgpu_comm_unique_id(uint8_t id[RTGPU_COMM_ID_BYTES]);
/* Blocks until all `world` ranks have called it with the same id. */
int rtgpu_comm_create(const uint8_t id[RTGPU_COMM_ID_BYTES], int rank, int world, rtgpu_comm** out);
void rtgpu_comm_destroy(rtgpu_comm* comm);
/* How long rtgpu_gather_tiles_device may wait for its transfers (default 60 000 ms, or RTGPU_COMM_TIMEOUT_MS).  The reference's
 * merge reads files and cannot hang (io/TileFile.cpp:29-70); a send or receive whose peer died after the communicator was
 * made would.  When the limit passes, the communicator is aborted (ncclCommAbort), the call returns RTGPU_ERR_COMM and the
 * handle refuses further use.  0: the gather is only enqueued, as a caller that waits on the stream itself may prefer. */
int rtgpu_comm_set_timeout_ms(rtgpu_comm* comm, int timeout_ms);
/* Floats in the compact buffer of `rank` (0 on bad arguments). */
uint64_t rtgpu_tile_buffer_floats(const rtgpu_render_settings* settings, int tile_size, int rank, int world);
/* Renderer::render for the squares of `rank` into its compact device buffer; returns when the work is complete. */
int rtgpu_render_tiles_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                              int tile_size, int rank, int world, float* d_tiles, rtgpu_render_stats* stats, void* hip_stream);
/* The gather on hip_stream: ranks > 0 send d_tiles, rank 0 receives and writes the tile's pixels of d_frame
 * (width * height * 3 floats on rank 0's device; ignored elsewhere).  Every rank of the communicator must call.  With more
 * than one rank the call returns when its transfers have completed or the communicator's time limit has passed (see
 * rtgpu_comm_set_timeout_ms). */
int rtgpu_gather_tiles_device(rtgpu_comm* comm, const rtgpu_render_settings* settings, int tile_size,
                              const float* d_tiles, float* d_frame, void* hip_stream);
/* The last step of the gather alone: d_all_tiles holds the `world` compact buffers one after the other, each at a multiple of
 * rtgpu_tile_buffer_floats(.., rank 0, world) floats (how rank 0 lays them out).  For checking the square order on one GPU. */
int rtgpu_scatter_tiles_device(const rtgpu_render_settings* settings, int tile_size, int world, const float* d_all_tiles,
                               float* d_frame, void* hip_stream);
/* A copy of a built scene on another device of this process (device-to-device; no rebuild). */
int rtgpu_scene_replicate(const rtgpu_scene* src, int device, rtgpu_scene** out);

/* Work counters of a render, for the roofline model of the integrator's traversal stages: the frame is rendered with every ray
 * traversed on its own by one lane (nearer child first, pruned by its own hits; the reference's counterpart is BVHDiag,
 * shape/bvh/BVHNode.h:11-15, summed into the --perffci image) and the child boxes / triangles tested are added up per class of
 * ray.  The image is the one rtgpu_render_device writes. */
typedef struct rtgpu_render_work {
    uint64_t extension_rays, extension_boxes, extension_triangles;
    uint64_t shadow_rays, shadow_boxes, shadow_triangles;
    uint64_t primary_rays, primary_boxes, primary_triangles;
} rtgpu_render_work;
int rtgpu_render_work_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                             float* d_rgb_out, rtgpu_render_work* work, void* hip_stream);

/* Device time of a render by stage (HIP events around every kernel launch, one batch of paths in flight at a time so that
 * launches do not overlap): what the reference prints as one number, "Rendering time" (exe/Raytracer.cpp:339-341), split the
 * way the wavefront pipeline is.  A stage's average launch duration is its ms / launches.  The image is the one
 * rtgpu_render_device writes; the call itself is slower than rtgpu_render_device (no overlap of batches). */
typedef struct rtgpu_render_profile {
    double primary_ms, shade_ms, trace_ms, gather_ms, resolve_ms;
    double total_ms;
    uint32_t primary_launches, shade_launches, trace_launches, gather_launches, resolve_launches;
    uint32_t reserved;
} rtgpu_render_profile;
int rtgpu_render_profile_device(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                                float* d_rgb_out, rtgpu_render_profile* profile, void* hip_stream);

/* rtgpu_render_device without any allocation.  No counterpart in the reference: it is for callers that cannot allocate
 * device memory while they run -- tight serving loops, harnesses that forbid hipMalloc, code between the begin and end of a
 * stream capture (every call it makes can be captured; replaying such a graph has not been made to work on ROCm 7.2 and is
 * not supported).
 * The integrator's workspace is a fixed array of RTGPU_STATIC_WORKSPACE_BYTES inside the library (each GPU the library
 * runs on holds it until the process ends; its address is looked up when a scene is made on the device), paths are rendered
 * in batches that fit it, all work is enqueued on hip_stream and the call returns without waiting (the caller synchronises;
 * no ray statistics).  Nothing is read back, so every batch enqueues all 2 * max_path_length + 2 passes over its full path
 * count.  Calls that overlap on one GPU share the array and race: one call at a time per GPU.  RTGPU_ERR_LIMIT when one
 * pixel's geometry_aa * material_aa paths (about 140 bytes each) do not fit the array; n_gpus must be 0 or 1.  The image is
 * bit-identical to rtgpu_render_device's. */
#define RTGPU_STATIC_WORKSPACE_BYTES (64u << 20)
int rtgpu_render_device_static(const rtgpu_scene* scene, const rtgpu_camera* camera, const rtgpu_render_settings* settings,
                               float* d_rgb_out, void* hip_stream);

/* PerspectiveCamera(fov) + setTransform (camera/PerspectiveCamera.cpp:7-19): fills *out from a
 * horizontal field of view in degrees and a row-major object->world 4x4. */
int rtgpu_camera_from_transform(double fov_x_degrees, const double transform[16], rtgpu_camera* out);

/* Tree statistics of a built scene (LOGSTAT keys of shape/bvh/BVHBuilder.h:329-337). */
typedef struct rtgpu_scene_info {
    uint64_t n_triangles;     /* over all meshes */
    uint64_t n_nodes;         /* flattened inner nodes over all trees */
    uint64_t n_leaves;
    uint32_t max_depth;
    uint32_t max_leaf_size;
    uint64_t device_bytes;    /* HBM held by the scene */
    double build_ms;
} rtgpu_scene_info;
int rtgpu_scene_get_info(const rtgpu_scene* scene, rtgpu_scene_info* out);

/* Per-ray traversal work counters, for the roofline model: d_counts receives 2 uint32 per ray
 * {child boxes tested, triangles tested} of the ray traversed on its own, nearer child first and
 * pruned by its own hits (the reference's counterpart is BVHDiag, shape/bvh/BVHNode.h:11-15).
 * Also writes the hits.  The cooperative kernel behind rtgpu_trace_closest_device tests
 * somewhat more boxes than this (subtrees taken over by idle lanes are entered early). */
int rtgpu_trace_closest_count_device(const rtgpu_scene* scene, const float* d_origins, const float* d_dirs,
                                     uint64_t n, uint32_t* d_counts, void* hip_stream);

/* ---- photon map ---------------------------------------------------------------------------- */

/* PhotonMapMode (scene/renderable/Scene.h) as the executable's --pmmode sets it.  "full": a diffuse vertex reads
 * the map instead of sampling once `depth` earlier vertices of its path were diffuse (specularity < 0.8),
 * material/DiffuseMaterial.cpp:39-80.  "caustics": only photons that arrive after a specular bounce are stored; a
 * diffuse bounce whose next vertex turns out specular ends the path with the map's estimate instead, and glass
 * behind a diffuse or rough bounce is black (DiffuseMaterial.cpp:104-135, GlassMaterial.cpp:31-38). */
typedef enum rtgpu_photon_map_mode {
    RTGPU_PM_NONE = 0,
    RTGPU_PM_CAUSTICS = 1,
    RTGPU_PM_FULL = 2
} rtgpu_photon_map_mode;

typedef struct rtgpu_photon { /* photonmapping/Photon.h:7-24 */
    float pos[3];
    float normal[3];  /* surfaceNormal: the hit's shading normal */
    float energy[3];  /* the photon's energy when it arrived */
    uint32_t is_caustic;
} rtgpu_photon;

/* PhotonMapBuilder::buildPhotonMap (photonmapping/PhotonMapBuilder.cpp:15-41): traces photons from every point
 * and area light (PhotonTracer::tracePhotons, photonmapping/PhotonTracer.cpp:8-172, at most 12 vertices each),
 * builds the search tree over them on the GPU's host and keeps both with the scene; later renders of the scene
 * use the map (Scene::getPhotonMap).  rays_per_*: the executable's --pmrayspointlamp / --pmraysarealamp.
 * Photons are kept in emission order (light, ray, vertex), so the map depends only on the arguments. */
int rtgpu_scene_build_photon_map(rtgpu_scene* scene, int32_t mode, int32_t depth, uint64_t rays_per_point_light,
                                 uint64_t rays_per_area_light, uint32_t seed, uint64_t* n_photons);
/* The executable's --savepm / --loadpm as memory transfers: copy the photons out (at most `capacity`; *n receives
 * the number the map holds) or install a photon list (mode NONE removes the map). */
int rtgpu_scene_get_photons(const rtgpu_scene* scene, rtgpu_photon* out, uint64_t capacity, uint64_t* n);
int rtgpu_scene_set_photons(rtgpu_scene* scene, const rtgpu_photon* photons, uint64_t n, int32_t mode, int32_t depth);

/* The radiance estimate's gather, KDTree::getElementsNearestTo as DiffuseMaterial calls it
 * (material/DiffuseMaterial.cpp:62-75): the 20 nearest photons whose surfaceNormal has a non-negative dot product
 * with `direction`.  count = photons found, max_dist = the search radius the reference ends with (distance of the
 * 20th photon; FLT_MAX when fewer than 20 but at least one were found; 1e9 when none), energy = their energies
 * summed nearest first.  Host arrays: points / directions n * 3 floats. */
typedef struct rtgpu_photon_gather {
    uint32_t count;
    float max_dist;
    float energy[3];
} rtgpu_photon_gather;
int rtgpu_scene_gather_photons(const rtgpu_scene* scene, const float* points, const float* directions, uint64_t n,
                               rtgpu_photon_gather* out);

/* ---- scene ingestion ------------------------------------------------------------------- */

/* A binary glTF file flattened into a scene description in host memory: loadGLTFScene (io/GLTF.cpp:682-705:
 * LoadBinaryFromFile, then nodes / primitives / materials / lights / cameras of the default scene) followed by
 * the flattening of DynamicScene::build (scene/dynamic/DynamicScene.cpp:108-145), or of soupifyScene (:17-106,
 * the reference's --soupify: one world-space mesh, per-triangle materials) when `soupify` is non-zero.
 * image_aspect_ratio = width / height of the frame, as loadGLTFScene takes it (the camera's x field of view
 * follows from the file's yfov).  PNG images only; no GPU is touched. */
typedef struct rtgpu_gltf rtgpu_gltf;
int rtgpu_gltf_load(const char* glb_path, float image_aspect_ratio, int soupify, rtgpu_gltf** out);
/* PathResolver::setWorkDir (utility/PathResolver.cpp:22-24, set from --workdir in exe/Raytracer.cpp:112): what "#WORKDIR#"
 * stands for in a path the file names (the scene's "EnvironmentHDRIFilePath").  Process-wide like the reference's
 * singleton; empty until set.  Call before rtgpu_gltf_load. */
void rtgpu_gltf_set_workdir(const char* directory);
/* The description to hand to rtgpu_scene_create; its arrays live until rtgpu_gltf_free. */
const rtgpu_scene_desc* rtgpu_gltf_scene_desc(const rtgpu_gltf* g);
/* Renderer::findCamera (renderer/Renderer.cpp:304-320): the last camera whose node (or parent) carries
 * IsMainCamera, else the first; RTGPU_ERR_NO_CAMERA without one.  Thin-lens parameters from the node's
 * FocalDistance / Aperture / FStop / SensorSize (io/GLTF.cpp:598-629). */
int rtgpu_gltf_camera(const rtgpu_gltf* g, rtgpu_camera* out);
uint32_t rtgpu_gltf_camera_count(const rtgpu_gltf* g);
void rtgpu_gltf_free(rtgpu_gltf* g);

/* ---- frame-buffer output ---------------------------------------------------------------- */

/* write_to_exr_file (io/EXRWriter.cpp:6-72): channels B, G, R as half floats (the executable's --outputtype exr,
 * exe/Raytracer.cpp:347-350) or full floats; uncompressed scan lines.  rgb = width * height * 3 floats, the
 * FrameBuffer layout. */
int rtgpu_write_exr(const char* path, const float* rgb, int32_t width, int32_t height, int full_floats);
/* write_to_png_file (io/PNG.cpp:7-28): pixel * 2^exposure, ^(1 / gamma), * 255 truncated and clamped to 255;
 * 8-bit RGB.  The executable's defaults are exposure -2.5, gamma 2.2 (exe/Raytracer.cpp:69-70). */
int rtgpu_write_png(const char* path, const float* rgb, int32_t width, int32_t height, double exposure, double gamma);

#ifdef __cplusplus
}
#endif
#endif /* RTGPU_H */
