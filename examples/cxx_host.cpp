// cxx_host.cpp -- a host program in the reference's language that drives the GPU path through
// include/rtgpu_adapter.hpp the way raytracer/src/exe/Raytracer.cpp:325-341 drives Renderer::render
// and raytracer/test/BVH.cpp:18-49 drives Scene::traceRay: build a scene description, trace a few
// rays, render a tile, write it as a TILE file (io/TileFile.cpp:8-27).
//
//   cxx_host <output.tile> [width height geometryAA materialAA seed]
//
// The scene is small and fixed so that tests can rebuild it elsewhere and compare bit for bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "rtgpu_adapter.hpp"

namespace {

void identity(double m[16])
{
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
void translation(double m[16], double x, double y, double z)
{
    identity(m);
    m[3] = x; m[7] = y; m[11] = z;
}

// write_to_tile_file, io/TileFile.cpp:8-27
void writeTile(const std::vector<float>& pixels, int width, int height, const rtgpu::Tile& tile, const char* path)
{
    std::ofstream out(path, std::ios::out | std::ios::trunc | std::ios::binary);
    out.write("TILE", 4);
    const int res[2] = {width, height};
    out.write(reinterpret_cast<const char*>(res), sizeof(res));
    out.write(reinterpret_cast<const char*>(&tile), sizeof(tile));
    for (int y = tile.yStart; y < tile.yEnd; ++y)
        out.write(reinterpret_cast<const char*>(&pixels[3 * ((size_t)y * width + tile.xStart)]), sizeof(float) * 3 * (tile.xEnd - tile.xStart));
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <output.tile> [width height geometryAA materialAA seed]\n", argv[0]); return -1; }
    const int width = argc > 3 ? std::atoi(argv[2]) : 64, height = argc > 3 ? std::atoi(argv[3]) : 48;
    rtgpu::RenderSettings settings;
    settings.geometryAAModifier = argc > 4 ? std::atoi(argv[4]) : 4;
    settings.materialAAModifier = argc > 5 ? std::atoi(argv[5]) : 4;
    settings.seed = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 7u;
    try {
        // ---- geometry: a floor, a back wall and a pyramid in one mesh with per-triangle materials ----
        const float vertices[] = {
            -4, 0, -4,   -4, 0, 4,   4, 0, 4,   4, 0, -4,        // floor
            -4, 0, -4,   4, 0, -4,   4, 5, -4,  -4, 5, -4,       // back wall
            -1, 0, 1.5f,   1, 0, 1.5f,   1, 0, 3.5f,   -1, 0, 3.5f,   0, 1.5f, 2.5f, // pyramid in front of the spheres
        };
        const float normals[] = {
            0, 1, 0,  0, 1, 0,  0, 1, 0,  0, 1, 0,
            0, 0, 1,  0, 0, 1,  0, 0, 1,  0, 0, 1,
            -0.5f, 0.5f, -0.5f,  0.5f, 0.5f, -0.5f,  0.5f, 0.5f, 0.5f,  -0.5f, 0.5f, 0.5f,  0, 1, 0,
        };
        const uint32_t indices[] = {0, 1, 2,  0, 2, 3,  4, 5, 6,  4, 6, 7,  8, 12, 9,  9, 12, 10,  10, 12, 11,  11, 12, 8};
        const uint32_t triMaterial[] = {0, 0, 1, 1, 2, 2, 2, 2};
        rtgpu_mesh_desc mesh{};
        mesh.vertices = vertices; mesh.normals = normals; mesh.indices = indices; mesh.tri_material = triMaterial;
        mesh.n_vertices = 13; mesh.n_triangles = 8;

        rtgpu_material materials[4] = {};
        materials[0].type = RTGPU_MAT_DIFFUSE; materials[0].color[0] = 0.7f; materials[0].color[1] = 0.7f; materials[0].color[2] = 0.65f; materials[0].intensity = 1.0;
        materials[1].type = RTGPU_MAT_DIFFUSE; materials[1].color[0] = 0.25f; materials[1].color[1] = 0.35f; materials[1].color[2] = 0.7f; materials[1].intensity = 1.0;
        materials[2].type = RTGPU_MAT_GLOSSY; materials[2].color[0] = 0.9f; materials[2].color[1] = 0.85f; materials[2].color[2] = 0.8f; materials[2].intensity = 1.0; materials[2].roughness = 0.3f;
        materials[3].type = RTGPU_MAT_DIFFUSE; materials[3].color[0] = 0.8f; materials[3].color[1] = 0.3f; materials[3].color[2] = 0.2f; materials[3].intensity = 1.0;

        rtgpu_instance_desc instances[3] = {};
        instances[0].shape = 0; instances[0].material = 0;
        identity(instances[0].transform); identity(instances[0].inverse);
        for (int k = 0; k < 2; ++k) {                                 // the two unit spheres of test/BVH.cpp:18-49, lifted onto the floor
            const double x = k == 0 ? -2.0 : 2.0;
            instances[1 + k].shape = RTGPU_SHAPE_SPHERE; instances[1 + k].material = 3;
            translation(instances[1 + k].transform, x, 1.0, 0.0);
            translation(instances[1 + k].inverse, -x, -1.0, 0.0);
        }
        rtgpu_point_light light{};
        light.pos[0] = 1.0f; light.pos[1] = 4.0f; light.pos[2] = 3.0f;
        light.color[0] = light.color[1] = light.color[2] = 1.0f;
        light.intensity = 120.0;

        rtgpu_scene_desc desc{};
        desc.meshes = &mesh; desc.n_meshes = 1;
        desc.instances = instances; desc.n_instances = 3;
        desc.materials = materials; desc.n_materials = 4;
        desc.point_lights = &light; desc.n_point_lights = 1;
        rtgpu::GpuScene scene(desc);

        // ---- Scene::traceRay / testVisibility, as in test/BVH.cpp ----
        const float origin[3] = {0, 1, 0}, left[3] = {-1, 0, 0}, right[3] = {1, 0, 0}, up[3] = {0, 1, 0};
        const auto hitLeft = scene.traceRay(origin, left), hitRight = scene.traceRay(origin, right), hitUp = scene.traceRay(origin, up);
        std::printf("traceRay -x: inst %u t %.9g\n", hitLeft ? hitLeft->inst : 0xFFFFFFFFu, hitLeft ? hitLeft->t : -1.0f);
        std::printf("traceRay +x: inst %u t %.9g\n", hitRight ? hitRight->inst : 0xFFFFFFFFu, hitRight ? hitRight->t : -1.0f);
        std::printf("traceRay +y: %s\n", hitUp ? "hit" : "miss");
        std::printf("testVisibility -x within 0.5: %d, within 2: %d\n", (int)scene.testVisibility(origin, left, 0.5f), (int)scene.testVisibility(origin, left, 2.0f));

        // ---- a batch of rays kept in page-locked arrays (rtgpu::HostArray): the same answers without staging copies ----
        {
            const uint64_t n = 1000;
            rtgpu::HostArray<float> o(3 * n), d(3 * n);
            rtgpu::HostArray<rtgpu_hit> hits(n);
            for (uint64_t i = 0; i < n; ++i) {
                o[3 * i] = 0; o[3 * i + 1] = 1; o[3 * i + 2] = 0;
                d[3 * i] = (i & 1) ? 1.0f : -1.0f; d[3 * i + 1] = 0; d[3 * i + 2] = 0;
            }
            scene.traceRays(o.data(), d.data(), n, hits.data());
            uint64_t ok = 0;
            for (uint64_t i = 0; i < n; ++i) ok += hits[i].inst == ((i & 1) ? 2u : 1u) && hits[i].t == 1.0f;
            std::printf("traceRays on page-locked arrays: %llu of %llu as expected\n", (unsigned long long)ok, (unsigned long long)n);
        }

        // ---- Renderer::render ----
        const double cameraTransform[16] = {1, 0, 0, 0.0,  0, 1, 0, 2.0,  0, 0, 1, 7.0,  0, 0, 0, 1};  // looks down -z from (0, 2, 7)
        rtgpu_camera camera;
        rtgpu::check(rtgpu_camera_from_transform(55.0, cameraTransform, &camera));
        std::vector<float> pixels((size_t)width * height * 3, 0.0f);
        const rtgpu::Tile tile{0, 0, width, height};
        int calls = 0;
        rtgpu::GpuRenderer renderer(scene, camera);
        const rtgpu_render_stats stats = renderer.render(pixels.data(), width, height, tile, settings,
                                                         [&](const std::string&, float, bool) { ++calls; });
        std::printf("render %dx%d: %llu primary, %llu extension, %llu shadow rays, %.2f ms on the device, %d progress calls\n", width, height,
                    (unsigned long long)stats.primary_rays, (unsigned long long)stats.extension_rays, (unsigned long long)stats.shadow_rays,
                    stats.device_ms, calls);
        writeTile(pixels, width, height, tile, argv[1]);

        // ---- errors arrive as the reference's exception types ----
        try {
            rtgpu_scene_desc bad = desc;
            bad.n_materials = 0;
            rtgpu::GpuScene s(bad);
            std::printf("error case: not rejected\n");
            return 2;
        } catch (const std::invalid_argument& e) {
            std::printf("invalid_argument: %s\n", e.what());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
