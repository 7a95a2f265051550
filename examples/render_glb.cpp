// render_glb.cpp -- the render branch of the reference's executable (raytracer/src/exe/Raytracer.cpp:52-341)
// over the GPU path: the same option names for what this library covers, loadGLTFScene + build replaced by
// rtgpu_gltf_load + rtgpu_scene_create, Renderer::render replaced by the adapter's GpuRenderer::render.
//
//   render_glb [--workdir DIR] --scene scene.glb --width 500 --height 500 [--xstart .. --ystart .. --xend .. --yend ..]
//              [--soupify] [--aageometry 4] [--aamaterial 4] [--seed 0] [--outputtype exr|png|ppm|tile] [--output output]
//              [--exposure -2.5] [--gamma 2.2] [--pmmode none|caustics|full] [--pmdepth 0]
//              [--pmrayspointlamp 10000000] [--pmraysarealamp 10000000] [--savepm] [--loadpm] [--pmfile FILE]
//   render_glb --width W --height H --mergetiles a.tile b.tile ... [--outputtype ..] [--output ..]
//
// Paths follow the executable: "#WORKDIR#" in a path is replaced by the work directory (utility/PathResolver.cpp),
// relative scene and output paths are taken from it, an output path without extension gets the output type as one
// (exe/Raytracer.cpp:104-191), missing output directories are created (:345).
//
// Output: "exr" (half floats, io/EXRWriter.cpp) and "png" (exposure + gamma, io/PNG.cpp) through the library's
// writers, "tile" = the TILE exchange file of io/TileFile.cpp:8-27, "ppm" = the text PPM of io/PPMFile.cpp:4-24.
// --mergetiles pastes TILE files into one frame instead of rendering (exe/Raytracer.cpp:260-277, io/TileFile.cpp:29-70).
// --savepm / --loadpm keep the photons in a file of this program's own (the reference dumps its tree nodes with their
// pointers, photonmapping/KDTree.h serialize: not a format another program can read); --preview and --perffci are
// accepted and ignored with a note (no window, no per-pixel cost image).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtgpu_adapter.hpp"

namespace {

// write_to_tile_file, io/TileFile.cpp:8-27
void writeTile(const std::vector<float>& pixels, int width, int height, const rtgpu::Tile& tile, const std::string& path)
{
    std::ofstream out(path, std::ios::out | std::ios::trunc | std::ios::binary);
    out.write("TILE", 4);
    const int res[2] = {width, height};
    out.write(reinterpret_cast<const char*>(res), sizeof(res));
    out.write(reinterpret_cast<const char*>(&tile), sizeof(tile));
    for (int y = tile.yStart; y < tile.yEnd; ++y)
        out.write(reinterpret_cast<const char*>(&pixels[3 * ((size_t)y * width + tile.xStart)]), sizeof(float) * 3 * (tile.xEnd - tile.xStart));
}

// load_from_tile_file, io/TileFile.cpp:29-70
void loadTile(std::vector<float>& pixels, int width, int height, const std::string& path)
{
    std::ifstream in(path, std::ios::in | std::ios::binary);
    char magic[4] = {0, 0, 0, 0};
    in.read(magic, sizeof(magic));
    if (!in || std::string(magic, 4) != "TILE") throw std::runtime_error("Bad file");
    int res[2];
    in.read(reinterpret_cast<char*>(res), sizeof(res));
    if (!in || res[0] != width || res[1] != height) throw std::runtime_error("Resolution mismatch");
    rtgpu::Tile t;
    in.read(reinterpret_cast<char*>(&t), sizeof(t));
    if (!in || t.xStart > t.xEnd || t.yStart > t.yEnd || t.xStart < 0 || t.xEnd < 0 || t.yStart < 0 || t.yEnd < 0 || t.xEnd > width || t.yEnd > height)
        throw std::runtime_error("Invalid tile geometry specification");
    for (int y = t.yStart; y < t.yEnd; ++y) {
        in.read(reinterpret_cast<char*>(&pixels[3 * ((size_t)y * width + t.xStart)]), sizeof(float) * 3 * (t.xEnd - t.xStart));
        if (!in) throw std::runtime_error("Bad file");
    }
}

// write_to_ppm, io/PPMFile.cpp:4-24
void writePpm(const std::vector<float>& pixels, int width, int height, const std::string& path)
{
    std::ofstream out(path, std::ios::out | std::ios::trunc);
    out << "P3" << std::endl << width << " " << height << std::endl << 255 << std::endl;
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float* p = &pixels[3 * ((size_t)y * width + x)];
            out << static_cast<int>(p[0] * 255) << " " << static_cast<int>(p[1] * 255) << " " << static_cast<int>(p[2] * 255) << " ";
        }
        out << std::endl;
    }
}

// PathResolver::resolve, utility/PathResolver.cpp:3-20
std::string resolvePath(const std::string& in, const std::string& workdir)
{
    std::string str = in;
    size_t pos;
    while ((pos = str.find("#WORKDIR#")) != std::string::npos) str.replace(pos, 9, workdir);
    return str;
}

// The photons of a map, in emission order: "RTGPUPM1", count, records of rtgpu_photon.
void savePhotons(const std::vector<rtgpu_photon>& photons, const std::string& path)
{
    std::ofstream out(path, std::ios::out | std::ios::trunc | std::ios::binary);
    const uint64_t n = photons.size();
    out.write("RTGPUPM1", 8);
    out.write(reinterpret_cast<const char*>(&n), sizeof(n));
    out.write(reinterpret_cast<const char*>(photons.data()), (std::streamsize)(n * sizeof(rtgpu_photon)));
    if (!out) throw std::runtime_error("Could not write the photon map file.");
}

std::vector<rtgpu_photon> loadPhotons(const std::string& path)
{
    std::ifstream in(path, std::ios::in | std::ios::binary);
    char magic[8] = {};
    uint64_t n = 0;
    in.read(magic, 8);
    in.read(reinterpret_cast<char*>(&n), sizeof(n));
    if (!in || std::memcmp(magic, "RTGPUPM1", 8) != 0) throw std::runtime_error("Not a photon map file of this program.");
    const uint64_t left = (uint64_t)std::filesystem::file_size(path) - 16u;
    if (n > left / sizeof(rtgpu_photon)) throw std::runtime_error("Photon map file is truncated.");
    std::vector<rtgpu_photon> photons(n);
    in.read(reinterpret_cast<char*>(photons.data()), (std::streamsize)(n * sizeof(rtgpu_photon)));
    if (!in) throw std::runtime_error("Photon map file is truncated.");
    return photons;
}

// unsigned long options: plain integers, as the executable takes them; "1E7" (its documented default) is read too
unsigned long long parseCount(const std::string& v)
{
    char* end = nullptr;
    const unsigned long long n = std::strtoull(v.c_str(), &end, 10);
    if (end && *end == '\0') return n;
    const double d = std::strtod(v.c_str(), &end);
    return (end && *end == '\0' && d >= 0 && d < 1.8e19) ? (unsigned long long)d : 0ull;
}

} // namespace

int main(int argc, char** argv)
{
    namespace fs = std::filesystem;
    std::map<std::string, std::string> opt = {{"workdir", fs::current_path().string()}, {"scene", "scene.glb"}, {"width", "500"}, {"height", "500"}, {"xstart", "0"}, {"ystart", "0"},
                                              {"outputtype", "exr"}, {"output", "output"}, {"exposure", "-2.5"}, {"gamma", "2.2"}, {"aageometry", "4"}, {"aamaterial", "4"}, {"seed", "0"},
                                              {"pmmode", "none"}, {"pmdepth", "0"}, {"pmfile", ""}, {"pmrayspointlamp", "10000000"}, {"pmraysarealamp", "10000000"}, {"perffci", ""},
                                              {"gpus", "1"}};
    bool soupify = false, savePm = false, loadPm = false;
    std::vector<std::string> tileFilesToMerge;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0) { std::fprintf(stderr, "unexpected argument %s\n", argv[i]); return -1; }
        a = a.substr(2);
        if (a == "soupify") { soupify = true; continue; }
        if (a == "savepm") { savePm = true; continue; }
        if (a == "loadpm") { loadPm = true; continue; }
        if (a == "preview") { std::fprintf(stderr, "--preview: no preview window in this program\n"); continue; }
        if (a == "mergetiles") {   // multitoken: every following word up to the next option
            while (i + 1 < argc && std::strncmp(argv[i + 1], "--", 2) != 0) tileFilesToMerge.push_back(argv[++i]);
            continue;
        }
        if (a == "help") { std::printf("options: --workdir --scene --width --height --xstart --ystart --xend --yend --soupify --aageometry --aamaterial --seed --outputtype --output --exposure --gamma --mergetiles --pmmode --pmdepth --pmrayspointlamp --pmraysarealamp --savepm --loadpm --pmfile --gpus\n"); return 0; }
        if (opt.find(a) == opt.end() && a != "xend" && a != "yend") { std::fprintf(stderr, "unrecognised option '--%s'\n", a.c_str()); return -1; }
        if (i + 1 >= argc) { std::fprintf(stderr, "option --%s needs a value\n", a.c_str()); return -1; }
        opt[a] = argv[++i];
    }
    // exe/Raytracer.cpp:104-112
    const std::string workDir = opt["workdir"];
    if (!fs::is_directory(workDir)) { std::fprintf(stderr, "The specified work directory does not exist or is not a folder.\n"); return -1; }
    const int width = std::atoi(opt["width"].c_str()), height = std::atoi(opt["height"].c_str());
    if (width <= 0 || height <= 0) { std::fprintf(stderr, "Invalid frame size.\n"); return -1; }
    const int xstart = std::atoi(opt["xstart"].c_str()), ystart = std::atoi(opt["ystart"].c_str());
    const int xend = opt.count("xend") ? std::atoi(opt["xend"].c_str()) : (width - xstart);   // exe/Raytracer.cpp:119-120
    const int yend = opt.count("yend") ? std::atoi(opt["yend"].c_str()) : (height - ystart);
    if (xstart >= xend || xstart >= width || xend > width || xstart < 0 || xend < 0) { std::fprintf(stderr, "Invalid xstart or xend.\n"); return -1; }
    if (ystart >= yend || ystart >= height || yend > height || ystart < 0 || yend <= 0) { std::fprintf(stderr, "Invalid ystart or yend.\n"); return -1; }
    const rtgpu::Tile tile{xstart, ystart, xend, yend};
    rtgpu::RenderSettings settings;
    settings.geometryAAModifier = std::atoi(opt["aageometry"].c_str());
    settings.materialAAModifier = std::atoi(opt["aamaterial"].c_str());
    settings.seed = (uint32_t)std::strtoul(opt["seed"].c_str(), nullptr, 10);
    // not an option of the reference (which runs one process per tile and merges TILE files): the frame's tile is shared out
    // over that many GPUs inside rtgpu_render and gathered with RCCL
    settings.gpus = std::atoi(opt["gpus"].c_str());
    if (settings.gpus < 1 || (settings.gpus > 1 && settings.gpus > rtgpu_device_count())) { std::fprintf(stderr, "Invalid number of GPUs (%d present).\n", rtgpu_device_count()); return -1; }
    const std::string type = opt["outputtype"];
    if (type != "exr" && type != "png" && type != "ppm" && type != "tile") { std::fprintf(stderr, "Invalid output type: %s\n", type.c_str()); return -1; }
    // exe/Raytracer.cpp:142-155
    fs::path outputfile = resolvePath(opt["output"], workDir);
    if (outputfile.is_relative()) outputfile = fs::path(workDir) / outputfile;
    if (!outputfile.has_filename()) outputfile.replace_filename("output");
    if (!outputfile.has_extension()) outputfile.replace_extension(type);
    if (!opt["perffci"].empty()) std::fprintf(stderr, "--perffci: no per-pixel cost image in this program\n");
    // exe/Raytracer.cpp:182-191
    fs::path sceneFile = resolvePath(opt["scene"], workDir);
    if (sceneFile.is_relative()) sceneFile = fs::path(workDir) / sceneFile;
    if (tileFilesToMerge.empty() && !fs::is_regular_file(sceneFile)) {
        std::fprintf(stderr, "The specified scene file does not exist or is not a file. (%s)\n", sceneFile.string().c_str());
        return -1;
    }
    // exe/Raytracer.cpp:193-242
    fs::path photonMapFilePath;
    if (loadPm || savePm) {
        photonMapFilePath = resolvePath(opt["pmfile"], workDir);
        if (loadPm && !fs::is_regular_file(photonMapFilePath)) { std::fprintf(stderr, "The specified photon map file does not exist or is not a file.\n"); return -1; }
    }
    const std::string pm = opt["pmmode"];
    int pmMode;
    if (pm == "none") pmMode = RTGPU_PM_NONE;
    else if (pm == "caustics") pmMode = RTGPU_PM_CAUSTICS;
    else if (pm == "full") pmMode = RTGPU_PM_FULL;
    else { std::fprintf(stderr, "Invalid photonmapping mode!\n"); return -1; }
    if (settings.geometryAAModifier <= 0 || settings.materialAAModifier <= 0) { std::fprintf(stderr, "Invalid AA settings!\n"); return -1; }
    const unsigned long long pmPoint = parseCount(opt["pmrayspointlamp"]), pmArea = parseCount(opt["pmraysarealamp"]);
    if (pmPoint == 0 || pmArea == 0) { std::fprintf(stderr, "Invalid PM raycount settings!\n"); return -1; }
    const int pmDepth = std::atoi(opt["pmdepth"].c_str());

    std::vector<float> pixels((size_t)width * height * 3, 0.0f);
    rtgpu_gltf* file = nullptr;
    try {
        if (!tileFilesToMerge.empty()) {   // exe/Raytracer.cpp:260-277
            std::printf("Merging tile files...\n");
            for (const std::string& f : tileFilesToMerge) {
                const std::string filePath = resolvePath(f, workDir);
                std::printf("%s\n", filePath.c_str());
                loadTile(pixels, width, height, filePath);
            }
        } else {
            std::printf("Loading scene.\n");
            rtgpu_gltf_set_workdir(workDir.c_str());   // PathResolver::get().setWorkDir(workDir), exe/Raytracer.cpp:112
            rtgpu::check(rtgpu_gltf_load(sceneFile.string().c_str(), static_cast<float>(width) / height, soupify ? 1 : 0, &file));
            rtgpu_camera camera;
            rtgpu::check(rtgpu_gltf_camera(file, &camera));          // Renderer::findCamera: "No camera in scene."
            rtgpu::GpuScene scene(*rtgpu_gltf_scene_desc(file));     // DynamicScene::build
            rtgpu_gltf_free(file);
            file = nullptr;
            if (pmMode != RTGPU_PM_NONE) {   // exe/Raytracer.cpp:296-322
                if (loadPm) {
                    std::printf("Loading photon map...\n");
                    const std::vector<rtgpu_photon> photons = loadPhotons(photonMapFilePath.string());
                    scene.setPhotons(photons, (rtgpu_photon_map_mode)pmMode, pmDepth);
                    std::printf("Loaded %llu photons\n", (unsigned long long)photons.size());
                } else {
                    std::printf("Building photon map...\n");
                    const uint64_t nPhotons = scene.buildPhotonMap((rtgpu_photon_map_mode)pmMode, pmDepth, pmPoint, pmArea, settings.seed);
                    std::printf("Building photon KD-tree (%llu photons)\n", (unsigned long long)nPhotons);
                }
                if (savePm) {
                    std::printf("Saving photon map...\n");
                    savePhotons(scene.photons(), photonMapFilePath.string());
                }
            }
            std::printf("Rendering...\n");
            std::printf("Geometry AA level = %d\nMaterial AA level = %d\n", settings.geometryAAModifier, settings.materialAAModifier);
            rtgpu::GpuRenderer renderer(scene, camera);
            const rtgpu_render_stats stats = renderer.render(pixels.data(), width, height, tile, settings,
                [](const std::string& job, float progress, bool done) {
                    std::printf("%s - %g%%\r", job.c_str(), progress * 100);
                    if (done) std::printf("\n");
                    std::fflush(stdout);
                });
            std::printf("Done\n");
            std::printf("Rendering time: %g milliseconds\n", stats.device_ms);
            std::printf("rays: primary %llu extension %llu shadow %llu\n", (unsigned long long)stats.primary_rays,
                        (unsigned long long)stats.extension_rays, (unsigned long long)stats.shadow_rays);
        }
        // exe/Raytracer.cpp:344-365
        if (outputfile.has_parent_path()) fs::create_directories(outputfile.parent_path());
        const std::string outFile = outputfile.string();
        std::printf("Writing output to %s\n", outFile.c_str());
        if (type == "exr") rtgpu::check(rtgpu_write_exr(outFile.c_str(), pixels.data(), width, height, 0));
        else if (type == "png") rtgpu::check(rtgpu_write_png(outFile.c_str(), pixels.data(), width, height, std::atof(opt["exposure"].c_str()), std::atof(opt["gamma"].c_str())));
        else if (type == "tile") writeTile(pixels, width, height, tile, outFile);
        else writePpm(pixels, width, height, outFile);
    } catch (const std::exception& e) {
        if (file) rtgpu_gltf_free(file);
        std::fprintf(stderr, "%s\n", e.what());
        return -1;
    }
    return 0;
}
