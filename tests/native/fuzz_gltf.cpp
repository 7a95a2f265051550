// Host-only harness: the .glb loader (raytracer_amd/csrc/gltf_bridge.cpp, compiled in here) on randomly damaged
// copies of a valid file, for a build with -fsanitize=address,undefined.  Usage: fuzz_gltf <valid.glb> <scratch.glb> <n>
// or, to damage the Radiance .hdr file a scene names instead: fuzz_gltf <valid.hdr> <the .hdr path the .glb names> <n> <scene.glb>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "rt_scene.h"
namespace rt { int set_error(int code, const std::string&) { return code; } }
extern "C" int rtgpu_camera_from_transform(double, const double*, rtgpu_camera* out) { std::memset(out, 0, sizeof *out); return 0; }
void rtgpu_scene::release() {}
#include "gltf_bridge.cpp"
int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const std::string scratch = argv[2];
    const int iterations = std::atoi(argv[3]);
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::mt19937 rng(123);
    int ok = 0, err = 0;
    for (int k = 0; k < iterations; ++k) {
        std::vector<char> d = data;
        int n = 1 + rng() % 6;
        for (int i = 0; i < n; ++i) d[rng() % d.size()] = (char)(rng() & 255);
        if (k % 5 == 0) d.resize(12 + rng() % (d.size() - 12));
        std::ofstream o(scratch, std::ios::binary); o.write(d.data(), d.size()); o.close();
        rtgpu_gltf* g = nullptr;
        if (rtgpu_gltf_load(argc > 4 ? argv[4] : scratch.c_str(), 1.5f, k & 1, &g) == 0) { rtgpu_camera c; rtgpu_gltf_camera(g, &c); rtgpu_gltf_free(g); ++ok; } else ++err;
    }
    std::printf("ok %d err %d\n", ok, err);
}
