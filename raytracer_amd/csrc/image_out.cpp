// image_out.cpp -- the frame-buffer writers of the reference's executable that need an encoder:
// write_to_exr_file (io/EXRWriter.cpp:6-72: channels B, G, R, half or full floats) and write_to_png_file
// (io/PNG.cpp:7-28: exposure, gamma, 8-bit RGB).  The reference encodes with tinyexr and lodepng (vendored
// third-party code); here an uncompressed scan-line OpenEXR file and a zlib-deflated PNG are written directly.
// Host code only.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_scene.h"

namespace {

// IEEE float -> half, round to nearest even, overflow to infinity
uint16_t float_to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mant = x & 0x007FFFFFu;
    const int exp = (int)((x >> 23) & 0xFFu);
    if (exp == 255) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0u));
    const int e = exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        const uint32_t m = mant | 0x00800000u;
        const int shift = 14 - e;
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h; // may carry into the exponent, which is the right result
    return (uint16_t)(sign | h);
}

void put(std::vector<uint8_t>& out, const void* p, size_t n) { out.insert(out.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
void put_str(std::vector<uint8_t>& out, const char* s) { put(out, s, std::strlen(s) + 1); }
void put_i32(std::vector<uint8_t>& out, int32_t v) { put(out, &v, 4); }
void put_f32(std::vector<uint8_t>& out, float v) { put(out, &v, 4); }
void attr(std::vector<uint8_t>& out, const char* name, const char* type, const std::vector<uint8_t>& body)
{
    put_str(out, name);
    put_str(out, type);
    put_i32(out, (int32_t)body.size());
    put(out, body.data(), body.size());
}

bool write_all(const char* path, const std::vector<uint8_t>& data)
{
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data.data(), 1, data.size(), f) == data.size();
    return (std::fclose(f) == 0) && ok;
}

void be32(std::vector<uint8_t>& out, uint32_t v)
{
    const uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
    put(out, b, 4);
}
void png_chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& body)
{
    be32(out, (uint32_t)body.size());
    const size_t start = out.size();
    put(out, tag, 4);
    put(out, body.data(), body.size());
    be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

} // namespace

extern "C" {

int rtgpu_write_exr(const char* path, const float* rgb, int32_t width, int32_t height, int full_floats)
{
    if (!path || !rgb || width <= 0 || height <= 0) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "bad image arguments");
    return rt::guarded([&]() -> int {   // the frame-sized vectors below can fail to allocate
    std::vector<uint8_t> out;
    const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    put(out, magic, 8);
    {
        std::vector<uint8_t> ch;
        for (const char* name : {"B", "G", "R"}) { // io/EXRWriter.cpp:46-50
            put_str(ch, name);
            put_i32(ch, full_floats ? 2 : 1);       // FLOAT : HALF
            const uint8_t lin[4] = {0, 0, 0, 0};
            put(ch, lin, 4);
            put_i32(ch, 1);
            put_i32(ch, 1);
        }
        ch.push_back(0);
        attr(out, "channels", "chlist", ch);
    }
    attr(out, "compression", "compression", {0});
    std::vector<uint8_t> win;
    put_i32(win, 0); put_i32(win, 0); put_i32(win, width - 1); put_i32(win, height - 1);
    attr(out, "dataWindow", "box2i", win);
    attr(out, "displayWindow", "box2i", win);
    attr(out, "lineOrder", "lineOrder", {0});
    std::vector<uint8_t> one, zero2;
    put_f32(one, 1.0f);
    put_f32(zero2, 0.0f); put_f32(zero2, 0.0f);
    attr(out, "pixelAspectRatio", "float", one);
    attr(out, "screenWindowCenter", "v2f", zero2);
    attr(out, "screenWindowWidth", "float", one);
    out.push_back(0);
    const size_t sample = full_floats ? 4 : 2, lineBytes = 3 * sample * (size_t)width;
    uint64_t offset = out.size() + 8ull * (uint64_t)height;
    for (int y = 0; y < height; ++y) { put(out, &offset, 8); offset += 8 + lineBytes; }
    for (int y = 0; y < height; ++y) {
        put_i32(out, y);
        put_i32(out, (int32_t)lineBytes);
        for (int c = 2; c >= 0; --c) // B, G, R planes of the line
            for (int x = 0; x < width; ++x) {
                const float v = rgb[3 * ((size_t)y * width + x) + c];
                if (full_floats) put_f32(out, v);
                else { const uint16_t h = float_to_half(v); put(out, &h, 2); }
            }
    }
    if (!write_all(path, out)) return rt::set_error(RTGPU_ERR_IO, std::string("cannot write ") + path);
    return RTGPU_OK;
    });
}

int rtgpu_write_png(const char* path, const float* rgb, int32_t width, int32_t height, double exposure, double gamma)
{
    if (!path || !rgb || width <= 0 || height <= 0) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "bad image arguments");
    return rt::guarded([&]() -> int {
    const double exposureFactor = std::pow(2, exposure);
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (int y = 0; y < height; ++y) {
        raw.push_back(0); // filter: none
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) {
                // (pixel * exposureFactor).pow(1 / gamma): RGB * double narrows per component, RGB::pow(double) (film/RGB.h:85-88, 132-142)
                const float scaled = (float)(rgb[3 * ((size_t)y * width + x) + c] * exposureFactor);
                const float toned = (float)std::pow((double)scaled, 1 / gamma);
                raw.push_back((uint8_t)std::max(0, std::min(255, static_cast<int>(toned * 255))));
            }
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(bound);
    if (compress2(comp.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) return rt::set_error(RTGPU_ERR_IO, "deflate failed");
    comp.resize(bound);
    std::vector<uint8_t> out;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    put(out, sig, 8);
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width);
    be32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0}; // 8 bit, RGB (LCT_RGB)
    put(ihdr, tail, 5);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", comp);
    png_chunk(out, "IEND", {});
    if (!write_all(path, out)) return rt::set_error(RTGPU_ERR_IO, std::string("cannot write ") + path);
    return RTGPU_OK;
    });
}

} // extern "C"
