// gltf_bridge.cpp -- binary glTF (.glb) -> rtgpu_scene_desc, the scene-ingestion side of the drop-in.
//
// Restates loadGLTFScene (io/GLTF.cpp:655-705: nodes, primitives, the material tree of loadMaterial
// :323-455, lights and camera parameters from node "extras", the environment from scene "extras")
// and the flattening of DynamicScene::build / soupifyScene (scene/dynamic/DynamicScene.cpp:17-160).
// The reference parses with tinygltf + stb_image (third-party, vendored there); here the container,
// the JSON, PNG images (zlib) and Radiance .hdr environment maps are read directly.  JPEG images and
// "VideoImageMapping" materials have no decoder here and are reported as unsupported.
// Host code only: nothing in this file touches the GPU.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rt_scene.h"

namespace {

struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- JSON ------------------------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Int, Real, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const std::string& key) const
    {
        if (kind != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }
    bool is_number() const { return kind == Int || kind == Real; }
    int as_int(int dflt) const { return is_number() ? (int)num : dflt; }
    int get_int(const std::string& key, int dflt) const
    {
        const Json* v = find(key);
        return v ? v->as_int(dflt) : dflt;
    }
    double get_num(const std::string& key, double dflt) const
    {
        const Json* v = find(key);
        return v && v->is_number() ? v->num : dflt;
    }
    std::string get_str(const std::string& key, const std::string& dflt = "") const
    {
        const Json* v = find(key);
        return v && v->kind == String ? v->str : dflt;
    }
    size_t size() const { return kind == Array ? arr.size() : 0; }
};

struct JsonParser {
    const char* p;
    const char* end;
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    [[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("glTF JSON: ") + what); }
    Json value(int depth)
    {
        if (depth > 64) bad("nesting too deep");
        ws();
        if (p >= end) bad("unexpected end");
        Json j;
        if (*p == '{') {
            ++p;
            j.kind = Json::Object;
            ws();
            if (p < end && *p == '}') { ++p; return j; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') bad("expected a key");
                std::string key = string();
                ws();
                if (p >= end || *p != ':') bad("expected ':'");
                ++p;
                j.obj.emplace_back(std::move(key), value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return j; }
                bad("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            ++p;
            j.kind = Json::Array;
            ws();
            if (p < end && *p == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return j; }
                bad("expected ',' or ']'");
            }
        }
        if (*p == '"') { j.kind = Json::String; j.str = string(); return j; }
        if (end - p >= 4 && !std::strncmp(p, "true", 4)) { p += 4; j.kind = Json::Bool; j.b = true; return j; }
        if (end - p >= 5 && !std::strncmp(p, "false", 5)) { p += 5; j.kind = Json::Bool; j.b = false; return j; }
        if (end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; return j; }
        // number: an integer token stays an Int (the loader's extras treat ints and reals differently, io/GLTF.cpp:143-170)
        const char* s = p;
        bool real = false;
        if (p < end && (*p == '-' || *p == '+')) ++p;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
            if (*p == '.' || *p == 'e' || *p == 'E') real = true;
            ++p;
        }
        if (p == s) bad("unexpected character");
        j.kind = real ? Json::Real : Json::Int;
        j.num = std::strtod(std::string(s, p).c_str(), nullptr);
        return j;
    }
    std::string string()
    {
        ++p; // opening quote
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) bad("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (end - p < 5) bad("bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                    break;
                }
                default: out += *p; // \" \\ \/
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) bad("unterminated string");
        ++p;
        return out;
    }
};

// ---- extras helpers, io/GLTF.cpp:143-200 ------------------------------------------------------
bool extras_bool(const Json* extras, const char* name, bool dflt = false)
{
    const Json* v = extras ? extras->find(name) : nullptr;
    if (!v) return dflt;
    if (v->kind == Json::Bool) return v->b;
    if (v->kind == Json::Int) return (int)v->num != 0;
    if (v->kind == Json::Real) return v->num > 0;
    return dflt;
}
double as_double(const Json& v, double dflt)
{
    if (v.kind == Json::Bool) return v.b ? 1.0 : 0.0;
    if (v.kind == Json::Int) return (double)(int)v.num;
    if (v.kind == Json::Real) return v.num;
    return dflt;
}
double extras_double(const Json* extras, const char* name, double dflt = 0.0)
{
    const Json* v = extras ? extras->find(name) : nullptr;
    return v ? as_double(*v, dflt) : dflt;
}
void extras_color(const Json* extras, const char* name, float out[3])
{
    const Json* v = extras ? extras->find(name) : nullptr;
    if (!v || v->kind != Json::Array || v->arr.size() < 3) return;
    for (int k = 0; k < 3; ++k) out[k] = (float)as_double(v->arr[k], 1.0);
}

// ---- 4x4 double matrices, row-major (math/Transformation.cpp) ------------------------------------
struct Mat4 {
    double m[16];
};
Mat4 identity()
{
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}
Mat4 mul(const Mat4& a, const Mat4& b)
{
    Mat4 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a.m[i * 4 + k] * b.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}
bool invert(const Mat4& a, Mat4& out) // Gauss-Jordan with partial pivoting (Matrix4d::inverse in the reference)
{
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { w[i][j] = a.m[i * 4 + j]; w[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 8; ++j) std::swap(w[piv][j], w[c][j]);
        const double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = w[r][c];
            if (f != 0.0)
                for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out.m[i * 4 + j] = w[i][4 + j];
    return true;
}
struct Xform { // Transformation: matrix + inverse
    Mat4 m = identity(), inv = identity();
};
Xform append(const Xform& a, const Xform& b) { return Xform{mul(a.m, b.m), mul(b.inv, a.inv)}; } // math/Transformation.h:66-69
Xform translate(double x, double y, double z)
{
    Xform t;
    t.m.m[3] = x; t.m.m[7] = y; t.m.m[11] = z;
    t.inv.m[3] = -x; t.inv.m[7] = -y; t.inv.m[11] = -z;
    return t;
}
Xform scale(double x, double y, double z) // math/Transformation.cpp:18-33
{
    Xform t;
    t.m.m[0] = x; t.m.m[5] = y; t.m.m[10] = z;
    t.inv.m[0] = 1 / x; t.inv.m[5] = 1 / y; t.inv.m[10] = 1 / z;
    return t;
}
Xform rotate_quaternion(double i, double j, double k, double r) // math/Transformation.cpp:104-117
{
    const double s = 1.0 / (r * r + i * i + j * j + k * k);
    Xform t;
    const double v[16] = {1 - 2 * s * ((j * j) + (k * k)), 2 * s * (i * j - k * r), 2 * s * (i * k + j * r), 0,
                          2 * s * (i * j + k * r), 1.0 - 2 * s * ((i * i) + (k * k)), 2 * s * (j * k - i * r), 0,
                          2 * s * (i * k - j * r), 2 * s * (j * k + i * r), 1 - 2 * s * ((i * i) + (j * j)), 0,
                          0, 0, 0, 1};
    std::memcpy(t.m.m, v, sizeof v);
    if (!invert(t.m, t.inv)) throw std::runtime_error("glTF: singular rotation");
    return t;
}
// Transformation::transform(Point) / (Vector3), math/Transformation.h:71-87
void xf_point(const Mat4& M, const float p[3], float out[3])
{
    double v[4];
    for (int i = 0; i < 4; ++i) v[i] = ((M.m[i * 4] * p[0] + M.m[i * 4 + 1] * p[1]) + M.m[i * 4 + 2] * p[2]) + M.m[i * 4 + 3] * 1.0;
    for (int i = 0; i < 3; ++i) out[i] = (float)(v[i] / v[3]);
}
void xf_vector(const Mat4& M, const float p[3], float out[3])
{
    for (int i = 0; i < 3; ++i) out[i] = (float)(((M.m[i * 4] * p[0] + M.m[i * 4 + 1] * p[1]) + M.m[i * 4 + 2] * p[2]) + M.m[i * 4 + 3] * 0.0);
}

// ---- base64 (data: URIs) and PNG -----------------------------------------------------------------
std::vector<uint8_t> base64_decode(const char* s, size_t n)
{
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else continue; // padding, whitespace
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return out;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// PNG -> RGBA8, what the reference's loader gets from stb_image when it asks for 4 components (tinygltf's
// LoadImageData, io/lib/tiny_gltf.h:1677-1771; loadImage, io/GLTF.cpp:292-316).  stb_image's conventions, from its
// source text (io/lib/stb_image.h: stbi__depth_scale_table, stbi__compute_transparency, stbi__expand_png_palette):
// grey samples of 1 / 2 / 4 bits are scaled by 0xff / 0x55 / 0x11, palette indices are not; a tRNS chunk of a grey or
// RGB image is a colour key (alpha 0 where the pixel equals it, 255 elsewhere), of a palette image one alpha per entry;
// both interlace methods are read.  A 16-bit image reaches loadImage with 8 bytes per pixel and is refused there.
void decode_png(const uint8_t* data, size_t n, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || std::memcmp(data, sig, 8)) throw std::runtime_error("glTF image: not a PNG");
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    bool hasTrns = false;
    std::vector<uint8_t> idat, palette, trns;
    width = height = 0;
    while (pos + 12 <= n) {
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (pos + 12 + (size_t)len > n) throw std::runtime_error("glTF image: truncated PNG");
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            width = be32(body); height = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) { trns.assign(body, body + len); hasTrns = true; }
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) throw std::runtime_error("glTF image: bad PNG size");
    if (interlace > 1) throw std::runtime_error("glTF image: bad PNG interlace method");
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) throw std::runtime_error("glTF image: PNG bit depth must be 1, 2, 4, 8 or 16");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("glTF image: bad PNG colour type");
    }
    if ((ctype == 3 && depth == 16) || (depth < 8 && ctype != 0 && ctype != 3)) throw std::runtime_error("glTF image: bad PNG colour type");
    if (depth == 16) throw std::runtime_error("Can't load image data");   // io/GLTF.cpp:311-313: image.size() != width * height * component
    if (ctype == 3 && (palette.empty() || palette.size() % 3 != 0 || palette.size() > 768)) throw std::runtime_error("glTF image: bad PNG palette");
    static const uint8_t scaleOf[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    const uint8_t scale = ctype == 0 ? scaleOf[depth] : 1;
    uint8_t key[3] = {0, 0, 0};
    bool keyed = false;
    if (hasTrns) {
        if (ctype == 3) {
            if (trns.size() > palette.size() / 3) throw std::runtime_error("glTF image: bad PNG tRNS length");
        } else {
            if (ctype == 4 || ctype == 6) throw std::runtime_error("glTF image: PNG tRNS with alpha");
            if (trns.size() != (size_t)channels * 2) throw std::runtime_error("glTF image: bad PNG tRNS length");
            for (int k = 0; k < channels; ++k) key[k] = (uint8_t)(trns[2 * k + 1] * scale);
            keyed = true;
        }
    }
    const size_t bitsPerPixel = (size_t)channels * depth;
    const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;   // distance of the filters' "left" byte
    // the reduced images of the file: one, or the seven passes of Adam7
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass whole = {0, 0, 1, 1};
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    const Pass* passes = interlace ? adam7 : &whole;
    const int nPasses = interlace ? 7 : 1;
    size_t rawSize = 0;
    for (int p = 0; p < nPasses; ++p) {
        const uint32_t pw = width > passes[p].x0 ? (width - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const uint32_t ph = height > passes[p].y0 ? (height - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (pw && ph) rawSize += (((size_t)pw * bitsPerPixel + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw(rawSize);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
        throw std::runtime_error("glTF image: PNG data does not inflate");
    rgba.assign((size_t)width * height * 4, 0);
    const uint8_t* row = raw.data();
    for (int p = 0; p < nPasses; ++p) {
        const Pass& P = passes[p];
        const uint32_t pw = width > P.x0 ? (width - P.x0 + P.dx - 1) / P.dx : 0;
        const uint32_t ph = height > P.y0 ? (height - P.y0 + P.dy - 1) / P.dy : 0;
        if (!pw || !ph) continue;
        const size_t stride = ((size_t)pw * bitsPerPixel + 7) / 8;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (uint32_t y = 0; y < ph; ++y, row += stride + 1) {
            const int filter = row[0];
            for (size_t x = 0; x < stride; ++x) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                int pred = 0;
                switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: {
                    const int q = a + b - c, pa = std::abs(q - a), pb = std::abs(q - b), pc = std::abs(q - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: throw std::runtime_error("glTF image: bad PNG filter");
                }
                cur[x] = (uint8_t)(row[1 + x] + pred);
            }
            uint8_t* outRow = rgba.data() + (size_t)(P.y0 + y * P.dy) * width * 4;
            for (uint32_t x = 0; x < pw; ++x) {
                uint8_t smp[4] = {0, 0, 0, 0};
                if (depth == 8) {
                    for (int k = 0; k < channels; ++k) smp[k] = cur[(size_t)x * channels + k];
                } else {   // 1, 2 or 4 bits, one channel, most significant bits first
                    const size_t bit = (size_t)x * depth;
                    smp[0] = (uint8_t)((cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u));
                }
                uint8_t r, g, b, a = 255;
                if (ctype == 0) {
                    r = g = b = (uint8_t)(smp[0] * scale);
                    if (keyed && r == key[0]) a = 0;
                } else if (ctype == 2) {
                    r = smp[0]; g = smp[1]; b = smp[2];
                    if (keyed && r == key[0] && g == key[1] && b == key[2]) a = 0;
                } else if (ctype == 3) {
                    const size_t idx = smp[0];
                    if (idx * 3 + 2 >= palette.size()) throw std::runtime_error("glTF image: PNG palette index out of range");
                    r = palette[idx * 3]; g = palette[idx * 3 + 1]; b = palette[idx * 3 + 2];
                    if (idx < trns.size()) a = trns[idx];
                } else if (ctype == 4) { r = g = b = smp[0]; a = smp[1]; }
                else { r = smp[0]; g = smp[1]; b = smp[2]; a = smp[3]; }
                uint8_t* out = outRow + (size_t)(P.x0 + x * P.dx) * 4;
                out[0] = r; out[1] = g; out[2] = b; out[3] = a;
            }
            prev.swap(cur);
        }
    }
}

// Radiance .hdr -> RGBA floats, what read_hdr_file (io/HDR.cpp:5-18) gets from stb_image's stbi_loadf with 4
// components.  stb_image's rules, from its source text (io/lib/stb_image.h, stbi__hdr_load / stbi__hdr_convert): the
// first line is "#?RADIANCE" or "#?RGBE", a header line "FORMAT=32-bit_rle_rgbe" is required, the size line is
// "-Y <height> +X <width>"; rows are run-length coded per component when 8 <= width < 32768 and the row starts with
// 2 2 <width, 2 bytes>, flat RGBE otherwise; a pixel is mantissa * 2^(e - 136) in float, (0, 0, 0) for e = 0, alpha 1.
// (A flat row met after coded rows makes stb_image restart at the first row; such a file is refused here.)
void decode_hdr(const std::vector<uint8_t>& file, std::vector<float>& rgba, uint32_t& width, uint32_t& height)
{
    size_t pos = 0;
    auto line = [&]() {
        std::string t;
        while (pos < file.size() && file[pos] != '\n') t.push_back((char)file[pos++]);
        if (pos < file.size()) ++pos;
        return t;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") throw std::runtime_error("Failed to load HDR file");
    bool valid = false;
    for (;;) {
        const std::string t = line();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") valid = true;
    }
    if (!valid) throw std::runtime_error("Failed to load HDR file");
    const std::string size = line();
    if (size.compare(0, 3, "-Y ") != 0) throw std::runtime_error("Failed to load HDR file");
    char* end = nullptr;
    const long h = std::strtol(size.c_str() + 3, &end, 10);
    while (*end == ' ') ++end;
    if (std::strncmp(end, "+X ", 3) != 0) throw std::runtime_error("Failed to load HDR file");
    const long w = std::strtol(end + 3, nullptr, 10);
    if (w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h >= (1ull << 28)) throw std::runtime_error("Failed to load HDR file");
    width = (uint32_t)w; height = (uint32_t)h;
    rgba.assign((size_t)w * h * 4, 0.0f);
    auto need = [&](size_t n) { if (pos + n > file.size()) throw std::runtime_error("Failed to load HDR file"); };
    auto convert = [&](const uint8_t* in, float* out) {
        if (in[3] != 0) {
            const float f1 = (float)std::ldexp(1.0f, (int)in[3] - (int)(128 + 8));
            out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1;
        } else {
            out[0] = out[1] = out[2] = 0.0f;
        }
        out[3] = 1.0f;
    };
    const bool mayBeCoded = w >= 8 && w < 32768;
    bool flat = !mayBeCoded;
    if (mayBeCoded) {
        need(4);
        flat = file[pos] != 2 || file[pos + 1] != 2 || (file[pos + 2] & 0x80);   // decided by the first row
    }
    if (flat) {
        need((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; ++i) convert(&file[pos + 4 * i], &rgba[4 * i]);
        return;
    }
    std::vector<uint8_t> scan((size_t)w * 4);
    for (long j = 0; j < h; ++j) {
        need(4);
        if (file[pos] != 2 || file[pos + 1] != 2 || (file[pos + 2] & 0x80)) throw std::runtime_error("Failed to load HDR file");
        const long len = ((long)file[pos + 2] << 8) | file[pos + 3];
        pos += 4;
        if (len != w) throw std::runtime_error("Failed to load HDR file");
        for (int k = 0; k < 4; ++k) {
            long i = 0;
            while (w - i > 0) {
                need(1);
                int count = file[pos++];
                if (count > 128) {
                    need(1);
                    const uint8_t value = file[pos++];
                    count -= 128;
                    if (count > w - i) throw std::runtime_error("Failed to load HDR file");
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = value;
                } else {
                    if (count > w - i) throw std::runtime_error("Failed to load HDR file");
                    if (count == 0) throw std::runtime_error("Failed to load HDR file");   // would never end
                    need((size_t)count);
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = file[pos++];
                }
            }
        }
        for (long i = 0; i < w; ++i) convert(&scan[(size_t)i * 4], &rgba[((size_t)j * w + i) * 4]);
    }
}

// PathResolver (utility/PathResolver.cpp): "#WORKDIR#" in a path stands for the directory set with setWorkDir
std::string g_workdir;
std::string resolve_path(const std::string& in)
{
    std::string str = in;
    size_t at;
    if (g_workdir.find("#WORKDIR#") != std::string::npos) return str;   // the replacement would never end
    while ((at = str.find("#WORKDIR#")) != std::string::npos) str.replace(at, 9, g_workdir);
    return str;
}

std::vector<uint8_t> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to load gltf: cannot open " + path);
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> data((size_t)std::max<std::streamoff>(n, 0));
    if (n > 0) f.read((char*)data.data(), n);
    return data;
}

} // namespace

// ---- the loaded scene -----------------------------------------------------------------------------
struct rtgpu_gltf {
    struct Mesh {
        std::vector<float> vertices, normals, texcoords;
        std::vector<uint32_t> indices, triMaterial;
    };
    struct Tex {
        std::vector<uint8_t> rgba;
        uint32_t width = 0, height = 0;
        double gamma = 1.0;
    };
    struct Camera {
        double xFov = 0, focalDistance = 0, aperture = 0;
        bool isMain = false;
        Mat4 transform;
    };
    std::vector<std::unique_ptr<Mesh>> meshes;
    std::vector<rtgpu_instance_desc> instances;
    std::vector<rtgpu_material> materials;
    std::vector<rtgpu_point_light> pointLights;
    std::vector<rtgpu_area_light> areaLights;
    std::vector<rtgpu_directional_light> dirLights;
    std::vector<std::unique_ptr<Tex>> textures;
    std::vector<Camera> cameras;
    int hasEnv = 0;   // 1: colour, 2: image map
    float envColor[3] = {0, 0, 0};
    double envIntensity = 0;
    std::vector<float> envMap;
    uint32_t envW = 0, envH = 0;
    // views handed to the caller
    std::vector<rtgpu_mesh_desc> meshDescs;
    std::vector<rtgpu_texture> texDescs;
    rtgpu_scene_desc desc{};
};

namespace {

struct Loader {
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
    std::string baseDir;
    float aspect;
    rtgpu_gltf& out;
    std::map<std::pair<int, int>, uint32_t> meshCache;     // (mesh, primitive) -> out.meshes index (meshCache, io/GLTF.cpp:460-471)
    std::map<int, uint32_t> materialCache;                 // glTF material (or -1) -> material id (:473-493)
    std::map<std::pair<int, int>, uint32_t> textureCache;  // (image, gamma * 10) -> texture index

    Loader(rtgpu_gltf& o, float a) : aspect(a), out(o) {}

    const Json& list(const char* name) const
    {
        static const Json empty;
        const Json* v = root.find(name);
        return v && v->kind == Json::Array ? *v : empty;
    }
    const Json& item(const char* name, int idx) const
    {
        const Json& l = list(name);
        if (idx < 0 || (size_t)idx >= l.size()) throw std::runtime_error(std::string("glTF: index out of range in ") + name);
        return l.arr[(size_t)idx];
    }

    // one accessor read the way loadIndicesImpl / loadVec3s / loadVec2s walk it (io/GLTF.cpp:33-138): from the
    // accessor's start to the end of its buffer view, one element per stride
    // A byte offset / length / stride of the file as a size: the JSON number must be finite, non-negative and an
    // integer below 2^53 (a double holds those exactly; anything else would make the cast undefined or the sums wrap).
    static size_t byte_count(const Json& obj, const char* name)
    {
        const double v = obj.get_num(name, 0);
        if (!(v >= 0.0) || !(v < 9007199254740992.0) || v != std::floor(v)) throw std::runtime_error(std::string("glTF: bad ") + name);
        return (size_t)v;
    }
    template <typename F>
    void walk_accessor(const Json& acc, size_t elemSize, F&& f)
    {
        const Json& bv = item("bufferViews", acc.get_int("bufferView", -1));
        const size_t viewOff = byte_count(bv, "byteOffset");
        const size_t start = viewOff + byte_count(acc, "byteOffset");   // both < 2^53: no wrap
        const size_t end = viewOff + byte_count(bv, "byteLength");
        size_t stride = byte_count(bv, "byteStride");
        if (stride == 0) stride = elemSize;
        const int bi = bv.get_int("buffer", -1);
        if (bi < 0 || (size_t)bi >= buffers.size()) throw std::runtime_error("glTF: buffer index out of range");
        const std::vector<uint8_t>& buf = buffers[(size_t)bi];
        for (size_t i = start; i < end; i += stride) {
            if (i > buf.size() || elemSize > buf.size() - i) throw std::runtime_error("glTF: accessor reads past its buffer");
            f(buf.data() + i);
        }
    }
    void load_indices(const Json& acc, std::vector<uint32_t>& idx)
    {
        if (acc.get_str("type") != "SCALAR") throw Unsupported("Unsupported index accessor");
        const int ct = acc.get_int("componentType", 0);
        size_t sz;
        if (ct == 5124 || ct == 5125) sz = 4;
        else if (ct == 5122 || ct == 5123) sz = 2;
        else if (ct == 5120 || ct == 5121) sz = 1;
        else throw Unsupported("Unsupported index component type");
        walk_accessor(acc, sz, [&](const uint8_t* p) {
            uint32_t v = 0;
            std::memcpy(&v, p, sz); // little endian
            idx.push_back(v);
        });
        idx.resize(idx.size() / 3 * 3); // whole triangles only (:49-55)
    }
    void load_floats(const Json& acc, int comps, std::vector<float>& out_)
    {
        if (acc.get_int("componentType", 0) != 5126 || acc.get_str("type") != (comps == 3 ? "VEC3" : "VEC2"))
            throw Unsupported("Unsupported vector accessor");
        walk_accessor(acc, sizeof(float) * (size_t)comps, [&](const uint8_t* p) {
            float v[3];
            std::memcpy(v, p, sizeof(float) * (size_t)comps);
            out_.insert(out_.end(), v, v + comps);
        });
    }

    // loadPrimitiveShape, io/GLTF.cpp:202-256
    std::unique_ptr<rtgpu_gltf::Mesh> load_shape(const Json& prim)
    {
        if (prim.get_int("mode", 4) != 4) throw Unsupported("Unsupported primitive type");
        auto m = std::make_unique<rtgpu_gltf::Mesh>();
        if (prim.get_int("indices", -1) != -1) load_indices(item("accessors", prim.get_int("indices", -1)), m->indices);
        const Json* attrs = prim.find("attributes");
        static const Json none;
        if (!attrs) attrs = &none;
        if (attrs->has("POSITION")) load_floats(item("accessors", attrs->get_int("POSITION", -1)), 3, m->vertices);
        if (attrs->has("NORMAL")) load_floats(item("accessors", attrs->get_int("NORMAL", -1)), 3, m->normals);
        else throw Unsupported("Implicit normals are not supported");
        if (attrs->has("TEXCOORD_0")) load_floats(item("accessors", attrs->get_int("TEXCOORD_0", -1)), 2, m->texcoords);
        // The walks above run to the end of their buffer views, whatever the accessors' counts say, so accessors that
        // share a view (positions followed by normals, say) give arrays of different lengths; the reference hands them to
        // TriangleMesh as they are and one index triple addresses all three (:246-254).  Here every index must be inside
        // every array (the reference would read past its vectors), and the arrays are then brought to one length -- the
        // entries added or dropped are ones no triangle names.
        const size_t nv = m->vertices.size() / 3, nn = m->normals.size() / 3, nt = m->texcoords.size() / 2;
        for (uint32_t i : m->indices) {
            if (i >= nv) throw std::runtime_error("glTF: vertex index out of range");
            if (i >= nn) throw std::runtime_error("glTF: normal index out of range");
            if (!m->texcoords.empty() && i >= nt) throw std::runtime_error("glTF: texture coordinate index out of range");
        }
        m->normals.resize(nv * 3, 0.0f);
        if (!m->texcoords.empty()) m->texcoords.resize(nv * 2, 0.0f);
        return m;
    }

    // loadImage / loadTexture, io/GLTF.cpp:284-321 (PNG only)
    uint32_t load_texture(int textureIdx, double gamma, const Json& gltfMaterial)
    {
        if (const Json* ex = gltfMaterial.find("extras"))
            if (ex->has("VideoImageMapping")) throw Unsupported("VideoImageMapping materials (JPEG frame sequences) are not supported");
        const int imageIdx = item("textures", textureIdx).get_int("source", -1);
        const auto key = std::make_pair(imageIdx, (int)std::lround(gamma * 10));
        auto it = textureCache.find(key);
        if (it != textureCache.end()) return it->second;
        const Json& img = item("images", imageIdx);
        std::vector<uint8_t> bytes;
        if (img.has("bufferView")) {
            const Json& bv = item("bufferViews", img.get_int("bufferView", -1));
            const int bi = bv.get_int("buffer", -1);
            if (bi < 0 || (size_t)bi >= buffers.size()) throw std::runtime_error("glTF: buffer index out of range");
            const size_t off = byte_count(bv, "byteOffset"), len = byte_count(bv, "byteLength");
            const size_t have = buffers[(size_t)bi].size();
            if (off > have || len > have - off) throw std::runtime_error("glTF: image view past its buffer");
            bytes.assign(buffers[(size_t)bi].begin() + (std::ptrdiff_t)off, buffers[(size_t)bi].begin() + (std::ptrdiff_t)(off + len));
        } else {
            bytes = load_uri(img.get_str("uri"));
        }
        if (bytes.size() >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8) throw Unsupported("glTF image: JPEG needs a decoder this library does not have");
        auto tex = std::make_unique<rtgpu_gltf::Tex>();
        decode_png(bytes.data(), bytes.size(), tex->rgba, tex->width, tex->height);
        tex->gamma = gamma;
        out.textures.push_back(std::move(tex));
        const uint32_t id = (uint32_t)out.textures.size() - 1;
        textureCache[key] = id;
        return id;
    }
    std::vector<uint8_t> load_uri(const std::string& uri)
    {
        if (uri.rfind("data:", 0) == 0) {
            const size_t comma = uri.find(',');
            if (comma == std::string::npos) throw std::runtime_error("glTF: malformed data URI");
            return base64_decode(uri.c_str() + comma + 1, uri.size() - comma - 1);
        }
        if (uri.empty()) throw std::runtime_error("glTF: buffer or image without data");
        return read_file(baseDir + uri);
    }

    uint32_t add_material(const rtgpu_material& m)
    {
        out.materials.push_back(m);
        return (uint32_t)out.materials.size() - 1;
    }
    static rtgpu_material blank(int type)
    {
        rtgpu_material m{};
        m.type = type;
        m.color[0] = m.color[1] = m.color[2] = 1.0f;
        m.intensity = 1.0;
        return m;
    }

    // loadMaterial, io/GLTF.cpp:323-455
    uint32_t load_material(const Json& mat, const Json* nodeProps)
    {
        rtgpu_material diffuse = blank(RTGPU_MAT_DIFFUSE);
        diffuse.color[0] = diffuse.color[1] = diffuse.color[2] = 0.0f; // DiffuseMaterial::diffuseColor = BLACK
        rtgpu_material glossy = blank(RTGPU_MAT_GLOSSY), coatGlossy = blank(RTGPU_MAT_GLOSSY);
        static const Json none;
        const Json* pbr = mat.find("pbrMetallicRoughness");
        if (!pbr) pbr = &none;

        float baseColor[3] = {1.0f, 1.0f, 1.0f};
        if (const Json* t = pbr->find("baseColorTexture")) diffuse.texture = 1 + load_texture(t->get_int("index", -1), 2.2, mat); // sRGB to raw radiance
        if (const Json* f = pbr->find("baseColorFactor")) {
            if (f->kind == Json::Array && f->arr.size() >= 3) {
                for (int k = 0; k < 3; ++k) baseColor[k] = (float)f->arr[(size_t)k].num;
                std::memcpy(diffuse.color, baseColor, 12);
                std::memcpy(glossy.color, baseColor, 12);
            }
        }
        if (pbr->has("roughnessFactor")) glossy.roughness = (float)pbr->get_num("roughnessFactor", 0.0);
        if (const Json* t = mat.find("normalTexture")) {
            const uint32_t tex = 1 + load_texture(t->get_int("index", -1), 1.0, mat);
            diffuse.normal_map = glossy.normal_map = coatGlossy.normal_map = tex;
        }
        const bool clearCoat = extras_bool(nodeProps, "Material.ClearCoat", false);
        if (clearCoat) coatGlossy.roughness = (float)extras_double(nodeProps, "Material.ClearCoatRoughness", 0);

        uint32_t result = add_material(diffuse);
        if (pbr->has("metallicFactor")) {
            const double metallic = pbr->get_num("metallicFactor", 0.0);
            if (metallic > 0) {
                rtgpu_material mix = blank(RTGPU_MAT_CONSTMIX);
                mix.mix_factor = metallic;
                mix.first = result;
                mix.second = add_material(glossy);
                result = add_material(mix);
            }
        }
        if (clearCoat) {
            rtgpu_material coat = blank(RTGPU_MAT_FRESNELMIX);
            coat.roughness = (float)extras_double(nodeProps, "Material.ClearCoatIOR", 1.45f);
            coat.first = result;
            coat.second = add_material(coatGlossy);
            result = add_material(coat);
        }
        const double ior = extras_double(nodeProps, "Material.IOR", 1.0);
        const double transmission = extras_double(nodeProps, "Material.Transmission", 0.0);
        const double attenuation = extras_double(nodeProps, "Material.AttenuationStrength", 5.0);
        if (transmission > 0) {
            rtgpu_material glass = blank(RTGPU_MAT_GLASS);
            glass.intensity = ior;
            std::memcpy(glass.color, baseColor, 12);
            glass.roughness = (float)attenuation;
            rtgpu_material mix = blank(RTGPU_MAT_CONSTMIX);
            mix.first = result;
            mix.second = add_material(glass);
            mix.mix_factor = transmission;
            result = add_material(mix);
        }
        const Json* ef = mat.find("emissiveFactor");
        const Json* et = mat.find("emissiveTexture");
        if (ef || et) {
            rtgpu_material emissive = blank(RTGPU_MAT_EMISSIVE);
            emissive.color[0] = emissive.color[1] = emissive.color[2] = 0.0f; // EmissiveMaterial::color = BLACK
            bool active = false;
            if (et) { emissive.texture = 1 + load_texture(et->get_int("index", -1), 1.0, mat); active = true; }
            if (ef && ef->kind == Json::Array && ef->arr.size() >= 3) {
                for (int k = 0; k < 3; ++k) emissive.color[k] = (float)ef->arr[(size_t)k].num;
                active = !(emissive.color[0] == 0.0f && emissive.color[1] == 0.0f && emissive.color[2] == 0.0f);
            }
            emissive.intensity = extras_double(nodeProps, "Material.EmissionStrength", 1.0);
            if (active) {
                rtgpu_material add = blank(RTGPU_MAT_ADD);
                add.first = add_material(emissive);
                add.second = result;
                result = add_material(add);
            }
        }
        const double alpha = extras_double(nodeProps, "Material.Alpha", 1.0);
        if (alpha < 1.0) {
            rtgpu_material mix = blank(RTGPU_MAT_CONSTMIX);
            mix.first = add_material(blank(RTGPU_MAT_TRANSPARENT));
            mix.second = result;
            mix.mix_factor = alpha;
            result = add_material(mix);
        }
        return result;
    }

    // loadPrimitive, io/GLTF.cpp:457-496
    void load_primitive(int meshIdx, int primIdx, const Json& prim, const Json* nodeProps, const Xform& world)
    {
        const auto key = std::make_pair(meshIdx, primIdx);
        auto mi = meshCache.find(key);
        if (mi == meshCache.end()) {
            out.meshes.push_back(load_shape(prim));
            mi = meshCache.emplace(key, (uint32_t)out.meshes.size() - 1).first;
        }
        const int matIdx = prim.get_int("material", -1);
        auto ci = materialCache.find(matIdx);
        if (ci == materialCache.end()) {
            uint32_t id;
            if (matIdx != -1) {
                id = load_material(item("materials", matIdx), nodeProps);
            } else {
                rtgpu_material d = blank(RTGPU_MAT_DIFFUSE);
                d.intensity = 0.8;
                id = add_material(d);
            }
            ci = materialCache.emplace(matIdx, id).first;
        }
        rtgpu_instance_desc inst{};
        inst.shape = (int32_t)mi->second;
        inst.material = ci->second;
        std::memcpy(inst.transform, world.m.m, sizeof inst.transform);
        std::memcpy(inst.inverse, world.inv.m, sizeof inst.inverse);
        out.instances.push_back(inst);
    }

    // loadNode (io/GLTF.cpp:498-648) fused with the depth-first flattening of DynamicScene::build
    // (scene/dynamic/DynamicScene.cpp:108-145): lights and cameras take the accumulated transform
    void load_node(int nodeIdx, const Xform& parentXf, const Json* parent, int depth)
    {
        if (depth > 256) throw std::runtime_error("glTF: node hierarchy too deep (cycle?)");
        const Json& node = item("nodes", nodeIdx);
        Xform local;
        if (const Json* mtx = node.find("matrix")) {
            if (mtx->kind != Json::Array || mtx->arr.size() != 16) throw std::runtime_error("glTF: node matrix must have 16 entries");
            for (int c = 0; c < 4; ++c) // column-major in the file
                for (int r = 0; r < 4; ++r) local.m.m[r * 4 + c] = mtx->arr[(size_t)(c * 4 + r)].num;
            if (!invert(local.m, local.inv)) throw std::runtime_error("glTF: singular node matrix");
        } else {
            auto vec = [&](const char* name, size_t n, double* v) {
                const Json* a = node.find(name);
                if (!a || a->kind != Json::Array || a->arr.size() < n) return false;
                for (size_t k = 0; k < n; ++k) v[k] = a->arr[k].num;
                return true;
            };
            double v[4];
            if (vec("translation", 3, v)) local = append(local, translate(v[0], v[1], v[2]));
            if (vec("rotation", 4, v)) local = append(local, rotate_quaternion(v[0], v[1], v[2], v[3]));
            if (vec("scale", 3, v)) local = append(local, scale(v[0], v[1], v[2]));
        }
        const Xform world = append(parentXf, local);
        const Json* extras = node.find("extras");
        const int meshIdx = node.get_int("mesh", -1), camIdx = node.get_int("camera", -1);

        if (extras_bool(extras, "IsAreaLight")) {
            if (meshIdx == -1) throw std::runtime_error("Invalid area light definition");
            const Json* prims = item("meshes", meshIdx).find("primitives");
            if (!prims || prims->size() != 1) throw std::runtime_error("Invalid area light definition");
            auto shape = load_shape(prims->arr[0]);
            if (shape->vertices.size() != 9 || shape->indices.size() != 3) throw std::runtime_error("Invalid area light definition: must be triangular");
            rtgpu_area_light L{};
            xf_point(world.m, &shape->vertices[3 * shape->indices[0]], L.a); // AreaLight::applyTransform, light/AreaLight.cpp:14-19
            xf_point(world.m, &shape->vertices[3 * shape->indices[1]], L.b);
            xf_point(world.m, &shape->vertices[3 * shape->indices[2]], L.c);
            L.intensity = extras_double(extras, "LightIntensity", 500);
            L.color[0] = L.color[1] = L.color[2] = 1.0f;
            extras_color(extras, "LightColor", L.color);
            out.areaLights.push_back(L);
        } else if (extras_bool(extras, "IsPointLight")) {
            rtgpu_point_light L{};
            L.intensity = extras_double(extras, "LightIntensity", 500);
            L.color[0] = L.color[1] = L.color[2] = 1.0f;
            extras_color(extras, "LightColor", L.color);
            const float zero[3] = {0, 0, 0};
            xf_point(world.m, zero, L.pos); // PointLight::applyTransform, light/PointLight.cpp
            if (!(L.intensity < 1E-6)) out.pointLights.push_back(L);
        } else if (extras_bool(extras, "IsDirectionalLight")) {
            rtgpu_directional_light L{};
            L.intensity = extras_double(extras, "LightIntensity", 500);
            L.angle = (float)(0.5 * extras_double(extras, "DirectionalLight.Angle", 0.009180));
            L.color[0] = L.color[1] = L.color[2] = 1.0f;
            extras_color(extras, "LightColor", L.color);
            const float down[3] = {0, -1, 0};
            float d[3];
            xf_vector(world.m, down, d); // DirectionalLight::applyTransform: transform, then normalize
            const float len = std::sqrt(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2]));
            for (int k = 0; k < 3; ++k) L.direction[k] = d[k] / len;
            if (!(L.intensity < 1E-6)) out.dirLights.push_back(L);
        } else if (meshIdx != -1) {
            const Json* prims = item("meshes", meshIdx).find("primitives");
            const size_t n = prims ? prims->size() : 0;
            for (size_t k = 0; k < n; ++k) load_primitive(meshIdx, (int)k, prims->arr[k], extras, world);
        } else if (camIdx != -1) {
            const Json& cam = item("cameras", camIdx);
            if (cam.get_str("type") == "perspective") {
                static const Json none;
                const Json* persp = cam.find("perspective");
                if (!persp) persp = &none;
                const double kPi = 3.14159265358979323;
                const double xFovRad = std::atan(std::tan(persp->get_num("yfov", 0.0) / 2.0) * aspect) * 2.0;
                rtgpu_gltf::Camera c;
                c.xFov = (xFovRad / kPi) * 180.0;
                c.transform = world.m;
                c.isMain = extras_bool(extras, "IsMainCamera") || extras_bool(parent, "IsMainCamera");
                c.focalDistance = (extras && extras->has("FocalDistance")) ? extras_double(extras, "FocalDistance", 0.0) : extras_double(parent, "FocalDistance", 0.0);
                if (extras && extras->has("Aperture")) c.aperture = extras_double(extras, "Aperture", 0.0);
                else if (parent && parent->has("Aperture")) c.aperture = extras_double(parent, "Aperture", 0.0);
                else if (c.focalDistance > 0) {
                    const double sensorSize = extras_double(extras, "SensorSize", 0.036);
                    const double distToSensor = (0.5 * sensorSize) / std::tan(0.5 * xFovRad);
                    const double focalLength = 1.0 / (1.0 / c.focalDistance + 1.0 / distToSensor);
                    if (extras && extras->has("FStop")) c.aperture = 0.5 * focalLength / extras_double(extras, "FStop", 0.0);
                    else if (parent && parent->has("FStop")) c.aperture = 0.5 * focalLength / extras_double(parent, "FStop", 0.0);
                }
                out.cameras.push_back(c);
            }
        }
        if (const Json* children = node.find("children"))
            for (const Json& c : children->arr) load_node(c.as_int(-1), world, extras, depth + 1);
    }

    // loadEnvironmentInfo, io/GLTF.cpp:650-680
    void load_environment(const Json& scene)
    {
        const Json* extras = scene.find("extras");
        const double intensity = extras_double(extras, "EnvironmentIntensity", 0.0);
        if (!(intensity > 0)) return;
        if (extras->has("EnvironmentHDRIFilePath")) {
            const std::string hdrPath = extras->get_str("EnvironmentHDRIFilePath");
            std::vector<uint8_t> file;
            try {
                file = read_file(resolve_path(hdrPath));
            } catch (const std::exception&) {
                throw std::runtime_error("Failed to load HDR file");   // io/HDR.cpp:10-13
            }
            decode_hdr(file, out.envMap, out.envW, out.envH);
            out.envIntensity = intensity;
            out.hasEnv = 2;
        } else if (extras->has("EnvironmentColor")) {
            extras_color(extras, "EnvironmentColor", out.envColor);
            out.envIntensity = intensity;
            out.hasEnv = 1;
        }
    }

    void parse_container(const std::string& path)
    {
        const std::vector<uint8_t> file = read_file(path);
        const size_t slash = path.find_last_of('/');
        baseDir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
        auto le32 = [&](size_t off) { uint32_t v; std::memcpy(&v, file.data() + off, 4); return v; };
        if (file.size() < 20 || le32(0) != 0x46546C67u) throw std::runtime_error("Failed to load gltf: not a binary glTF file");
        if (le32(4) != 2u) throw std::runtime_error("Failed to load gltf: container version must be 2");
        const size_t total = std::min<size_t>(le32(8), file.size());
        size_t pos = 12;
        std::vector<uint8_t> bin;
        bool haveJson = false, haveBin = false;
        while (pos + 8 <= total) {
            const uint32_t len = le32(pos), type = le32(pos + 4);
            if (pos + 8 + (size_t)len > total) throw std::runtime_error("Failed to load gltf: truncated chunk");
            if (type == 0x4E4F534Au && !haveJson) {
                JsonParser jp{(const char*)file.data() + pos + 8, (const char*)file.data() + pos + 8 + len};
                root = jp.value(0);
                haveJson = true;
            } else if (type == 0x004E4942u && !haveBin) {
                bin.assign(file.begin() + (std::ptrdiff_t)(pos + 8), file.begin() + (std::ptrdiff_t)(pos + 8 + len));
                haveBin = true;
            }
            pos += 8 + (size_t)((len + 3u) & ~3u);
        }
        if (!haveJson || root.kind != Json::Object) throw std::runtime_error("Failed to load gltf: no JSON chunk");
        const Json& bl = list("buffers");
        for (size_t i = 0; i < bl.size(); ++i) {
            const Json& b = bl.arr[i];
            if (!b.has("uri")) {
                if (i != 0 || !haveBin) throw std::runtime_error("Failed to load gltf: buffer without data");
                buffers.push_back(std::move(bin));
            } else {
                buffers.push_back(load_uri(b.get_str("uri")));
            }
        }
    }

    void run(const std::string& path)
    {
        parse_container(path);
        const Json& scenes = list("scenes");
        int sceneIdx = root.get_int("scene", 0); // model.defaultScene
        if (scenes.size() == 0) throw std::runtime_error("Failed to load gltf: no scene");
        if (sceneIdx < 0 || (size_t)sceneIdx >= scenes.size()) sceneIdx = 0;
        const Json& scene = scenes.arr[(size_t)sceneIdx];
        if (const Json* nodes = scene.find("nodes"))
            for (const Json& n : nodes->arr) load_node(n.as_int(-1), Xform{}, nullptr, 0);
        load_environment(scene);
    }
};

// DynamicScene::soupifyScene (scene/dynamic/DynamicScene.cpp:17-106): every mesh instance is appended to one
// mesh in world space (TriangleMesh::appendMeshes + applyTransform, shape/TriangleMesh.cpp:295-326: vertices
// as points, normals as plain vectors, not renormalised), materials become per-triangle ranges (CompositeMaterial)
void soupify(rtgpu_gltf& g)
{
    auto merged = std::make_unique<rtgpu_gltf::Mesh>();
    bool anyUV = false;
    for (const auto& inst : g.instances)
        if (!g.meshes[(size_t)inst.shape]->texcoords.empty()) anyUV = true;
    for (const auto& inst : g.instances) {
        const rtgpu_gltf::Mesh& m = *g.meshes[(size_t)inst.shape];
        Mat4 M;
        std::memcpy(M.m, inst.transform, sizeof M.m);
        const uint32_t base = (uint32_t)(merged->vertices.size() / 3);
        const size_t nv = m.vertices.size() / 3;
        for (size_t v = 0; v < nv; ++v) {
            float p[3], n[3];
            xf_point(M, &m.vertices[3 * v], p);
            xf_vector(M, &m.normals[3 * v], n);
            merged->vertices.insert(merged->vertices.end(), p, p + 3);
            merged->normals.insert(merged->normals.end(), n, n + 3);
            if (anyUV) {
                const float uv[2] = {m.texcoords.empty() ? 0.0f : m.texcoords[2 * v], m.texcoords.empty() ? 0.0f : m.texcoords[2 * v + 1]};
                merged->texcoords.insert(merged->texcoords.end(), uv, uv + 2);
            }
        }
        for (uint32_t i : m.indices) merged->indices.push_back(base + i);
        merged->triMaterial.insert(merged->triMaterial.end(), m.indices.size() / 3, inst.material);
    }
    if (g.instances.empty()) return;
    rtgpu_instance_desc one{};
    one.shape = 0;
    one.material = g.instances[0].material;
    const Mat4 id = identity();
    std::memcpy(one.transform, id.m, sizeof one.transform);
    std::memcpy(one.inverse, id.m, sizeof one.inverse);
    g.meshes.clear();
    g.meshes.push_back(std::move(merged));
    g.instances.assign(1, one);
}

void finish(rtgpu_gltf& g)
{
    for (const auto& m : g.meshes) {
        rtgpu_mesh_desc d{};
        d.vertices = m->vertices.data();
        d.normals = m->normals.data();
        d.indices = m->indices.data();
        d.n_vertices = (uint32_t)(m->vertices.size() / 3);
        d.n_triangles = (uint32_t)(m->indices.size() / 3);
        d.tri_material = m->triMaterial.empty() ? nullptr : m->triMaterial.data();
        d.texcoords = m->texcoords.empty() ? nullptr : m->texcoords.data();
        g.meshDescs.push_back(d);
    }
    for (const auto& t : g.textures) {
        rtgpu_texture d{};
        d.rgba = t->rgba.data();
        d.width = t->width;
        d.height = t->height;
        d.gamma = t->gamma;
        g.texDescs.push_back(d);
    }
    rtgpu_scene_desc& d = g.desc;
    d = rtgpu_scene_desc{};
    d.meshes = g.meshDescs.data(); d.n_meshes = (uint32_t)g.meshDescs.size();
    d.instances = g.instances.data(); d.n_instances = (uint32_t)g.instances.size();
    d.materials = g.materials.data(); d.n_materials = (uint32_t)g.materials.size();
    d.point_lights = g.pointLights.data(); d.n_point_lights = (uint32_t)g.pointLights.size();
    d.area_lights = g.areaLights.data(); d.n_area_lights = (uint32_t)g.areaLights.size();
    d.directional_lights = g.dirLights.data(); d.n_directional_lights = (uint32_t)g.dirLights.size();
    d.textures = g.texDescs.data(); d.n_textures = (uint32_t)g.texDescs.size();
    d.has_environment = g.hasEnv;
    d.env_map = g.envMap.empty() ? nullptr : g.envMap.data();
    d.env_map_width = g.envW;
    d.env_map_height = g.envH;
    std::memcpy(d.env_color, g.envColor, 12);
    d.env_intensity = g.envIntensity;
}

} // namespace

extern "C" {

void rtgpu_gltf_set_workdir(const char* directory) { g_workdir = directory ? directory : ""; }

int rtgpu_gltf_load(const char* glb_path, float image_aspect_ratio, int soupify_scene, rtgpu_gltf** out)
{
    if (!glb_path || !out) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    auto g = std::make_unique<rtgpu_gltf>();
    try {
        Loader loader(*g, image_aspect_ratio);
        loader.run(glb_path);
        if (soupify_scene) soupify(*g);
        finish(*g);
    } catch (const Unsupported& e) {
        return rt::set_error(RTGPU_ERR_UNSUPPORTED, e.what());
    } catch (const std::bad_alloc&) {
        return rt::set_error(RTGPU_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception& e) {
        return rt::set_error(RTGPU_ERR_IO, e.what());
    }
    *out = g.release();
    return RTGPU_OK;
}

const rtgpu_scene_desc* rtgpu_gltf_scene_desc(const rtgpu_gltf* g) { return g ? &g->desc : nullptr; }

uint32_t rtgpu_gltf_camera_count(const rtgpu_gltf* g) { return g ? (uint32_t)g->cameras.size() : 0u; }

int rtgpu_gltf_camera(const rtgpu_gltf* g, rtgpu_camera* out)
{
    if (!g || !out) return rt::set_error(RTGPU_ERR_INVALID_ARGUMENT, "null argument");
    // Renderer::findCamera, renderer/Renderer.cpp:304-320: the last camera marked as main, else the first
    const rtgpu_gltf::Camera* cam = nullptr;
    for (const auto& c : g->cameras)
        if (c.isMain) cam = &c;
    if (!cam) {
        if (g->cameras.empty()) return rt::set_error(RTGPU_ERR_NO_CAMERA, "No camera in scene.");
        cam = &g->cameras[0];
    }
    const int rc = rtgpu_camera_from_transform(cam->xFov, cam->transform.m, out);
    if (rc != RTGPU_OK) return rc;
    out->focal_distance = cam->focalDistance;
    out->aperture = cam->aperture;
    return RTGPU_OK;
}

void rtgpu_gltf_free(rtgpu_gltf* g) { delete g; }

} // extern "C"
