// scene_config.cpp -- cameras.json / *_vr.json readers, config discovery, sRGB + 8-bit presentation,
// PNG / PPM writers, and their C-ABI entry points (include/msplat.h).  See scene_config.hpp for the
// reference lines each piece follows.  Host only; no third-party code.
#include "scene_config.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <algorithm>

#include "../../include/msplat.h"

namespace {

// ---- a small JSON reader: objects, arrays, numbers, strings, true/false/null -----------------------
struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::map<std::string, JValue> obj;

    const JValue& at(size_t i) const
    {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    const JValue& at(const std::string& k) const
    {
        if (kind != Object) throw std::runtime_error("json: not an object");
        auto it = obj.find(k);
        if (it == obj.end()) throw std::runtime_error("json: key '" + k + "' not found");
        return it->second;
    }
    float f() const
    {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return (float)num;
    }
};

class JParser
{
public:
    explicit JParser(const std::string& text) : s(text) {}
    JValue Parse()
    {
        JValue v = Value();
        Ws();
        if (p != s.size()) throw std::runtime_error("json: trailing characters");
        return v;
    }

private:
    const std::string& s;
    size_t p = 0;
    void Ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
    char Peek() { Ws(); if (p >= s.size()) throw std::runtime_error("json: unexpected end"); return s[p]; }
    void Expect(char c) { if (Peek() != c) throw std::runtime_error(std::string("json: expected '") + c + "'"); ++p; }
    JValue Value()
    {
        const char c = Peek();
        JValue v;
        if (c == '{') {
            ++p;
            v.kind = JValue::Object;
            if (Peek() == '}') { ++p; return v; }
            for (;;) {
                JValue k = StringValue();
                Expect(':');
                v.obj[k.str] = Value();
                if (Peek() == ',') { ++p; continue; }
                Expect('}');
                return v;
            }
        }
        if (c == '[') {
            ++p;
            v.kind = JValue::Array;
            if (Peek() == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(Value());
                if (Peek() == ',') { ++p; continue; }
                Expect(']');
                return v;
            }
        }
        if (c == '"') return StringValue();
        if (s.compare(p, 4, "true") == 0) { p += 4; v.kind = JValue::Bool; v.b = true; return v; }
        if (s.compare(p, 5, "false") == 0) { p += 5; v.kind = JValue::Bool; return v; }
        if (s.compare(p, 4, "null") == 0) { p += 4; return v; }
        char* end = nullptr;
        v.num = std::strtod(s.c_str() + p, &end);
        if (end == s.c_str() + p) throw std::runtime_error("json: unexpected character");
        p = (size_t)(end - s.c_str());
        v.kind = JValue::Number;
        return v;
    }
    JValue StringValue()
    {
        Expect('"');
        JValue v;
        v.kind = JValue::String;
        while (p < s.size() && s[p] != '"') {
            if (s[p] == '\\' && p + 1 < s.size()) {
                const char e = s[p + 1];
                v.str += (e == 'n') ? '\n' : (e == 't') ? '\t' : e;     // \uXXXX is kept verbatim (not needed here)
                p += 2;
            } else {
                v.str += s[p++];
            }
        }
        if (p >= s.size()) throw std::runtime_error("json: unterminated string");
        ++p;
        return v;
    }
};

bool ReadAll(const std::string& path, std::string& out)
{
    std::ifstream f(path, std::ios::binary);
    if (f.fail()) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

// ---- PNG helpers -------------------------------------------------------------------------------------
uint32_t Crc32(const uint8_t* d, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ d[i]) & 255] ^ (crc >> 8);
    return ~crc;
}

void Be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }

void Chunk(std::vector<uint8_t>& png, const char type[4], const std::vector<uint8_t>& data)
{
    Be32(png, (uint32_t)data.size());
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    Be32(png, Crc32(png.data() + start, png.size() - start));
}

}  // namespace

// ------------------------------------------------------------------------------------------
// CamerasConfig
// ------------------------------------------------------------------------------------------

bool CamerasConfig::ImportJson(const std::string& jsonFilename)
{
    std::string text;
    if (!ReadAll(jsonFilename, text)) return false;
    try {
        const JValue data = JParser(text).Parse();
        if (data.kind != JValue::Array) throw std::runtime_error("json: top-level array expected");
        for (const JValue& o : data.arr) {
            (void)o.at("id").f();
            const JValue& jp = o.at("position");
            const JValue& jr = o.at("rotation");
            const float width = o.at("width").f(), height = o.at("height").f();
            const float fx = o.at("fx").f();
            (void)o.at("fy").f();
            Camera c;
            // camerasconfig.cpp:47-48 (both angles from fx, as the reference does)
            c.fov[0] = 2.0f * atanf(width / (2.0f * fx));
            c.fov[1] = 2.0f * atanf(height / (2.0f * fx));
            // camerasconfig.cpp:38-53: the JSON rotation is row-major; columns 1 and 2 are negated so that
            // -z is forward and +y is up; the position is the translation column
            for (int r = 0; r < 3; ++r) {
                c.mat[0 * 4 + r] = jr.at(r).at(0).f();
                c.mat[1 * 4 + r] = -jr.at(r).at(1).f();
                c.mat[2 * 4 + r] = -jr.at(r).at(2).f();
                c.mat[3 * 4 + r] = jp.at(r).f();
            }
            c.mat[3] = c.mat[7] = c.mat[11] = 0.0f;
            c.mat[15] = 1.0f;
            cameraVec.push_back(c);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "[msplat][E] CamerasConfig::ImportJson exception: %s\n", e.what());
        return false;
    }
    return true;
}

void CamerasConfig::EstimateFloorPlane(float normalOut[3], float posOut[3]) const
{
    // camerasconfig.cpp:69-95: average camera up vector; plane offset = average of dot(pos, up)
    float up[3] = {0, 0, 0};
    if (cameraVec.empty()) {
        normalOut[0] = 0; normalOut[1] = 1; normalOut[2] = 0;
        posOut[0] = posOut[1] = posOut[2] = 0;
        return;
    }
    const float wgt = 1.0f / (float)cameraVec.size();
    for (const Camera& c : cameraVec)
        for (int k = 0; k < 3; ++k) up[k] += wgt * c.mat[4 + k];
    const float len = std::sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]);
    if (len > 0.0f) { up[0] /= len; up[1] /= len; up[2] /= len; }
    else { up[0] = 0; up[1] = 1; up[2] = 0; }
    float dist = 0.0f;
    for (const Camera& c : cameraVec) dist += wgt * (c.mat[12] * up[0] + c.mat[13] * up[1] + c.mat[14] * up[2]);
    for (int k = 0; k < 3; ++k) {
        normalOut[k] = up[k];
        posOut[k] = up[k] * dist;
    }
}

// ------------------------------------------------------------------------------------------
// VrConfig
// ------------------------------------------------------------------------------------------

VrConfig::VrConfig()
{
    std::memset(floorMat, 0, sizeof(floorMat));
    floorMat[0] = floorMat[5] = floorMat[10] = floorMat[15] = 1.0f;
}

void VrConfig::SetFloorMat(const float floorMatIn[16]) { std::memcpy(floorMat, floorMatIn, sizeof(floorMat)); }

bool VrConfig::ImportJson(const std::string& jsonFilename)
{
    std::string text;
    if (!ReadAll(jsonFilename, text)) return false;
    try {
        const JValue obj = JParser(text).Parse();
        const JValue& m = obj.at("floorMat");       // vrconfig.cpp:31-35: rows of the matrix
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) floorMat[c * 4 + r] = m.at(r).at(c).f();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "[msplat][E] VrConfig::ImportJson exception: %s\n", e.what());
        return false;
    }
    return true;
}

bool VrConfig::ExportJson(const std::string& jsonFilename) const
{
    std::ofstream f(jsonFilename);
    if (f.fail()) return false;
    f << "{\n    \"floorMat\": [";
    for (int r = 0; r < 4; ++r) {
        f << "[";
        for (int c = 0; c < 4; ++c) f << floorMat[c * 4 + r] << (c < 3 ? ", " : "");
        f << (r < 3 ? "], " : "]]");
    }
    f << "\n}";
    return true;
}

// ------------------------------------------------------------------------------------------
// config discovery (app.cpp:89-142)
// ------------------------------------------------------------------------------------------

std::string FindConfigFile(const std::string& plyFilename, const std::string& configFilename)
{
    namespace fs = std::filesystem;
    std::error_code ec;
    const fs::path ply(plyFilename);
    if (!fs::is_regular_file(ply, ec)) {
        std::fprintf(stderr, "[msplat][E] PLY file does not exist or is not a file: \"%s\"\n", plyFilename.c_str());
        return "";
    }
    fs::path dir = ply.parent_path();
    for (int i = 0; i < 3; ++i) {      // the PLY's directory, its parent and grandparent
        const fs::path cand = dir / configFilename;
        if (fs::is_regular_file(cand, ec)) return cand.string();
        if (!dir.has_parent_path()) break;
        dir = dir.parent_path();
    }
    return "";
}

std::string MakeVrConfigFilename(const std::string& plyFilename)
{
    const std::filesystem::path ply(plyFilename);
    return (ply.parent_path() / (ply.stem().string() + "_vr.json")).string();
}

// ------------------------------------------------------------------------------------------
// presentation
// ------------------------------------------------------------------------------------------

float LinearToSRGB(float linear)
{
    return linear <= 0.0031308f ? 12.92f * linear : 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
}

float SRGBToLinear(float srgb) { return srgb <= 0.04045f ? srgb / 12.92f : powf((srgb + 0.055f) / 1.055f, 2.4f); }

void PresentRGBA8(const float* rgba, int width, int height, bool flipY, bool encodeSRGB, uint8_t* out)
{
    // an RGBA8 target clamps to [0,1] and rounds to nearest (the reference's default back buffer)
    for (int y = 0; y < height; ++y) {
        const float* src = rgba + (size_t)(flipY ? height - 1 - y : y) * width * 4;
        uint8_t* dst = out + (size_t)y * width * 4;
        for (int x = 0; x < width * 4; ++x) {
            float v = src[x];
            if (encodeSRGB && (x & 3) != 3) v = LinearToSRGB(v);
            v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
            dst[x] = (uint8_t)(v * 255.0f + 0.5f);
        }
    }
}

bool WritePNG(const std::string& filename, const uint8_t* rgba8, int width, int height)
{
    if (width <= 0 || height <= 0) return false;
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    Be32(ihdr, (uint32_t)width);
    Be32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, 6, 0, 0, 0};   // 8-bit RGBA, deflate, no filter method, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    Chunk(png, "IHDR", ihdr);
    // zlib stream of "stored" deflate blocks (no compression: keeps the writer dependency-free)
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * ((size_t)width * 4 + 1));
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);   // filter type None
        raw.insert(raw.end(), rgba8 + (size_t)y * width * 4, rgba8 + (size_t)(y + 1) * width * 4);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 255)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 255)); z.push_back((uint8_t)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
        if (n == 0) break;
    }
    Be32(z, (b << 16) | a);
    Chunk(png, "IDAT", z);
    Chunk(png, "IEND", {});
    std::ofstream f(filename, std::ios::binary);
    if (f.fail()) return false;
    f.write(reinterpret_cast<const char*>(png.data()), (std::streamsize)png.size());
    return !f.fail();
}

bool WritePPM(const std::string& filename, const uint8_t* rgba8, int width, int height)
{
    std::ofstream f(filename, std::ios::binary);
    if (f.fail()) return false;
    f << "P6\n" << width << " " << height << "\n255\n";
    for (size_t i = 0; i < (size_t)width * height; ++i) f.write(reinterpret_cast<const char*>(rgba8 + i * 4), 3);
    return !f.fail();
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int msplat_cameras_import_json(const char* path, float* mats16_out, float* fovs2_out, uint32_t cap, uint32_t* count_out)
{
    if (!path || !count_out) return MSPLAT_ERR_INVALID_ARG;
    CamerasConfig cc;
    if (!cc.ImportJson(path)) return MSPLAT_ERR_IO;
    *count_out = (uint32_t)cc.GetNumCameras();
    const uint32_t n = std::min<uint32_t>(cap, *count_out);
    for (uint32_t i = 0; i < n; ++i) {
        if (mats16_out) std::memcpy(mats16_out + 16 * i, cc.GetCameraVec()[i].mat, 64);
        if (fovs2_out) std::memcpy(fovs2_out + 2 * i, cc.GetCameraVec()[i].fov, 8);
    }
    return MSPLAT_OK;
}

int msplat_cameras_floor_plane(const char* path, float normal_out[3], float pos_out[3])
{
    if (!path || !normal_out || !pos_out) return MSPLAT_ERR_INVALID_ARG;
    CamerasConfig cc;
    if (!cc.ImportJson(path)) return MSPLAT_ERR_IO;
    cc.EstimateFloorPlane(normal_out, pos_out);
    return MSPLAT_OK;
}

int msplat_vrconfig_import_json(const char* path, float floor_mat_out[16])
{
    if (!path || !floor_mat_out) return MSPLAT_ERR_INVALID_ARG;
    VrConfig vc;
    if (!vc.ImportJson(path)) return MSPLAT_ERR_IO;
    std::memcpy(floor_mat_out, vc.GetFloorMat(), 64);
    return MSPLAT_OK;
}

int msplat_vrconfig_export_json(const char* path, const float floor_mat[16])
{
    if (!path || !floor_mat) return MSPLAT_ERR_INVALID_ARG;
    VrConfig vc;
    vc.SetFloorMat(floor_mat);
    return vc.ExportJson(path) ? MSPLAT_OK : MSPLAT_ERR_IO;
}

int msplat_find_config_file(const char* ply_path, const char* config_name, char* out, uint32_t cap)
{
    if (!ply_path || !config_name || !out || cap == 0) return MSPLAT_ERR_INVALID_ARG;
    const std::string r = FindConfigFile(ply_path, config_name);
    if (r.empty() || r.size() + 1 > cap) { out[0] = 0; return r.empty() ? MSPLAT_ERR_IO : MSPLAT_ERR_INVALID_ARG; }
    std::memcpy(out, r.c_str(), r.size() + 1);
    return MSPLAT_OK;
}

int msplat_write_image(const char* path, const float* rgba, int width, int height, int encode_srgb)
{
    if (!path || !rgba || width <= 0 || height <= 0) return MSPLAT_ERR_INVALID_ARG;
    std::vector<uint8_t> px((size_t)width * height * 4);
    PresentRGBA8(rgba, width, height, /*flipY=*/true, encode_srgb != 0, px.data());
    const std::string p(path);
    const bool ppm = p.size() > 4 && p.compare(p.size() - 4, 4, ".ppm") == 0;
    return (ppm ? WritePPM(p, px.data(), width, height) : WritePNG(p, px.data(), width, height)) ? MSPLAT_OK : MSPLAT_ERR_IO;
}

}  // extern "C"
