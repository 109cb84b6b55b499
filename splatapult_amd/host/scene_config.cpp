// scene_config.cpp -- cameras.json / *_vr.json readers, config discovery, sRGB + 8-bit presentation,
// PNG / PPM writers, and their C-ABI entry points (include/msplat.h).  See scene_config.hpp for the
// reference lines each piece follows.  Host only; no third-party code.
#include "scene_config.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
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
    int depth = 0;                         // nesting guard: the configs this reads are 3 levels deep
    struct DepthGuard {
        int& d;
        explicit DepthGuard(int& dd) : d(dd) { if (++d > 64) throw std::runtime_error("json: nesting deeper than 64"); }
        ~DepthGuard() { --d; }
    };
    JValue Value()
    {
        DepthGuard guard(depth);
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

// ---- PNG reader: zlib inflate (RFC 1950/1951) + the five PNG filters; 8-bit gray / gray+alpha / RGB / RGBA,
// non-interlaced -- the formats the reference's Image::Load accepts (core/image.cpp:72-101) ----------------
namespace {

struct BitReader
{
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t acc = 0;
    int cnt = 0;
    bool fail = false;
    uint32_t Bits(int k)
    {
        while (cnt < k) {
            if (pos >= n) { fail = true; return 0; }
            acc |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        acc = (k == 32) ? 0 : (acc >> k);
        cnt -= k;
        return v;
    }
};

// canonical Huffman decoder (counts per length + sorted symbols)
struct Huffman
{
    uint16_t count[16] = {0};
    std::vector<uint16_t> symbol;
    bool Build(const uint8_t* lengths, int n)
    {
        for (auto& c : count) c = 0;
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;          // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len];
        symbol.assign(n, 0);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int Decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= (int)br.Bits(1);
            if (br.fail) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// max_out: the caller knows the exact decoded size (PNG: h * (1 + w * channels)); a stream that grows past it is
// rejected at once instead of expanding a crafted file to gigabytes first
bool Inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t max_out)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n < 6 || (src[0] & 15) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 32)) return false;   // zlib header
    BitReader br{src + 2, n - 2};
    for (;;) {
        const uint32_t final = br.Bits(1), type = br.Bits(2);
        if (br.fail) return false;
        if (type == 0) {
            br.acc = 0; br.cnt = 0;                                       // to the byte boundary
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > br.n || out.size() + len > max_out) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                lit.Build(lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                dist.Build(lengths, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)br.Bits(5) + 257, ndist = (int)br.Bits(5) + 1, ncode = (int)br.Bits(4) + 4;
                if (br.fail || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.Bits(3);
                Huffman lencode;
                if (!lencode.Build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.Decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                    uint8_t prev = 0;
                    int rep;
                    if (sym == 16) { if (idx == 0) return false; prev = lengths[idx - 1]; rep = 3 + (int)br.Bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.Bits(3);
                    else rep = 11 + (int)br.Bits(7);
                    if (br.fail || idx + rep > nlen + ndist) return false;
                    while (rep--) lengths[idx++] = prev;
                }
                if (lengths[256] == 0 || !lit.Build(lengths, nlen) || !dist.Build(lengths + nlen, ndist)) return false;
            }
            for (;;) {
                int sym = lit.Decode(br);
                if (sym < 0) return false;
                if (sym < 256) {
                    if (out.size() >= max_out) return false;
                    out.push_back((uint8_t)sym);
                    continue;
                }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) return false;
                const size_t len = lbase[sym] + br.Bits(lext[sym]);
                const int ds = dist.Decode(br);
                if (ds < 0 || ds >= 30) return false;
                const size_t d = dbase[ds] + br.Bits(dext[ds]);
                if (br.fail || d > out.size() || out.size() + len > max_out) return false;
                const size_t from = out.size() - d;
                for (size_t k = 0; k < len; ++k) out.push_back(out[from + k]);
            }
        } else {
            return false;
        }
        if (final) return true;
    }
}

}  // namespace

bool ReadPNG(const std::string& filename, std::vector<uint8_t>& rgba8, int& width, int& height)
{
    std::string file;
    if (!ReadAll(filename, file) || file.size() < 8 + 25 || std::memcmp(file.data(), "\x89PNG\r\n\x1a\n", 8) != 0) return false;
    const uint8_t* b = reinterpret_cast<const uint8_t*>(file.data());
    auto be32 = [](const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
    size_t pos = 8;
    std::vector<uint8_t> idat;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t n = be32(b + pos);
        if (pos + 12 + (size_t)n > file.size()) return false;
        const uint8_t* typ = b + pos + 4;
        const uint8_t* d = typ + 4;
        if (Crc32(typ, n + 4) != be32(d + n)) return false;
        if (!std::memcmp(typ, "IHDR", 4) && n == 13) {
            w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (!std::memcmp(typ, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + n);
        } else if (!std::memcmp(typ, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)n;
    }
    const int channels = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (w == 0 || h == 0 || w > 16384 || h > 16384 || depth != 8 || channels == 0 || interlace != 0) return false;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + (size_t)w * channels));
    const size_t expect = (size_t)h * (1 + (size_t)w * channels);
    if (!Inflate(idat.data(), idat.size(), raw, expect) || raw.size() != expect) return false;
    const size_t bpp = channels, rowBytes = (size_t)w * channels;
    std::vector<uint8_t> cur(rowBytes), prev(rowBytes, 0);
    rgba8.assign((size_t)w * h * 4, 255);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* in = raw.data() + (size_t)y * (rowBytes + 1);
        const int filter = in[0];
        if (filter > 4) return false;
        for (size_t i = 0; i < rowBytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (filter == 1) pred = a;
            else if (filter == 2) pred = bb;
            else if (filter == 3) pred = (a + bb) >> 1;
            else if (filter == 4) {
                const int pp = a + bb - c, pa = std::abs(pp - a), pb = std::abs(pp - bb), pc = std::abs(pp - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
            }
            cur[i] = (uint8_t)(in[1 + i] + pred);
        }
        uint8_t* o = rgba8.data() + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t* s = cur.data() + (size_t)x * channels;
            if (channels <= 2) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = s[0]; if (channels == 2) o[4 * x + 3] = s[1]; }
            else { o[4 * x] = s[0]; o[4 * x + 1] = s[1]; o[4 * x + 2] = s[2]; if (channels == 4) o[4 * x + 3] = s[3]; }
        }
        std::swap(cur, prev);
    }
    width = (int)w;
    height = (int)h;
    return true;
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

// PNG -> RGBA8, top row first.  rgba8_out may be NULL to query the size; returns MSPLAT_ERR_INVALID_ARG when cap
// (bytes) is too small.
int msplat_read_image(const char* path, uint8_t* rgba8_out, uint64_t cap, uint32_t* width_out, uint32_t* height_out)
{
    if (!path || !width_out || !height_out) return MSPLAT_ERR_INVALID_ARG;
    std::vector<uint8_t> px;
    int w = 0, h = 0;
    if (!ReadPNG(path, px, w, h)) return MSPLAT_ERR_IO;
    *width_out = (uint32_t)w;
    *height_out = (uint32_t)h;
    if (!rgba8_out) return MSPLAT_OK;
    if (cap < px.size()) return MSPLAT_ERR_INVALID_ARG;
    std::memcpy(rgba8_out, px.data(), px.size());
    return MSPLAT_OK;
}

}  // extern "C"
