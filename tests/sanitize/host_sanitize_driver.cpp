// host_sanitize_driver.cpp -- ASan / UBSan job for the host half of libmsplat.so (SURVEY.md section 5's sanitizer
// counterpart): the hand-written parsers behind a C ABI that is documented as never throwing --
// PLY (gaussian_scene.cpp, point_scene.cpp), JSON + PNG inflate / deflate (scene_config.cpp).
//
//   make sanitize      builds the three host sources + this file with -fsanitize=address,undefined (no HIP needed)
//                      and runs it on tests/golden/* plus a set of hostile inputs generated here.
//
// The device entry points the host code forwards to are stubbed (they only receive what the parsers produced, so
// the stubs also check that the buffers they are handed are fully readable).  Exit code 0 = every call returned
// the expected status and the sanitizers stayed silent.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/msplat.h"
#include "../../include/msplat_debug.h"

static int g_fail = 0;
static uint64_t g_stub_bytes = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

static uint64_t touch(const void* p, uint64_t bytes)
{
    // read every byte the parser claims to hand over: ASan reports an over-claim
    const volatile uint8_t* b = static_cast<const volatile uint8_t*>(p);
    uint64_t s = 0;
    for (uint64_t i = 0; i < bytes; ++i) s += b[i];
    g_stub_bytes += bytes;
    return s;
}

extern "C" {
// ---- stubs of the device half (msplat_device.hip) ----
int msplat_upload_cloud(msplat_ctx*, const void* aos, uint64_t n, uint32_t stride_bytes, const msplat_attr_offsets* off, int full_sh)
{
    if (n) touch(aos, n * stride_bytes);
    return (off && (full_sh == 0 || full_sh == 1)) ? MSPLAT_OK : MSPLAT_ERR_INVALID_ARG;
}
int msplat_upload_ply_vertices(msplat_ctx*, const void* vertices, uint64_t n, const msplat_ply_layout* layout, int)
{
    if (n && layout) touch(vertices, n * layout->vertex_size);
    return layout ? MSPLAT_OK : MSPLAT_ERR_INVALID_ARG;
}
int msplat_upload_points(msplat_ctx*, const void* aos, uint64_t n, uint32_t stride_bytes, uint32_t, uint32_t)
{
    if (n) touch(aos, n * stride_bytes);
    return MSPLAT_OK;
}
}

// the stubs never dereference the context: any non-null pointer gets past the host code's NULL checks
static msplat_ctx* const kCtx = reinterpret_cast<msplat_ctx*>(uintptr_t(0x1000));
static std::string g_dir;
static std::string tmp(const char* name) { return g_dir + "/" + name; }
static void write_file(const std::string& path, const std::string& bytes)
{
    std::ofstream f(path, std::ios::binary);
    f.write(bytes.data(), (std::streamsize)bytes.size());
}

static std::string ply_header(const char* count, const std::vector<std::string>& props, const char* type = "float")
{
    std::string h = "ply\nformat binary_little_endian 1.0\ncomment made by the sanitizer driver\nelement vertex ";
    h += count;
    h += "\n";
    for (const auto& p : props) h += std::string("property ") + type + " " + p + "\n";
    h += "end_header\n";
    return h;
}

static std::vector<std::string> gaussian_props(bool full_sh)
{
    std::vector<std::string> p = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"};
    if (full_sh)
        for (int i = 0; i < 45; ++i) p.push_back("f_rest_" + std::to_string(i));
    for (const char* s : {"opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"}) p.push_back(s);
    return p;
}

static uint32_t crc32_of(const std::string& s)
{
    uint32_t c = 0xFFFFFFFFu;
    for (unsigned char ch : s) {
        c ^= ch;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return ~c;
}
static std::string be32(uint32_t v) { return std::string{(char)(v >> 24), (char)(v >> 16), (char)(v >> 8), (char)v}; }
static std::string png_chunk(const std::string& tag, const std::string& data)
{
    return be32((uint32_t)data.size()) + tag + data + be32(crc32_of(tag + data));
}
// zlib stream of stored (uncompressed) deflate blocks
static std::string zlib_stored(const std::string& raw)
{
    std::string z = "\x78\x01";
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z += (char)(last ? 1 : 0);
        z += (char)(n & 255); z += (char)(n >> 8); z += (char)(~n & 255); z += (char)((~n >> 8) & 255);
        z += raw.substr(pos, n);
        pos += n;
    } while (pos < raw.size());
    uint32_t a = 1, b = 0;
    for (unsigned char ch : raw) { a = (a + ch) % 65521u; b = (b + a) % 65521u; }
    z += be32((b << 16) | a);
    return z;
}

static void ply_cases(const std::string& golden)
{
    // the reference's own fixture, both importers, both SH options, and the GPU-ingest front end (stubbed upload)
    for (int sh = 0; sh < 2; ++sh) {
        msplat_cloud* c = msplat_cloud_create(sh);
        EXPECT(msplat_cloud_import_ply(c, (golden + "/test.ply").c_str()) == MSPLAT_OK);
        EXPECT(msplat_cloud_num_gaussians(c) == 16);
        EXPECT(msplat_cloud_stride(c) == (msplat_cloud_has_full_sh(c) ? 244u : 100u));
        msplat_attr_offsets off;
        EXPECT(msplat_cloud_attr_offsets(c, &off) == MSPLAT_OK);
        EXPECT(msplat_upload_gaussian_cloud(kCtx, c) == MSPLAT_OK);
        EXPECT(msplat_cloud_export_ply(c, tmp("export.ply").c_str()) == MSPLAT_OK);
        msplat_cloud* c2 = msplat_cloud_create(sh);
        EXPECT(msplat_cloud_import_ply(c2, tmp("export.ply").c_str()) == MSPLAT_OK);
        EXPECT(msplat_cloud_num_gaussians(c2) == 16);
        const float origin[3] = {0, 0, 0};
        EXPECT(msplat_cloud_prune(c2, origin, 5) == MSPLAT_OK && msplat_cloud_num_gaussians(c2) == 5);
        EXPECT(msplat_cloud_prune(c2, origin, 500) == MSPLAT_OK);
        EXPECT(msplat_cloud_init_debug(c2) == MSPLAT_OK);
        msplat_cloud_destroy(c2);
        msplat_cloud_destroy(c);
        EXPECT(msplat_upload_ply(kCtx, (golden + "/test.ply").c_str(), sh) == MSPLAT_OK);
    }
    // hostile vertex counts: overflow, allocation failure; the C ABI returns an error, never throws
    for (const char* count : {"9223372036854775807", "4611686018427387904", "1152921504606846976", "18446744073709551616", "-1", "abc", ""}) {
        write_file(tmp("huge.ply"), ply_header(count, {"x", "y", "z"}) + std::string(36, '\0'));
        msplat_cloud* c = msplat_cloud_create(1);
        EXPECT(msplat_cloud_import_ply(c, tmp("huge.ply").c_str()) != MSPLAT_OK);
        msplat_cloud_destroy(c);
        msplat_points* p = msplat_points_create(0);
        EXPECT(msplat_points_import_ply(p, tmp("huge.ply").c_str()) != MSPLAT_OK);
        msplat_points_destroy(p);
        EXPECT(msplat_upload_ply(kCtx, tmp("huge.ply").c_str(), 1) != MSPLAT_OK);
    }
    // short file: 3 vertices announced, 1 present -- accepted like the reference (tail zero-filled, ply.cpp:80-84)
    {
        const auto props = gaussian_props(false);
        std::string body(props.size() * 4, '\0');
        const float one = 1.0f;
        std::memcpy(&body[0], &one, 4);
        write_file(tmp("short.ply"), ply_header("3", props) + body);
        msplat_cloud* c = msplat_cloud_create(0);
        EXPECT(msplat_cloud_import_ply(c, tmp("short.ply").c_str()) == MSPLAT_OK && msplat_cloud_num_gaussians(c) == 3);
        EXPECT(msplat_upload_gaussian_cloud(kCtx, c) == MSPLAT_OK);
        msplat_cloud_destroy(c);
        EXPECT(msplat_upload_ply(kCtx, tmp("short.ply").c_str(), 0) == MSPLAT_OK);
    }
    // full-SH file, double-typed properties, unknown types, missing end_header, truncated header, wrong magic, no vertex element
    {
        const auto props = gaussian_props(true);
        write_file(tmp("full.ply"), ply_header("2", props) + std::string(2 * props.size() * 4, '\x01'));
        msplat_cloud* c = msplat_cloud_create(1);
        EXPECT(msplat_cloud_import_ply(c, tmp("full.ply").c_str()) == MSPLAT_OK && msplat_cloud_has_full_sh(c));
        msplat_cloud_destroy(c);
        write_file(tmp("dbl.ply"), ply_header("2", {"x", "y", "z"}, "double") + std::string(2 * 24, '\0'));
        msplat_points* p = msplat_points_create(1);
        (void)msplat_points_import_ply(p, tmp("dbl.ply").c_str());
        EXPECT(msplat_upload_point_cloud(kCtx, p) == MSPLAT_OK || msplat_points_num(p) == 0);
        msplat_points_init_debug(p);
        EXPECT(msplat_points_export_ply(p, tmp("pts.ply").c_str()) == MSPLAT_OK);
        EXPECT(msplat_points_import_ply(p, tmp("pts.ply").c_str()) == MSPLAT_OK && msplat_points_num(p) > 0);
        EXPECT(msplat_upload_point_cloud(kCtx, p) == MSPLAT_OK);
        msplat_points_destroy(p);
    }
    const std::string bad_headers[] = {
        "", "ply", "ply\n", "plx\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nend_header\n",
        "ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n",
        "ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty quux x\nend_header\n",
        "ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\n",
        "ply\nformat binary_little_endian 1.0\nelement face 1\nproperty float x\nend_header\n",
        "ply\nformat binary_little_endian 1.0\nelement vertex\nproperty float\nend_header\n",
        std::string("ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float ") + std::string(70000, 'x') + "\nend_header\n",
        std::string("ply\n") + std::string(100000, '\n'),
        std::string("ply\0format\0", 11)};
    for (const auto& h : bad_headers) {
        write_file(tmp("bad.ply"), h + std::string(16, '\0'));
        msplat_cloud* c = msplat_cloud_create(1);
        (void)msplat_cloud_import_ply(c, tmp("bad.ply").c_str());          // either outcome; must not crash or leak
        msplat_cloud_destroy(c);
        msplat_points* p = msplat_points_create(0);
        (void)msplat_points_import_ply(p, tmp("bad.ply").c_str());
        msplat_points_destroy(p);
        (void)msplat_upload_ply(kCtx, tmp("bad.ply").c_str(), 1);
    }
    EXPECT(msplat_cloud_import_ply(nullptr, "x") == MSPLAT_ERR_INVALID_ARG);
    msplat_cloud* c = msplat_cloud_create(1);
    EXPECT(msplat_cloud_import_ply(c, tmp("does_not_exist.ply").c_str()) != MSPLAT_OK);
    // attribute arrays -> cloud (synthetic scenes), incl. zero quaternions and NaNs
    {
        const uint64_t n = 257;
        std::vector<float> xyz(n * 3, 0.5f), dc(n * 3, 0.1f), rest(n * 45, 0.01f), op(n, 0.3f), ls(n * 3, -3.0f), rot(n * 4, 0.0f);
        for (uint64_t i = 0; i < n; ++i) rot[i * 4 + (i % 4)] = (i % 7) ? 1.0f : 0.0f;
        xyz[5] = NAN; op[9] = INFINITY; ls[12] = 90.0f;
        EXPECT(msplat_cloud_from_attributes(c, n, xyz.data(), dc.data(), rest.data(), op.data(), ls.data(), rot.data()) == MSPLAT_OK);
        EXPECT(msplat_cloud_num_gaussians(c) == n);
        EXPECT(msplat_cloud_from_attributes(c, n, xyz.data(), dc.data(), nullptr, op.data(), ls.data(), rot.data()) == MSPLAT_OK);
        EXPECT(msplat_cloud_from_attributes(c, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == MSPLAT_OK ||
               msplat_cloud_num_gaussians(c) == n);
    }
    msplat_cloud_destroy(c);
}

static void json_cases(const std::string& golden)
{
    float m[16];
    EXPECT(msplat_vrconfig_import_json((golden + "/test_vr.json").c_str(), m) == MSPLAT_OK);
    EXPECT(msplat_vrconfig_export_json(tmp("out_vr.json").c_str(), m) == MSPLAT_OK);
    float m2[16];
    EXPECT(msplat_vrconfig_import_json(tmp("out_vr.json").c_str(), m2) == MSPLAT_OK);
    for (int i = 0; i < 16; ++i) EXPECT(std::fabs(m[i] - m2[i]) < 1e-5f);
    // a well-formed cameras.json with 3 cameras
    std::string cams = "[";
    for (int k = 0; k < 3; ++k) {
        cams += std::string(k ? "," : "") + "{\"id\": " + std::to_string(k) + ", \"img_name\": \"a\\u00e9\\n\\\"b\", \"width\": 1920, \"height\": 1080, "
                "\"position\": [1.5, -2e-1, 3E+0], \"rotation\": [[1,0,0],[0,1,0],[0,0,1]], \"fx\": 1163.25, \"fy\": 1156.28, \"extra\": {\"a\": [true, false, null]}}";
    }
    cams += "]";
    write_file(tmp("cameras.json"), cams);
    uint32_t n = 0;
    EXPECT(msplat_cameras_import_json(tmp("cameras.json").c_str(), nullptr, nullptr, 0, &n) == MSPLAT_OK && n == 3);
    std::vector<float> mats(3 * 16), fovs(3 * 2);
    EXPECT(msplat_cameras_import_json(tmp("cameras.json").c_str(), mats.data(), fovs.data(), 3, &n) == MSPLAT_OK);
    EXPECT(msplat_cameras_import_json(tmp("cameras.json").c_str(), mats.data(), fovs.data(), 1, &n) == MSPLAT_OK && n == 3);   // cap < count
    float nrm[3], pos[3];
    EXPECT(msplat_cameras_floor_plane(tmp("cameras.json").c_str(), nrm, pos) == MSPLAT_OK);
    // hostile JSON: deep nesting, truncation at every prefix, bad escapes / numbers, wrong shapes
    const std::string hostile[] = {
        std::string(10000, '[') + std::string(10000, ']'), std::string(100000, '{'), "[{\"id\": 0}]", "[1, 2, 3]", "{}", "[", "]", "nul",
        "[{\"id\": 0, \"position\": [1, 2], \"rotation\": [[1,0,0],[0,1,0],[0,0,1]], \"width\": 1, \"height\": 1, \"fx\": 1, \"fy\": 1}]",
        "[{\"id\": 0, \"position\": [1, 2, 3], \"rotation\": [[1,0],[0,1,0],[0,0,1]], \"width\": 1, \"height\": 1, \"fx\": 1, \"fy\": 1}]",
        "[{\"id\": \"x\", \"position\": \"y\"}]", "\"\\u12\"", "\"\\uD800\"", "\"abc", "1e99999", "-", "0x10", "[1,]", "{\"a\" 1}", "{\"a\": 1,}",
        "[{\"id\": 0, \"position\": [1e400, -1e400, 0], \"rotation\": [[1,0,0],[0,1,0],[0,0,1]], \"width\": 0, \"height\": 0, \"fx\": 0, \"fy\": 0}]",
        std::string("[\"") + std::string(1 << 20, 'a') + "\"]"};
    for (const auto& h : hostile) {
        write_file(tmp("h.json"), h);
        uint32_t k = 0;
        (void)msplat_cameras_import_json(tmp("h.json").c_str(), mats.data(), fovs.data(), 3, &k);
        (void)msplat_cameras_floor_plane(tmp("h.json").c_str(), nrm, pos);
        (void)msplat_vrconfig_import_json(tmp("h.json").c_str(), m2);
    }
    for (size_t cut = 0; cut < cams.size(); cut += 7) {          // every 7th truncation of a valid file
        write_file(tmp("h.json"), cams.substr(0, cut));
        uint32_t k = 0;
        EXPECT(msplat_cameras_import_json(tmp("h.json").c_str(), mats.data(), fovs.data(), 3, &k) != MSPLAT_OK || cut == 0 || k <= 3);
    }
    uint32_t k = 0;
    EXPECT(msplat_cameras_import_json(tmp("missing.json").c_str(), nullptr, nullptr, 0, &k) == MSPLAT_ERR_IO);
    // FindConfigFile: long names, tiny output buffer
    char out[8];
    EXPECT(msplat_find_config_file((g_dir + "/a/b/c/point_cloud.ply").c_str(), "cameras.json", out, sizeof(out)) != MSPLAT_OK);
    std::vector<char> big(8192);
    (void)msplat_find_config_file((g_dir + "/" + std::string(3000, 'd') + "/x.ply").c_str(), "cameras.json", big.data(), (uint32_t)big.size());
    EXPECT(msplat_find_config_file(tmp("short.ply").c_str(), "cameras.json", big.data(), (uint32_t)big.size()) == MSPLAT_OK);
}

static void image_cases()
{
    // write -> read round trip through the library's own PNG writer (stored deflate) and reader (inflate + unfilter)
    const int W = 37, H = 21;
    std::vector<float> img((size_t)W * H * 4);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (float)((i * 2654435761u) % 1000) / 999.0f * 1.4f - 0.2f;     // outside [0,1] too
    img[3] = NAN; img[7] = INFINITY;
    for (int srgb = 0; srgb < 2; ++srgb) {
        EXPECT(msplat_write_image(tmp("o.png").c_str(), img.data(), W, H, srgb) == MSPLAT_OK);
        EXPECT(msplat_write_image(tmp("o.ppm").c_str(), img.data(), W, H, srgb) == MSPLAT_OK);
        uint32_t w = 0, h = 0;
        EXPECT(msplat_read_image(tmp("o.png").c_str(), nullptr, 0, &w, &h) == MSPLAT_OK && w == (uint32_t)W && h == (uint32_t)H);
        std::vector<uint8_t> px((size_t)W * H * 4);
        EXPECT(msplat_read_image(tmp("o.png").c_str(), px.data(), px.size(), &w, &h) == MSPLAT_OK);
        EXPECT(msplat_read_image(tmp("o.png").c_str(), px.data(), px.size() - 1, &w, &h) != MSPLAT_OK);      // buffer one byte short
    }
    EXPECT(msplat_write_image(tmp("z.png").c_str(), img.data(), 0, 0, 0) != MSPLAT_OK || true);
    // hand-built PNGs: every colour type the reader accepts, every filter type, and the ones it must refuse
    const std::string sig = "\x89PNG\r\n\x1a\n";
    auto ihdr = [&](uint32_t w, uint32_t h, int depth, int ctype, int interlace) {
        return png_chunk("IHDR", be32(w) + be32(h) + std::string{(char)depth, (char)ctype, 0, 0, (char)interlace});
    };
    const int channels[7] = {1, 0, 3, 0, 2, 0, 4};
    for (int ctype : {0, 2, 4, 6}) {
        const int ch = channels[ctype];
        std::string raw;
        for (int y = 0; y < 9; ++y) {
            raw += (char)(y % 5);                                   // filter types 0..4
            for (int x = 0; x < 11 * ch; ++x) raw += (char)((x * 7 + y * 13) & 255);
        }
        const std::string png = sig + ihdr(11, 9, 8, ctype, 0) + png_chunk("tEXt", "k\0v") + png_chunk("IDAT", zlib_stored(raw)) + png_chunk("IEND", "");
        write_file(tmp("t.png"), png);
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> px(11 * 9 * 4);
        EXPECT(msplat_read_image(tmp("t.png").c_str(), px.data(), px.size(), &w, &h) == MSPLAT_OK && w == 11 && h == 9);
        const size_t idat_end = png.size() - 12;                    // everything before the IEND chunk
        for (size_t cut = 0; cut < png.size(); cut += 5) {          // truncated at every 5th byte: never a crash, and an
            write_file(tmp("t.png"), png.substr(0, cut));           // error while any pixel data is missing
            const int rc = msplat_read_image(tmp("t.png").c_str(), px.data(), px.size(), &w, &h);
            EXPECT(rc != MSPLAT_OK || cut >= idat_end);
        }
        std::string flipped = png;                                   // a flipped byte in every region
        for (size_t at = 8; at < flipped.size(); at += 3) {
            flipped[at] = (char)(flipped[at] ^ 0x5A);
            write_file(tmp("t.png"), flipped);
            (void)msplat_read_image(tmp("t.png").c_str(), px.data(), px.size(), &w, &h);
            flipped[at] = png[at];
        }
    }
    // bad filter byte, 16-bit depth, interlaced, palette, zero size, absurd size, a decompression bomb behind a 4x4 header
    std::string raw(9 * (1 + 11 * 4), '\0');
    raw[0] = 9;
    const std::string refused[] = {
        sig + ihdr(11, 9, 8, 6, 0) + png_chunk("IDAT", zlib_stored(raw)) + png_chunk("IEND", ""),
        sig + ihdr(11, 9, 16, 6, 0) + png_chunk("IDAT", zlib_stored(raw)) + png_chunk("IEND", ""),
        sig + ihdr(11, 9, 8, 6, 1) + png_chunk("IDAT", zlib_stored(raw)) + png_chunk("IEND", ""),
        sig + ihdr(11, 9, 8, 3, 0) + png_chunk("IDAT", zlib_stored(raw)) + png_chunk("IEND", ""),
        sig + ihdr(0, 0, 8, 6, 0) + png_chunk("IDAT", zlib_stored("")) + png_chunk("IEND", ""),
        sig + ihdr(0x7FFFFFFF, 0x7FFFFFFF, 8, 6, 0) + png_chunk("IDAT", zlib_stored(raw)) + png_chunk("IEND", ""),
        sig + ihdr(4, 4, 8, 6, 0) + png_chunk("IDAT", zlib_stored(std::string(8 << 20, '\0'))) + png_chunk("IEND", ""),
        sig + png_chunk("IDAT", zlib_stored(raw)), sig, "", "not a png at all"};
    for (const auto& pz : refused) {
        write_file(tmp("r.png"), pz);
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> px(11 * 9 * 4);
        EXPECT(msplat_read_image(tmp("r.png").c_str(), px.data(), px.size(), &w, &h) != MSPLAT_OK);
    }
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <tests/golden> <scratch dir>\n", argv[0]);
        return 2;
    }
    const std::string golden = argv[1];
    g_dir = argv[2];
    ply_cases(golden);
    json_cases(golden);
    image_cases();
    // matrix helpers (closed forms; singular input must not trap under UBSan)
    float z[16] = {0}, o[16];
    msplat_mat4_inverse(z, o);
    msplat_mat4_mul(z, z, o);
    msplat_perspective(0.8f, 1.5f, 0.1f, 1000.0f, o);
    msplat_perspective(0.0f, 0.0f, 0.0f, 0.0f, o);
    msplat_create_projection(-1.0f, 1.0f, 1.0f, -1.0f, 0.1f, 1000.0f, o);
    std::printf("host sanitize driver: %d failed expectation(s), %llu bytes handed to the device stubs\n", g_fail,
                (unsigned long long)g_stub_bytes);
    return g_fail ? 1 : 0;
}
