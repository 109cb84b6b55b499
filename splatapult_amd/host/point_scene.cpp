// point_scene.cpp -- PointCloud (see point_scene.hpp) and its C-ABI wrappers (include/msplat.h).
#include "point_scene.hpp"

#include <exception>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "../../include/msplat.h"
#include "scene_config.hpp"

namespace {

// the in-memory record: position.xyzw then colour.rgba, 32 bytes (pointcloud.cpp:19-23)
struct PointRecord {
    float position[4];
    float color[4];
};

}  // namespace

PointCloud::PointCloud(bool useLinearColorsIn) : linearColours_(useLinearColorsIn) {}

void PointCloud::SetUpAttributes()
{
    posAttr_ = BinaryAttribute(BinaryAttribute::Type::Float, offsetof(PointRecord, position));
    rgbAttr_ = BinaryAttribute(BinaryAttribute::Type::Float, offsetof(PointRecord, color));
}

void PointCloud::Reserve(size_t n)
{
    count_ = n;
    recordBytes_ = sizeof(PointRecord);
    SetUpAttributes();
    records_.reset(new PointRecord[n ? n : 1], [](void* p) { delete[] static_cast<PointRecord*>(p); });
}

bool PointCloud::ImportPly(const std::string& plyFilename)
{
    std::ifstream plyFile(plyFilename, std::ios::binary);
    if (!plyFile.is_open()) {
        std::fprintf(stderr, "[msplat][E] failed to open %s\n", plyFilename.c_str());
        return false;
    }
    Ply ply;
    if (!ply.Parse(plyFile)) {
        std::fprintf(stderr, "[msplat][E] Error parsing ply file \"%s\"\n", plyFilename.c_str());
        return false;
    }
    BinaryAttribute px, py, pz, red, green, blue;
    if (!ply.GetProperty("x", px) || !ply.GetProperty("y", py) || !ply.GetProperty("z", pz)) {
        std::fprintf(stderr, "[msplat][E] Error parsing ply file \"%s\", missing position property\n", plyFilename.c_str());
        return false;                                           // pointcloud.cpp:55-60
    }
    const bool useDoubles = px.type == BinaryAttribute::Type::Double && py.type == BinaryAttribute::Type::Double &&
                            pz.type == BinaryAttribute::Type::Double;
    if (!ply.GetProperty("red", red) || !ply.GetProperty("green", green) || !ply.GetProperty("blue", blue))
        // logged, not fatal (pointcloud.cpp:66-71): the unset attributes then read as 0 -> black points
        std::fprintf(stderr, "[msplat][E] Error parsing ply file \"%s\", missing color property\n", plyFilename.c_str());

    Reserve(ply.GetVertexCount());
    PointRecord* pd = static_cast<PointRecord*>(records_.get());
    size_t i = 0;
    ply.ForEachVertex([&](const void* v, size_t) {
        float p[3];
        if (useDoubles) {
            p[0] = (float)px.Read<double>(v); p[1] = (float)py.Read<double>(v); p[2] = (float)pz.Read<double>(v);
        } else {
            p[0] = px.Read<float>(v); p[1] = py.Read<float>(v); p[2] = pz.Read<float>(v);
        }
        // Reference quirk kept on purpose: with linearColours_ it is the POSITIONS that go through SRGBToLinear,
        // not the colours (pointcloud.cpp:84-95,109-120).
        for (int k = 0; k < 3; ++k) pd[i].position[k] = linearColours_ ? SRGBToLinear(p[k]) : p[k];
        pd[i].position[3] = 1.0f;
        pd[i].color[0] = (float)red.Read<uint8_t>(v) / 255.0f;
        pd[i].color[1] = (float)green.Read<uint8_t>(v) / 255.0f;
        pd[i].color[2] = (float)blue.Read<uint8_t>(v) / 255.0f;
        pd[i].color[3] = 1.0f;
        ++i;
    });
    return true;
}

bool PointCloud::ExportPly(const std::string& plyFilename) const
{
    std::ofstream plyFile(plyFilename, std::ios::binary);
    if (!plyFile.is_open()) {
        std::fprintf(stderr, "[msplat][E] failed to open %s\n", plyFilename.c_str());
        return false;
    }
    // x y z nx ny nz (float) red green blue (uchar): pointcloud.cpp:143-151
    Ply ply;
    const char* fnames[6] = {"x", "y", "z", "nx", "ny", "nz"};
    const char* cnames[3] = {"red", "green", "blue"};
    for (const char* nm : fnames) ply.AddProperty(nm, BinaryAttribute::Type::Float);
    for (const char* nm : cnames) ply.AddProperty(nm, BinaryAttribute::Type::UChar);
    BinaryAttribute f[6], c[3];
    for (int k = 0; k < 6; ++k) ply.GetProperty(fnames[k], f[k]);
    for (int k = 0; k < 3; ++k) ply.GetProperty(cnames[k], c[k]);
    ply.AllocData(count_);
    const PointRecord* pd = static_cast<const PointRecord*>(records_.get());
    size_t i = 0;
    ply.ForEachVertexMut([&](void* v, size_t) {
        for (int k = 0; k < 3; ++k) f[k].Write<float>(v, pd[i].position[k]);
        for (int k = 3; k < 6; ++k) f[k].Write<float>(v, 0.0f);
        for (int k = 0; k < 3; ++k) c[k].Write<uint8_t>(v, (uint8_t)(pd[i].color[k] * 255.0f));   // truncation, :185-187
        ++i;
    });
    ply.Dump(plyFile);
    return true;
}

void PointCloud::InitDebugCloud()
{
    // three axis lines of 5 points each, 0.2 apart, coloured r / g / b (pointcloud.cpp:199-258)
    const int kPerAxis = 5;
    const float delta = 1.0f / (float)kPerAxis;
    Reserve((size_t)kPerAxis * 3);
    PointRecord* pd = static_cast<PointRecord*>(records_.get());
    for (int axis = 0; axis < 3; ++axis)
        for (int i = 0; i < kPerAxis; ++i) {
            PointRecord& p = pd[axis * kPerAxis + i];
            for (int k = 0; k < 3; ++k) {
                p.position[k] = (k == axis) ? i * delta : 0.0f;
                p.color[k] = (k == axis) ? 1.0f : 0.0f;
            }
            p.position[3] = 1.0f;
            p.color[3] = 1.0f;
        }
}

void PointCloud::ForEachPosition(const ForEachPositionCallback& cb) const
{
    posAttr_.ForEach<float>(GetRawDataPtr(), GetStride(), GetNumPoints(), cb);
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
struct msplat_points {
    PointCloud pc;
    explicit msplat_points(bool linear) : pc(linear) {}
};

extern "C" {

msplat_points* msplat_points_create(int use_linear_colors) { return new msplat_points(use_linear_colors != 0); }
void msplat_points_destroy(msplat_points* p) { delete p; }
int msplat_points_import_ply(msplat_points* p, const char* path)
{
    if (!p || !path) return MSPLAT_ERR_INVALID_ARG;
    try {
        return p->pc.ImportPly(path) ? MSPLAT_OK : MSPLAT_ERR_IO;
    } catch (const std::exception&) {      // the C ABI never throws (allocation failure on a garbled file)
        return MSPLAT_ERR_IO;
    }
}
int msplat_points_export_ply(const msplat_points* p, const char* path)
{
    if (!p || !path) return MSPLAT_ERR_INVALID_ARG;
    try {
        return p->pc.ExportPly(path) ? MSPLAT_OK : MSPLAT_ERR_IO;
    } catch (const std::exception&) {
        return MSPLAT_ERR_IO;
    }
}
void msplat_points_init_debug(msplat_points* p) { if (p) p->pc.InitDebugCloud(); }
uint64_t msplat_points_num(const msplat_points* p) { return p ? p->pc.GetNumPoints() : 0; }
uint32_t msplat_points_stride(const msplat_points* p) { return p ? (uint32_t)p->pc.GetStride() : 0; }
const void* msplat_points_data(const msplat_points* p) { return p ? p->pc.GetRawDataPtr() : nullptr; }

int msplat_upload_point_cloud(msplat_ctx* ctx, const msplat_points* p)
{
    if (!ctx || !p) return MSPLAT_ERR_INVALID_ARG;
    return msplat_upload_points(ctx, p->pc.GetRawDataPtr(), p->pc.GetNumPoints(), (uint32_t)p->pc.GetStride(),
                                (uint32_t)p->pc.GetPositionAttrib().offset, (uint32_t)p->pc.GetColorAttrib().offset);
}

}  // extern "C"
