// gaussian_scene.cpp -- Ply / GaussianCloud implementation (host side of the drop-in boundary)
// plus the msplat_cloud_* and msplat_mat4_* C-ABI entry points (include/msplat.h).
//
// Behavioural contract follows /root/reference/src/ply.cpp:72-281 and
// /root/reference/src/gaussiancloud.cpp:86-122,138-365,505-657; glm's closed forms
// (quat->mat3, mat3/mat4 products, inverse, perspective) are restated since glm is not available.
// Built with -ffp-contract=off so the arithmetic below runs exactly as written.
#include "gaussian_scene.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <exception>
#include <sstream>

#include "../../include/msplat.h"

#pragma clang fp contract(off)

namespace {

void LogE(const char* fmt, const char* a = "", const char* b = "")
{
    std::fprintf(stderr, "[msplat][E] ");
    std::fprintf(stderr, fmt, a, b);
}

// next header line that is not a comment (ply.cpp:28-39)
bool NextLine(std::ifstream& f, std::string& line)
{
    while (std::getline(f, line)) {
        if (line.compare(0, 7, "comment") != 0) return true;
    }
    return false;
}

struct TypeName {
    const char* a;
    const char* b;
    BinaryAttribute::Type t;
};
const TypeName kTypeNames[] = {
    {"char", "int8", BinaryAttribute::Type::Char},     {"uchar", "uint8", BinaryAttribute::Type::UChar},
    {"short", "int16", BinaryAttribute::Type::Short},  {"ushort", "uint16", BinaryAttribute::Type::UShort},
    {"int", "int32", BinaryAttribute::Type::Int},      {"uint", "uint32", BinaryAttribute::Type::UInt},
    {"float", "float32", BinaryAttribute::Type::Float}, {"double", "float64", BinaryAttribute::Type::Double},
};

// record layouts (gaussiancloud.cpp:32-56): float offsets
constexpr int kBaseFloats = 25;   // 100 B
constexpr int kFullFloats = 61;   // 244 B
constexpr int kOffPos = 0, kOffR0 = 4, kOffG0 = 8, kOffB0 = 12, kOffCov0 = 16, kOffCov1 = 19, kOffCov2 = 22;
constexpr int kOffR1 = 25, kOffR2 = 29, kOffR3 = 33, kOffG1 = 37, kOffG2 = 41, kOffG3 = 45;
constexpr int kOffB1 = 49, kOffB2 = 53, kOffB3 = 57;

// 3x3 column-major helpers, glm operator* ordering: out[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2]
void Mul3(const float a[9], const float b[9], float out[9])
{
    float t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            float s = a[0 * 3 + r] * b[c * 3 + 0];
            s = s + a[1 * 3 + r] * b[c * 3 + 1];
            s = s + a[2 * 3 + r] * b[c * 3 + 2];
            t[c * 3 + r] = s;
        }
    std::memcpy(out, t, sizeof(t));
}
void Transpose3(const float a[9], float out[9])
{
    float t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) t[c * 3 + r] = a[r * 3 + c];
    std::memcpy(out, t, sizeof(t));
}

// gaussiancloud.cpp:86-94 -- Sigma = R S S^T R^T with R from the normalised quaternion (w,x,y,z)
void CovFromRotScale(const float rot[4], const float scale[3], float V[9])
{
    float w = rot[0], x = rot[1], y = rot[2], z = rot[3];
    const float len = std::sqrt((w * w + x * x) + (y * y + z * z));
    if (len <= 0.0f) {
        w = 1.0f;
        x = y = z = 0.0f;
    } else {
        const float inv = 1.0f / len;
        w *= inv; x *= inv; y *= inv; z *= inv;
    }
    const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    const float R[9] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz),        2.0f * (xz - wy),
                        2.0f * (xy - wz),        1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx),
                        2.0f * (xz + wy),        2.0f * (yz - wx),        1.0f - 2.0f * (xx + yy)};
    const float S[9] = {scale[0], 0, 0, 0, scale[1], 0, 0, 0, scale[2]};
    float St[9], Rt[9], A[9], B[9];
    Transpose3(S, St);
    Transpose3(R, Rt);
    Mul3(R, S, A);
    Mul3(A, St, B);
    Mul3(B, Rt, V);
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi sweeps (replaces Eigen::SelfAdjointEigenSolver,
// gaussiancloud.cpp:96-117).  evec columns = eigenvectors, eval ascending.
void JacobiEigen3(const float Vin[9], float evec[9], float eval[3])
{
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) a[r][c] = 0.5 * ((double)Vin[c * 3 + r] + (double)Vin[r * 3 + c]);
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-30) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (std::fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int i, int j) { return a[i][i] < a[j][j]; });
    for (int k = 0; k < 3; ++k) {
        eval[k] = (float)a[order[k]][order[k]];
        for (int r = 0; r < 3; ++r) evec[k * 3 + r] = (float)v[r][order[k]];
    }
}

// rotation matrix (column-major, det +1) -> quaternion (w,x,y,z)
void QuatFromMat3(const float R[9], float q[4])
{
    const float m00 = R[0], m11 = R[4], m22 = R[8];
    const float tr = m00 + m11 + m22;
    float w, x, y, z;
    if (tr > 0.0f) {
        const float s = std::sqrt(tr + 1.0f) * 2.0f;
        w = 0.25f * s;
        x = (R[1 * 3 + 2] - R[2 * 3 + 1]) / s;
        y = (R[2 * 3 + 0] - R[0 * 3 + 2]) / s;
        z = (R[0 * 3 + 1] - R[1 * 3 + 0]) / s;
    } else if (m00 > m11 && m00 > m22) {
        const float s = std::sqrt(1.0f + m00 - m11 - m22) * 2.0f;
        w = (R[1 * 3 + 2] - R[2 * 3 + 1]) / s;
        x = 0.25f * s;
        y = (R[1 * 3 + 0] + R[0 * 3 + 1]) / s;
        z = (R[2 * 3 + 0] + R[0 * 3 + 2]) / s;
    } else if (m11 > m22) {
        const float s = std::sqrt(1.0f + m11 - m00 - m22) * 2.0f;
        w = (R[2 * 3 + 0] - R[0 * 3 + 2]) / s;
        x = (R[1 * 3 + 0] + R[0 * 3 + 1]) / s;
        y = 0.25f * s;
        z = (R[2 * 3 + 1] + R[1 * 3 + 2]) / s;
    } else {
        const float s = std::sqrt(1.0f + m22 - m00 - m11) * 2.0f;
        w = (R[0 * 3 + 1] - R[1 * 3 + 0]) / s;
        x = (R[2 * 3 + 0] + R[0 * 3 + 2]) / s;
        y = (R[2 * 3 + 1] + R[1 * 3 + 2]) / s;
        z = 0.25f * s;
    }
    const float n = std::sqrt(w * w + x * x + y * y + z * z);
    q[0] = w / n; q[1] = x / n; q[2] = y / n; q[3] = z / n;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Ply
// ------------------------------------------------------------------------------------------

bool Ply::Parse(std::ifstream& plyFile)
{
    if (!ParseHeader(plyFile)) return false;
    // The reference allocates vertexSize * vertexCount straight from the header (ply.cpp:80-84) and lets
    // std::bad_alloc fly on a garbled count; here the product must not wrap and the allocation failure is a
    // parse error (the C ABI never throws).  A merely short file is still accepted like in the reference:
    // the tail stays zero-filled.
    if (vertexSize != 0 && vertexCount > (size_t)0x7FFFFFFFFFFFFFFFull / vertexSize) {
        LogE("Invalid ply file, vertex block size overflows (%s vertices of %s bytes)\n", std::to_string(vertexCount).c_str(),
             std::to_string(vertexSize).c_str());
        return false;
    }
    try {
        AllocData(vertexCount);
    } catch (const std::exception&) {
        LogE("Invalid ply file, cannot allocate %s vertices of %s bytes\n", std::to_string(vertexCount).c_str(),
             std::to_string(vertexSize).c_str());
        vertexCount = 0;
        data.clear();
        return false;
    }
    // one bulk read of the vertex block (ply.cpp:80-84); a short file leaves the tail zero-filled
    plyFile.read(reinterpret_cast<char*>(data.data()), (std::streamsize)(vertexSize * vertexCount));
    return true;
}

void Ply::Dump(std::ofstream& plyFile) const
{
    DumpHeader(plyFile);
    plyFile.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)(vertexSize * vertexCount));
}

bool Ply::GetProperty(const std::string& key, BinaryAttribute& attributeOut) const
{
    auto it = propertyMap.find(key);
    if (it == propertyMap.end()) return false;
    attributeOut = it->second;
    return true;
}

void Ply::AddProperty(const std::string& key, BinaryAttribute::Type type)
{
    // offsets accumulate in declaration order; a duplicate name keeps its first slot but still
    // consumes space (emplace semantics of ply.cpp:106-112)
    BinaryAttribute attrib(type, vertexSize);
    propertyMap.emplace(key, attrib);
    vertexSize += attrib.size;
}

void Ply::AllocData(size_t numVertices)
{
    vertexCount = numVertices;
    data.assign(vertexSize * numVertices, 0);
}

void Ply::ForEachVertex(const VertexCallback& cb) const
{
    const uint8_t* p = data.data();
    for (size_t i = 0; i < vertexCount; ++i, p += vertexSize) cb(p, vertexSize);
}

void Ply::ForEachVertexMut(const VertexCallbackMut& cb)
{
    uint8_t* p = data.data();
    for (size_t i = 0; i < vertexCount; ++i, p += vertexSize) cb(p, vertexSize);
}

bool Ply::ParseHeader(std::ifstream& plyFile)
{
    std::string line;
    if (!NextLine(plyFile, line) || line != "ply") {
        LogE("Invalid ply file\n");
        return false;
    }
    if (!NextLine(plyFile, line)) {
        LogE("Unexpected error reading next line\n");
        return false;
    }
    if (line == "format binary_big_endian 1.0") {
        LogE("Unsupported ply file, only binary_little_endian supported\n");
        return false;
    }
    if (line != "format binary_little_endian 1.0") {
        LogE("Invalid ply file, expected format\n");
        return false;
    }
    if (!NextLine(plyFile, line)) {
        LogE("Unexpected error reading next line\n");
        return false;
    }
    {
        std::istringstream ss(line);
        std::string a, b;
        size_t count = 0;
        if (!(ss >> a >> b >> count) || a != "element" || b != "vertex") {
            LogE("Invalid ply file, expected \"element vertex {number}\"\n");
            return false;
        }
        vertexCount = count;
    }
    for (;;) {
        if (!NextLine(plyFile, line)) {
            LogE("unexpected error reading line\n");
            return false;
        }
        if (line == "end_header") return true;
        std::istringstream ss(line);
        std::string kw, ty, name;
        ss >> kw >> ty >> name;
        if (kw != "property") {
            LogE("Invalid header, expected property\n");
            return false;
        }
        BinaryAttribute::Type t = BinaryAttribute::Type::Unknown;
        for (const TypeName& tn : kTypeNames)
            if (ty == tn.a || ty == tn.b) t = tn.t;
        if (t == BinaryAttribute::Type::Unknown) {
            LogE("Unsupported type \"%s\" for property \"%s\"\n", ty.c_str(), name.c_str());
            return false;
        }
        AddProperty(name, t);
    }
}

void Ply::DumpHeader(std::ofstream& plyFile) const
{
    plyFile << "ply\nformat binary_little_endian 1.0\nelement vertex " << vertexCount << "\n";
    std::vector<std::pair<size_t, const std::string*>> order;
    order.reserve(propertyMap.size());
    for (const auto& kv : propertyMap) order.emplace_back(kv.second.offset, &kv.first);
    std::sort(order.begin(), order.end());
    for (const auto& o : order) {
        const BinaryAttribute& a = propertyMap.at(*o.second);
        const char* tn = "unknown";
        for (const TypeName& t : kTypeNames)
            if (t.t == a.type) tn = t.a;
        plyFile << "property " << tn << " " << *o.second << "\n";
    }
    plyFile << "end_header\n";
}

// ------------------------------------------------------------------------------------------
// GaussianCloud
// ------------------------------------------------------------------------------------------

GaussianCloud::GaussianCloud(const Options& options) : opt(options) {}

void GaussianCloud::InitAttribs()
{
    using T = BinaryAttribute::Type;
    auto at = [](int floatOff) { return BinaryAttribute(T::Float, (size_t)floatOff * sizeof(float)); };
    posWithAlphaAttrib = at(kOffPos);
    r_sh0Attrib = at(kOffR0);
    g_sh0Attrib = at(kOffG0);
    b_sh0Attrib = at(kOffB0);
    cov3_col0Attrib = at(kOffCov0);
    cov3_col1Attrib = at(kOffCov1);
    cov3_col2Attrib = at(kOffCov2);
    if (hasFullSH) {
        r_sh1Attrib = at(kOffR1); r_sh2Attrib = at(kOffR2); r_sh3Attrib = at(kOffR3);
        g_sh1Attrib = at(kOffG1); g_sh2Attrib = at(kOffG2); g_sh3Attrib = at(kOffG3);
        b_sh1Attrib = at(kOffB1); b_sh2Attrib = at(kOffB2); b_sh3Attrib = at(kOffB3);
    }
}

void GaussianCloud::Allocate(size_t n, bool fullSH)
{
    numGaussians = n;
    gaussianSize = (fullSH ? kFullFloats : kBaseFloats) * sizeof(float);
    float* p = new float[std::max<size_t>(n, 1) * (fullSH ? kFullFloats : kBaseFloats)]();
    data.reset(p, [](void* q) { delete[] static_cast<float*>(q); });
}

namespace {
// one vertex: gaussiancloud.cpp:254-361
inline void BuildRecord(float* o, bool fullSH, const float xyz[3], const float f_dc[3], const float* f_rest45,
                        float opacity, const float logScale[3], const float rot[4])
{
    o[0] = xyz[0];
    o[1] = xyz[1];
    o[2] = xyz[2];
    o[3] = 1.0f / (1.0f + expf(-opacity));   // ComputeAlphaFromOpacity, gaussiancloud.cpp:119-122
    static const int lo[3] = {kOffR0, kOffG0, kOffB0};
    static const int hi[3] = {kOffR1, kOffG1, kOffB1};
    for (int c = 0; c < 3; ++c) {
        o[lo[c]] = f_dc[c];
        if (fullSH) {
            for (int k = 1; k < 4; ++k) o[lo[c] + k] = f_rest45[c * 15 + k - 1];
            for (int k = 4; k < 16; ++k) o[hi[c] + k - 4] = f_rest45[c * 15 + k - 1];
        } else {
            o[lo[c] + 1] = o[lo[c] + 2] = o[lo[c] + 3] = 0.0f;
        }
    }
    const float sc[3] = {expf(logScale[0]), expf(logScale[1]), expf(logScale[2])};
    float V[9];
    CovFromRotScale(rot, sc, V);
    for (int k = 0; k < 9; ++k) o[kOffCov0 + k] = V[k];
}
}  // namespace

bool GaussianCloud::ImportPly(const std::string& plyFilename)
{
    std::ifstream plyFile(plyFilename, std::ios::binary);
    if (!plyFile.is_open()) {
        LogE("failed to open %s\n", plyFilename.c_str());
        return false;
    }
    Ply ply;
    if (!ply.Parse(plyFile)) {
        LogE("Error parsing ply file \"%s\"\n", plyFilename.c_str());
        return false;
    }
    // property lookup (gaussiancloud.cpp:160-230): missing mandatory properties are logged and read as 0
    BinaryAttribute px, py, pz, fdc[3], frest[45], opac, sc[3], rt[4];
    if (!ply.GetProperty("x", px) || !ply.GetProperty("y", py) || !ply.GetProperty("z", pz))
        LogE("Error parsing ply file \"%s\", missing position property\n", plyFilename.c_str());
    for (int i = 0; i < 3; ++i)
        if (!ply.GetProperty("f_dc_" + std::to_string(i), fdc[i]))
            LogE("Error parsing ply file \"%s\", missing f_dc property\n", plyFilename.c_str());
    hasFullSH = false;
    if (opt.importFullSH) {
        hasFullSH = true;
        for (int i = 0; i < 45; ++i)
            if (!ply.GetProperty("f_rest_" + std::to_string(i), frest[i])) {
                std::fprintf(stderr, "[msplat][W] PLY file \"%s\", missing f_rest property\n", plyFilename.c_str());
                hasFullSH = false;
                break;
            }
    }
    if (!ply.GetProperty("opacity", opac))
        LogE("Error parsing ply file \"%s\", missing opacity property\n", plyFilename.c_str());
    for (int i = 0; i < 3; ++i)
        if (!ply.GetProperty("scale_" + std::to_string(i), sc[i]))
            LogE("Error parsing ply file \"%s\", missing scale property\n", plyFilename.c_str());
    for (int i = 0; i < 4; ++i)
        if (!ply.GetProperty("rot_" + std::to_string(i), rt[i]))
            LogE("Error parsing ply file \"%s\", missing rot property\n", plyFilename.c_str());

    InitAttribs();
    Allocate(ply.GetVertexCount(), hasFullSH);
    const size_t strideF = gaussianSize / sizeof(float);
    float* out = static_cast<float*>(data.get());
    const bool full = hasFullSH;
    ply.ForEachVertex([&](const void* v, size_t) {
        const float xyz[3] = {px.Read<float>(v), py.Read<float>(v), pz.Read<float>(v)};
        const float dc[3] = {fdc[0].Read<float>(v), fdc[1].Read<float>(v), fdc[2].Read<float>(v)};
        float rest[45];
        if (full)
            for (int i = 0; i < 45; ++i) rest[i] = frest[i].Read<float>(v);
        const float ls[3] = {sc[0].Read<float>(v), sc[1].Read<float>(v), sc[2].Read<float>(v)};
        const float q[4] = {rt[0].Read<float>(v), rt[1].Read<float>(v), rt[2].Read<float>(v), rt[3].Read<float>(v)};
        BuildRecord(out, full, xyz, dc, rest, opac.Read<float>(v), ls, q);
        out += strideF;
    });
    return true;
}

bool GaussianCloud::FromAttributes(size_t n, const float* xyz, const float* f_dc, const float* f_rest,
                                   const float* opacity, const float* logScale, const float* rot)
{
    if (n && (!xyz || !f_dc || !opacity || !logScale || !rot)) return false;
    hasFullSH = opt.importFullSH && f_rest != nullptr;
    InitAttribs();
    Allocate(n, hasFullSH);
    const size_t strideF = gaussianSize / sizeof(float);
    float* out = static_cast<float*>(data.get());
    for (size_t i = 0; i < n; ++i)
        BuildRecord(out + i * strideF, hasFullSH, xyz + 3 * i, f_dc + 3 * i, hasFullSH ? f_rest + 45 * i : nullptr,
                    opacity[i], logScale + 3 * i, rot + 4 * i);
    return true;
}

bool GaussianCloud::ExportPly(const std::string& plyFilename) const
{
    // gaussiancloud.cpp:367-503.  Unlike the reference (which reads 15 floats past r_sh0 and so
    // scrambles full-SH exports) the coefficients are gathered per attribute.
    std::ofstream plyFile(plyFilename, std::ios::binary);
    if (!plyFile.is_open()) {
        LogE("failed to open %s\n", plyFilename.c_str());
        return false;
    }
    Ply ply;
    using T = BinaryAttribute::Type;
    std::vector<std::string> names = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"};
    const bool exportRest = opt.exportFullSH;
    if (exportRest)
        for (int i = 0; i < 45; ++i) names.push_back("f_rest_" + std::to_string(i));
    for (const char* s : {"opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"}) names.push_back(s);
    for (const auto& nm : names) ply.AddProperty(nm, T::Float);
    ply.AllocData(numGaussians);
    const size_t strideF = gaussianSize / sizeof(float);
    const float* g = static_cast<const float*>(data.get());
    const size_t outFloats = names.size();
    ply.ForEachVertexMut([&](void* v, size_t) {
        float* o = static_cast<float*>(v);
        size_t k = 0;
        o[k++] = g[0]; o[k++] = g[1]; o[k++] = g[2];
        o[k++] = 0.0f; o[k++] = 0.0f; o[k++] = 0.0f;
        o[k++] = g[kOffR0]; o[k++] = g[kOffG0]; o[k++] = g[kOffB0];
        if (exportRest) {
            static const int lo[3] = {kOffR0, kOffG0, kOffB0};
            static const int hi[3] = {kOffR1, kOffG1, kOffB1};
            for (int c = 0; c < 3; ++c)
                for (int j = 1; j < 16; ++j)
                    o[k++] = (j < 4) ? g[lo[c] + j] : (hasFullSH ? g[hi[c] + j - 4] : 0.0f);
        }
        o[k++] = -logf((1.0f / g[3]) - 1.0f);   // ComputeOpacityFromAlpha, gaussiancloud.cpp:124-127
        float evec[9], eval[3];
        JacobiEigen3(g + kOffCov0, evec, eval);
        // make it a proper rotation (det +1), as the reference does before glm::quat(R)
        const float det = evec[0] * (evec[4] * evec[8] - evec[5] * evec[7]) - evec[3] * (evec[1] * evec[8] - evec[2] * evec[7]) +
                          evec[6] * (evec[1] * evec[5] - evec[2] * evec[4]);
        if (det < 0.0f)
            for (float& e : evec) e = -e;
        float q[4];
        QuatFromMat3(evec, q);
        for (int j = 0; j < 3; ++j) o[k++] = logf(sqrtf(std::max(eval[j], 0.0f)));
        for (int j = 0; j < 4; ++j) o[k++] = q[j];
        assert(k == outFloats);
        (void)outFloats;
        g += strideF;
    });
    ply.Dump(plyFile);
    return true;
}

void GaussianCloud::InitDebugCloud()
{
    // gaussiancloud.cpp:505-578: RGB axis gizmo, 5 splats per axis + a white one at the origin.
    // (The reference leaves hasFullSH=false with a 244 B stride; we keep that observable behaviour.)
    const int NUM = 5;
    hasFullSH = false;
    InitAttribs();
    Allocate(NUM * 3 + 1, true);
    float* g = static_cast<float*>(data.get());
    const float DELTA = 1.0f / (float)NUM, COV = 0.005f;
    const float SH_C0 = 0.28209479177387814f, ONE = 1.0f / (2.0f * SH_C0), ZERO = -1.0f / (2.0f * SH_C0);
    auto put = [&](int idx, float x, float y, float z, float r, float gg, float b) {
        float* o = g + (size_t)idx * kFullFloats;
        o[0] = x; o[1] = y; o[2] = z; o[3] = 1.0f;
        o[kOffR0] = r; o[kOffG0] = gg; o[kOffB0] = b;
        o[kOffCov0 + 0] = COV; o[kOffCov1 + 1] = COV; o[kOffCov2 + 2] = COV;
    };
    for (int i = 0; i < NUM; ++i) {
        put(i, i * DELTA + DELTA, 0, 0, ONE, ZERO, ZERO);
        put(NUM + i, 0, i * DELTA + DELTA, 0, ZERO, ONE, ZERO);
        put(2 * NUM + i, 0, 0, i * DELTA + DELTA + 0.0001f, ZERO, ZERO, ONE);
    }
    put(3 * NUM, 0, 0, 0, ONE, ONE, ONE);
}

void GaussianCloud::PruneSplats(const float origin[3], uint32_t numSplats)
{
    // gaussiancloud.cpp:581-626: keep the numSplats nearest to origin, ordered by distance
    if (!data || (size_t)numSplats >= numGaussians) return;
    const size_t strideF = gaussianSize / sizeof(float);
    const float* g = static_cast<const float*>(data.get());
    std::vector<std::pair<float, uint32_t>> dist(numGaussians);
    for (size_t i = 0; i < numGaussians; ++i) {
        const float* p = g + i * strideF;
        const float dx = origin[0] - p[0], dy = origin[1] - p[1], dz = origin[2] - p[2];
        dist[i] = {std::sqrt(dx * dx + dy * dy + dz * dz), (uint32_t)i};
    }
    std::stable_sort(dist.begin(), dist.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    float* nd = new float[std::max<size_t>(numSplats, 1) * strideF];
    for (uint32_t i = 0; i < numSplats; ++i) std::memcpy(nd + i * strideF, g + dist[i].second * strideF, gaussianSize);
    numGaussians = numSplats;
    data.reset(nd, [](void* q) { delete[] static_cast<float*>(q); });
}

void GaussianCloud::ForEachPosWithAlpha(const ForEachPosWithAlphaCallback& cb) const
{
    posWithAlphaAttrib.ForEach<float>(GetRawDataPtr(), GetStride(), GetNumGaussians(), cb);
}

// ------------------------------------------------------------------------------------------
// C ABI: scene data + host matrices
// ------------------------------------------------------------------------------------------

struct msplat_cloud {
    GaussianCloud gc;
    explicit msplat_cloud(bool fullSH) : gc(GaussianCloud::Options{fullSH, fullSH}) {}
};

#define CM(m, c, r) ((m)[(c) * 4 + (r)])

extern "C" {

msplat_cloud* msplat_cloud_create(int import_full_sh) { return new msplat_cloud(import_full_sh != 0); }
void msplat_cloud_destroy(msplat_cloud* c) { delete c; }

int msplat_cloud_import_ply(msplat_cloud* c, const char* path)
{
    try {
        if (!c || !path) return MSPLAT_ERR_INVALID_ARG;
        return c->gc.ImportPly(path) ? MSPLAT_OK : MSPLAT_ERR_IO;
    } catch (const std::exception& e) {
        LogE("%s: %s\n", __func__, e.what());
        return MSPLAT_ERR_IO;
    }
}

int msplat_cloud_export_ply(msplat_cloud* c, const char* path)
{
    try {
        if (!c || !path) return MSPLAT_ERR_INVALID_ARG;
        return c->gc.ExportPly(path) ? MSPLAT_OK : MSPLAT_ERR_IO;
    } catch (const std::exception& e) {
        LogE("%s: %s\n", __func__, e.what());
        return MSPLAT_ERR_IO;
    }
}

int msplat_cloud_init_debug(msplat_cloud* c)
{
    try {
        if (!c) return MSPLAT_ERR_INVALID_ARG;
        c->gc.InitDebugCloud();
        return MSPLAT_OK;
    } catch (const std::exception& e) {
        LogE("%s: %s\n", __func__, e.what());
        return MSPLAT_ERR_INVALID_ARG;
    }
}

int msplat_cloud_prune(msplat_cloud* c, const float origin[3], uint32_t keep)
{
    try {
        if (!c || !origin) return MSPLAT_ERR_INVALID_ARG;
        c->gc.PruneSplats(origin, keep);
        return MSPLAT_OK;
    } catch (const std::exception& e) {
        LogE("%s: %s\n", __func__, e.what());
        return MSPLAT_ERR_INVALID_ARG;
    }
}

int msplat_cloud_from_attributes(msplat_cloud* c, uint64_t n, const float* xyz, const float* f_dc,
                                 const float* f_rest, const float* opacity, const float* log_scale,
                                 const float* rot)
{
    try {
        if (!c) return MSPLAT_ERR_INVALID_ARG;
        return c->gc.FromAttributes((size_t)n, xyz, f_dc, f_rest, opacity, log_scale, rot) ? MSPLAT_OK
                                                                                            : MSPLAT_ERR_INVALID_ARG;
    } catch (const std::exception& e) {
        LogE("%s: %s\n", __func__, e.what());
        return MSPLAT_ERR_INVALID_ARG;
    }
}

uint64_t msplat_cloud_num_gaussians(const msplat_cloud* c) { return c ? c->gc.GetNumGaussians() : 0; }
uint64_t msplat_cloud_stride(const msplat_cloud* c) { return c ? c->gc.GetStride() : 0; }
uint64_t msplat_cloud_total_size(const msplat_cloud* c) { return c ? c->gc.GetTotalSize() : 0; }
const void* msplat_cloud_raw_data(const msplat_cloud* c) { return c ? c->gc.GetRawDataPtr() : nullptr; }
int msplat_cloud_has_full_sh(const msplat_cloud* c) { return c && c->gc.HasFullSH(); }

int msplat_cloud_attr_offsets(const msplat_cloud* c, msplat_attr_offsets* o)
{
    if (!c || !o) return MSPLAT_ERR_INVALID_ARG;
    const GaussianCloud& g = c->gc;
    std::memset(o, 0, sizeof(*o));
    o->pos_with_alpha = (uint32_t)g.GetPosWithAlphaAttrib().offset;
    o->r_sh0 = (uint32_t)g.GetR_SH0Attrib().offset;
    o->g_sh0 = (uint32_t)g.GetG_SH0Attrib().offset;
    o->b_sh0 = (uint32_t)g.GetB_SH0Attrib().offset;
    o->cov3_col0 = (uint32_t)g.GetCov3_Col0Attrib().offset;
    o->cov3_col1 = (uint32_t)g.GetCov3_Col1Attrib().offset;
    o->cov3_col2 = (uint32_t)g.GetCov3_Col2Attrib().offset;
    if (g.HasFullSH()) {
        o->r_sh1 = (uint32_t)g.GetR_SH1Attrib().offset; o->r_sh2 = (uint32_t)g.GetR_SH2Attrib().offset;
        o->r_sh3 = (uint32_t)g.GetR_SH3Attrib().offset;
        o->g_sh1 = (uint32_t)g.GetG_SH1Attrib().offset; o->g_sh2 = (uint32_t)g.GetG_SH2Attrib().offset;
        o->g_sh3 = (uint32_t)g.GetG_SH3Attrib().offset;
        o->b_sh1 = (uint32_t)g.GetB_SH1Attrib().offset; o->b_sh2 = (uint32_t)g.GetB_SH2Attrib().offset;
        o->b_sh3 = (uint32_t)g.GetB_SH3Attrib().offset;
    }
    return MSPLAT_OK;
}

int msplat_upload_gaussian_cloud(msplat_ctx* ctx, const msplat_cloud* c)
{
    if (!ctx || !c) return MSPLAT_ERR_INVALID_ARG;
    msplat_attr_offsets off;
    msplat_cloud_attr_offsets(c, &off);
    return msplat_upload_cloud(ctx, c->gc.GetRawDataPtr(), c->gc.GetNumGaussians(), (uint32_t)c->gc.GetStride(), &off,
                               c->gc.HasFullSH() ? 1 : 0);
}

int msplat_upload_ply(msplat_ctx* ctx, const char* path, int import_full_sh)
{
    try {
        if (!ctx || !path) return MSPLAT_ERR_INVALID_ARG;
        std::ifstream f(path, std::ios::binary);
        if (!f.is_open()) {
            LogE("failed to open %s\n", path);
            return MSPLAT_ERR_IO;
        }
        Ply ply;
        if (!ply.Parse(f)) {
            LogE("Error parsing ply file \"%s\"\n", path);
            return MSPLAT_ERR_IO;
        }
        msplat_ply_layout L;
        L.vertex_size = (uint32_t)ply.GetVertexSize();
        auto off = [&](const std::string& name) -> int32_t {
            BinaryAttribute a;
            if (!ply.GetProperty(name, a) || a.type != BinaryAttribute::Type::Float) return -1;
            return (int32_t)a.offset;
        };
        L.x = off("x"); L.y = off("y"); L.z = off("z");
        for (int i = 0; i < 3; ++i) L.f_dc[i] = off("f_dc_" + std::to_string(i));
        for (int i = 0; i < 45; ++i) L.f_rest[i] = off("f_rest_" + std::to_string(i));
        L.opacity = off("opacity");
        for (int i = 0; i < 3; ++i) L.scale[i] = off("scale_" + std::to_string(i));
        for (int i = 0; i < 4; ++i) L.rot[i] = off("rot_" + std::to_string(i));
        return msplat_upload_ply_vertices(ctx, ply.GetRawData(), ply.GetVertexCount(), &L, import_full_sh);
    } catch (const std::exception& e) {
        LogE("%s: %s\n", __func__, e.what());
        return MSPLAT_ERR_IO;
    }
}

// ---- matrices (glm closed forms; used by Sort/Render exactly as splatrenderer.cpp:161,175,327) ----

void msplat_mat4_mul(const float a[16], const float b[16], float out[16])
{
    float t[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = CM(a, 0, r) * CM(b, c, 0);
            s = s + CM(a, 1, r) * CM(b, c, 1);
            s = s + CM(a, 2, r) * CM(b, c, 2);
            s = s + CM(a, 3, r) * CM(b, c, 3);
            t[c * 4 + r] = s;
        }
    std::memcpy(out, t, sizeof(t));
}

void msplat_mat4_inverse(const float m[16], float out[16])
{
    // adjugate from 2x2 sub-determinants of column pairs (the glm::inverse formulation)
    const float s00 = CM(m,2,2) * CM(m,3,3) - CM(m,3,2) * CM(m,2,3);
    const float s02 = CM(m,1,2) * CM(m,3,3) - CM(m,3,2) * CM(m,1,3);
    const float s03 = CM(m,1,2) * CM(m,2,3) - CM(m,2,2) * CM(m,1,3);
    const float s04 = CM(m,2,1) * CM(m,3,3) - CM(m,3,1) * CM(m,2,3);
    const float s06 = CM(m,1,1) * CM(m,3,3) - CM(m,3,1) * CM(m,1,3);
    const float s07 = CM(m,1,1) * CM(m,2,3) - CM(m,2,1) * CM(m,1,3);
    const float s08 = CM(m,2,1) * CM(m,3,2) - CM(m,3,1) * CM(m,2,2);
    const float s10 = CM(m,1,1) * CM(m,3,2) - CM(m,3,1) * CM(m,1,2);
    const float s11 = CM(m,1,1) * CM(m,2,2) - CM(m,2,1) * CM(m,1,2);
    const float s12 = CM(m,2,0) * CM(m,3,3) - CM(m,3,0) * CM(m,2,3);
    const float s14 = CM(m,1,0) * CM(m,3,3) - CM(m,3,0) * CM(m,1,3);
    const float s15 = CM(m,1,0) * CM(m,2,3) - CM(m,2,0) * CM(m,1,3);
    const float s16 = CM(m,2,0) * CM(m,3,2) - CM(m,3,0) * CM(m,2,2);
    const float s18 = CM(m,1,0) * CM(m,3,2) - CM(m,3,0) * CM(m,1,2);
    const float s19 = CM(m,1,0) * CM(m,2,2) - CM(m,2,0) * CM(m,1,2);
    const float s20 = CM(m,2,0) * CM(m,3,1) - CM(m,3,0) * CM(m,2,1);
    const float s22 = CM(m,1,0) * CM(m,3,1) - CM(m,3,0) * CM(m,1,1);
    const float s23 = CM(m,1,0) * CM(m,2,1) - CM(m,2,0) * CM(m,1,1);
    const float F0[4] = {s00, s00, s02, s03}, F1[4] = {s04, s04, s06, s07}, F2[4] = {s08, s08, s10, s11};
    const float F3[4] = {s12, s12, s14, s15}, F4[4] = {s16, s16, s18, s19}, F5[4] = {s20, s20, s22, s23};
    const float V0[4] = {CM(m,1,0), CM(m,0,0), CM(m,0,0), CM(m,0,0)};
    const float V1[4] = {CM(m,1,1), CM(m,0,1), CM(m,0,1), CM(m,0,1)};
    const float V2[4] = {CM(m,1,2), CM(m,0,2), CM(m,0,2), CM(m,0,2)};
    const float V3[4] = {CM(m,1,3), CM(m,0,3), CM(m,0,3), CM(m,0,3)};
    float adj[16];
    for (int i = 0; i < 4; ++i) {
        const float sgnA = (i & 1) ? -1.0f : 1.0f;
        const float sgnB = -sgnA;
        adj[0 * 4 + i] = ((V1[i] * F0[i] - V2[i] * F1[i]) + V3[i] * F2[i]) * sgnA;
        adj[1 * 4 + i] = ((V0[i] * F0[i] - V2[i] * F3[i]) + V3[i] * F4[i]) * sgnB;
        adj[2 * 4 + i] = ((V0[i] * F1[i] - V1[i] * F3[i]) + V3[i] * F5[i]) * sgnA;
        adj[3 * 4 + i] = ((V0[i] * F2[i] - V1[i] * F4[i]) + V2[i] * F5[i]) * sgnB;
    }
    const float det = (CM(m,0,0) * adj[0] + CM(m,0,1) * adj[4]) + (CM(m,0,2) * adj[8] + CM(m,0,3) * adj[12]);
    const float ood = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = adj[i] * ood;
}

void msplat_perspective(float fovy, float aspect, float zn, float zf, float out[16])
{
    // glm::perspective (right-handed, z in [-1,1]) as used at app.cpp:1042
    const float t = tanf(fovy / 2.0f);
    std::memset(out, 0, 16 * sizeof(float));
    CM(out, 0, 0) = 1.0f / (aspect * t);
    CM(out, 1, 1) = 1.0f / t;
    CM(out, 2, 2) = -(zf + zn) / (zf - zn);
    CM(out, 2, 3) = -1.0f;
    CM(out, 3, 2) = -(2.0f * zf * zn) / (zf - zn);
}

void msplat_create_projection(float tanL, float tanR, float tanU, float tanD, float zn, float zf, float out[16])
{
    // util.cpp:420-480 (OpenGL clip space): asymmetric frustum from tangents of the half angles
    const float w = tanR - tanL, h = tanU - tanD;
    std::memset(out, 0, 16 * sizeof(float));
    out[0] = 2 / w;
    out[5] = 2 / h;
    out[8] = (tanR + tanL) / w;
    out[9] = (tanU + tanD) / h;
    out[11] = -1;
    if (zf <= zn) {   // far plane at infinity
        out[10] = -1;
        out[14] = -(zn + zn);
    } else {
        out[10] = -(zf + zn) / (zf - zn);
        out[14] = -(zf * (zn + zn)) / (zf - zn);
    }
}

}  // extern "C"
