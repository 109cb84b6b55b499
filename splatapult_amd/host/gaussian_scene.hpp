// gaussian_scene.hpp -- dependency-free (no glm / Eigen) re-implementation of the reference's
// scene-data surface that feeds the splat renderer:
//   BinaryAttribute  (/root/reference/src/core/binaryattribute.h:12-111)
//   Ply              (/root/reference/src/ply.h:19-46, ply.cpp:72-281)
//   GaussianCloud    (/root/reference/src/gaussiancloud.h:17-91, gaussiancloud.cpp:138-365,633-657)
// Same class names, method names, argument meaning and return conventions, so code written
// against the reference keeps compiling; the implementations are new.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

// A typed field at a fixed byte offset inside a record.
class BinaryAttribute
{
public:
    enum class Type { Unknown, Char, UChar, Short, UShort, Int, UInt, Float, Double, NumTypes };

    BinaryAttribute() = default;
    BinaryAttribute(Type typeIn, size_t offsetIn) : type(typeIn), size(SizeOf(typeIn)), offset(offsetIn) {}

    static size_t SizeOf(Type t)
    {
        switch (t) {
        case Type::Char: case Type::UChar: return 1;
        case Type::Short: case Type::UShort: return 2;
        case Type::Int: case Type::UInt: case Type::Float: return 4;
        case Type::Double: return 8;
        default: return 0;
        }
    }

    // pointer to this attribute inside the record at `data`; nullptr for an unset attribute
    template <typename T> const T* Get(const void* data) const
    {
        return type == Type::Unknown ? nullptr
                                     : reinterpret_cast<const T*>(static_cast<const uint8_t*>(data) + offset);
    }
    template <typename T> T* Get(void* data)
    {
        return type == Type::Unknown ? nullptr : reinterpret_cast<T*>(static_cast<uint8_t*>(data) + offset);
    }
    // value, or 0 for an unset attribute (the reference's Read<T> contract)
    template <typename T> const T Read(const void* data) const
    {
        if (type == Type::Unknown) return T(0);
        T v;
        std::memcpy(&v, static_cast<const uint8_t*>(data) + offset, sizeof(T));
        return v;
    }
    template <typename T> bool Write(void* data, const T& val)
    {
        if (type == Type::Unknown) return false;
        std::memcpy(static_cast<uint8_t*>(data) + offset, &val, sizeof(T));
        return true;
    }
    template <typename T>
    void ForEachMut(void* data, size_t stride, size_t count, const std::function<void(T*)>& cb)
    {
        uint8_t* p = static_cast<uint8_t*>(data) + offset;
        for (size_t i = 0; i < count; ++i, p += stride) cb(reinterpret_cast<T*>(p));
    }
    template <typename T>
    void ForEach(const void* data, size_t stride, size_t count, const std::function<void(const T*)>& cb) const
    {
        const uint8_t* p = static_cast<const uint8_t*>(data) + offset;
        for (size_t i = 0; i < count; ++i, p += stride) cb(reinterpret_cast<const T*>(p));
    }

    Type type = Type::Unknown;
    size_t size = 0;
    size_t offset = 0;
};

// Binary little-endian PLY with a single "element vertex" block of scalar properties.
class Ply
{
public:
    Ply() = default;
    bool Parse(std::ifstream& plyFile);
    void Dump(std::ofstream& plyFile) const;

    bool GetProperty(const std::string& key, BinaryAttribute& attributeOut) const;
    void AddProperty(const std::string& key, BinaryAttribute::Type type);
    void AllocData(size_t numVertices);

    using VertexCallback = std::function<void(const void*, size_t)>;
    void ForEachVertex(const VertexCallback& cb) const;
    using VertexCallbackMut = std::function<void(void*, size_t)>;
    void ForEachVertexMut(const VertexCallbackMut& cb);

    size_t GetVertexCount() const { return vertexCount; }
    // extensions (not in the reference): bulk access for the GPU ingest path
    size_t GetVertexSize() const { return vertexSize; }
    const uint8_t* GetRawData() const { return data.data(); }

protected:
    bool ParseHeader(std::ifstream& plyFile);
    void DumpHeader(std::ofstream& plyFile) const;

    std::unordered_map<std::string, BinaryAttribute> propertyMap;
    std::vector<uint8_t> data;
    size_t vertexCount = 0;
    size_t vertexSize = 0;
};

class GaussianCloud
{
public:
    struct Options
    {
        bool importFullSH;
        bool exportFullSH;
    };

    explicit GaussianCloud(const Options& options);

    bool ImportPly(const std::string& plyFilename);
    bool ExportPly(const std::string& plyFilename) const;
    void InitDebugCloud();
    // only keep the nearest splats (origin = float[3])
    void PruneSplats(const float origin[3], uint32_t numGaussians);

    // extension: same per-vertex math as ImportPly, from attribute arrays (f_rest may be null)
    bool FromAttributes(size_t n, const float* xyz, const float* f_dc, const float* f_rest,
                        const float* opacity, const float* logScale, const float* rot);

    size_t GetNumGaussians() const { return numGaussians; }
    size_t GetStride() const { return gaussianSize; }
    size_t GetTotalSize() const { return GetNumGaussians() * gaussianSize; }
    void* GetRawDataPtr() { return data.get(); }
    const void* GetRawDataPtr() const { return data.get(); }

    const BinaryAttribute& GetPosWithAlphaAttrib() const { return posWithAlphaAttrib; }
    const BinaryAttribute& GetR_SH0Attrib() const { return r_sh0Attrib; }
    const BinaryAttribute& GetR_SH1Attrib() const { return r_sh1Attrib; }
    const BinaryAttribute& GetR_SH2Attrib() const { return r_sh2Attrib; }
    const BinaryAttribute& GetR_SH3Attrib() const { return r_sh3Attrib; }
    const BinaryAttribute& GetG_SH0Attrib() const { return g_sh0Attrib; }
    const BinaryAttribute& GetG_SH1Attrib() const { return g_sh1Attrib; }
    const BinaryAttribute& GetG_SH2Attrib() const { return g_sh2Attrib; }
    const BinaryAttribute& GetG_SH3Attrib() const { return g_sh3Attrib; }
    const BinaryAttribute& GetB_SH0Attrib() const { return b_sh0Attrib; }
    const BinaryAttribute& GetB_SH1Attrib() const { return b_sh1Attrib; }
    const BinaryAttribute& GetB_SH2Attrib() const { return b_sh2Attrib; }
    const BinaryAttribute& GetB_SH3Attrib() const { return b_sh3Attrib; }
    const BinaryAttribute& GetCov3_Col0Attrib() const { return cov3_col0Attrib; }
    const BinaryAttribute& GetCov3_Col1Attrib() const { return cov3_col1Attrib; }
    const BinaryAttribute& GetCov3_Col2Attrib() const { return cov3_col2Attrib; }

    using ForEachPosWithAlphaCallback = std::function<void(const float*)>;
    void ForEachPosWithAlpha(const ForEachPosWithAlphaCallback& cb) const;

    bool HasFullSH() const { return hasFullSH; }

protected:
    void InitAttribs();
    void Allocate(size_t n, bool fullSH);

    std::shared_ptr<void> data;

    BinaryAttribute posWithAlphaAttrib;
    BinaryAttribute r_sh0Attrib, r_sh1Attrib, r_sh2Attrib, r_sh3Attrib;
    BinaryAttribute g_sh0Attrib, g_sh1Attrib, g_sh2Attrib, g_sh3Attrib;
    BinaryAttribute b_sh0Attrib, b_sh1Attrib, b_sh2Attrib, b_sh3Attrib;
    BinaryAttribute cov3_col0Attrib, cov3_col1Attrib, cov3_col2Attrib;

    size_t numGaussians = 0;
    size_t gaussianSize = 0;

    Options opt;
    bool hasFullSH = false;
};
