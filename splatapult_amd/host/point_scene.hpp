// point_scene.hpp -- the SfM point cloud that the reference can draw instead of the splats ('c' key), SURVEY.md
// 8f-4: PointCloud (/root/reference/src/pointcloud.h:15-48, pointcloud.cpp:18-264).  Same class and method names
// as the reference; built on this repo's own Ply / BinaryAttribute (gaussian_scene.hpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

#include "gaussian_scene.hpp"

class PointCloud
{
public:
    // useLinearColorsIn = App passes isFramebufferSRGBEnabled (app.cpp:216)
    explicit PointCloud(bool useLinearColorsIn);

    bool ImportPly(const std::string& plyFilename);
    bool ExportPly(const std::string& plyFilename) const;
    void InitDebugCloud();

    size_t GetNumPoints() const { return count_; }
    size_t GetStride() const { return recordBytes_; }
    size_t GetTotalSize() const { return GetNumPoints() * GetStride(); }
    void* GetRawDataPtr() { return records_.get(); }
    const void* GetRawDataPtr() const { return records_.get(); }

    const BinaryAttribute& GetPositionAttrib() const { return posAttr_; }
    const BinaryAttribute& GetColorAttrib() const { return rgbAttr_; }

    using ForEachPositionCallback = std::function<void(const float*)>;
    void ForEachPosition(const ForEachPositionCallback& cb) const;

private:
    // one interleaved record per point (position + colour, see point_scene.cpp); the accessors above describe it
    void Reserve(size_t n);
    void SetUpAttributes();

    size_t count_ = 0, recordBytes_ = 0;
    std::shared_ptr<void> records_;
    BinaryAttribute posAttr_, rgbAttr_;
    const bool linearColours_;
};
