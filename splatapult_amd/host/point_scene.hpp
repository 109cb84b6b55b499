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

    size_t GetNumPoints() const { return numPoints; }
    size_t GetStride() const { return pointSize; }
    size_t GetTotalSize() const { return GetNumPoints() * GetStride(); }
    void* GetRawDataPtr() { return data.get(); }
    const void* GetRawDataPtr() const { return data.get(); }

    const BinaryAttribute& GetPositionAttrib() const { return positionAttrib; }
    const BinaryAttribute& GetColorAttrib() const { return colorAttrib; }

    using ForEachPositionCallback = std::function<void(const float*)>;
    void ForEachPosition(const ForEachPositionCallback& cb) const;

protected:
    void InitAttribs();
    void Alloc(size_t n);

    std::shared_ptr<void> data;
    BinaryAttribute positionAttrib;
    BinaryAttribute colorAttrib;
    size_t numPoints = 0;
    size_t pointSize = 0;
    bool useLinearColors;
};
