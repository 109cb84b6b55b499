// scene_config.hpp -- the config files that travel with a scene and image output
// (8f-3), dependency-free (no glm, nlohmann-json or libpng):
//   CamerasConfig  (/root/reference/src/camerasconfig.h:12-32, camerasconfig.cpp:20-95)  cameras.json
//   VrConfig       (/root/reference/src/vrconfig.h:12-23, vrconfig.cpp:20-65)            <scene>_vr.json
//   FindConfigFile / MakeVrConfigFilename (/root/reference/src/app.cpp:89-142)
//   LinearToSRGB / SRGBToLinear (/root/reference/src/core/util.cpp:357-380), 8-bit presentation as the
//   desktop blit does it (shader/desktop_frag.glsl:19-37 into an RGBA8 back buffer), PNG / PPM writers.
// Same class and method names as the reference; matrices are float[16] column-major instead of glm::mat4.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct Camera
{
    float mat[16];   // inverse view matrix (camera-to-world), column-major
    float fov[2];
};

class CamerasConfig
{
public:
    CamerasConfig() = default;
    bool ImportJson(const std::string& jsonFilename);
    const std::vector<Camera>& GetCameraVec() const { return cameraVec; }
    size_t GetNumCameras() const { return cameraVec.size(); }
    void EstimateFloorPlane(float normalOut[3], float posOut[3]) const;

protected:
    std::vector<Camera> cameraVec;
};

class VrConfig
{
public:
    VrConfig();
    bool ImportJson(const std::string& jsonFilename);
    bool ExportJson(const std::string& jsonFilename) const;
    const float* GetFloorMat() const { return floorMat; }
    void SetFloorMat(const float floorMatIn[16]);

protected:
    float floorMat[16];
};

// searches the directory of plyFilename, its parent and grandparent for configFilename ("" if not found)
std::string FindConfigFile(const std::string& plyFilename, const std::string& configFilename);
// <dir>/<stem>_vr.json
std::string MakeVrConfigFilename(const std::string& plyFilename);

float LinearToSRGB(float linear);
float SRGBToLinear(float srgb);

// W x H float RGBA (row 0 = GL bottom row) -> 8-bit RGBA, top row first when flipY (image files)
void PresentRGBA8(const float* rgba, int width, int height, bool flipY, bool encodeSRGB, uint8_t* out);
bool WritePNG(const std::string& filename, const uint8_t* rgba8, int width, int height);
// 8-bit gray / gray+alpha / RGB / RGBA non-interlaced PNG (what core/image.cpp:72-101 accepts) -> RGBA8, top row first
bool ReadPNG(const std::string& filename, std::vector<uint8_t>& rgba8, int& width, int& height);
bool WritePPM(const std::string& filename, const uint8_t* rgba8, int width, int height);
