// example_render.cpp -- the reference's per-frame sequence (App::Render, /root/reference/src/app.cpp:1037-1068)
// written against the drop-in C++ surface: GaussianCloud::ImportPly -> SplatRenderer::Init ->
// Sort + Render into an explicit RGBA32F framebuffer, dumped as a binary PPM-like float file.
//
//   g++ -std=c++17 -I. splatapult_amd/host/example_render.cpp -Lsplatapult_amd/lib -lmsplat -o example_render
//   ./example_render scene.ply out.f32 [width height] [--nosh] [--frames-in-flight N] [--devices 0,1,2,...]
// With --frames-in-flight N the same frame is issued N + 1 times round-robin over N contexts that share the cloud
// (SplatRenderer::SetFramesInFlight); the last one is written.  With --devices the frame's bin rows are dealt to the
// listed GPUs (SplatRenderer::ConfigureDevices, msplat_group_*): same pixels.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "msplat_host.hpp"

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s scene.ply out.f32 [width height] [--nosh]\n", argv[0]);
        return 2;
    }
    int W = 1024, H = 768;   // the reference's default window (sdl_main.cpp:92)
    bool nosh = false;
    if (argc >= 5 && argv[3][0] != '-') { W = std::atoi(argv[3]); H = std::atoi(argv[4]); }
    int inFlight = 1;
    std::vector<int> devices;
    for (int i = 3; i < argc; ++i) {
        nosh = nosh || !std::strcmp(argv[i], "--nosh");
        if (!std::strcmp(argv[i], "--frames-in-flight") && i + 1 < argc) inFlight = std::atoi(argv[i + 1]);
        if (!std::strcmp(argv[i], "--devices") && i + 1 < argc)
            for (const char* p = argv[i + 1]; *p;) {
                devices.push_back(std::atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
    }

    auto cloud = std::make_shared<GaussianCloud>(GaussianCloud::Options{!nosh, false});
    if (!cloud->ImportPly(argv[1])) return 1;

    SplatRenderer renderer;
    renderer.SetFramesInFlight(inFlight);
    if (devices.size() > 1) renderer.ConfigureDevices(devices, MSPLAT_BANDS_BLOCK_INTERLEAVED, 2);
    if (!renderer.Init(cloud, /*isFramebufferSRGBEnabled=*/false, /*useRgcSortOverride=*/false)) return 1;

    // app.cpp:73-75,1039-1042: camera at the identity pose pulled back along +Z, 45 degree fovy
    msplat::mat4 cameraMat{}, projMat{};
    for (int i = 0; i < 4; ++i) cameraMat.m[i * 4 + i] = 1.0f;
    cameraMat.m[14] = 5.0f;
    const float zn = 0.1f, zf = 1000.0f;
    msplat_perspective(45.0f * 3.14159265358979f / 180.0f, (float)W / (float)H, zn, zf, projMat.m);
    msplat::vec4 viewport{{0.0f, 0.0f, (float)W, (float)H}};
    msplat::vec2 nearFar{{zn, zf}};

    std::vector<float> fb((size_t)W * H * 4);
    renderer.SetRenderTarget(fb.data(), 0, /*isDevicePointer=*/false);
    for (int k = 0; k < (inFlight > 1 ? inFlight + 1 : 1); ++k) {
        renderer.Sort(cameraMat, projMat, viewport, nearFar);       // moves on to the next context
        renderer.Render(cameraMat, projMat, viewport, nearFar);
    }
    renderer.Synchronize();

    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    std::fwrite(fb.data(), sizeof(float), fb.size(), f);
    std::fclose(f);
    std::printf("%zu splats -> %dx%d RGBA32F (row 0 = bottom) written to %s\n", cloud->GetNumGaussians(), W, H, argv[2]);
    return 0;
}
