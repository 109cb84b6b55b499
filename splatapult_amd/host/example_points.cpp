// example_points.cpp -- the reference's point-cloud view (App::Render with opt.drawPointCloud,
// /root/reference/src/app.cpp:1061-1064) written against the drop-in C++ surface: PointCloud -> PointRenderer::Init
// -> Render, with several frames in flight on the splat side to exercise SplatRenderer::SetFramesInFlight.
//
//   g++ -std=c++17 -I. splatapult_amd/host/example_points.cpp -Lsplatapult_amd/lib -lmsplat -o example_points
//   ./example_points out_points.f32 width height [input.ply] [sprite.png]
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "msplat_host.hpp"
#include "scene_config.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s out.f32 width height [input.ply] [sprite.png]\n", argv[0]);
        return 2;
    }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    auto points = std::make_shared<PointCloud>(/*useLinearColors=*/false);
    if (argc >= 5) {
        if (!points->ImportPly(argv[4])) return 1;
    } else {
        points->InitDebugCloud();            // pointcloud.cpp:199-258: three coloured axis lines
    }

    PointRenderer renderer;
    std::vector<uint8_t> sprite;
    if (argc >= 6) {                         // the reference loads texture/sphere.png (pointrenderer.cpp:54)
        int sw = 0, sh = 0;
        if (!ReadPNG(argv[5], sprite, sw, sh)) return 1;
        renderer.SetSprite(sprite.data(), (uint32_t)sw, (uint32_t)sh);
    }
    if (!renderer.Init(points, /*isFramebufferSRGBEnabled=*/false)) return 1;

    msplat::mat4 cameraMat{}, projMat{};
    for (int i = 0; i < 4; ++i) cameraMat.m[i * 4 + i] = 1.0f;
    cameraMat.m[12] = 0.4f;
    cameraMat.m[13] = 0.4f;
    cameraMat.m[14] = 2.5f;
    const float zn = 0.1f, zf = 1000.0f;
    msplat_perspective(45.0f * 3.14159265358979f / 180.0f, (float)W / (float)H, zn, zf, projMat.m);
    msplat::vec4 viewport{{0.0f, 0.0f, (float)W, (float)H}};
    msplat::vec2 nearFar{{zn, zf}};

    std::vector<float> fb((size_t)W * H * 4);
    renderer.SetRenderTarget(fb.data(), 0, /*isDevicePointer=*/false);
    renderer.Render(cameraMat, projMat, viewport, nearFar);          // sorts and draws (pointrenderer.cpp:113-196)

    FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 1;
    std::fwrite(fb.data(), sizeof(float), fb.size(), f);
    std::fclose(f);
    std::printf("%zu points -> %dx%d RGBA32F written to %s\n", points->GetNumPoints(), W, H, argv[1]);
    return 0;
}
