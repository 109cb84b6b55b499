// msplat_host.hpp -- C++ drop-in shim: the reference's SplatRenderer class surface
// (/root/reference/src/splatrenderer.h:23-67) on top of the C ABI in include/msplat.h.
//
//   bool Init(std::shared_ptr<GaussianCloud>, bool isFramebufferSRGBEnabled, bool useRgcSortOverride);
//   void Sort  (cameraMat, projMat, viewport, nearFar);
//   void Render(cameraMat, projMat, viewport, nearFar);
//
// Matrix/vector arguments are templated on anything that is laid out like glm's types (a mat4 is
// 16 contiguous column-major floats, vec4/vec2 are 4/2 floats): glm::mat4, glm::vec4, glm::vec2 or
// the msplat::mat4/vec4/vec2 PODs below all work unchanged, so App::Render's two call sites
// (app.cpp:603-607, :1067-1068) compile as they are.
//
// The reference draws into "whatever GL framebuffer is bound"; here the target is explicit:
// SetRenderTarget(ptr, pitch, isDevice) once (or per frame), then Render writes RGBA32F/16F rows,
// row 0 = GL bottom row, alpha = 1.
#pragma once

#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

#include "../../include/msplat.h"
#include "../../include/msplat_debug.h"
#include "gaussian_scene.hpp"
#include "point_scene.hpp"

namespace msplat {
struct mat4 { float m[16]; };   // column-major, m[col*4 + row]
struct vec4 { float v[4]; };
struct vec2 { float v[2]; };
}  // namespace msplat

class SplatRenderer
{
public:
    SplatRenderer() = default;
    ~SplatRenderer() { DestroyContexts(); }
    SplatRenderer(const SplatRenderer&) = delete;
    SplatRenderer& operator=(const SplatRenderer&) = delete;

    // optional, before Init: device ordinal, framebuffer format (MSPLAT_FB_*), stream (hipStream_t)
    void Configure(int device, int fbFormat, void* stream = nullptr, float tEpsilon = -1.0f)
    {
        cfg.device = device;
        cfg.fb_format = fbFormat;
        cfg.stream = stream;
        cfg.t_epsilon = tEpsilon;
    }

    // optional, before Init: render on several GPUs from this one process (msplat_group_* in msplat.h).  The
    // cloud is replicated, the screen's bin rows are partitioned over `devices` (MSPLAT_BANDS_*: contiguous bands by
    // default) and every device writes its rows into the render target, which must be host memory or memory of
    // devices[0] (the other devices reach it through the xGMI peer mapping).  Sort / Render are unchanged.
    // bandCull: Sort also drops splats that cannot reach a device's rows (mono rendering only: every Render must use its
    // Sort's camera).  Frames in flight are not combined with a device group.
    void ConfigureDevices(const std::vector<int>& devicesIn, int bandKind = MSPLAT_BANDS_CONTIGUOUS, int blockRows = 1,
                          bool bandCull = false)
    {
        groupDevices.assign(devicesIn.begin(), devicesIn.end());
        groupKind = bandKind;
        groupBlockRows = blockRows;
        groupBandCull = bandCull;
    }

    // optional, before Init, with ConfigureDevices: how the other devices' rows reach devices[0]'s render target --
    // MSPLAT_EXCHANGE_PEER_STORE (default: their compositors store through the peer mapping), MSPLAT_EXCHANGE_RCCL (grouped
    // ncclSend / ncclRecv over communicators from ncclCommInitAll) or MSPLAT_EXCHANGE_COPY.  Not available (RCCL missing, a
    // device listed twice): logged, the group keeps its previous exchange.
    void SetGroupExchange(int exchange) { groupExchange = exchange; }

    // One process per GPU (the other multi-GPU shape): this process renders the bin rows of rank `rank` of `world`
    // (msplat_band_plan's layouts) and, after Render into a device target, ExchangeBands() gathers every rank's rows into rank
    // `root`'s target with ONE group of ncclSend / ncclRecv on the context's stream (msplat_band_exchange; `comm` = the host's
    // ncclComm_t).  Call SetBandPlan after Init.
    bool SetBandPlan(int rank, int world, int rowsFull, int bandKind = MSPLAT_BANDS_CONTIGUOUS, int blockRows = 1, bool bandCull = false)
    {
        int32_t first = 0, count = 0, block = 1, stride = 1;
        if (msplat_band_plan(bandKind, rowsFull, world, rank, blockRows, &first, &count, &block, &stride) != MSPLAT_OK) {
            std::fprintf(stderr, "[msplat][E] SetBandPlan: %s\n", msplat_last_error(nullptr));
            return false;
        }
        bandRank = rank; bandWorld = world; bandKindSet = bandKind; bandBlockRows = blockRows;
        for (msplat_ctx* h : ctxs)
            if (msplat_set_band_layout(h, first, count, block, stride) != MSPLAT_OK || msplat_set_band_cull(h, bandCull ? 1 : 0) != MSPLAT_OK) {
                std::fprintf(stderr, "[msplat][E] SetBandPlan: %s\n", msplat_last_error(h));
                return false;
            }
        return true;
    }
    bool ExchangeBands(void* comm, int root, int width, int height, int flags = 0)        // flags: MSPLAT_EXCHANGE_WIRE_FP16
    {
        if (!ctx || !targetIsDevice || !target) return false;
        const uint64_t pitch = targetPitch ? targetPitch : (uint64_t)width * (cfg.fb_format == MSPLAT_FB_RGBA16F ? 8u : 16u);
        const int rc = msplat_band_exchange(ctx, comm, bandRank, bandWorld, root, bandKindSet, bandBlockRows, target, pitch, width, height, flags);
        if (rc != MSPLAT_OK) std::fprintf(stderr, "[msplat][E] ExchangeBands: %s\n", msplat_group_last_error(nullptr));
        return rc == MSPLAT_OK;
    }

    // optional, before Init: number of frames in flight (default 1).  With depth > 1 every Sort moves on to
    // the next of `depth` contexts (own stream + per-frame buffers, one shared cloud), so successive frames
    // overlap on the GPU the way a GL driver overlaps queued frames; the following Render(s) use the context
    // of the latest Sort.  Each context creates its own stream (a stream given to Configure is only used
    // with depth 1): use WaitOnStream / Synchronize before consuming a frame, and one render target per
    // frame in flight.
    void SetFramesInFlight(int depth) { framesInFlight = depth < 1 ? 1 : depth; }

    // msplat_config.two_pass (before Init): MSPLAT_TWO_PASS_AUTO (default) / _ON / _OFF -- Render in two passes with occlusion
    // feedback, the same pixels (include/msplat.h).  TwoPassInfo: what the current context's latest two-pass Render did
    // (msplat_get_two_pass_info; out[0] == 0: it ran in one pass).
    void SetTwoPass(int mode) { cfg.two_pass = mode; }
    bool TwoPassInfo(uint64_t out[8]) const { return ctx != nullptr && msplat_get_two_pass_info(ctx, out) == MSPLAT_OK; }

    // splatrenderer.cpp:50-151.  false after logging on failure.  The cloud is copied to the device and
    // not retained; useRgcSortOverride is accepted and ignored (one HIP sort replaces both GL sorters).
    bool Init(std::shared_ptr<GaussianCloud> gaussianCloud, bool isFramebufferSRGBEnabledIn, bool useRgcSortOverrideIn)
    {
        (void)useRgcSortOverrideIn;
        DestroyContexts();
        cfg.struct_size = sizeof(cfg);
        cfg.srgb = isFramebufferSRGBEnabledIn ? 1 : 0;
        if (groupDevices.size() > 1) return InitGroup(*gaussianCloud);
        msplat_config c = cfg;
        if (framesInFlight > 1) {
            c.stream = nullptr;
            c.compositor_waves = 1280;       // frames share the CUs (measured r3: 768 .. 2048)
            c.frame_mode = MSPLAT_FRAMES_IN_FLIGHT;   // kernels that co-schedule well with other frames' kernels (+2-3 %)
            c.async_submit = 1;                       // a worker thread per context issues its launches (msplat.h)
        }
        // r6: an even number >= 4 of frames in flight: the contexts' streams alternate between the even and the odd CU positions of
        // every XCD -- two frames per half of the CUs instead of four on all of them (+3-5 %)
        const bool halves = framesInFlight >= 4 && framesInFlight % 2 == 0 && cfg.cu_partition == MSPLAT_CU_ALL;
        for (int k = 0; k < framesInFlight; ++k) {
            msplat_ctx* h = nullptr;
            if (halves) c.cu_partition = MSPLAT_CU_EVEN + (k & 1);
            if (msplat_create(&h, &c) != MSPLAT_OK) {
                std::fprintf(stderr, "[msplat][E] %s\n", msplat_last_error(nullptr));
                DestroyContexts();
                return false;
            }
            ctxs.push_back(h);
        }
        ctx = ctxs[0];
        cur = framesInFlight - 1;            // the first Sort lands on context 0
        msplat_attr_offsets off{};
        off.pos_with_alpha = (uint32_t)gaussianCloud->GetPosWithAlphaAttrib().offset;
        off.r_sh0 = (uint32_t)gaussianCloud->GetR_SH0Attrib().offset;
        off.g_sh0 = (uint32_t)gaussianCloud->GetG_SH0Attrib().offset;
        off.b_sh0 = (uint32_t)gaussianCloud->GetB_SH0Attrib().offset;
        off.cov3_col0 = (uint32_t)gaussianCloud->GetCov3_Col0Attrib().offset;
        off.cov3_col1 = (uint32_t)gaussianCloud->GetCov3_Col1Attrib().offset;
        off.cov3_col2 = (uint32_t)gaussianCloud->GetCov3_Col2Attrib().offset;
        if (gaussianCloud->HasFullSH()) {
            off.r_sh1 = (uint32_t)gaussianCloud->GetR_SH1Attrib().offset;
            off.r_sh2 = (uint32_t)gaussianCloud->GetR_SH2Attrib().offset;
            off.r_sh3 = (uint32_t)gaussianCloud->GetR_SH3Attrib().offset;
            off.g_sh1 = (uint32_t)gaussianCloud->GetG_SH1Attrib().offset;
            off.g_sh2 = (uint32_t)gaussianCloud->GetG_SH2Attrib().offset;
            off.g_sh3 = (uint32_t)gaussianCloud->GetG_SH3Attrib().offset;
            off.b_sh1 = (uint32_t)gaussianCloud->GetB_SH1Attrib().offset;
            off.b_sh2 = (uint32_t)gaussianCloud->GetB_SH2Attrib().offset;
            off.b_sh3 = (uint32_t)gaussianCloud->GetB_SH3Attrib().offset;
        }
        if (msplat_upload_cloud(ctx, gaussianCloud->GetRawDataPtr(), gaussianCloud->GetNumGaussians(),
                                (uint32_t)gaussianCloud->GetStride(), &off, gaussianCloud->HasFullSH() ? 1 : 0) != MSPLAT_OK) {
            std::fprintf(stderr, "[msplat][E] %s\n", msplat_last_error(ctx));
            return false;
        }
        for (size_t k = 1; k < ctxs.size(); ++k)
            if (msplat_attach_cloud(ctxs[k], ctxs[0]) != MSPLAT_OK) {
                std::fprintf(stderr, "[msplat][E] %s\n", msplat_last_error(ctxs[k]));
                return false;
            }
        return true;
    }

    // splatrenderer.cpp:153-312
    template <class Mat4, class Vec4, class Vec2>
    void Sort(const Mat4& cameraMat, const Mat4& projMat, const Vec4& viewport, const Vec2& nearFar)
    {
        static_assert(sizeof(Mat4) == 64 && sizeof(Vec4) == 16 && sizeof(Vec2) == 8, "glm-compatible layout expected");
        if (group) {
            const int grc = msplat_group_sort(group, reinterpret_cast<const float*>(&cameraMat), reinterpret_cast<const float*>(&projMat),
                                              reinterpret_cast<const float*>(&viewport), reinterpret_cast<const float*>(&nearFar));
            if (grc != MSPLAT_OK) std::fprintf(stderr, "[msplat][%c] Sort: %s\n", Level(grc), msplat_group_last_error(group));
            return;
        }
        if (ctxs.empty()) return;
        cur = (cur + 1) % (int)ctxs.size();
        ctx = ctxs[cur];
        const int rc = msplat_sort(ctx, reinterpret_cast<const float*>(&cameraMat), reinterpret_cast<const float*>(&projMat),
                                   reinterpret_cast<const float*>(&viewport), reinterpret_cast<const float*>(&nearFar));
        if (rc != MSPLAT_OK)         // void, like the reference
            std::fprintf(stderr, "[msplat][%c] Sort: %s\n", Level(rc), msplat_last_error(ctx));
    }

    // splatrenderer.cpp:315-343 (+ the GL pipeline behind glDrawElements)
    template <class Mat4, class Vec4, class Vec2>
    void Render(const Mat4& cameraMat, const Mat4& projMat, const Vec4& viewport, const Vec2& nearFar)
    {
        static_assert(sizeof(Mat4) == 64 && sizeof(Vec4) == 16 && sizeof(Vec2) == 8, "glm-compatible layout expected");
        if (!target) {
            std::fprintf(stderr, "[msplat][E] Render: no render target set (SetRenderTarget)\n");
            return;
        }
        if (group) {
            const int grc = msplat_group_render(group, reinterpret_cast<const float*>(&cameraMat), reinterpret_cast<const float*>(&projMat),
                                                reinterpret_cast<const float*>(&viewport), reinterpret_cast<const float*>(&nearFar), target,
                                                targetPitch, targetIsDevice ? 1 : 0);
            if (grc != MSPLAT_OK) std::fprintf(stderr, "[msplat][%c] Render: %s\n", Level(grc), msplat_group_last_error(group));
            return;
        }
        const int rc = msplat_render(ctx, reinterpret_cast<const float*>(&cameraMat), reinterpret_cast<const float*>(&projMat),
                                     reinterpret_cast<const float*>(&viewport), reinterpret_cast<const float*>(&nearFar), target,
                                     targetPitch, targetIsDevice ? 1 : 0);
        if (rc != MSPLAT_OK) std::fprintf(stderr, "[msplat][%c] Render: %s\n", Level(rc), msplat_last_error(ctx));
    }

    // Both eyes of the latest Sort in one chain of launches (msplat_render_stereo): what the XR callback does with two Render
    // calls (app.cpp:603-607), for device targets at half the launches.  target1: the second eye's image (same pitch / kind as
    // the SetRenderTarget one, which receives the first eye).
    template <class Mat4, class Vec4, class Vec2>
    void RenderStereo(const Mat4& cameraMat0, const Mat4& projMat0, const Mat4& cameraMat1, const Mat4& projMat1, const Vec4& viewport,
                      const Vec2& nearFar, void* target1)
    {
        static_assert(sizeof(Mat4) == 64 && sizeof(Vec4) == 16 && sizeof(Vec2) == 8, "glm-compatible layout expected");
        if (!target || !target1) {
            std::fprintf(stderr, "[msplat][E] RenderStereo: no render target set (SetRenderTarget / target1)\n");
            return;
        }
        if (group) {         // a device group renders its rows per view
            Render(cameraMat0, projMat0, viewport, nearFar);
            void* keep = target;
            target = target1;
            Render(cameraMat1, projMat1, viewport, nearFar);
            target = keep;
            return;
        }
        const int rc = msplat_render_stereo(ctx, reinterpret_cast<const float*>(&cameraMat0), reinterpret_cast<const float*>(&projMat0),
                                            reinterpret_cast<const float*>(&cameraMat1), reinterpret_cast<const float*>(&projMat1),
                                            reinterpret_cast<const float*>(&viewport), reinterpret_cast<const float*>(&nearFar), target,
                                            target1, targetPitch, targetIsDevice ? 1 : 0);
        if (rc != MSPLAT_OK) std::fprintf(stderr, "[msplat][%c] RenderStereo: %s\n", Level(rc), msplat_last_error(ctx));
    }

    // replaces "the currently bound GL framebuffer" (app.cpp:1000-1035)
    void SetRenderTarget(void* rgba, uint64_t pitchBytes, bool isDevicePointer)
    {
        target = rgba;
        targetPitch = pitchBytes;
        targetIsDevice = isDevicePointer;
    }

    // emulated depth buffer for targets that have one in the GL app (app.cpp:163; 24 = default back buffer)
    void SetDepthTest(int depthBits)
    {
        for (msplat_ctx* h : ctxs) msplat_set_depth_test(h, depthBits);
    }

    // the blend as the GL app's own render target performs it (MSPLAT_ROP_RGBA8: default back buffer, MSPLAT_ROP_RGBA16F: --fp16)
    void SetTargetEmulation(int rop)
    {
        for (msplat_ctx* h : ctxs) msplat_set_target_emulation(h, rop);
    }

    // blocks until every frame in flight has finished
    // (also where a pair-buffer overflow of an earlier device-target Render is reported, see msplat_render)
    void Synchronize()
    {
        if (group && msplat_group_synchronize(group) != MSPLAT_OK)
            std::fprintf(stderr, "[msplat][E] Synchronize: %s\n", msplat_group_last_error(group));
        for (msplat_ctx* h : ctxs) {
            // (MSPLAT_ERR_PAIR_OVERFLOW_EARLIER: a queued call of an async_submit context found an earlier frame's overflow)
            const int rc = msplat_synchronize(h);
            if (rc != MSPLAT_OK) std::fprintf(stderr, "[msplat][%c] Synchronize: %s\n", Level(rc), msplat_last_error(h));
        }
    }
    // device-side join: `stream` (hipStream_t) waits for the frame issued last (the latest Sort's context)
    void WaitOnStream(void* stream)
    {
        if (!ctx) return;
        const int rc = msplat_stream_wait(ctx, stream);
        if (rc != MSPLAT_OK) std::fprintf(stderr, "[msplat][%c] WaitOnStream: %s\n", Level(rc), msplat_last_error(ctx));
    }

    // reverse join: the context the NEXT Sort will use waits for `event` (hipEvent_t), e.g. recorded after the
    // consumer of the render target that frame is going to overwrite
    void NextFrameWaitEvent(void* event)
    {
        if (!ctxs.empty()) msplat_wait_event(ctxs[(cur + 1) % (int)ctxs.size()], event);
    }
    int GetFrameSlot() const { return cur; }

    msplat_ctx* GetContext() { return group ? msplat_group_context(group, 0) : ctx; }   // the context of the latest Sort
    msplat_group* GetGroup() { return group; }

public:
    uint32_t numBlocksPerWorkgroup = 1024;   // accepted and ignored (splatrenderer.h:39)

protected:
    bool InitGroup(GaussianCloud& cloud)
    {
        std::vector<int32_t> devs(groupDevices.begin(), groupDevices.end());
        if (msplat_group_create(&group, devs.data(), (uint32_t)devs.size(), &cfg) != MSPLAT_OK) {
            std::fprintf(stderr, "[msplat][E] %s\n", msplat_group_last_error(nullptr));
            group = nullptr;
            return false;
        }
        msplat_group_set_layout(group, groupKind, groupBlockRows);
        msplat_group_set_band_cull(group, groupBandCull ? 1 : 0);
        if (groupExchange != MSPLAT_EXCHANGE_PEER_STORE && msplat_group_set_exchange(group, groupExchange) != MSPLAT_OK)
            std::fprintf(stderr, "[msplat][W] SetGroupExchange: %s\n", msplat_group_last_error(group));
        msplat_attr_offsets off{};
        off.pos_with_alpha = (uint32_t)cloud.GetPosWithAlphaAttrib().offset;
        off.r_sh0 = (uint32_t)cloud.GetR_SH0Attrib().offset;
        off.g_sh0 = (uint32_t)cloud.GetG_SH0Attrib().offset;
        off.b_sh0 = (uint32_t)cloud.GetB_SH0Attrib().offset;
        off.cov3_col0 = (uint32_t)cloud.GetCov3_Col0Attrib().offset;
        off.cov3_col1 = (uint32_t)cloud.GetCov3_Col1Attrib().offset;
        off.cov3_col2 = (uint32_t)cloud.GetCov3_Col2Attrib().offset;
        if (cloud.HasFullSH()) {
            off.r_sh1 = (uint32_t)cloud.GetR_SH1Attrib().offset; off.r_sh2 = (uint32_t)cloud.GetR_SH2Attrib().offset;
            off.r_sh3 = (uint32_t)cloud.GetR_SH3Attrib().offset; off.g_sh1 = (uint32_t)cloud.GetG_SH1Attrib().offset;
            off.g_sh2 = (uint32_t)cloud.GetG_SH2Attrib().offset; off.g_sh3 = (uint32_t)cloud.GetG_SH3Attrib().offset;
            off.b_sh1 = (uint32_t)cloud.GetB_SH1Attrib().offset; off.b_sh2 = (uint32_t)cloud.GetB_SH2Attrib().offset;
            off.b_sh3 = (uint32_t)cloud.GetB_SH3Attrib().offset;
        }
        if (msplat_group_upload_cloud(group, cloud.GetRawDataPtr(), cloud.GetNumGaussians(), (uint32_t)cloud.GetStride(), &off,
                                      cloud.HasFullSH() ? 1 : 0) != MSPLAT_OK) {
            std::fprintf(stderr, "[msplat][E] %s\n", msplat_group_last_error(group));
            return false;
        }
        return true;
    }

    // log level of a status code: MSPLAT_ERR_PAIR_OVERFLOW_EARLIER reports a PAST frame, the call itself did its work
    static char Level(int rc) { return rc == MSPLAT_ERR_PAIR_OVERFLOW_EARLIER ? 'W' : 'E'; }

    void DestroyContexts()
    {
        if (group) msplat_group_destroy(group);
        group = nullptr;
        for (msplat_ctx* h : ctxs) msplat_destroy(h);
        ctxs.clear();
        ctx = nullptr;
    }

    msplat_group* group = nullptr;       // set when ConfigureDevices named more than one device
    std::vector<int> groupDevices;
    int groupKind = MSPLAT_BANDS_CONTIGUOUS, groupBlockRows = 1;
    bool groupBandCull = false;
    int groupExchange = MSPLAT_EXCHANGE_PEER_STORE;
    int bandRank = 0, bandWorld = 1, bandKindSet = MSPLAT_BANDS_CONTIGUOUS, bandBlockRows = 1;

    std::vector<msplat_ctx*> ctxs;       // one per frame in flight; ctxs[0] owns the cloud
    msplat_ctx* ctx = nullptr;           // == ctxs[cur]
    int cur = 0;
    int framesInFlight = 1;
    msplat_config cfg{sizeof(msplat_config), 0, MSPLAT_FB_RGBA32F, 0, -1.0f, 0, nullptr, 0, 0, 0, 0};
    void* target = nullptr;
    uint64_t targetPitch = 0;
    bool targetIsDevice = false;
};

// The SfM point-cloud view: PointRenderer's surface (/root/reference/src/pointrenderer.h:23-57).
// Render sorts and draws in one call, like the reference (pointrenderer.cpp:113-196).
class PointRenderer
{
public:
    PointRenderer() = default;
    ~PointRenderer() { msplat_destroy(ctx); }
    PointRenderer(const PointRenderer&) = delete;
    PointRenderer& operator=(const PointRenderer&) = delete;

    void Configure(int device, int fbFormat, void* stream = nullptr)
    {
        cfg.device = device;
        cfg.fb_format = fbFormat;
        cfg.stream = stream;
    }
    // optional, before Init: the sprite (the reference loads texture/sphere.png, pointrenderer.cpp:54-64);
    // RGBA8, top row first as decoded from the file (ReadPNG).  Default: the library's built-in sphere.
    void SetSprite(const uint8_t* rgba8, uint32_t width, uint32_t height)
    {
        sprite.assign(rgba8, rgba8 + (size_t)width * height * 4);
        spriteW = width;
        spriteH = height;
    }

    bool Init(std::shared_ptr<PointCloud> pointCloud, bool isFramebufferSRGBEnabledIn)
    {
        msplat_destroy(ctx);
        ctx = nullptr;
        cfg.struct_size = sizeof(cfg);
        cfg.srgb = isFramebufferSRGBEnabledIn ? 1 : 0;
        if (msplat_create(&ctx, &cfg) != MSPLAT_OK) {
            std::fprintf(stderr, "[msplat][E] %s\n", msplat_last_error(nullptr));
            return false;
        }
        if (msplat_upload_points(ctx, pointCloud->GetRawDataPtr(), pointCloud->GetNumPoints(), (uint32_t)pointCloud->GetStride(),
                                 (uint32_t)pointCloud->GetPositionAttrib().offset,
                                 (uint32_t)pointCloud->GetColorAttrib().offset) != MSPLAT_OK ||
            msplat_set_point_sprite(ctx, sprite.empty() ? nullptr : sprite.data(), spriteW, spriteH) != MSPLAT_OK) {
            std::fprintf(stderr, "[msplat][E] %s\n", msplat_last_error(ctx));
            return false;
        }
        return true;
    }

    template <class Mat4, class Vec4, class Vec2>
    void Render(const Mat4& cameraMat, const Mat4& projMat, const Vec4& viewport, const Vec2& nearFar)
    {
        static_assert(sizeof(Mat4) == 64 && sizeof(Vec4) == 16 && sizeof(Vec2) == 8, "glm-compatible layout expected");
        if (!target) {
            std::fprintf(stderr, "[msplat][E] Render: no render target set (SetRenderTarget)\n");
            return;
        }
        const float* c = reinterpret_cast<const float*>(&cameraMat);
        const float* p = reinterpret_cast<const float*>(&projMat);
        const float* v = reinterpret_cast<const float*>(&viewport);
        const float* nf = reinterpret_cast<const float*>(&nearFar);
        if (msplat_sort(ctx, c, p, v, nf) != MSPLAT_OK ||
            msplat_render(ctx, c, p, v, nf, target, targetPitch, targetIsDevice ? 1 : 0) != MSPLAT_OK)
            std::fprintf(stderr, "[msplat][E] PointRenderer::Render: %s\n", msplat_last_error(ctx));
    }

    void SetRenderTarget(void* rgba, uint64_t pitchBytes, bool isDevicePointer)
    {
        target = rgba;
        targetPitch = pitchBytes;
        targetIsDevice = isDevicePointer;
    }
    msplat_ctx* GetContext() { return ctx; }

protected:
    msplat_ctx* ctx = nullptr;
    msplat_config cfg{sizeof(msplat_config), 0, MSPLAT_FB_RGBA32F, 0, -1.0f, 0, nullptr, 0, 0, 0, 0};
    std::vector<uint8_t> sprite;
    uint32_t spriteW = 0, spriteH = 0;
    void* target = nullptr;
    uint64_t targetPitch = 0;
    bool targetIsDevice = false;
};
