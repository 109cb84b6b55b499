// msplat_device.hip -- context management + C-ABI entry points of libmsplat.so that touch the GPU.
// Host-only scene code (Ply / GaussianCloud / matrices) lives in msplat_host.cpp.
//
// Drop-in boundary: include/msplat.h (replaces SplatRenderer::Init/Sort/Render,
// /root/reference/src/splatrenderer.cpp:50-343).  There is NO CPU fallback: without a HIP device
// msplat_create fails with MSPLAT_ERR_NO_DEVICE.
#include "msplat_kernels.hip.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/msplat.h"
#include "../../include/msplat_debug.h"

using namespace msplat;

namespace {

thread_local std::string g_last_error;

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

// The uploaded cloud (read-only after upload).  Held by shared_ptr so that several contexts -- one per
// frame in flight, each with its own stream and per-frame buffers -- can render the same cloud
// (msplat_attach_cloud); a re-upload allocates a new store when the old one is still shared.
struct CloudStore {
    int device = 0;
    Buf pos4, recs;
    // spatial storage order (r4): slot j holds the splat uploaded as order_host[j]; one bounding box per kBoxSplats slots
    bool reordered = false;
    std::vector<uint32_t> order_host;
    Buf boxes;
    uint32_t nboxes = 0;
    ~CloudStore()
    {
        (void)hipSetDevice(device);
        if (pos4.p) (void)hipFree(pos4.p);
        if (recs.p) (void)hipFree(recs.p);
        if (boxes.p) (void)hipFree(boxes.p);
    }
};

// clouds of at least this many splats are stored in Morton order (msplat_config.spatial_order = AUTO): below it a pass-0
// chunk is a large part of the cloud and there is nothing to skip
constexpr uint64_t kSpatialMinSplats = 1ull << 18;

}  // namespace

// msplat_config.async_submit (r4): msplat_sort and device-output msplat_render of the context return at once and their launches
// are issued by a worker thread that belongs to the context.  One frame is ~14 launches = ~55 us of host time; with several frames
// in flight issued round-robin from ONE thread the k-th context starts k x 55 us after the first, and a short block of frames
// (the driver times 20) pays that stagger at its start and again at its end.  With a worker per context the caller's thread only
// copies 38 floats per call.  Everything else on the context first waits for the worker to have ISSUED what is queued (never
// for the GPU); a failed queued call is reported by the next msplat_synchronize.
struct AsyncWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv, cv_idle;
    std::deque<std::function<int()>> q;
    bool busy = false, quit = false;
    int err_code = 0;
    std::string err_msg;
    // a queued call found that an EARLIER device-output frame overflowed the pair buffer (its own work was done): kept until
    // the next msplat_synchronize / msplat_stream_wait hands it to the caller (it used to be dropped)
    bool warn = false;
    std::string warn_msg;

    void run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || !q.empty(); });
            if (q.empty()) return;              // quit, nothing left
            std::function<int()> t = std::move(q.front());
            q.pop_front();
            busy = true;
            lk.unlock();
            t();
            lk.lock();
            busy = false;
            if (q.empty()) cv_idle.notify_all();
        }
    }
    void post(std::function<int()> t)
    {
        std::lock_guard<std::mutex> lk(mu);
        q.push_back(std::move(t));
        cv.notify_one();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [&] { return q.empty() && !busy; });
    }
};

constexpr uint32_t kFusedMaxChunks = 1u << 20;     // scan-free passes up to this many chunk rows: with the two-level group tables (r3) a
                                               // prefix is <= nchunks / 128 + 34 rows, so every table qualifies (r2, one level: 8192 rows,
                                               // beyond that the radix_scan kernels)

struct msplat_ctx {
    msplat_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_partition = MSPLAT_CU_ALL;      // what the context's own stream really got (msplat_config.cu_partition)
    std::string err;

    // cloud
    uint64_t N = 0;
    bool full_sh = false;
    bool has_cloud = false;
    bool has_sort = false;
    std::shared_ptr<CloudStore> store;   // owns pos4 / recs (possibly shared with other contexts)
    Buf pos4;       // view: float4[N]  (x, y, z, bound)  -- the reference's posVec (splatrenderer.cpp:106-111)
    Buf recs;       // view: padded AoS, 16 (full SH) or 8 float4 per splat, reference float offsets preserved
    hipEvent_t join_ev = nullptr;        // msplat_stream_wait
    // sort state
    Buf keyA, keyB, valA, valB;   // uint32[N]; final sorted result in keyA/valA
    Buf hist;       // uint32[256 * hist_stride]
    uint32_t hist_stride = 0;
    // scan-free path (msplat_sort.hip.h, radix_upsweep): per-group histogram sums, one row of 256 per 32 chunks
    Buf gsumS[2];   // sort passes alternate between the two
    Buf gsumB1, gsumB2;     // binning: column pass / row pass
    uint32_t gsumS_rows = 0, gsumB1_rows = 0, gsumB2_rows = 0;
    uint32_t gsupS = 0, gsupB1 = 0, gsupB2 = 0;     // supergroup rows at the head of each table (the group rows follow)
    // wide-digit 3-pass sort (r3, msplat_sort.hip.h ws_*): histogram rows of up to 2048 digits per chunk, one group
    // table per pass, one visibility bit per splat, the visible set's minimum key per frame parity (counters[10..11])
    Buf wsHist, wsGsum[3], vmask;
    Buf live_list, live_cnt;    // live bounding boxes of the latest Sort that ran box_cull_kernel (spatially ordered clouds)
    int spatial_mode = 0;       // msplat_config.spatial_order: 0 auto, 1 always, 2 never
    bool last_sort_listed = false;   // the latest Sort's pass 0 walked the listed boxes only
    FrameParams last_sort_fp{};
    uint32_t ws_items = 8, ws_gshift = 4, ws_gsum_words = 0;
    uint32_t ws_threads = 512;  // 256 (4 waves, 40 / 56 KB of LDS) for contexts that share the GPU with other frames
    bool wide_sort = true;      // MSPLAT_SORT=lsd8 (or no lane-ordered LDS atomics): the four 8-bit passes
    bool wide_sort_cfg = true;  // what the context asked for; wide_sort = what the uploaded cloud gets (alloc_cloud_buffers)
    uint32_t sort_parity = 0;
    bool tables_dirty = false;  // a launch failed: clear every self-cleaning table before the next frame
    // per-bin pair counts taken by the row pass's upsweep (r3): list offsets + work order (no kernel of their own)
    Buf bincnt;
    // (XCD-contiguous chunk ranges: on for the sort's downsweeps -- 6 M splats 196 -> 185 us, no change at 1 M -- and the row
    //  pass's downsweep -- 6 M / 4096^2 binning 347 -> 327 us: a column's chunks write adjacent runs of every row --, off for the
    //  column pass's downsweep, where they were measured slower: 453 -> 509 us.  Fixed since r4.)
    Buf heavy, heavy_flag;      // column pass: chunks with far more pairs than the others are split over several workgroups
    uint32_t render_parity = 0;
    bool scan_free = true;  // MSPLAT_SCAN_KERNELS=1 forces the 3-kernel (upsweep, scan, downsweep) passes
    Buf totals;     // uint32[256]  digit totals of the current radix pass (rows in binning pass 2)
    Buf totals1;    // uint32[256]  column totals of binning pass 1
    Buf counters;   // uint32[16]: 0=V, 1=D, 2=overflow, 4=drawn, 6..7=pairs16 (u64), 8=probe
    Buf queue;      // uint32[kQueueShards * kQueueStride]: the compositors' sharded work queue heads
    // render state
    uint64_t rank_cap = 0;      // draw-order ranks rec2d / rect / hist1 ... are sized for: N, or 2 N + 64 once msplat_render_stereo ran
    Buf rec2d;      // float4[3*N]
    Buf rect;       // uint32[N]
    Buf zq;         // uint32[N] quantised window depth per rank (only with msplat_set_depth_test)
    int depth_bits = 0;
    int rop = 0;            // render-target emulation for the draw-order compositor (msplat_set_target_emulation)
    // point-cloud mode: pos4 = positions, recs = float4 colours, sprite = float4 mip chain
    bool point_mode = false;
    Buf sprite;
    SpriteParams sprite_params{};
    Buf tile_start; // uint32[65537]
    Buf tile_order; // uint32[65536] bins by descending list length, then uint32[65536] = 0, 1, 2, ... (the order used when persistent
                    // waves pull the items dynamically: measured, the heaviest-first order buys nothing there)
    Buf hist1;      // uint32[256 * hist1_stride]
    uint32_t hist1_stride = 0;
    Buf pairsA, pairsB;   // uint32[pair_cap]
    uint64_t pair_cap = 0;
    Buf hist2;      // uint32[256 * hist2_stride]
    uint32_t hist2_stride = 0;
    Buf fb;         // internal framebuffer for host-output renders
    Buf probe;      // uint32[8 * work items] compositor probe (msplat_set_tile_probe)
    bool probe_on = false;
    // device-output renders never synchronise: a pair-buffer overflow is left in host-mapped memory by the
    // binning kernel and picked up by the next call on the context (poll_async_overflow)
    uint32_t* h_flags = nullptr;   // host view   [0] = pairs needed by an overflowed device-output render
    uint32_t* d_flags = nullptr;   // device view of the same words
    // band
    // owned bin rows: blocks of band_block rows starting at band_first, band_first + band_stride, ..., at most band_count
    // rows (0 = as many as the image has); banded == false: the whole image (msplat_set_band_layout)
    bool banded = false;
    int band_first = 0, band_block = 1, band_stride = 1, band_count = 0;
    bool band_cull = false;
    // last frame
    FrameParams last_fp{};
    bool has_render = false;
    // timing
    // per-stage hipEvent sets, recorded on every `timing_stride`-th call so that the event markers
    // (a few us of pipeline bubble each) do not perturb a throughput run; averaged by msplat_get_timings
    static constexpr int kEvSets = 32;
    hipEvent_t ev[kEvSets][13]{};  // [6],[7] = exact dispatch start/stop of the compositor (hipExtLaunchKernelGGL); [8..12]: second pass of a two-pass frame
    bool ev_ok = false;
    int timing_stride = 1;
    uint64_t sort_calls = 0, render_calls = 0;
    uint32_t sort_sets = 0, render_sets = 0;      // sets recorded since the last msplat_get_timings
    int cur_render_set = -1;
    bool comp_kernel_timed = false;
    uint32_t comp_kernel_sets_mask = 0;
    uint32_t two_pass_sets_mask = 0;         // timing sets recorded by two-pass frames
    // two-pass frame with occlusion feedback (msplat_occlusion.hip.h)
    Buf occ, occ_mask, occ_fin, occ_state, occ_live, occ_boxdead, occ_unf;
    int two_pass_mode = MSPLAT_TWO_PASS_AUTO;
    float occ_frac = 0.15f;                  // share of the visible splats that goes into pass 1 (0.15-0.3 is the optimum of the blobs, 0.02 of a camera inside a scene)
    uint32_t occ_streak = 0;                 // consecutive two-pass frames submitted (their feedback describes two-pass frames)
    uint32_t occ_seq = 0, occ_change_seq = 0;   // number of the latest two-pass frame; first frame that ran with the current share
    uint32_t occ_off = 0;                    // AUTO: frames left of a single-pass period after two passes did not pay
    uint32_t occ_strikes = 0, occ_backoff = 1024; // ... decided after three looks; the pause doubles every time
    int occ_state_auto = 0;                  // AUTO: 0 one pass, 1 probing (occ_probe_left frames), 2 waiting for the probe's feedback, 3 two passes
    uint32_t occ_probe_left = 0, occ_no_shrink = 0, occ_wait_frames = 0;
    bool occ_pinned = false;                 // msplat_debug_two_pass: the share is fixed
    bool last_render_two_pass = false;
    uint64_t frames_rendered = 0, frames_two_pass = 0;

    std::unique_ptr<AsyncWorker> worker;     // msplat_config.async_submit
    bool atomic_rank = true;    // LDS atomics hand out ranks in lane order (probed at create)
    int comp_waves = 8192;      // compositor grid (persistent waves; measured best of 2k..8k); msplat_config.compositor_waves overrides
    bool comp_waves_auto = true;   // nobody chose a pool size: up to 20 k work items every item gets its own wave (r2: a wave
                                   // that pulls a second item pays an atomic + two dependent loads; 17.6 k items: -8 %)
    uint64_t device_bytes = 0;
};

namespace {

int fail(msplat_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (c) c->err = buf;
    return code;
}

// waits until the context's worker thread (if any) has issued everything queued; no-op on the worker itself
thread_local const msplat_ctx* g_on_worker_of = nullptr;
inline void drain_async(msplat_ctx* c)
{
    if (c && c->worker && g_on_worker_of != c) c->worker->drain();
}

#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), MSPLAT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                      \
    } while (0)

int buf_alloc(msplat_ctx* c, Buf& b, size_t bytes)
{
    if (b.p && b.bytes >= bytes) return MSPLAT_OK;
    if (b.p) {
        (void)hipFree(b.p);
        c->device_bytes -= b.bytes;
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0) bytes = 16;
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    c->device_bytes += bytes;
    return MSPLAT_OK;
}

void buf_free(msplat_ctx* c, Buf& b)
{
    if (b.p) {
        (void)hipFree(b.p);
        c->device_bytes -= b.bytes;
    }
    b.p = nullptr;
    b.bytes = 0;
}

inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

constexpr size_t kProbeWords = 8;                                           // per compositor work item
constexpr size_t kProbeBytes = (size_t)65536 * 8 * kProbeWords * sizeof(uint32_t);   // 256x256 bins x 4 quadrants x 2 halves

int grid_for(uint32_t nchunks)
{
    // grid-stride kernels: enough workgroups to fill 256 CUs x 8, never more than the chunk count
    const uint32_t cap = 256u * 8u;
    return (int)std::max(1u, std::min(nchunks, cap));
}

// host matrix helpers (implemented in msplat_host.cpp)
}  // namespace

extern "C" {

const char* msplat_version_string(void) { return "msplat 0.1 (gfx950, HIP)"; }

int msplat_tile_size(void) { return kBin; }

const char* msplat_last_error(const msplat_ctx* ctx)
{
    drain_async(const_cast<msplat_ctx*>(ctx));          // (a queued call may be writing the text)
    if (ctx) return ctx->err.c_str();
    return g_last_error.c_str();
}

int msplat_create(msplat_ctx** out, const msplat_config* cfg)
{
    if (!out) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: out is NULL");
    *out = nullptr;
    msplat_config c{};
    c.struct_size = sizeof(msplat_config);
    c.t_epsilon = -1.0f;
    if (cfg) {
        // the struct grows at its end: a caller built against an older header passes a shorter struct_size
        constexpr size_t kMinConfig = offsetof(msplat_config, rank_mode);      // (spatial_order, added in r4, defaults to AUTO too)
        if (cfg->struct_size < kMinConfig || cfg->struct_size > sizeof(msplat_config))
            return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: config struct_size %u not in [%zu, %zu]",
                        cfg->struct_size, kMinConfig, sizeof(msplat_config));
        memcpy(&c, cfg, cfg->struct_size);
        c.struct_size = sizeof(msplat_config);
    }
    if (c.rank_mode != MSPLAT_RANK_AUTO && c.rank_mode != MSPLAT_RANK_BALLOT)
        return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: bad rank_mode %d", c.rank_mode);
    if (c.fb_format != MSPLAT_FB_RGBA32F && c.fb_format != MSPLAT_FB_RGBA16F)
        return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: bad fb_format %d", c.fb_format);
    if (c.spatial_order < MSPLAT_SPATIAL_AUTO || c.spatial_order > MSPLAT_SPATIAL_OFF)
        return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: bad spatial_order %d", c.spatial_order);
    if (c.two_pass < MSPLAT_TWO_PASS_AUTO || c.two_pass > MSPLAT_TWO_PASS_OFF)
        return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: bad two_pass %d", c.two_pass);
    if (c.cu_partition < MSPLAT_CU_ALL || c.cu_partition > MSPLAT_CU_ODD)
        return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: bad cu_partition %d", c.cu_partition);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, MSPLAT_ERR_NO_DEVICE, "msplat_create: no HIP device (%s); there is no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (c.device < 0 || c.device >= ndev)
        return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_create: device %d out of range [0,%d)", c.device, ndev);
    msplat_ctx* ctx = new msplat_ctx;
    ctx->cfg = c;
    ctx->device = c.device;
    if (ctx->cfg.t_epsilon < 0.0f) ctx->cfg.t_epsilon = 1.0f / 16384.0f;
    ctx->two_pass_mode = c.two_pass;
    e = hipSetDevice(ctx->device);
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, MSPLAT_ERR_NO_DEVICE, "hipSetDevice(%d): %s", c.device, hipGetErrorString(e));
    }
    if (c.stream) {
        ctx->stream = (hipStream_t)c.stream;
    } else {
        // cu_partition (r6): the stream runs on the even or the odd CU positions of every XCD.  Mask bit 8 c + x is CU c of XCD x, and a
        // set bit enables CU c in EVERY XCD (tools/ubench_cumask.hip): a partition by XCD does not exist.  Two of four frames in
        // flight per half: each stream's launches compete with one other stream's instead of three (sort and binning stages -27 %,
        // projection and compositor slower, the frame 3-5 % faster: tools/gpu_r6_q.sh).  Without the extension: every CU, as before.
        e = hipErrorNotSupported;
        if (c.cu_partition == MSPLAT_CU_EVEN || c.cu_partition == MSPLAT_CU_ODD) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount == 256) {
                uint32_t mask[8] = {0};
                for (int i = 0; i < 256; ++i)
                    if ((i / 8) % 2 == c.cu_partition - MSPLAT_CU_EVEN) mask[i / 32] |= 1u << (i % 32);
                e = hipExtStreamCreateWithCUMask(&ctx->stream, 8, mask);
                ctx->cu_partition = e == hipSuccess ? c.cu_partition : MSPLAT_CU_ALL;
            }
            if (e != hipSuccess) (void)hipGetLastError();
        }
        if (e != hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return fail(nullptr, MSPLAT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    if (c.enable_timing > 0) {
        bool ok = true;
        for (auto& set : ctx->ev)
            for (auto& ev : set) ok = ok && (hipEventCreate(&ev) == hipSuccess);
        ctx->ev_ok = ok;
        ctx->timing_stride = c.enable_timing;
    }
    int rc = buf_alloc(ctx, ctx->totals, 256 * sizeof(uint32_t));
    if (rc == MSPLAT_OK) rc = buf_alloc(ctx, ctx->counters, 16 * sizeof(uint32_t));
    if (rc == MSPLAT_OK) rc = buf_alloc(ctx, ctx->totals1, 256 * sizeof(uint32_t));
    if (rc == MSPLAT_OK) rc = buf_alloc(ctx, ctx->queue, kQueueShards * kQueueStride * sizeof(uint32_t));
    if (rc == MSPLAT_OK) rc = buf_alloc(ctx, ctx->tile_start, (65536 + 1024 + 16) * sizeof(uint32_t));
    if (rc == MSPLAT_OK) rc = buf_alloc(ctx, ctx->bincnt, (65536 + 1024 + 16) * sizeof(uint32_t));
    if (rc == MSPLAT_OK && hipMemsetAsync(ctx->bincnt.p, 0, ctx->bincnt.bytes, ctx->stream) != hipSuccess) rc = MSPLAT_ERR_HIP;
    if (rc == MSPLAT_OK) rc = buf_alloc(ctx, ctx->tile_order, 2 * 65536 * sizeof(uint32_t));

    if (c.compositor_waves > 0) { ctx->comp_waves = std::max(64, (int)c.compositor_waves); ctx->comp_waves_auto = false; }
    if (rc == MSPLAT_OK) {
        if (hipHostMalloc((void**)&ctx->h_flags, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&ctx->d_flags, ctx->h_flags, 0) != hipSuccess)
            rc = fail(ctx, MSPLAT_ERR_HIP, "msplat_create: cannot allocate the host-mapped status words");
        else
            std::memset(ctx->h_flags, 0, 64);
    }
    if (rc == MSPLAT_OK && hipMemsetAsync(ctx->counters.p, 0, 16 * sizeof(uint32_t), ctx->stream) != hipSuccess)
        rc = MSPLAT_ERR_HIP;
    if (rc == MSPLAT_OK && hipMemsetAsync((uint32_t*)ctx->counters.p + 10, 0xFF, 2 * sizeof(uint32_t), ctx->stream) != hipSuccess)
        rc = MSPLAT_ERR_HIP;      // minimum key of the visible set, one word per frame parity
    if (rc == MSPLAT_OK) {
        // feature probe: stable ranks straight from LDS atomics need lane-ordered ds_add_rtn
        uint32_t* bad = (uint32_t*)ctx->counters.p + 8;
        hipLaunchKernelGGL(iota_kernel, dim3(65536 / kThreads), dim3(kThreads), 0, ctx->stream, (uint32_t*)ctx->tile_order.p + 65536);
        hipLaunchKernelGGL(lds_atomic_order_probe, dim3(64), dim3(kThreads), 0, ctx->stream, bad);
        uint32_t hbad = 1;
        if (hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(ctx, MSPLAT_ERR_HIP, "LDS atomic order probe failed to run");
        ctx->atomic_rank = (hbad == 0) && c.rank_mode != MSPLAT_RANK_BALLOT;
        ctx->scan_free = getenv("MSPLAT_SCAN_KERNELS") == nullptr;
        // frames in flight: kernels that co-schedule well (msplat.h); AUTO and any stale padding value = one frame at a time
        ctx->wide_sort = true;
        ctx->ws_threads = c.frame_mode == MSPLAT_FRAMES_IN_FLIGHT ? (uint32_t)kWsThreadsSmall : (uint32_t)kWsThreads;
        if (const char* sk = getenv("MSPLAT_SORT")) ctx->wide_sort = std::string(sk) != "lsd8";
        if (!ctx->atomic_rank) ctx->wide_sort = false;       // the wide kernels rank with lane-ordered LDS atomics only
        ctx->spatial_mode = c.spatial_order;
        if (ctx->wide_sort) {
            // ws_downsweep needs 72 / 104 KB of dynamic LDS with 512 threads (40 / 56 KB with 256).  The attribute belongs to
            // the function ON A DEVICE (a kernel object per device): it is requested once per device, with that device current
            // (a process-wide flag left devices 1.. of a msplat_group without it)
            static std::mutex lds_mu;
            static int lds_state[64];          // per device ordinal: 0 = not asked yet, 1 = granted, -1 = refused
            const int dslot = ctx->device & 63;
            std::lock_guard<std::mutex> lk(lds_mu);
            if (lds_state[dslot] == 0) lds_state[dslot] = [] {
                bool ok = true;
                auto want = [&](const void* f, size_t bytes) { ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess; };
                want(reinterpret_cast<const void*>(&ws_downsweep<true, 8, kWsThreads>), ws_downsweep_lds(8, kWsThreads));
                want(reinterpret_cast<const void*>(&ws_downsweep<false, 8, kWsThreads>), ws_downsweep_lds(8, kWsThreads));
                want(reinterpret_cast<const void*>(&ws_downsweep<true, 16, kWsThreads>), ws_downsweep_lds(16, kWsThreads));
                want(reinterpret_cast<const void*>(&ws_downsweep<false, 16, kWsThreads>), ws_downsweep_lds(16, kWsThreads));
                want(reinterpret_cast<const void*>(&ws_downsweep<true, 8, kWsThreadsSmall>), ws_downsweep_lds(8, kWsThreadsSmall));
                want(reinterpret_cast<const void*>(&ws_downsweep<false, 8, kWsThreadsSmall>), ws_downsweep_lds(8, kWsThreadsSmall));
                want(reinterpret_cast<const void*>(&ws_downsweep<true, 16, kWsThreadsSmall>), ws_downsweep_lds(16, kWsThreadsSmall));
                want(reinterpret_cast<const void*>(&ws_downsweep<false, 16, kWsThreadsSmall>), ws_downsweep_lds(16, kWsThreadsSmall));
                return ok ? 1 : -1;
            }();
            if (lds_state[dslot] < 0) { (void)hipGetLastError(); ctx->wide_sort = false; }
        }
        ctx->wide_sort_cfg = ctx->wide_sort;
    }
    if (rc != MSPLAT_OK) {
        std::string msg = ctx->err;
        msplat_destroy(ctx);
        g_last_error = msg;
        return rc;
    }
    if (c.async_submit != 0) {
        try {
            ctx->worker.reset(new AsyncWorker);
            AsyncWorker* w = ctx->worker.get();
            const msplat_ctx* self = ctx;
            const int dev = ctx->device;
            w->th = std::thread([w, self, dev] {
                (void)hipSetDevice(dev);
                g_on_worker_of = self;
                w->run();
            });
        } catch (const std::exception&) {
            ctx->worker.reset();          // no thread: the context simply issues its own launches
        }
    }
    *out = ctx;
    return MSPLAT_OK;
}

void msplat_destroy(msplat_ctx* ctx)
{
    if (!ctx) return;
    if (ctx->worker) {
        ctx->worker->drain();
        { std::lock_guard<std::mutex> lk(ctx->worker->mu); ctx->worker->quit = true; }
        ctx->worker->cv.notify_all();
        if (ctx->worker->th.joinable()) ctx->worker->th.join();
        ctx->worker.reset();
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pos4 = Buf{};
    ctx->recs = Buf{};
    ctx->store.reset();
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    Buf* all[] = {&ctx->keyA, &ctx->keyB, &ctx->valA, &ctx->valB, &ctx->hist, &ctx->gsumS[0], &ctx->gsumS[1], &ctx->gsumB1, &ctx->gsumB2,
                  &ctx->totals, &ctx->counters, &ctx->rec2d, &ctx->rect, &ctx->totals1, &ctx->tile_start, &ctx->tile_order,
                  &ctx->hist1, &ctx->pairsA, &ctx->pairsB, &ctx->hist2, &ctx->fb, &ctx->probe, &ctx->zq, &ctx->sprite, &ctx->queue, &ctx->occ, &ctx->occ_mask, &ctx->occ_fin,
                  &ctx->occ_state, &ctx->occ_live, &ctx->occ_boxdead, &ctx->occ_unf,
                  &ctx->wsHist, &ctx->wsGsum[0], &ctx->wsGsum[1], &ctx->wsGsum[2], &ctx->vmask, &ctx->bincnt, &ctx->heavy, &ctx->heavy_flag,
                  &ctx->live_list, &ctx->live_cnt};
    for (Buf* b : all) buf_free(ctx, *b);
    if (ctx->ev_ok)
        for (auto& set : ctx->ev)
            for (auto& ev : set) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int ensure_pair_capacity(msplat_ctx* ctx, uint64_t cap);

// A device-output render (msplat_render with out_is_device = 1) cannot report a pair-buffer overflow when it is
// issued: the count is only known on the device.  bin1_downsweep leaves the number of pairs the frame needed in
// host-mapped memory; every later msplat_sort / msplat_render / msplat_synchronize on the context looks at that
// word (a plain host load, no synchronisation), grows the buffer when the capacity is automatic, and reports
// MSPLAT_ERR_PAIR_OVERFLOW once: the frame that overflowed is missing splats in its last bin columns.
// Returns MSPLAT_OK when nothing is pending; `msg` receives the text the caller reports after doing its own work.
static int poll_async_overflow(msplat_ctx* ctx, std::string& msg)
{
    if (!ctx->h_flags) return MSPLAT_OK;
    uint32_t need = __atomic_load_n(ctx->h_flags, __ATOMIC_RELAXED);
    if (need == 0) return MSPLAT_OK;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(ctx, MSPLAT_ERR_HIP, "hipStreamSynchronize failed while handling a pair-buffer overflow");
    need = std::max(need, __atomic_load_n(ctx->h_flags, __ATOMIC_RELAXED));     // a later frame may have needed more
    __atomic_store_n(ctx->h_flags, 0u, __ATOMIC_RELAXED);
    const uint64_t old_cap = ctx->pair_cap;
    char buf[384];
    if (ctx->cfg.pair_capacity == 0 && (uint64_t)need + (need >> 2) + 1024 <= 0x7FFFFFFFull &&
        ensure_pair_capacity(ctx, (uint64_t)need + (need >> 2) + 1024) == MSPLAT_OK) {
        snprintf(buf, sizeof(buf), "an earlier device-output render overflowed the pair buffer (needed %u pairs, capacity "
                 "was %llu): that frame lacks splats in its last bin columns; capacity grown to %llu", need,
                 (unsigned long long)old_cap, (unsigned long long)ctx->pair_cap);
    } else {
        snprintf(buf, sizeof(buf), "an earlier device-output render overflowed the pair buffer (needed %u pairs, capacity "
                 "%llu%s): that frame lacks splats in its last bin columns", need, (unsigned long long)old_cap,
                 ctx->cfg.pair_capacity ? ", fixed by msplat_config.pair_capacity" : "");
    }
    msg = buf;
    return MSPLAT_ERR_PAIR_OVERFLOW;
}

// the overflow warning a queued call left behind (async_submit), once
static bool take_async_warning(msplat_ctx* ctx, std::string& msg)
{
    if (!ctx->worker) return false;
    std::lock_guard<std::mutex> lk(ctx->worker->mu);
    if (!ctx->worker->warn) return false;
    ctx->worker->warn = false;
    msg = ctx->worker->warn_msg;
    return true;
}

int msplat_synchronize(msplat_ctx* ctx)
{
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (ctx->worker && g_on_worker_of != ctx) {
        ctx->worker->drain();
        int code = 0;
        std::string msg;
        { std::lock_guard<std::mutex> lk(ctx->worker->mu); code = ctx->worker->err_code; msg = ctx->worker->err_msg; ctx->worker->err_code = 0; }
        if (code) {
            (void)hipSetDevice(ctx->device);
            (void)hipStreamSynchronize(ctx->stream);
            return fail(ctx, code, "a queued msplat_sort / msplat_render failed: %s", msg.c_str());
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::string msg;
    if (poll_async_overflow(ctx, msg) != MSPLAT_OK) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW, "%s", msg.c_str());
    if (take_async_warning(ctx, msg)) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", msg.c_str());
    return MSPLAT_OK;
}

// per-work-item compositor counters for the following renders (performance analysis / bench statistics;
// costs a few clock reads per batch, so it is off by default)
int msplat_set_tile_probe(msplat_ctx* ctx, int enable)
{
    drain_async(ctx);
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (enable && !ctx->probe.p) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        int rc = buf_alloc(ctx, ctx->probe, kProbeBytes);
        if (rc) return rc;
    }
    ctx->probe_on = enable != 0;
    return MSPLAT_OK;
}

static int prepare_cloud_buffers(msplat_ctx* ctx, uint64_t n, bool full_sh, const std::shared_ptr<CloudStore>& share);

// Emulated depth buffer for subsequent renders (app.cpp:163 enables GL_DEPTH_TEST, which is live
// on targets with a depth attachment).  bits = 0: colour-only target, the default.
int msplat_set_depth_test(msplat_ctx* ctx, int depth_bits)
{
    drain_async(ctx);
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (depth_bits != 0 && depth_bits != 24 && depth_bits != 32)
        return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_set_depth_test: depth_bits must be 0, 24 or 32 (got %d)", depth_bits);
    if (depth_bits != 0) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        int rc = buf_alloc(ctx, ctx->zq, std::max<uint64_t>(ctx->N, 1) * 4);
        if (rc) return rc;
    }
    ctx->depth_bits = depth_bits;
    return MSPLAT_OK;
}

// What the render target does to the running colour, for diffing against the GL app's own output (SURVEY 8a-12,
// src/app.cpp:1012-1020): MSPLAT_ROP_NONE = float accumulation, rounded once at the end (default, the colour-only fp32 FBO);
// MSPLAT_ROP_RGBA8 = the default back buffer (clamp to [0,1] + 8-bit unorm after every blend, GL 4.6 17.3.6);
// MSPLAT_ROP_RGBA16F = the --fp16 target (fp16 rounding after every blend).  Renders then walk every bin list in draw
// order without early termination (the draw-order compositor of msplat_set_depth_test, with or without a depth buffer).
int msplat_set_target_emulation(msplat_ctx* ctx, int rop)
{
    drain_async(ctx);
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (rop != MSPLAT_ROP_NONE && rop != MSPLAT_ROP_RGBA8 && rop != MSPLAT_ROP_RGBA16F)
        return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_set_target_emulation: rop must be MSPLAT_ROP_NONE, _RGBA8 or _RGBA16F (got %d)", rop);
    if (rop != MSPLAT_ROP_NONE && ctx->point_mode)
        return fail(ctx, MSPLAT_ERR_UNSUPPORTED, "msplat_set_target_emulation: the point-cloud sprite compositor has no render-target "
                    "emulation (the context holds a point cloud)");
    ctx->rop = rop;
    return MSPLAT_OK;
}

// Frames in flight: `ctx` renders `owner`'s cloud (no copy).  Each context keeps its own stream and
// per-frame buffers, so consecutive frames issued round-robin over several contexts overlap on the GPU
// (one frame's latency-bound sort/binning launches fill the gaps of another frame's compositor).
int msplat_attach_cloud(msplat_ctx* ctx, msplat_ctx* owner)
{
    drain_async(ctx);
    drain_async(owner);
    if (!ctx || !owner) return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_attach_cloud: NULL argument");
    if (ctx == owner) return MSPLAT_OK;
    if (!owner->has_cloud || !owner->store) return fail(ctx, MSPLAT_ERR_NO_CLOUD, "msplat_attach_cloud: owner has no cloud");
    if (owner->device != ctx->device)
        return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_attach_cloud: contexts are on different devices (%d, %d)",
                    ctx->device, owner->device);
    ctx->point_mode = owner->point_mode;
    int rc = prepare_cloud_buffers(ctx, owner->N, owner->full_sh, owner->store);
    if (rc) return rc;
    ctx->has_cloud = true;       // uploads are synchronous: the store is complete
    return MSPLAT_OK;
}

void* msplat_get_stream(msplat_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int msplat_get_fb_format(const msplat_ctx* ctx) { return ctx ? ctx->cfg.fb_format : -1; }


// Makes `stream` (a hipStream_t, NULL = the legacy default stream) wait for everything enqueued so far on
// the context's stream, without blocking the host.
int msplat_stream_wait(msplat_ctx* ctx, void* stream)
{
    drain_async(ctx);
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((hipStream_t)stream == ctx->stream) {
        std::string msg;
        if (take_async_warning(ctx, msg)) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", msg.c_str());
        return MSPLAT_OK;
    }
    if (!ctx->join_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->join_ev, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->join_ev, 0));
    std::string msg;
    if (take_async_warning(ctx, msg)) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", msg.c_str());
    return MSPLAT_OK;
}

// The context's stream waits for `event` (a hipEvent_t recorded by the caller, e.g. after the consumer of a
// framebuffer that the next frame on this context will overwrite).
int msplat_wait_event(msplat_ctx* ctx, void* event)
{
    drain_async(ctx);
    if (!ctx || !event) return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_wait_event: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return MSPLAT_OK;
}

#include "msplat_upload.hip.inc"

#include "msplat_bands.hip.inc"

static int make_frame_params(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                             const float viewport[4], const float nearFar[2], FrameParams& fp)
{
    if (!cameraMat || !projMat || !viewport || !nearFar)
        return fail(ctx, MSPLAT_ERR_INVALID_ARG, "NULL matrix/viewport argument");
    std::memset(&fp, 0, sizeof(fp));
    msplat_mat4_inverse(cameraMat, fp.view);               // splatrenderer.cpp:161 / :327
    msplat_mat4_mul(projMat, fp.view, fp.mvp);             // splatrenderer.cpp:175
    std::memcpy(fp.proj, projMat, 16 * sizeof(float));
    fp.eye[0] = cameraMat[12]; fp.eye[1] = cameraMat[13]; fp.eye[2] = cameraMat[14];   // :328
    fp.W = viewport[2];
    fp.H = viewport[3];
    fp.zn = nearFar[0];
    fp.zf = nearFar[1];
    fp.X0 = viewport[0] * (0.00001f * nearFar[0]);         // splat_vert.glsl:160
    fp.Y0 = viewport[1];
    fp.width = (int)viewport[2];
    fp.height = (int)viewport[3];
    if (fp.width < 1 || fp.height < 1 || fp.width > 8192 || fp.height > 8192)
        return fail(ctx, MSPLAT_ERR_UNSUPPORTED, "viewport %dx%d outside [1,8192]^2 (256x256 bins of 32 px)",
                    fp.width, fp.height);
    fp.tiles_x = (fp.width + kBin - 1) / kBin;
    const int rows_full = (fp.height + kBin - 1) / kBin;
    fp.banded = ctx->banded ? 1 : 0;
    fp.band_first = ctx->band_first;
    fp.band_block = ctx->band_block;
    fp.band_stride = ctx->band_stride;
    fp.band_inv_stride = 1.0f / (float)std::max(1, ctx->band_stride);
    fp.tiles_y = rows_full;
    if (ctx->banded) {           // owned rows below rows_full, at most band_count of them
        int owned = 0;
        if (rows_full > ctx->band_first) {
            const int d = rows_full - ctx->band_first, k = d / ctx->band_stride, j = d - k * ctx->band_stride;
            owned = k * ctx->band_block + std::min(j, ctx->band_block);
        }
        if (ctx->band_count > 0) owned = std::min(owned, ctx->band_count);
        fp.tiles_y = owned;
    }
    fp.full_sh = ctx->full_sh ? 1 : 0;
    fp.srgb = ctx->cfg.srgb ? 1 : 0;
    fp.t_eps = ctx->cfg.t_epsilon;
    fp.band_cull = (ctx->band_cull && ctx->banded && !ctx->point_mode) ? 1 : 0;   // points carry no footprint bound
    fp.depth_bits = ctx->depth_bits;
    fp.rop = ctx->rop;
    fp.views = 1;
    fp.rows_view = rows_full;
    fp.view_scale2 = 0.0f;
    for (int c = 0; c < 3; ++c)
        fp.view_scale2 = std::max(fp.view_scale2, fp.view[c * 4] * fp.view[c * 4] + fp.view[c * 4 + 1] * fp.view[c * 4 + 1] +
                                                      fp.view[c * 4 + 2] * fp.view[c * 4 + 2]);
    return MSPLAT_OK;
}

// exclusive scan of a chunk-major histogram table along the chunks.  The element count is only known on the device,
// so the variant is picked from the cloud size: up to 2 M splats (a few thousand chunk rows at most) the 16-workgroup
// version, beyond that one workgroup per digit.
static void launch_scan(hipStream_t s, bool small, uint32_t* hist, uint32_t hist_stride, const uint32_t* d_n,
                        uint32_t n_static, uint32_t n_cap, uint32_t chunk, uint32_t* totals)
{
    if (small)
        hipLaunchKernelGGL(radix_scan_small, dim3(kScanSmallBlocks), dim3(kThreads), 0, s, hist, d_n, n_static, n_cap, chunk,
                           totals);
    else
        hipLaunchKernelGGL(radix_scan, dim3(256), dim3(kThreads), 0, s, hist, hist_stride, d_n, n_static, n_cap, chunk,
                           totals);
}

// The scan-free passes keep their tables clean for the NEXT frame from inside the frame (every group table is zeroed by a
// later kernel of the same frame, the bin counts by their consumer, the minimum-key words by the other parity's pass 0), so
// a frame whose launches did not all go out leaves them in an unknown state: after any launch error the context is marked
// dirty and the next msplat_sort restores every table with memsets before it issues its kernels.
static int clear_frame_tables(msplat_ctx* ctx)
{
    hipStream_t s = ctx->stream;
    Buf* zero[] = {&ctx->gsumS[0], &ctx->gsumS[1], &ctx->gsumB1, &ctx->gsumB2, &ctx->wsGsum[0], &ctx->wsGsum[1], &ctx->wsGsum[2],
                   &ctx->bincnt};
    for (Buf* b : zero)
        if (b->p) HIP_TRY(ctx, hipMemsetAsync(b->p, 0, b->bytes, s));
    HIP_TRY(ctx, hipMemsetAsync((uint32_t*)ctx->counters.p + 10, 0xFF, 2 * sizeof(uint32_t), s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->queue.p, 0, ctx->queue.bytes, s));
    if (ctx->heavy.p) HIP_TRY(ctx, hipMemsetAsync(ctx->heavy.p, 0, ctx->heavy.bytes, s));
    ctx->tables_dirty = false;
    return MSPLAT_OK;
}

// Chunk-level cull (msplat_common.hip.h, box_live): for a spatially ordered cloud of which an EARLIER frame saw less than 70 %
// (host-mapped V, read without synchronising; 0 = no frame yet) Sort starts with box_cull_kernel and pass 0 walks the listed
// live boxes only.  Either form gives the same visible set, keys and order; the choice only matters for speed: the extra launch
// costs ~4 us, which a view of the whole cloud (BASELINE configs[1]: V = 0.99 N) would pay for nothing.
static LiveBoxes list_live_boxes(msplat_ctx* ctx, const FrameParams& fp, uint32_t last_V)
{
    LiveBoxes lb{nullptr, nullptr, 0u, (uint32_t)ctx->N};
    ctx->last_sort_fp = fp;
    ctx->last_sort_listed = false;
    const CloudStore* st = ctx->store.get();
    if (!st || !st->reordered || !st->boxes.p || ctx->point_mode) return lb;
    if (last_V == 0u || (uint64_t)last_V * 10u >= ctx->N * 7u) return lb;
    const uint32_t wgs = div_up(st->nboxes, kBoxGroup);
    hipLaunchKernelGGL(box_cull_kernel, dim3(wgs), dim3(kBoxGroup), 0, ctx->stream, (const CullBox*)st->boxes.p, st->nboxes, fp,
                       (uint32_t*)ctx->live_list.p, (uint32_t*)ctx->live_cnt.p);
    lb.list = (const uint32_t*)ctx->live_list.p;
    lb.cnt = (const uint32_t*)ctx->live_cnt.p;
    lb.wgs = wgs;
    ctx->last_sort_listed = true;
    return lb;
}

struct FrameArgs {            // the caller's four arrays, copied: a queued call outlives them
    float cam[16], proj[16], vp[4], nf[2];
    bool load(const float* c, const float* p, const float* v, const float* n)
    {
        if (!c || !p || !v || !n) return false;
        std::memcpy(cam, c, sizeof(cam)); std::memcpy(proj, p, sizeof(proj)); std::memcpy(vp, v, sizeof(vp)); std::memcpy(nf, n, sizeof(nf));
        return true;
    }
};

// result of a queued call: the first real failure is kept for msplat_synchronize; MSPLAT_ERR_PAIR_OVERFLOW_EARLIER (a past frame
// was composited from truncated lists; the buffer has grown since) is kept as a warning for the next msplat_synchronize /
// msplat_stream_wait, which return it once
static int note_async_result(msplat_ctx* ctx, int rc)
{
    if (rc == MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) {
        std::lock_guard<std::mutex> lk(ctx->worker->mu);
        ctx->worker->warn = true;
        ctx->worker->warn_msg = ctx->err;
    } else if (rc != MSPLAT_OK) {
        std::lock_guard<std::mutex> lk(ctx->worker->mu);
        if (ctx->worker->err_code == 0) { ctx->worker->err_code = rc; ctx->worker->err_msg = ctx->err; }
    }
    return rc;
}

static int sort_impl(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16], const float viewport[4], const float nearFar[2]);
static int render_impl(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16], const float viewport[4],
                       const float nearFar[2], void* rgba, uint64_t pitch_bytes, int out_is_device);

int msplat_sort(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                const float viewport[4], const float nearFar[2])
{
    if (ctx && ctx->worker && g_on_worker_of != ctx) {
        FrameArgs a;
        if (!a.load(cameraMat, projMat, viewport, nearFar)) {
            ctx->worker->drain();           // (ctx->err is the worker's while it runs)
            return fail(ctx, MSPLAT_ERR_INVALID_ARG, "NULL matrix/viewport argument");
        }
        ctx->worker->post([ctx, a] { return note_async_result(ctx, sort_impl(ctx, a.cam, a.proj, a.vp, a.nf)); });
        return MSPLAT_OK;
    }
    return sort_impl(ctx, cameraMat, projMat, viewport, nearFar);
}

static int sort_impl(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                     const float viewport[4], const float nearFar[2])
{
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (!ctx->has_cloud) return fail(ctx, MSPLAT_ERR_NO_CLOUD, "msplat_sort: no cloud uploaded");
    FrameParams fp;
    int rc = make_frame_params(ctx, cameraMat, projMat, viewport, nearFar, fp);
    if (rc) return rc;
    std::string pending_msg;
    const int pending = poll_async_overflow(ctx, pending_msg);     // the sort itself is still performed
    if (pending == MSPLAT_ERR_HIP) return pending;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t N = (uint32_t)ctx->N;
    uint32_t* counters = (uint32_t*)ctx->counters.p;
    uint32_t* d_V = counters + 0;
    uint32_t* hist = (uint32_t*)ctx->hist.p;
    uint32_t* totals = (uint32_t*)ctx->totals.p;
    const float4* pos = (const float4*)ctx->pos4.p;
    uint32_t *kA = (uint32_t*)ctx->keyA.p, *kB = (uint32_t*)ctx->keyB.p;
    uint32_t *vA = (uint32_t*)ctx->valA.p, *vB = (uint32_t*)ctx->valB.p;
    // chunk size by cloud size (msplat_common.hip.h, kSortItems): 2048 keys up to 2 M splats, 4096 beyond
    const bool large = ctx->N > (2u << 20);
    const uint32_t chunk = (uint32_t)kThreads * (large ? kSortItemsLarge : kSortItems);
    const int grid = grid_for(div_up(N, chunk));

    if (ctx->tables_dirty) {
        rc = clear_frame_tables(ctx);
        if (rc) return rc;
    }
    const bool timed = ctx->ev_ok && (ctx->sort_calls++ % ctx->timing_stride) == 0;
    const int tset = (int)(ctx->sort_sets % msplat_ctx::kEvSets);
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][0], s));
    if (ctx->wide_sort) {
        // three passes of 10 + (8..11) + (8..11) key bits (msplat_sort.hip.h, ws_*): 6 launches.
        // pass 0: positions -> raw keys + visibility bits in keyB / vmask -> (keyA, valA); pass 1: A -> B; pass 2: B -> A
        uint32_t* mk_cur = counters + 10 + (ctx->sort_parity & 1u);
        uint32_t* mk_next = counters + 10 + ((ctx->sort_parity ^ 1u) & 1u);
        ctx->sort_parity ^= 1u;
        uint32_t* whist = (uint32_t*)ctx->wsHist.p;
        unsigned long long* vm = (unsigned long long*)ctx->vmask.p;
        const int gsh = (int)ctx->ws_gshift;
        const uint32_t gw = ctx->ws_gsum_words;
        auto gt = [&](int pass) { return (uint32_t*)ctx->wsGsum[(pass + 3) % 3].p; };
        // Passes 1 and 2 see only the V splats that survived the cull.  With 8192-key chunks and V << N (a band-culled rank of a
        // multi-GPU frame keeps 17 %, a camera inside a scene 40 %) they would run on ~100 workgroups: they take 4096-key chunks
        // when an EARLIER frame's V (host-mapped word, read without synchronising) was below 2 M.  Any choice is correct at any V.
        const uint32_t last_V = ctx->h_flags ? __atomic_load_n(ctx->h_flags + 1, __ATOMIC_RELAXED) : 0u;
        const uint32_t items0 = ctx->ws_items;
        const uint32_t items12 = (items0 == 16u && last_V != 0u && (uint64_t)last_V + (last_V >> 2) <= (2u << 20)) ? 8u : items0;
        const LiveBoxes lb = list_live_boxes(ctx, fp, last_V);       // (launches box_cull_kernel when the view is a partial one)
        // pass 0 over listed boxes: fewer chunks than the cloud has, so 4096-key chunks keep the CUs covered (as for passes 1, 2)
        uint32_t items = lb.list != nullptr ? 8u : items0;
        int wgrid = grid_for(div_up(N, ctx->ws_threads * items));
        const uint32_t* dV = d_V;
#define MSPLAT_WS_T(KERNEL, CULLF, LDS, T, ...)                                                                          \
    do {                                                                                                                \
        if (items == 16u) hipLaunchKernelGGL((KERNEL<CULLF, 16, T>), dim3(wgrid), dim3(T), LDS(16, T), s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<CULLF, 8, T>), dim3(wgrid), dim3(T), LDS(8, T), s, __VA_ARGS__);                       \
    } while (0)
#define MSPLAT_WS(KERNEL, CULLF, LDS, ...)                                                                              \
    do {                                                                                                                \
        if (ctx->ws_threads == (uint32_t)kWsThreadsSmall) MSPLAT_WS_T(KERNEL, CULLF, LDS, kWsThreadsSmall, __VA_ARGS__); \
        else MSPLAT_WS_T(KERNEL, CULLF, LDS, kWsThreads, __VA_ARGS__);                                                  \
    } while (0)
#define MSPLAT_NO_LDS(I, T) 0
        // The upsweeps have no order to keep.  One frame at a time and 4096-key chunks (up to 2 M splats): twice the threads per
        // chunk, half the keys per thread -- 241 workgroups of 16 waves instead of 8 at 1 M: sort 57.4 -> 55.7 us.  Not at 6 M
        // (158 -> 166 us) and not for frames in flight (-0.5 %): `tools/archive/gpu_round3_q2.sh`.
#define MSPLAT_WS_UP(CULLF, ...)                                                                                        \
    do {                                                                                                                \
        if (items == 8u && ctx->ws_threads == (uint32_t)kWsThreads)                                                     \
            hipLaunchKernelGGL((ws_upsweep<CULLF, 4, 2 * kWsThreads>), dim3(wgrid), dim3(2 * kWsThreads), 0, s, __VA_ARGS__); \
        else MSPLAT_WS(ws_upsweep, CULLF, MSPLAT_NO_LDS, __VA_ARGS__);                                                    \
    } while (0)
        if (fp.band_cull)
            MSPLAT_WS_UP(2, (const uint32_t*)nullptr, pos, kB, vm, (const uint32_t*)nullptr, N, N, 0, mk_cur,
                      mk_next, whist, gt(0), gsh, gt(-1), gw, fp, lb);
        else
            MSPLAT_WS_UP(1, (const uint32_t*)nullptr, pos, kB, vm, (const uint32_t*)nullptr, N, N, 0, mk_cur,
                      mk_next, whist, gt(0), gsh, gt(-1), gw, fp, lb);
        MSPLAT_WS(ws_downsweep, true, ws_downsweep_lds, (const uint32_t*)kB, (const uint32_t*)nullptr, (const unsigned long long*)vm,
                  (const uint32_t*)nullptr, N, N, 0, (const uint32_t*)mk_cur, (const uint32_t*)whist, (const uint32_t*)gt(0), gsh, kA, vA,
                  d_V, lb);
        items = items12;
        wgrid = grid_for(div_up(N, ctx->ws_threads * items));
        MSPLAT_WS_UP(0, (const uint32_t*)kA, (const float4*)nullptr, (uint32_t*)nullptr,
                  (unsigned long long*)nullptr, dV, 0u, N, 1, mk_cur, mk_next, whist, gt(1), gsh, gt(0), gw, fp);
        MSPLAT_WS(ws_downsweep, false, ws_downsweep_lds, (const uint32_t*)kA, (const uint32_t*)vA, (const unsigned long long*)nullptr, dV,
                  0u, N, 1, (const uint32_t*)mk_cur, (const uint32_t*)whist, (const uint32_t*)gt(1), gsh, kB, vB, (uint32_t*)nullptr);
        MSPLAT_WS_UP(0, (const uint32_t*)kB, (const float4*)nullptr, (uint32_t*)nullptr,
                  (unsigned long long*)nullptr, dV, 0u, N, 2, mk_cur, mk_next, whist, gt(2), gsh, gt(1), gw, fp);
        MSPLAT_WS(ws_downsweep, false, ws_downsweep_lds, (const uint32_t*)kB, (const uint32_t*)vB, (const unsigned long long*)nullptr, dV,
                  0u, N, 2, (const uint32_t*)mk_cur, (const uint32_t*)whist, (const uint32_t*)gt(2), gsh, kA, vA, (uint32_t*)nullptr);
#undef MSPLAT_NO_LDS
#undef MSPLAT_WS
#undef MSPLAT_WS_T
#undef MSPLAT_WS_UP
        if (timed) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][1], s));
            ctx->sort_sets++;
        }
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            // e.g. the dynamic-LDS request was not honoured on this device: later frames take the four 8-bit passes, which
            // need no opt-in (this frame is lost; the tables are restored before the next one)
            ctx->tables_dirty = true;
            ctx->wide_sort = ctx->wide_sort_cfg = false;
            return fail(ctx, MSPLAT_ERR_HIP, "msplat_sort: a kernel launch of the three-pass sort failed (%s); the context falls back "
                        "to the 8-bit passes from the next frame on", hipGetErrorString(le));
        }
        ctx->has_sort = true;
        if (pending) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", pending_msg.c_str());
        return MSPLAT_OK;
    }
    // scan-free passes (2 launches each) while the chunk table is small, else upsweep + scan + downsweep
    const bool fused = ctx->scan_free && div_up(N, chunk) <= kFusedMaxChunks;
    auto gacc = [&](int pass) { return fused ? (uint32_t*)ctx->gsumS[pass & 1].p : nullptr; };
    auto gzero = [&](int pass) { return (uint32_t*)ctx->gsumS[(pass + 1) & 1].p; };      // always: keeps both tables clean
#define MSPLAT_UPSWEEP_B(MODE, BAND, ...)                                                                               \
    do {                                                                                                                \
        if (large) hipLaunchKernelGGL((radix_upsweep<MODE, kSortItemsLarge, BAND>), dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((radix_upsweep<MODE, kSortItems, BAND>), dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__);            \
    } while (0)
#define MSPLAT_DOWNSWEEP_B(MODE, BAND, ...)                                                                                      \
    do {                                                                                                                         \
        if (large) {                                                                                                             \
            if (ctx->atomic_rank) hipLaunchKernelGGL((radix_downsweep<MODE, true, true, kSortItemsLarge, BAND>), dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__); \
            else hipLaunchKernelGGL((radix_downsweep<MODE, true, false, kSortItemsLarge, BAND>), dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__);                 \
        } else {                                                                                                                 \
            if (ctx->atomic_rank) hipLaunchKernelGGL((radix_downsweep<MODE, true, true, kSortItems, BAND>), dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__);      \
            else hipLaunchKernelGGL((radix_downsweep<MODE, true, false, kSortItems, BAND>), dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__);                      \
        }                                                                                                                        \
    } while (0)
#define MSPLAT_UPSWEEP(MODE, ...) MSPLAT_UPSWEEP_B(MODE, true, __VA_ARGS__)
#define MSPLAT_DOWNSWEEP(MODE, ...) MSPLAT_DOWNSWEEP_B(MODE, true, __VA_ARGS__)
    // pass 0: cull + key fused into the first radix pass (presort_compute.glsl + byte 0 of the sort)
    const LiveBoxes lb = list_live_boxes(ctx, fp, ctx->h_flags ? __atomic_load_n(ctx->h_flags + 1, __ATOMIC_RELAXED) : 0u);
    // (r6: frames without the band-restricted cull -- every single-GPU frame -- run instantiations that leave its code out)
#define MSPLAT_PASS0(BAND)                                                                                                              \
    do {                                                                                                                                \
        MSPLAT_UPSWEEP_B(MODE_CULL, BAND, (const uint32_t*)nullptr, pos, (const uint32_t*)nullptr, N, N, 0, hist, ctx->hist_stride, gacc(0), \
                         gzero(0), ctx->gsumS_rows, fp, (const uint32_t*)nullptr, (uint32_t*)nullptr, ctx->gsupS, lb);                 \
        if (!fused) launch_scan(s, ctx->N <= (2u << 20), hist, ctx->hist_stride, nullptr, N, N, chunk, totals);                       \
        MSPLAT_DOWNSWEEP_B(MODE_CULL, BAND, (const uint32_t*)nullptr, (const uint32_t*)nullptr, pos, (const uint32_t*)nullptr, N, N, 0, \
                           (const uint32_t*)hist, ctx->hist_stride, (const uint32_t*)totals, kB, vB, d_V, (const uint32_t*)nullptr,    \
                           (const uint32_t*)gacc(0), (uint32_t*)nullptr, fp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, \
                           (uint32_t*)nullptr, 0, 0, ctx->gsupS, lb);                                                                  \
    } while (0)
    if (fp.band_cull) MSPLAT_PASS0(true); else MSPLAT_PASS0(false);
#undef MSPLAT_PASS0
    // passes 1..3 on the V survivors (V stays on the device; splatrenderer.cpp:195-204's readback is gone)
    for (int pass = 1; pass < 4; ++pass) {
        uint32_t* kin = (pass & 1) ? kB : kA;
        uint32_t* vin = (pass & 1) ? vB : vA;
        uint32_t* kout = (pass & 1) ? kA : kB;
        uint32_t* vout = (pass & 1) ? vA : vB;
        MSPLAT_UPSWEEP(MODE_KEYS, (const uint32_t*)kin, (const float4*)nullptr, (const uint32_t*)d_V, 0u, N, pass * 8, hist,
                       ctx->hist_stride, gacc(pass), gzero(pass), ctx->gsumS_rows, fp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                       ctx->gsupS);
        if (!fused) launch_scan(s, ctx->N <= (2u << 20), hist, ctx->hist_stride, d_V, 0u, N, chunk, totals);
        MSPLAT_DOWNSWEEP(MODE_KEYS, (const uint32_t*)kin, (const uint32_t*)vin, (const float4*)nullptr, (const uint32_t*)d_V, 0u, N,
                         pass * 8, (const uint32_t*)hist, ctx->hist_stride, (const uint32_t*)totals, kout, vout, (uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (const uint32_t*)gacc(pass), (uint32_t*)nullptr, fp, (uint32_t*)nullptr,
                         (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0, 0, ctx->gsupS);
    }
#undef MSPLAT_UPSWEEP
#undef MSPLAT_DOWNSWEEP
#undef MSPLAT_UPSWEEP_B
#undef MSPLAT_DOWNSWEEP_B
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][1], s));
        ctx->sort_sets++;
    }
    if (hipGetLastError() != hipSuccess) {
        ctx->tables_dirty = true;
        return fail(ctx, MSPLAT_ERR_HIP, "msplat_sort: a kernel launch failed");
    }
    ctx->has_sort = true;
    if (pending) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", pending_msg.c_str());
    return MSPLAT_OK;
}

// Should this Render run as two passes (msplat_occlusion.hip.h), and with what share of the visible splats in pass 1?  Host
// side only: reads what EARLIER frames of this context left in host-mapped memory (never waits); any answer gives the same
// pixels.  Feedback of a two-pass frame: [4] pairs of pass 1, [8] pairs of pass 2, [5] unfinished bins.
constexpr uint64_t kTwoPassMinSplats = 1u << 18;
constexpr uint32_t kTwoPassMinVisibleSerial = 1500000u;
constexpr int kProjGridTwoPass = 4096;       // one-wave workgroups of the grid-stride forms of project_kernel (9 fit a CU)
constexpr float kOccFracMin = 1.0f / 256.0f, kOccFracMax = 0.75f;
static bool occlusion_plan(msplat_ctx* ctx, const FrameParams& fp, bool stereo, float& frac)
{
    ctx->frames_rendered++;
    auto no = [&]() { ctx->occ_streak = 0; return false; };
    if (ctx->two_pass_mode == MSPLAT_TWO_PASS_OFF) return no();
    if (stereo || ctx->point_mode || ctx->depth_bits != 0 || ctx->rop != 0 || ctx->probe_on) return no();
    if (!ctx->scan_free) return no();            // (the chunk offset of pass 1's column pass exists in the scan-free form only)
    if (fp.tiles_x * fp.tiles_y <= 0 || (fp.tiles_x + 1) * (fp.tiles_y + 1) > kOccSatMax) return no();      // (the table of unfinished bins)
    const bool forced = ctx->two_pass_mode == MSPLAT_TWO_PASS_ON;
    if (!forced && (ctx->N < kTwoPassMinSplats || ctx->frames_rendered <= 8)) return no();
    // one frame at a time and few visible splats (a rank of a row-sharded frame, a 1 M-splat cloud): the nine extra launches
    // are latency the skipped work does not buy back (rank 3 of 8 of config 4, V = 1 M: 0.33 -> 0.38 ms; with frames in flight
    // the same rank gains 12 %)
    if (!forced && ctx->cfg.frame_mode != MSPLAT_FRAMES_IN_FLIGHT && ctx->h_flags &&
        __atomic_load_n(ctx->h_flags + 1, __ATOMIC_RELAXED) < kTwoPassMinVisibleSerial)
        return no();
    const size_t nbins = (size_t)fp.tiles_x * fp.tiles_y;

    // Feedback of an earlier two-pass frame (host-mapped, never waited for): [4] pairs of pass 1, [8] pairs of pass 2, [5] bins
    // pass 1 left unfinished, [9] / [10] the number of the frame the words of pass 1 / pass 2 describe.  Only a frame that ran
    // with the CURRENT share is evidence -- the host may be a hundred frames ahead of the GPU.
    bool fresh = false;
    float ratio = 0.0f, ufrac = 0.0f;
    if (ctx->h_flags) {
        const uint32_t s1 = __atomic_load_n(ctx->h_flags + 9, __ATOMIC_RELAXED), s2 = __atomic_load_n(ctx->h_flags + 10, __ATOMIC_RELAXED);
        const uint32_t d1 = __atomic_load_n(ctx->h_flags + 4, __ATOMIC_RELAXED), d2 = __atomic_load_n(ctx->h_flags + 8, __ATOMIC_RELAXED);
        if (s1 == s2 && s1 != 0u && (int32_t)(s1 - ctx->occ_change_seq) >= 0 && d1 != 0u) {
            fresh = true;
            ratio = (float)d2 / (float)d1;
            ufrac = (float)__atomic_load_n(ctx->h_flags + 5, __ATOMIC_RELAXED) / (float)nbins;
        }
    }
    auto set_share = [&](float next) {
        ctx->occ_frac = std::min(kOccFracMax, std::max(kOccFracMin, next));
        ctx->occ_change_seq = ctx->occ_seq + 1u;           // the first frame that runs with it
    };
    // the share: pass 2 should stay well below pass 1 (a flat optimum: tools/occlusion_potential.py)
    auto steer = [&]() {
        // (the step from "pass 2 has nothing to do" to "pass 2 does most of the work" can be one notch: after growing, the share
        //  is not shrunk again for 32 looks)
        if (ratio > 0.6f) {
            set_share(ctx->occ_frac * 1.25f);
            ctx->occ_no_shrink = 32u;
        } else if (ctx->occ_no_shrink != 0u) {
            --ctx->occ_no_shrink;
        } else if (ratio < 0.02f && ctx->occ_frac > kOccFracMin) {
            set_share(ctx->occ_frac * 0.75f);          // nothing left for pass 2: too much in pass 1
        } else if (ratio < 0.15f && ctx->occ_frac > kOccFracMin) {
            set_share(ctx->occ_frac * 0.85f);
        }
    };
    if (forced) {
        if (fresh && !ctx->occ_pinned) steer();
    } else {
        // AUTO.  A bin pass 1 does not finish costs a second look; where a third of the bins never saturate whatever the share
        // (a cloud seen from outside: its rim) and the cloud is small, the nine extra launches cost what the skipped splats save
        // (BASELINE config 2: 5830 -> 5770 frames/s at best, serial 0.27 -> 0.33 ms).  So two passes are PROBED: four frames,
        // then one pass again until their feedback is in (the probe must not cost a hundred slow frames because the host runs
        // ahead); more than 30 % unfinished is a strike (the next probe takes a larger share), three strikes a pause that doubles
        // every time; otherwise two passes stay on, steered, until three looks in a row say otherwise.
        enum { OFF = 0, PROBE = 1, WAIT = 2, ON = 3 };
        if (ctx->occ_state_auto == OFF) {
            if (ctx->occ_off != 0u) { --ctx->occ_off; return no(); }
            ctx->occ_state_auto = PROBE;
            ctx->occ_probe_left = 4u;
            ctx->occ_strikes = 0u;
            set_share(0.15f);
        }
        if (ctx->occ_state_auto == WAIT) {
            if (!fresh) {
                // feedback that never becomes fresh (the probe frames binned nothing: the camera looked away from the cloud):
                // back to one pass with the usual pause instead of waiting for ever
                if (++ctx->occ_wait_frames > 256u) {
                    ctx->occ_wait_frames = 0u;
                    ctx->occ_state_auto = OFF;
                    ctx->occ_off = ctx->occ_backoff;
                }
                return no();
            }
            ctx->occ_wait_frames = 0u;
            if (ufrac > 0.3f) {
                if (++ctx->occ_strikes >= 3u) {
                    ctx->occ_state_auto = OFF;
                    ctx->occ_off = ctx->occ_backoff;
                    ctx->occ_backoff = std::min(ctx->occ_backoff * 2u, 16384u);
                    return no();
                }
                set_share(ctx->occ_frac * 1.5f);
                ctx->occ_state_auto = PROBE;
                ctx->occ_probe_left = 4u;
            } else {
                ctx->occ_state_auto = ON;
                ctx->occ_strikes = 0u;
                ctx->occ_backoff = 1024u;
                steer();
            }
        } else if (ctx->occ_state_auto == ON && fresh && !ctx->occ_pinned) {
            if (ufrac > 0.3f) {
                if (++ctx->occ_strikes >= 3u) {
                    ctx->occ_state_auto = OFF;
                    ctx->occ_off = ctx->occ_backoff;
                    ctx->occ_backoff = std::min(ctx->occ_backoff * 2u, 16384u);
                    return no();
                }
                set_share(ctx->occ_frac * 1.5f);
            } else {
                ctx->occ_strikes = 0u;
                steer();
            }
        }
        if (ctx->occ_state_auto == PROBE) {
            if (ctx->occ_probe_left == 0u) {
                ctx->occ_state_auto = WAIT;
                ctx->occ_wait_frames = 0u;
                return no();
            }
            --ctx->occ_probe_left;
        }
    }
    if (buf_alloc(ctx, ctx->occ, 64) || buf_alloc(ctx, ctx->occ_mask, (size_t)kOccSatMax * 2 + 64) ||
        buf_alloc(ctx, ctx->occ_fin, nbins * 16 + 64) || buf_alloc(ctx, ctx->occ_state, (size_t)fp.width * fp.height * 16 + 64) ||
        buf_alloc(ctx, ctx->occ_live, ((size_t)ctx->N + 64) * 4) || buf_alloc(ctx, ctx->occ_boxdead, 2048 * 4 + 64) ||
        buf_alloc(ctx, ctx->occ_unf, nbins * 4 + 64)) {
        ctx->err.clear();                        // (no memory for the extra buffers: the frame runs in one pass)
        return no();
    }
    ctx->occ_seq++;                              // this frame's number (never 0)
    if (ctx->occ_seq == 0u) ctx->occ_seq = 1u;
    ctx->occ_streak++;
    frac = ctx->occ_frac;
    return true;
}

// One Render's decision (occlusion_plan), taken ONCE per Render: the retries of a host-output frame whose pair buffer overflowed
// re-issue the chain with the same plan (every retry used to advance the controller).
struct RenderPlan {
    bool decided = false, two_pass = false;
    float frac = 0.0f;
};

// ---- one Render = a chain of launches on the context's stream.  r5: one function per stage and one per kind of chain, instead of
// r4's single 260-line launch_render with a lambda inside.  What the stages share: -------------------------------
struct RenderChain {
    msplat_ctx* ctx;
    const FrameParams& fp;
    hipStream_t s;
    bool stereo;                 // two views in one chain: ranks [0, V) and [V1, V1 + V), bin rows stacked
    uint32_t N;                  // rank-indexed launches cover this many ranks (the cloud; 2 N + 64 with two views)
    uint32_t *d_Vsort, *d_D, *d_overflow, *d_queue;
    uint32_t* d_Vframe;          // ranks the binning walks (written by project_kernel for two views)
    uint32_t* occ;               // two-pass frames: cut / counters (msplat_occlusion.hip.h)
    int ntiles;
    uint32_t cap;
    bool timed;
    int tset, pgrid;
    void *d_out, *d_out1;
    size_t pitch;
    bool async_overflow_flag;
    // the compositor's schedule, decided where the bins are ordered (issue_binning) and used by issue_compositor
    bool ordered = true;
    uint32_t comp_items = 0, comp_pool = 0;
};

// vertex + geometry stage.  mode: PROJ_PLAIN / PROJ_TWO_VIEWS (chosen from the chain), PROJ_PASS1 or PROJ_LISTED (two-pass frames)
static void issue_projection(RenderChain& rc, int mode, float occ_frac)
{
    msplat_ctx* ctx = rc.ctx;
    const FrameParams& fp = rc.fp;
    hipStream_t s = rc.s;
    uint32_t *d_Vsort = rc.d_Vsort, *occ = rc.occ;
    const int pgrid = rc.pgrid;
    if (ctx->point_mode) {
        hipLaunchKernelGGL(point_project_kernel, dim3(pgrid), dim3(kProjThreads), 0, s, (const uint32_t*)ctx->valA.p, rc.d_Vframe,
                           (const float4*)ctx->pos4.p, (const float4*)ctx->recs.p, fp, ctx->sprite_params,
                           (float4*)ctx->rec2d.p, (uint32_t*)ctx->rect.p, ctx->depth_bits ? (uint32_t*)ctx->zq.p : nullptr);
        return;
    }
    const ProjParams pp = proj_params(fp);
    uint32_t* zq = (mode != PROJ_LISTED && ctx->depth_bits) ? (uint32_t*)ctx->zq.p : nullptr;
#define MSPLAT_PROJECT(SH, MODE, GRID, DV, EX, V1)                                                                          \
    hipLaunchKernelGGL((project_kernel<SH, MODE>), dim3(GRID), dim3(kProjThreads), 0, s, (const uint32_t*)ctx->valA.p, DV,  \
                       (const float4*)ctx->recs.p, pp, (float4*)ctx->rec2d.p, (uint32_t*)ctx->rect.p, zq, EX, V1)
    if (mode == PROJ_PASS1) {
        // (project_kernel's first pass computes the cut and leaves it in occ[0])
        const ProjExtra ex{nullptr, occ, nullptr, occ_frac};
        const int grid = std::min(pgrid, kProjGridTwoPass);
        if (ctx->full_sh) MSPLAT_PROJECT(true, PROJ_PASS1, grid, (const uint32_t*)d_Vsort, ex, ProjNoView1{0});
        else MSPLAT_PROJECT(false, PROJ_PASS1, grid, (const uint32_t*)d_Vsort, ex, ProjNoView1{0});
    } else if (mode == PROJ_LISTED) {
        // the listed ranks behind the cut (occ[1] of them)
        const ProjExtra ex{ctx->d_flags ? ctx->d_flags + 7 : (uint32_t*)nullptr, nullptr, (const uint32_t*)ctx->occ_live.p, 0.0f};
        const int grid = std::min(pgrid, kProjGridTwoPass);
        if (ctx->full_sh) MSPLAT_PROJECT(true, PROJ_LISTED, grid, (const uint32_t*)(occ + 1), ex, ProjNoView1{0});
        else MSPLAT_PROJECT(false, PROJ_LISTED, grid, (const uint32_t*)(occ + 1), ex, ProjNoView1{0});
    } else if (rc.stereo) {
        const ProjExtra ex{rc.d_Vframe, nullptr, nullptr, 0.0f};
        const ProjView1 v1 = proj_view1(fp);
        if (ctx->full_sh) MSPLAT_PROJECT(true, PROJ_TWO_VIEWS, pgrid, (const uint32_t*)d_Vsort, ex, v1);
        else MSPLAT_PROJECT(false, PROJ_TWO_VIEWS, pgrid, (const uint32_t*)d_Vsort, ex, v1);
    } else {
        const ProjExtra ex{nullptr, nullptr, nullptr, 0.0f};
        if (ctx->full_sh) MSPLAT_PROJECT(true, PROJ_PLAIN, pgrid, (const uint32_t*)d_Vsort, ex, ProjNoView1{0});
        else MSPLAT_PROJECT(false, PROJ_PLAIN, pgrid, (const uint32_t*)d_Vsort, ex, ProjNoView1{0});
    }
#undef MSPLAT_PROJECT
}

// bin lists over the current rectangles: column pass (bin1_*), row pass (radix_*<MODE_PAIR>), list offsets + work order.
// occ_pass: 0 one pass, 1 / 2 the passes of a two-pass frame; keep_overflow: second chain of a two-pass frame
static void issue_binning(RenderChain& rc, int keep_overflow, int occ_pass)
{
    msplat_ctx* ctx = rc.ctx;
    const FrameParams& fp = rc.fp;
    hipStream_t s = rc.s;
    const uint32_t N = rc.N, cap = rc.cap;
    const bool stereo = rc.stereo, async_overflow_flag = rc.async_overflow_flag;
    const int ntiles = rc.ntiles;
    uint32_t *d_Vsort = rc.d_Vsort, *d_D = rc.d_D, *d_overflow = rc.d_overflow, *d_queue = rc.d_queue, *occ = rc.occ, *d_Vframe = rc.d_Vframe;
    // second chain of a two-pass frame: the binning walks occ[4] ranks (V, or 0 when pass 1 left no bin unfinished)
    uint32_t* d_V = occ_pass == 2 ? occ + 4 : d_Vframe;
    const uint32_t* d_first = occ_pass == 1 ? occ : nullptr;      // pass 1 bins the ranks from the cut on

    // pass 1: stable partition by tile column, enumerated from the rank-ordered rectangles
    uint32_t* totals1 = (uint32_t*)ctx->totals1.p;
    uint32_t* totals2 = (uint32_t*)ctx->totals.p;
    const uint32_t bchunk = (uint32_t)kBinChunk;
    const int g1 = grid_for(div_up(N, bchunk));
    // heavy chunks of the column pass (bin1_upsweep): list per frame parity, helper workgroups in front of the downsweep's grid
    uint32_t* hv_cur = (uint32_t*)ctx->heavy.p + (ctx->render_parity & 1u) * (1u + kHeavyCap);
    uint32_t* hv_next = (uint32_t*)ctx->heavy.p + ((ctx->render_parity ^ 1u) & 1u) * (1u + kHeavyCap);
    ctx->render_parity ^= 1u;
    // helper workgroups for as many split chunks as an EARLIER frame asked for (host-mapped word, read without synchronising),
    // with headroom; a frame that needs more runs its extra heavy chunks unsplit and the next launch adapts
    const uint32_t last_heavy = ctx->h_flags ? __atomic_load_n(ctx->h_flags + 3, __ATOMIC_RELAXED) : 0u;
    const uint32_t heavy_slots = last_heavy != 0u ? std::min<uint32_t>(kHeavyCap, 2u * last_heavy + 8u) : 0u;
    const int nhelp = (int)(heavy_slots * (kHeavyParts - 1u));
    // scan-free variants while the chunk tables are small; the row pass's size (D) is only known on the device, so
    // its choice uses the D of an EARLIER frame that the binning kernel left in host-mapped memory (0 = none yet);
    // either variant is correct at any size, the choice only matters for speed
    // (r2, one-level group tables: the column pass's table was scanned by a kernel from 4096 rows on; with two levels every
    //  table that fits the supergroup rows is scan-free)
    const bool fused1 = ctx->scan_free && div_up(N, bchunk) <= kFusedMaxChunks;
    const uint32_t last_D = ctx->h_flags ? __atomic_load_n(ctx->h_flags + 2, __ATOMIC_RELAXED) : 0u;
    const bool fused2 = ctx->scan_free && last_D != 0u && div_up((uint64_t)last_D + (last_D >> 2), kPairChunk) <= kFusedMaxChunks &&
                        div_up(cap, kPairChunk) <= ctx->hist2_stride;                  // (the tables hold every chunk the capacity allows)
    uint32_t* gB1 = (uint32_t*)ctx->gsumB1.p;
    uint32_t* gB2 = (uint32_t*)ctx->gsumB2.p;
    // column pass: groups of 8 consecutive chunks share an XCD (xcd_grouped): the seams between the runs neighbouring chunks write
    // merge in that XCD's L2.  Measured r4 (serial binning, groups of 0 / 2 / 4 / 8 / 16 / 32): 6 M / 4096^2 328 / 320 / 316 / 315 /
    // 315 / 313 us, scene-like 6 M 201 / 194 / 194 / 195, 1 M unchanged (57); the fully XCD-contiguous mapping of r3 was slower.
    const int xcdg = 8;
    // (bin1_upsweep also clears the row pass's group table: its consumer, the previous frame's row downsweep, is long done)
#define MSPLAT_BIN1(CH)                                                                                                       \
    do {                                                                                                                      \
        hipLaunchKernelGGL(bin1_upsweep<CH>, dim3(g1), dim3(kThreads), 0, s, (const uint32_t*)ctx->rect.p, d_V,               \
                           (uint32_t*)ctx->hist1.p, ctx->hist1_stride, d_overflow, fused1 ? gB1 : nullptr, gB2,               \
                           ctx->gsumB2_rows, hv_cur, hv_next, (uint8_t*)ctx->heavy_flag.p, heavy_slots, ctx->gsupB1,          \
                           keep_overflow, d_first);                                                                           \
        if (!fused1)                                                                                                          \
            launch_scan(s, ctx->N <= (2u << 20), (uint32_t*)ctx->hist1.p, ctx->hist1_stride, d_V, 0u, N, bchunk, totals1);     \
        if (ctx->atomic_rank)                                                                                                 \
            hipLaunchKernelGGL((bin1_downsweep<true, CH>), dim3(g1 + nhelp), dim3(kThreads), 0, s,                            \
                               (const uint32_t*)ctx->rect.p, d_V, (const uint32_t*)ctx->hist1.p, ctx->hist1_stride,           \
                               (const uint32_t*)totals1, (uint32_t*)ctx->pairsA.p, cap, d_D, d_overflow, ctx->d_flags,        \
                               async_overflow_flag ? 1 : 0, fused1 ? (const uint32_t*)gB1 : nullptr, fused1 ? totals1 : nullptr, \
                               xcdg, (const uint32_t*)hv_cur, (const uint8_t*)ctx->heavy_flag.p,                               \
                               (uint32_t)nhelp, fp.tiles_x, ctx->gsupB1, (const uint32_t*)d_Vsort,                            \
                               (keep_overflow && ctx->d_flags) ? ctx->d_flags + 8 : (uint32_t*)nullptr, ctx->occ_seq, d_first); \
        else                                                                                                                  \
            hipLaunchKernelGGL((bin1_downsweep<false, CH>), dim3(g1 + nhelp), dim3(kThreads), 0, s,                           \
                               (const uint32_t*)ctx->rect.p, d_V, (const uint32_t*)ctx->hist1.p, ctx->hist1_stride,           \
                               (const uint32_t*)totals1, (uint32_t*)ctx->pairsA.p, cap, d_D, d_overflow, ctx->d_flags,        \
                               async_overflow_flag ? 1 : 0, fused1 ? (const uint32_t*)gB1 : nullptr, fused1 ? totals1 : nullptr, \
                               xcdg, (const uint32_t*)hv_cur, (const uint8_t*)ctx->heavy_flag.p,                               \
                               (uint32_t)nhelp, fp.tiles_x, ctx->gsupB1, (const uint32_t*)d_Vsort,                            \
                               (keep_overflow && ctx->d_flags) ? ctx->d_flags + 8 : (uint32_t*)nullptr, ctx->occ_seq, d_first); \
    } while (0)
    // (bin1_upsweep also clears the row pass's group table: its consumer, the previous frame's row downsweep, is long done)
    MSPLAT_BIN1(kBinChunk);
#undef MSPLAT_BIN1
    // pass 2: stable partition by tile row (one generic radix pass on the top byte); words become (tx<<24)|rank
    // The heaviest-first order of the bins only pays when every work item has its own wave (the hardware then starts the
    // waves in item order: 83 -> 97 us without it at config 2); persistent waves that pull items from the queue balance
    // themselves: they walk the bins in storage order (`tile_order` + 65536 holds 0, 1, 2, ...).
    const bool wave_comp = !ctx->point_mode && ctx->depth_bits == 0 && ctx->rop == 0;
    const uint32_t comp_items = (uint32_t)ntiles * 4u;
    // (every item on its own wave up to 20 k items, 40 k for two views in one chain: BASELINE configs[4] has 2 x 17.6 k)
    const uint32_t comp_pool = (ctx->comp_waves_auto && comp_items <= (stereo ? 40960u : 20480u)) ? comp_items : (uint32_t)ctx->comp_waves;
    const bool ordered = !(wave_comp && comp_pool < comp_items);
    // r3: the upsweep of the row pass also counts the pairs per bin, and one extra workgroup of its downsweep turns the
    // counts into the list offsets (+ the heaviest-first order when it is wanted).  (r3-r5: contexts with frames in flight
    // searched the offsets in the partitioned array with a kernel of their own, 1 % faster then; equal in r6 under the CU halves
    // -- 6290 / 6298, 2478-2514 / 2485-2494, 1363-1403 / 1384-1403 frames/s at 1 M, 6 M, 6 M / 4096^2 -- and removed: one launch less.)
    uint32_t* bincnt = (uint32_t*)ctx->bincnt.p;
    const int g2 = grid_for(div_up(cap, kPairChunk));
    hipLaunchKernelGGL(radix_upsweep<MODE_PAIR>, dim3(g2), dim3(kThreads), 0, s, (const uint32_t*)ctx->pairsA.p,
                       nullptr, d_D, 0u, cap, 24, (uint32_t*)ctx->hist2.p, ctx->hist2_stride, fused2 ? gB2 : nullptr, gB1,
                       ctx->gsumB1_rows, fp, (const uint32_t*)totals1, bincnt, ctx->gsupB2);
    if (!fused2)
        launch_scan(s, ctx->N <= (2u << 20), (uint32_t*)ctx->hist2.p, ctx->hist2_stride, d_D, 0u, cap, (uint32_t)kPairChunk, totals2);
    const int g2d = g2 + 1;
    if (ctx->atomic_rank)
        hipLaunchKernelGGL((radix_downsweep<MODE_PAIR, false, true>), dim3(g2d), dim3(kThreads), 0, s,
                           (const uint32_t*)ctx->pairsA.p, nullptr, nullptr, d_D, 0u, cap, 24,
                           (const uint32_t*)ctx->hist2.p, ctx->hist2_stride, (const uint32_t*)totals2,
                           (uint32_t*)ctx->pairsB.p, nullptr, nullptr, (const uint32_t*)totals1,
                           fused2 ? (const uint32_t*)gB2 : nullptr, fused2 ? totals2 : nullptr, fp, bincnt,
                           (uint32_t*)ctx->tile_start.p, (uint32_t*)ctx->tile_order.p, d_queue, ntiles, (ordered ? 1 : 0) | 2, ctx->gsupB2);
    else
        hipLaunchKernelGGL((radix_downsweep<MODE_PAIR, false, false>), dim3(g2d), dim3(kThreads), 0, s,
                           (const uint32_t*)ctx->pairsA.p, nullptr, nullptr, d_D, 0u, cap, 24,
                           (const uint32_t*)ctx->hist2.p, ctx->hist2_stride, (const uint32_t*)totals2,
                           (uint32_t*)ctx->pairsB.p, nullptr, nullptr, (const uint32_t*)totals1,
                           fused2 ? (const uint32_t*)gB2 : nullptr, fused2 ? totals2 : nullptr, fp, bincnt,
                           (uint32_t*)ctx->tile_start.p, (uint32_t*)ctx->tile_order.p, d_queue, ntiles, (ordered ? 1 : 0) | 2, ctx->gsupB2);
    rc.ordered = ordered;
    rc.comp_items = comp_items;
    rc.comp_pool = comp_pool;
}

// fragment stage + blend over the bin lists: the splat compositor (front to back), the draw-order compositor of the emulations,
// or the sprite compositor.  ev_c0 / ev_c1: the timing set's slots for the kernel's dispatch begin / end
static int issue_compositor(RenderChain& rc, int occ_pass, int ev_c0, int ev_c1)
{
    msplat_ctx* ctx = rc.ctx;
    const FrameParams& fp = rc.fp;
    hipStream_t s = rc.s;
    const uint32_t cap = rc.cap, comp_items = rc.comp_items, comp_pool = rc.comp_pool;
    const bool stereo = rc.stereo, timed = rc.timed, ordered = rc.ordered;
    const int ntiles = rc.ntiles, tset = rc.tset;
    uint32_t *d_queue = rc.d_queue, *occ = rc.occ;
    void *d_out = rc.d_out, *d_out1 = rc.d_out1;
    const size_t pitch = rc.pitch;
    uint32_t* fin = occ_pass ? (uint32_t*)ctx->occ_fin.p : nullptr;
    float4* state = occ_pass ? (float4*)ctx->occ_state.p : nullptr;
    const uint32_t* d_nbins = occ_pass == 2 ? occ + 2 : nullptr;      // second chain of a two-pass frame: the listed unfinished bins

    // persistent compositor: a fixed pool of waves pulls (bin, quadrant) items; never more waves than items
    const int cgrid = std::min(ntiles * 4, ctx->comp_waves);     // work items = (bin, quadrant) (the draw-order compositors)
    // (frames in flight: serialising the compositor launches of the contexts sharing a cloud with an event
    //  gate was measured r1 -- no gain over letting the hardware queues interleave them, dropped)
    if (ntiles > 0 && ctx->probe_on && ctx->probe.p && (ctx->point_mode || ctx->depth_bits != 0 || ctx->rop != 0))
        // the draw-order compositors do not write the probe: leave zeros, not an earlier frame's counters
        HIP_TRY(ctx, hipMemsetAsync(ctx->probe.p, 0, (size_t)ntiles * 8 * kProbeWords * sizeof(uint32_t), s));
    if (ntiles > 0 && ctx->point_mode) {
        // sprites in draw order (optionally against the emulated depth buffer)
        const uint32_t* zqp = ctx->depth_bits ? (const uint32_t*)ctx->zq.p : nullptr;
        if (ctx->cfg.fb_format == MSPLAT_FB_RGBA16F)
            hipLaunchKernelGGL(composite_points_kernel<true>, dim3(cgrid), dim3(kCompThreads), 0, s,
                               (const uint32_t*)ctx->tile_start.p, (const uint32_t*)ctx->pairsB.p,
                               (const float4*)ctx->rec2d.p, zqp, (const float4*)ctx->sprite.p, ctx->sprite_params, d_out,
                               pitch, fp, cap, (const uint32_t*)ctx->tile_order.p, d_queue, (uint32_t)ntiles * 4u);
        else
            hipLaunchKernelGGL(composite_points_kernel<false>, dim3(cgrid), dim3(kCompThreads), 0, s,
                               (const uint32_t*)ctx->tile_start.p, (const uint32_t*)ctx->pairsB.p,
                               (const float4*)ctx->rec2d.p, zqp, (const float4*)ctx->sprite.p, ctx->sprite_params, d_out,
                               pitch, fp, cap, (const uint32_t*)ctx->tile_order.p, d_queue, (uint32_t)ntiles * 4u);
        ctx->comp_kernel_timed = false;
    } else if (ntiles > 0 && (ctx->depth_bits != 0 || ctx->rop != 0)) {
        // emulated depth buffer: draw-order walk, no early termination
        if (ctx->cfg.fb_format == MSPLAT_FB_RGBA16F)
            hipLaunchKernelGGL(composite_depth_kernel<true>, dim3(cgrid), dim3(kCompThreads), 0, s,
                               (const uint32_t*)ctx->tile_start.p, (const uint32_t*)ctx->pairsB.p,
                               (const float4*)ctx->rec2d.p, (const uint32_t*)ctx->zq.p, d_out, pitch, fp, cap,
                               (const uint32_t*)ctx->tile_order.p, d_queue, (uint32_t)ntiles * 4u);
        else
            hipLaunchKernelGGL(composite_depth_kernel<false>, dim3(cgrid), dim3(kCompThreads), 0, s,
                               (const uint32_t*)ctx->tile_start.p, (const uint32_t*)ctx->pairsB.p,
                               (const float4*)ctx->rec2d.p, (const uint32_t*)ctx->zq.p, d_out, pitch, fp, cap,
                               (const uint32_t*)ctx->tile_order.p, d_queue, (uint32_t)ntiles * 4u);
        ctx->comp_kernel_timed = false;
    } else if (ntiles > 0) {
        // on sampled frames the dominant kernel gets exact dispatch begin/end events (the plain stream
        // markers around the stages can be processed while the previous kernel is still draining)
        hipEvent_t e0 = timed ? ctx->ev[tset][ev_c0] : nullptr, e1 = timed ? ctx->ev[tset][ev_c1] : nullptr;
        uint32_t* probe = ctx->probe_on ? (uint32_t*)ctx->probe.p : nullptr;
        if (probe) HIP_TRY(ctx, hipMemsetAsync(probe, 0, (size_t)ntiles * 8 * kProbeWords * sizeof(uint32_t), s));
        const uint32_t* ts = (const uint32_t*)ctx->tile_start.p;
        const uint32_t* pb = (const uint32_t*)ctx->pairsB.p;
        const float4* r2 = (const float4*)ctx->rec2d.p;
        const uint32_t* ord = occ_pass == 2 ? (const uint32_t*)ctx->occ_unf.p : (const uint32_t*)ctx->tile_order.p + (ordered ? 0 : 65536);
        const bool f16 = ctx->cfg.fb_format == MSPLAT_FB_RGBA16F;
        // wave issue priority by item number where the items are numbered heaviest-first (every item on its own wave); none for
        // persistent waves that walk the bins in storage order (r3, 4 frames in flight: none / by item number / by list length
        // = 5.82 / 5.76 / 5.79 k frames/s, i.e. no effect there)
        const int prio_mode = ordered ? 1 : 0;
        const int grid = (int)std::min<uint32_t>(comp_items, comp_pool);
        const CompParams cp = comp_params(fp);
        const CompExtra ex{d_out1, fin, state, d_nbins, probe};
        const int prio = occ_pass == 2 ? 0 : prio_mode;
#define MSPLAT_COMPOSITE(F16, OCC, TWO, PROBE)                                                                                     \
        hipExtLaunchKernelGGL((composite_kernel<F16, OCC, TWO, PROBE>), dim3(grid), dim3(kCompThreads), 0, s, e0, e1, 0, ts, pb, r2,  \
                              d_out, pitch, cp, cap, ord, d_queue, comp_items, prio, ex)
#define MSPLAT_COMPOSITE_F(F16)                                                                                                    \
        do {                                                                                                                       \
            if (occ_pass == 1) MSPLAT_COMPOSITE(F16, 1, false, false);                                                             \
            else if (occ_pass == 2) MSPLAT_COMPOSITE(F16, 2, false, false);                                                        \
            else if (stereo && probe) MSPLAT_COMPOSITE(F16, 0, true, true);                                                        \
            else if (stereo) MSPLAT_COMPOSITE(F16, 0, true, false);                                                                \
            else if (probe) MSPLAT_COMPOSITE(F16, 0, false, true);                                                                 \
            else MSPLAT_COMPOSITE(F16, 0, false, false);                                                                           \
        } while (0)
        // (occlusion_plan never chooses two passes for two views in one chain or while the probe is on)
        if (f16) MSPLAT_COMPOSITE_F(true); else MSPLAT_COMPOSITE_F(false);
#undef MSPLAT_COMPOSITE_F
#undef MSPLAT_COMPOSITE
        ctx->comp_kernel_timed = timed;
    } else {
        ctx->comp_kernel_timed = false;
    }
    return MSPLAT_OK;
}

// between the passes of a two-pass frame: the map of unfinished bins, dead boxes, the gate over the ranks behind the cut
static void issue_occlusion_gate(RenderChain& rc)
{
    msplat_ctx* ctx = rc.ctx;
    const FrameParams& fp = rc.fp;
    hipStream_t s = rc.s;
    const uint32_t N = rc.N;
    uint32_t *d_Vsort = rc.d_Vsort, *d_D = rc.d_D, *occ = rc.occ;
    hipLaunchKernelGGL(occ_mask_kernel, dim3(1), dim3(kOccMaskThreads), 0, s, (const uint32_t*)ctx->occ_fin.p, fp.tiles_x, fp.tiles_y,
                       (uint16_t*)ctx->occ_mask.p, occ, (const uint32_t*)d_Vsort, (uint32_t*)ctx->occ_unf.p);
    // spatially ordered cloud: whole boxes of 256 stored splats are dropped before any centre is fetched
    const uint32_t nboxes = (ctx->store && ctx->store->reordered && ctx->store->boxes.p) ? ctx->store->nboxes : 0u;
    const uint32_t boxwords = ((nboxes + (uint32_t)kThreads - 1u) / (uint32_t)kThreads) * ((uint32_t)kThreads / 32u);
    const bool use_boxes = nboxes != 0u && boxwords <= 2048u && ctx->occ_boxdead.p != nullptr;
    if (use_boxes)
        hipLaunchKernelGGL(occ_box_kernel, dim3(div_up(nboxes, kThreads)), dim3(kThreads), 0, s, (const CullBox*)ctx->store->boxes.p,
                           nboxes, fp, (const uint16_t*)ctx->occ_mask.p, (uint32_t*)ctx->occ_boxdead.p);
    hipLaunchKernelGGL(occ_gate_kernel, dim3(std::max(1u, div_up(N, kOccGateRanks))), dim3(kThreads), 0, s, (const uint32_t*)ctx->valA.p,
                       (const uint32_t*)d_Vsort, occ, (const float4*)ctx->pos4.p, (uint32_t*)ctx->rect.p, fp,
                       (const uint16_t*)ctx->occ_mask.p, (uint32_t*)ctx->occ_live.p, ctx->d_flags, (const uint32_t*)d_D, ctx->occ_seq,
                       use_boxes ? (const uint32_t*)ctx->occ_boxdead.p : (const uint32_t*)nullptr, boxwords);
}

static int launch_render(msplat_ctx* ctx, const FrameParams& fp, void* d_out, size_t pitch, bool async_overflow_flag,
                         void* d_out1 = nullptr, RenderPlan* plan = nullptr)
{
    hipStream_t s = ctx->stream;
    const bool stereo = fp.views == 2;
    uint32_t* counters = (uint32_t*)ctx->counters.p;
    const uint32_t N = stereo ? (uint32_t)(2 * ctx->N + 64) : (uint32_t)ctx->N;
    RenderChain rc{ctx, fp, s, stereo, N, counters + 0, counters + 1, counters + 2, (uint32_t*)ctx->queue.p,
                   stereo ? counters + 9 : counters + 0, (uint32_t*)ctx->occ.p, fp.tiles_x * fp.tiles_y, (uint32_t)ctx->pair_cap,
                   false, 0, (int)std::max(1u, div_up(N, kProjThreads)), d_out, d_out1, pitch, async_overflow_flag};
    // Two-pass frame with occlusion feedback (msplat_occlusion.hip.h): the nearest R1 splats first, then only what the bins
    // they did not saturate still need.  Same pixels; chosen once per Render (occlusion_plan).  The retries of a host-output
    // frame whose pair buffer overflowed reuse the plan AND count as the same Render: the sampling counter, the two-pass
    // statistics and the feedback's frame number advance once.
    RenderPlan local_plan;
    if (!plan) plan = &local_plan;
    const bool first_attempt = !plan->decided;
    if (first_attempt) {
        plan->two_pass = occlusion_plan(ctx, fp, stereo, plan->frac);
        plan->decided = true;
        ctx->render_calls++;
    }
    const bool two_pass = plan->two_pass;
    const bool timed = rc.timed = ctx->ev_ok && ((ctx->render_calls - 1) % ctx->timing_stride) == 0;
    const int tset = rc.tset = (int)(ctx->render_sets % msplat_ctx::kEvSets);
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][2], s));
    rc.occ = (uint32_t*)ctx->occ.p;              // (allocated by occlusion_plan with a context's first two-pass frame)

    // ---- the plain chain (also: both eyes in one chain, points, the emulations), or pass 1 of a two-pass frame ----
    issue_projection(rc, two_pass ? PROJ_PASS1 : PROJ_PLAIN, plan->frac);
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][3], s));
    issue_binning(rc, 0, two_pass ? 1 : 0);
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][4], s));
    int crc = issue_compositor(rc, two_pass ? 1 : 0, 6, 7);
    if (crc) return crc;
    if (two_pass) {
        // ---- pass 2: gate, the listed ranks, the bins pass 1 left unfinished ----
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][8], s));
        issue_occlusion_gate(rc);
        issue_projection(rc, PROJ_LISTED, 0.0f);
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][9], s));
        const bool timed1 = ctx->comp_kernel_timed;
        issue_binning(rc, 1, 2);
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][10], s));
        crc = issue_compositor(rc, 2, 11, 12);
        if (crc) return crc;
        ctx->comp_kernel_timed = ctx->comp_kernel_timed && timed1;
        if (first_attempt) ctx->frames_two_pass++;
    }
    ctx->last_render_two_pass = two_pass;
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[tset][5], s));
        if (ctx->comp_kernel_timed) ctx->comp_kernel_sets_mask |= 1u << tset; else ctx->comp_kernel_sets_mask &= ~(1u << tset);
        if (two_pass) ctx->two_pass_sets_mask |= 1u << tset; else ctx->two_pass_sets_mask &= ~(1u << tset);
        ctx->render_sets++;
    }
    if (hipGetLastError() != hipSuccess) {
        ctx->tables_dirty = true;
        return fail(ctx, MSPLAT_ERR_HIP, "msplat_render: a kernel launch failed");
    }
    return MSPLAT_OK;
}

int msplat_render(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                  const float viewport[4], const float nearFar[2],
                  void* rgba, uint64_t pitch_bytes, int out_is_device)
{
    if (ctx && ctx->worker && g_on_worker_of != ctx) {
        if (out_is_device && rgba) {
            FrameArgs a;
            if (!a.load(cameraMat, projMat, viewport, nearFar)) {
                ctx->worker->drain();
                return fail(ctx, MSPLAT_ERR_INVALID_ARG, "NULL matrix/viewport argument");
            }
            ctx->worker->post([ctx, a, rgba, pitch_bytes] {
                return note_async_result(ctx, render_impl(ctx, a.cam, a.proj, a.vp, a.nf, rgba, pitch_bytes, 1));
            });
            return MSPLAT_OK;
        }
        ctx->worker->drain();           // a host image is filled before the call returns: nothing to defer
    }
    return render_impl(ctx, cameraMat, projMat, viewport, nearFar, rgba, pitch_bytes, out_is_device);
}

static int render_impl(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                       const float viewport[4], const float nearFar[2],
                       void* rgba, uint64_t pitch_bytes, int out_is_device)
{
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (!ctx->has_cloud) return fail(ctx, MSPLAT_ERR_NO_CLOUD, "msplat_render: no cloud uploaded");
    if (!ctx->has_sort) return fail(ctx, MSPLAT_ERR_NO_SORT, "msplat_render: msplat_sort has not been called");
    if (!rgba) return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_render: rgba is NULL");
    FrameParams fp;
    int rc = make_frame_params(ctx, cameraMat, projMat, viewport, nearFar, fp);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bpp = ctx->cfg.fb_format == MSPLAT_FB_RGBA16F ? 8 : 16;
    const size_t tight = (size_t)fp.width * bpp;
    if (pitch_bytes == 0) pitch_bytes = tight;
    if (pitch_bytes < tight || pitch_bytes % bpp != 0)
        return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_render: pitch %llu too small / misaligned for width %d",
                    (unsigned long long)pitch_bytes, fp.width);
    ctx->last_fp = fp;
    if (ctx->tables_dirty && (rc = clear_frame_tables(ctx))) return rc;
    if (ctx->point_mode && !ctx->sprite.p && (rc = build_sprite(ctx, nullptr, 0, 0))) return rc;   // built-in sphere sprite
    std::string pending_msg;
    const int pending = poll_async_overflow(ctx, pending_msg);     // an EARLIER frame; this one is still rendered
    if (pending == MSPLAT_ERR_HIP) return pending;

    if (out_is_device) {
        rc = launch_render(ctx, fp, rgba, pitch_bytes, true);
        if (rc) return rc;
        ctx->has_render = true;
        if (pending) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", pending_msg.c_str());
        return MSPLAT_OK;
    }
    // host output: render into an internal device framebuffer, copy back, grow the pair buffer on overflow
    if ((rc = buf_alloc(ctx, ctx->fb, tight * fp.height))) return rc;
    RenderPlan plan;
    for (int attempt = 0; attempt < 6; ++attempt) {
        rc = launch_render(ctx, fp, ctx->fb.p, tight, false, nullptr, &plan);
        if (rc) return rc;
        uint32_t cnt[4];
        HIP_TRY(ctx, hipMemcpyAsync(cnt, ctx->counters.p, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (cnt[2] == 0) {
            if (ctx->banded) {
                // band mode: only the owned bin rows are written to the caller's buffer (as documented for
                // msplat_set_band), so several bands can be assembled in one host image
                for (int vy = 0; vy < fp.tiles_y;) {
                    const int y0 = band_real_row(fp, vy) * kBin;
                    int run = 1;                  // consecutive owned rows leave in one copy
                    while (vy + run < fp.tiles_y && band_real_row(fp, vy + run) == band_real_row(fp, vy) + run) ++run;
                    vy += run;
                    const int rows = std::min(kBin * run, fp.height - y0);
                    if (rows <= 0) break;
                    HIP_TRY(ctx, hipMemcpy2D((char*)rgba + (size_t)y0 * pitch_bytes, pitch_bytes,
                                             (const char*)ctx->fb.p + (size_t)y0 * tight, tight, tight, rows,
                                             hipMemcpyDeviceToHost));
                }
            } else {
                HIP_TRY(ctx, hipMemcpy2D(rgba, pitch_bytes, ctx->fb.p, tight, tight, fp.height, hipMemcpyDeviceToHost));
            }
            ctx->has_render = true;
            if (pending) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", pending_msg.c_str());
            return MSPLAT_OK;
        }
        // overflow: cnt[2] holds the required pair count
        const uint64_t need = (uint64_t)cnt[2] + (cnt[2] >> 2) + 1024;
        if (ctx->cfg.pair_capacity != 0 || need > 0x7FFFFFFFull)
            return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW, "pair buffer too small: need %u, capacity %llu", cnt[2],
                        (unsigned long long)ctx->pair_cap);
        if ((rc = ensure_pair_capacity(ctx, need))) return rc;
    }
    return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW, "pair buffer could not be grown");
}

// buffers indexed by draw-order rank, for 2 N + 64 ranks (two views in one chain)
static int ensure_stereo_ranks(msplat_ctx* ctx)
{
    const uint64_t need = 2 * std::max<uint64_t>(ctx->N, 1) + 64;
    if (ctx->rank_cap >= need) return MSPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if ((rc = buf_alloc(ctx, ctx->rec2d, need * 48))) return rc;
    if ((rc = buf_alloc(ctx, ctx->rect, need * 4))) return rc;
    if ((rc = buf_alloc(ctx, ctx->heavy_flag, (size_t)div_up(need, kBinChunk) + 64))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->heavy_flag.p, 0, ctx->heavy_flag.bytes, ctx->stream));
    ctx->hist1_stride = std::max(1u, div_up(need, kBinChunk));
    if ((rc = buf_alloc(ctx, ctx->hist1, (size_t)256 * ctx->hist1_stride * 4))) return rc;
    if ((rc = alloc_group_table(ctx, ctx->gsumB1, ctx->gsumB1_rows, ctx->hist1_stride, ctx->gsupB1))) return rc;
    ctx->rank_cap = need;
    return MSPLAT_OK;
}

static int render_stereo_impl(msplat_ctx* ctx, const float cam0[16], const float proj0[16], const float cam1[16], const float proj1[16],
                              const float viewport[4], const float nearFar[2], void* rgba0, void* rgba1, uint64_t pitch_bytes,
                              int out_is_device);

// Two views of ONE Sort (the reference's VR frame: Sort with the first eye, Render per eye -- src/app.cpp:603-607) as a single
// chain of launches: projection, binning and compositing each run once over both views' work instead of twice in a row (half
// the launches, twice the work per launch: 0.50 -> 0.44 ms at BASELINE configs[4]).  Per view the arithmetic, the bin lists and
// the pixels are exactly those of two msplat_render calls.  Falls back to those two calls for banded contexts, point clouds,
// depth-test / render-target emulation, clouds beyond 2^23 splats and viewports taller than 4096 (128 bin rows per view).
int msplat_render_stereo(msplat_ctx* ctx, const float cameraMat0[16], const float projMat0[16], const float cameraMat1[16],
                         const float projMat1[16], const float viewport[4], const float nearFar[2], void* rgba0, void* rgba1,
                         uint64_t pitch_bytes, int out_is_device)
{
    if (ctx && ctx->worker && g_on_worker_of != ctx) {
        if (out_is_device && rgba0 && rgba1) {
            FrameArgs a, b;
            if (!a.load(cameraMat0, projMat0, viewport, nearFar) || !b.load(cameraMat1, projMat1, viewport, nearFar)) {
                ctx->worker->drain();
                return fail(ctx, MSPLAT_ERR_INVALID_ARG, "NULL matrix/viewport argument");
            }
            ctx->worker->post([ctx, a, b, rgba0, rgba1, pitch_bytes] {
                return note_async_result(ctx, render_stereo_impl(ctx, a.cam, a.proj, b.cam, b.proj, a.vp, a.nf, rgba0, rgba1, pitch_bytes, 1));
            });
            return MSPLAT_OK;
        }
        ctx->worker->drain();
    }
    return render_stereo_impl(ctx, cameraMat0, projMat0, cameraMat1, projMat1, viewport, nearFar, rgba0, rgba1, pitch_bytes, out_is_device);
}

static int render_stereo_impl(msplat_ctx* ctx, const float cam0[16], const float proj0[16], const float cam1[16], const float proj1[16],
                              const float viewport[4], const float nearFar[2], void* rgba0, void* rgba1, uint64_t pitch_bytes,
                              int out_is_device)
{
    if (!ctx) return fail(nullptr, MSPLAT_ERR_INVALID_ARG, "ctx is NULL");
    if (!ctx->has_cloud) return fail(ctx, MSPLAT_ERR_NO_CLOUD, "msplat_render_stereo: no cloud uploaded");
    if (!ctx->has_sort) return fail(ctx, MSPLAT_ERR_NO_SORT, "msplat_render_stereo: msplat_sort has not been called");
    if (!rgba0 || !rgba1) return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_render_stereo: a target is NULL");
    FrameParams fp, fp1;
    int rc = make_frame_params(ctx, cam0, proj0, viewport, nearFar, fp);
    if (!rc) rc = make_frame_params(ctx, cam1, proj1, viewport, nearFar, fp1);
    if (rc) return rc;
    const bool batched = !ctx->banded && !ctx->point_mode && ctx->depth_bits == 0 && ctx->rop == 0 && ctx->N <= (1ull << 23) &&
                         2 * fp.tiles_y <= 256 && out_is_device;
    if (!batched) {      // the plain form: one render per view (host targets are filled one after the other anyway)
        rc = render_impl(ctx, cam0, proj0, viewport, nearFar, rgba0, pitch_bytes, out_is_device);
        if (rc && rc != MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) return rc;
        const int rc1 = render_impl(ctx, cam1, proj1, viewport, nearFar, rgba1, pitch_bytes, out_is_device);
        return rc1 ? rc1 : rc;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bpp = ctx->cfg.fb_format == MSPLAT_FB_RGBA16F ? 8 : 16;
    const size_t tight = (size_t)fp.width * bpp;
    if (pitch_bytes == 0) pitch_bytes = tight;
    if (pitch_bytes < tight || pitch_bytes % bpp != 0)
        return fail(ctx, MSPLAT_ERR_INVALID_ARG, "msplat_render_stereo: pitch %llu too small / misaligned for width %d",
                    (unsigned long long)pitch_bytes, fp.width);
    if ((rc = ensure_stereo_ranks(ctx))) return rc;
    fp.views = 2;
    fp.rows_view = fp.tiles_y;
    fp.tiles_y = 2 * fp.rows_view;
    std::memcpy(fp.view1, fp1.view, sizeof(fp.view1));
    std::memcpy(fp.proj1, fp1.proj, sizeof(fp.proj1));
    std::memcpy(fp.eye1, fp1.eye, sizeof(fp.eye1));
    ctx->last_fp = fp;
    if (ctx->tables_dirty && (rc = clear_frame_tables(ctx))) return rc;
    std::string pending_msg;
    const int pending = poll_async_overflow(ctx, pending_msg);
    if (pending == MSPLAT_ERR_HIP) return pending;
    rc = launch_render(ctx, fp, rgba0, pitch_bytes, true, rgba1);
    if (rc) return rc;
    ctx->has_render = true;
    if (pending) return fail(ctx, MSPLAT_ERR_PAIR_OVERFLOW_EARLIER, "%s", pending_msg.c_str());
    return MSPLAT_OK;
}

#include "msplat_getters.hip.inc"

}  // extern "C"
