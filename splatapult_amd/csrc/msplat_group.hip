// msplat_group.hip -- single-process multi-GPU rendering behind the C ABI (include/msplat.h, msplat_group_*).
//
// The reference is ONE single-threaded C++ process that calls SplatRenderer::Sort / Render once per frame
// (/root/reference/src/sdl_main.cpp:49-204, src/app.cpp:1067-1068).  A maintainer who wants the frame spread over the
// GPUs of a node keeps that shape: one msplat_group = one context per device, the cloud replicated, the screen's bin rows
// partitioned over the devices, msplat_group_sort / msplat_group_render with the arguments of
// msplat_sort / msplat_render.  The only exchange step is the row gather, and it is zero-copy where the hardware allows:
// device i's compositor stores its rows STRAIGHT into device 0's framebuffer through the peer mapping
// (hipDeviceEnablePeerAccess: every band travels over its own direct xGMI link, 7 links into device 0 concurrently, no
// ring), else its rows are staged locally and moved with one hipMemcpy2DAsync per run of consecutive rows.
//
// r5: the same gather over RCCL (the north star's "RCCL over xGMI only for the final row gather"), for both process shapes:
//   msplat_band_exchange       one process per GPU (the caller owns the ncclComm_t, e.g. made with ncclCommInitRank): rank `root`
//                              posts one ncclRecv per run of foreign bin rows straight into its framebuffer, the owners ncclSend
//                              their runs from where the compositor left them -- one ncclGroupStart/End, no pack / staging;
//   MSPLAT_EXCHANGE_RCCL       this file's one-process group with communicators from ncclCommInitAll.
// librccl is resolved with dlopen at the first use (the copy already in the process, e.g. PyTorch's, else /opt/rocm's): the
// library has no link-time dependency on it and loads on a box without RCCL.
//
// Host side: launching ~12 kernels on each of 8 devices from one thread costs more host time than the frame lasts on the
// GPUs, so every device beyond the first has a worker thread that issues its context's calls; the caller's thread drives
// device 0 and then waits for the workers to have ISSUED their work (the GPUs keep running asynchronously; device 0's
// stream is made to wait for the others' streams, so "synchronise device 0's stream" = "the frame is complete").
// Uses only the public C ABI of the single-device library plus HIP runtime calls.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msplat.h"
#include "../../include/msplat_debug.h"

namespace {

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = true, quit = false;
    int rc = MSPLAT_OK;

    void run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_task || quit; });
            if (quit) return;
            std::function<int()> t = std::move(task);
            has_task = false;
            lk.unlock();
            const int r = t();
            lk.lock();
            rc = r;
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> t)
    {
        std::lock_guard<std::mutex> lk(mu);
        task = std::move(t);
        has_task = true;
        done = false;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};


// ---- RCCL, resolved at run time --------------------------------------------------------------------------------------------
// (signatures: /opt/rocm/include/rccl/rccl.h:187-260,700-722,923-933; ncclUint8 = 1, ncclSuccess = 0)
struct Rccl {
    void* lib = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;             // why it is not available
    bool ok() const { return lib != nullptr; }
};

const Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        std::vector<std::string> names;
        if (const char* e = getenv("MSPLAT_RCCL_LIB")) names.push_back(e);
        void* h = nullptr;
        // the copy the process already uses (its communicators belong to THAT copy), then the system's
        // (MSPLAT_RCCL_LIB, when set, is the only candidate: a wrong path is reported, not papered over)
        if (names.empty())
            for (const char* n : {"librccl.so.1", "librccl.so"})
                if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const std::string& n : names)
            if (!h) h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (names.empty())
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* de = dlerror();          // (ONE call: dlerror() clears the error it returns -- ADVICE r5)
            x.why = std::string("librccl not found (") + (de ? de : "?") + "); set MSPLAT_RCCL_LIB";
            return x;
        }
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) { all = false; x.why = std::string("librccl lacks ") + n; } return p; };
        x.GroupStart = (int (*)())sym("ncclGroupStart");
        x.GroupEnd = (int (*)())sym("ncclGroupEnd");
        x.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        x.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        x.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
        x.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        x.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        if (all) x.lib = h;
        return x;
    }();
    return r;
}
constexpr int kNcclUint8 = 1;

// the runs of consecutive bin rows rank `rank` of `world` owns under a standard layout, as pixel rows [y0, y0 + nrows) of an
// image of H rows: f(y0, nrows).  (Runs end at block boundaries; contiguous bands are one run.)
template <class F>
int for_each_run(int32_t kind, int32_t block_rows, int32_t world, int32_t rank, int H, F&& f)
{
    const int T = msplat_tile_size(), rows_full = (H + T - 1) / T;
    int32_t first, count, block, stride;
    const int pr = msplat_band_plan(kind, rows_full, world, rank, block_rows, &first, &count, &block, &stride);
    if (pr) return pr;
    for (int v = 0; v < count;) {
        const int k = v / block, row = first + k * stride + (v - k * block);
        int run = std::min(block - (v - k * block), count - v);          // the rest of this block
        // (stride == block: the next block follows immediately -- contiguous bands arrive here as count rows in one block)
        const int y0 = row * T, nrows = std::min(run * T, H - y0);
        v += run;
        if (nrows <= 0) break;
        const int rc = f(y0, nrows);
        if (rc) return rc;
    }
    return MSPLAT_OK;
}

// ---- fp16 on the wire (MSPLAT_EXCHANGE_WIRE_FP16): the rows of an RGBA32F target travel as RGBA16F ---------------------------------
// One thread per pixel of a run: float4 <-> four halves (round to nearest even).  `rows` pitch rows of `width` pixels; the packed side
// is tight (width * 8 bytes per row).
__global__ __launch_bounds__(256) void pack_rows_f16(const char* __restrict__ src, size_t pitch, int width, int rows, uint2* __restrict__ dst)
{
    const size_t n = (size_t)width * rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t y = i / (size_t)width, x = i - y * (size_t)width;
        const float4 v = reinterpret_cast<const float4*>(src + y * pitch)[x];
        union { _Float16 h[4]; uint2 u; } pk;
        pk.h[0] = (_Float16)v.x; pk.h[1] = (_Float16)v.y; pk.h[2] = (_Float16)v.z; pk.h[3] = (_Float16)v.w;
        dst[i] = pk.u;
    }
}
__global__ __launch_bounds__(256) void unpack_rows_f16(const uint2* __restrict__ src, char* __restrict__ dst, size_t pitch, int width, int rows)
{
    const size_t n = (size_t)width * rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t y = i / (size_t)width, x = i - y * (size_t)width;
        union { _Float16 h[4]; uint2 u; } pk;
        pk.u = src[i];
        reinterpret_cast<float4*>(dst + y * pitch)[x] = make_float4((float)pk.h[0], (float)pk.h[1], (float)pk.h[2], (float)pk.h[3]);
    }
}

}  // namespace

struct msplat_group {
    std::vector<int> devices;
    std::vector<msplat_ctx*> ctx;
    std::vector<std::unique_ptr<Worker>> workers;      // [i - 1] drives ctx[i]
    std::vector<bool> peer_store;                      // ctx[i] stores into device 0's memory directly (the exchange in use)
    std::vector<bool> peer_ok;                         // ... and whether the peer mapping exists at all
    std::vector<void*> stage;                          // per-rank staging framebuffer when it may not
    std::vector<size_t> stage_bytes;
    int fb_format = MSPLAT_FB_RGBA32F;
    int kind = MSPLAT_BANDS_CONTIGUOUS, block_rows = 1;
    bool band_cull = false;
    int planned_rows = -1;                             // rows_full the contexts' layouts were set for
    hipEvent_t order_ev = nullptr;                     // "everything queued on context 0's stream so far" (msplat_group_render)
    int exchange = MSPLAT_EXCHANGE_PEER_STORE;         // what msplat_group_render is asked to use; last_exchange: what it used
    int last_exchange = MSPLAT_EXCHANGE_PEER_STORE;
    std::vector<void*> comms;                          // MSPLAT_EXCHANGE_RCCL: one ncclComm_t per device (ncclCommInitAll)
    std::string err;
    std::vector<std::string> rank_err;                 // [i]: text of a failure on rank i's worker thread that is not the
                                                       // context's own (HIP calls of the staged exchange); merged by for_all
};

namespace {

thread_local std::string g_group_error;

int gfail(msplat_group* g, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_group_error = buf;
    if (g) g->err = buf;
    return code;
}

// failure inside rank i's part of a for_all (possibly on a worker thread): the text goes to the rank's own slot -- several
// ranks may fail at once (they raced on g->err) -- and for_all reports it from the caller's thread
int rfail(msplat_group* g, uint32_t i, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g->rank_err[i] = buf;
    return code;
}

// runs f(i) for every rank: ranks >= 1 on their worker threads, rank 0 on the caller's; returns the first real error
// (MSPLAT_ERR_PAIR_OVERFLOW[_EARLIER] -- a deferred report about an EARLIER frame -- only if nothing worse happened)
int for_all(msplat_group* g, const std::function<int(uint32_t)>& f)
{
    const uint32_t n = (uint32_t)g->ctx.size();
    for (auto& e : g->rank_err) e.clear();
    for (uint32_t i = 1; i < n; ++i) g->workers[i - 1]->post([&f, i] { return f(i); });
    std::vector<int> rc(n, MSPLAT_OK);
    rc[0] = f(0);
    for (uint32_t i = 1; i < n; ++i) rc[i] = g->workers[i - 1]->wait();
    int soft = MSPLAT_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (rc[i] == MSPLAT_OK) continue;
        if (rc[i] == MSPLAT_ERR_PAIR_OVERFLOW || rc[i] == MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) {
            if (!soft) { soft = rc[i]; g->err = msplat_last_error(g->ctx[i]); }
            continue;
        }
        g->err = std::string("device ") + std::to_string(g->devices[i]) + ": " +
                 (g->rank_err[i].empty() ? std::string(msplat_last_error(g->ctx[i])) : g->rank_err[i]);
        g_group_error = g->err;
        return rc[i];
    }
    if (soft) g_group_error = g->err;
    return soft;
}

// (re)assigns the bin rows when the viewport's row count changes
int plan_rows(msplat_group* g, const float viewport[4])
{
    if (!viewport) return gfail(g, MSPLAT_ERR_INVALID_ARG, "NULL viewport");
    const int H = (int)viewport[3], T = msplat_tile_size();
    if (H < 1) return gfail(g, MSPLAT_ERR_INVALID_ARG, "viewport height %d", H);
    const int rows = (H + T - 1) / T;
    const uint32_t n = (uint32_t)g->ctx.size();
    if (rows == g->planned_rows) return MSPLAT_OK;
    for (uint32_t i = 0; i < n; ++i) {
        int rc;
        if (n == 1) {
            rc = msplat_set_band(g->ctx[i], 1, 0);
        } else {
            int32_t first, count, block, stride;
            rc = msplat_band_plan(g->kind, rows, (int32_t)n, (int32_t)i, g->block_rows, &first, &count, &block, &stride);
            if (rc) return gfail(g, rc, "%s", msplat_last_error(nullptr));
            rc = msplat_set_band_layout(g->ctx[i], first, count, block, stride);
            if (!rc) rc = msplat_set_band_cull(g->ctx[i], g->band_cull ? 1 : 0);
        }
        if (rc) return gfail(g, rc, "%s", msplat_last_error(g->ctx[i]));
    }
    g->planned_rows = rows;
    return MSPLAT_OK;
}

}  // namespace

extern "C" {

const char* msplat_group_last_error(const msplat_group* g) { return g ? g->err.c_str() : g_group_error.c_str(); }

int msplat_group_create(msplat_group** out, const int32_t* devices, uint32_t n, const msplat_config* cfg)
{
    if (!out) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_group_create: need 1..64 devices");
    if (cfg && cfg->stream && n > 1)
        return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_group_create: a caller stream can only be used with one device");
    msplat_config c{};
    c.struct_size = sizeof(c);
    c.t_epsilon = -1.0f;
    if (cfg) {
        if (cfg->struct_size > sizeof(c) || cfg->struct_size < 16)
            return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_group_create: bad config struct_size %u", cfg->struct_size);
        std::memcpy(&c, cfg, cfg->struct_size);
        c.struct_size = sizeof(c);
    }
    msplat_group* g = new msplat_group;
    g->fb_format = c.fb_format;
    for (uint32_t i = 0; i < n; ++i) {
        c.device = devices[i];
        msplat_ctx* h = nullptr;
        const int rc = msplat_create(&h, &c);
        if (rc) {
            gfail(nullptr, rc, "msplat_group_create: device %d: %s", devices[i], msplat_last_error(nullptr));
            msplat_group_destroy(g);
            return rc;
        }
        g->ctx.push_back(h);
        g->devices.push_back(devices[i]);
    }
    g->peer_store.assign(n, true);
    g->stage.assign(n, nullptr);
    g->stage_bytes.assign(n, 0);
    g->rank_err.assign(n, std::string());
    if (n > 1 && (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&g->order_ev, hipEventDisableTiming) != hipSuccess)) {
        gfail(nullptr, MSPLAT_ERR_HIP, "msplat_group_create: cannot create the ordering event on device %d", devices[0]);
        msplat_group_destroy(g);
        return MSPLAT_ERR_HIP;
    }
    // peer mapping towards device 0: the other devices' compositors then write their rows into its framebuffer directly
    for (uint32_t i = 1; i < n; ++i) {
        if (devices[i] == devices[0]) continue;            // same device (tests): plain device memory
        int can = 0;
        bool ok = hipSetDevice(devices[i]) == hipSuccess && hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can;
        if (ok) {
            const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        }
        (void)hipGetLastError();
        g->peer_store[i] = ok;
    }
    g->peer_ok = g->peer_store;
    for (uint32_t i = 1; i < n; ++i) {
        g->workers.emplace_back(new Worker);
        Worker* w = g->workers.back().get();
        w->th = std::thread([w] { w->run(); });
    }
    if (const char* e = getenv("MSPLAT_GROUP_EXCHANGE")) {
        const std::string x = e;
        int want = -1;
        if (x == "rccl") want = MSPLAT_EXCHANGE_RCCL;
        else if (x == "copy") want = MSPLAT_EXCHANGE_COPY;
        else if (x == "peer" || x == "peer_store") want = MSPLAT_EXCHANGE_PEER_STORE;
        if (want < 0) {                 // a typo ("nccl") must not silently select the default
            gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_group_create: MSPLAT_GROUP_EXCHANGE=%s: expected peer | peer_store | rccl | copy", e);
            msplat_group_destroy(g);
            return MSPLAT_ERR_INVALID_ARG;
        }
        const int rc = msplat_group_set_exchange(g, want);
        if (rc) {                       // asked for in the environment and not available: say so instead of silently doing something else
            gfail(nullptr, rc, "msplat_group_create: MSPLAT_GROUP_EXCHANGE=%s: %s", e, g->err.c_str());
            msplat_group_destroy(g);
            return rc;
        }
    }
    *out = g;
    return MSPLAT_OK;
}

int msplat_group_set_exchange(msplat_group* g, int32_t exchange)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    const uint32_t n = (uint32_t)g->ctx.size();
    if (exchange == MSPLAT_EXCHANGE_PEER_STORE || exchange == MSPLAT_EXCHANGE_COPY) {
        for (uint32_t i = 1; i < n; ++i) g->peer_store[i] = exchange == MSPLAT_EXCHANGE_PEER_STORE && g->peer_ok[i];
        g->exchange = exchange;
        return MSPLAT_OK;
    }
    if (exchange != MSPLAT_EXCHANGE_RCCL) return gfail(g, MSPLAT_ERR_INVALID_ARG, "msplat_group_set_exchange: unknown exchange %d", exchange);
    if (n == 1) { g->exchange = exchange; return MSPLAT_OK; }          // one device: there is nothing to exchange
    const Rccl& R = rccl();
    if (!R.ok()) return gfail(g, MSPLAT_ERR_UNSUPPORTED, "msplat_group_set_exchange: %s", R.why.c_str());
    if (g->comms.empty()) {
        // one communicator per device of this process (rccl.h:236).  RCCL refuses a device that is listed twice.
        std::vector<void*> comms(n, nullptr);
        const int rc = R.CommInitAll(comms.data(), (int)n, g->devices.data());
        if (rc != 0) {
            (void)hipGetLastError();
            return gfail(g, MSPLAT_ERR_UNSUPPORTED, "msplat_group_set_exchange: ncclCommInitAll over %u devices failed: %s", n,
                         R.GetErrorString(rc));
        }
        g->comms = comms;
    }
    for (uint32_t i = 1; i < n; ++i) g->peer_store[i] = false;          // rows are rendered locally and sent
    g->exchange = exchange;
    return MSPLAT_OK;
}

int msplat_group_get_exchange(const msplat_group* g) { return g ? g->last_exchange : -1; }

void msplat_group_destroy(msplat_group* g)
{
    if (!g) return;
    for (auto& w : g->workers) {
        { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
    }
    for (size_t i = 0; i < g->ctx.size(); ++i)
        if (g->ctx[i]) (void)msplat_synchronize(g->ctx[i]);
    for (void* c : g->comms)
        if (c && rccl().ok()) (void)rccl().CommDestroy(c);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        if (i < g->stage.size() && g->stage[i]) { (void)hipSetDevice(g->devices[i]); (void)hipFree(g->stage[i]); }
        msplat_destroy(g->ctx[i]);
    }
    if (g->order_ev) { (void)hipSetDevice(g->devices[0]); (void)hipEventDestroy(g->order_ev); }
    delete g;
}

uint32_t msplat_group_size(const msplat_group* g) { return g ? (uint32_t)g->ctx.size() : 0u; }

msplat_ctx* msplat_group_context(msplat_group* g, uint32_t i) { return (g && i < g->ctx.size()) ? g->ctx[i] : nullptr; }

int msplat_group_peer_store(const msplat_group* g, uint32_t i) { return (g && i < g->ctx.size() && g->peer_store[i]) ? 1 : 0; }

int msplat_group_upload_cloud(msplat_group* g, const void* aos, uint64_t n, uint32_t stride_bytes, const msplat_attr_offsets* off,
                              int full_sh)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    return for_all(g, [&](uint32_t i) { return msplat_upload_cloud(g->ctx[i], aos, n, stride_bytes, off, full_sh); });
}

int msplat_group_upload_gaussian_cloud(msplat_group* g, const msplat_cloud* c)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    return for_all(g, [&](uint32_t i) { return msplat_upload_gaussian_cloud(g->ctx[i], c); });
}

int msplat_group_upload_ply(msplat_group* g, const char* path, int import_full_sh)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    return for_all(g, [&](uint32_t i) { return msplat_upload_ply(g->ctx[i], path, import_full_sh); });
}

int msplat_group_set_layout(msplat_group* g, int32_t kind, int32_t block_rows)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    if (kind != MSPLAT_BANDS_CONTIGUOUS && kind != MSPLAT_BANDS_INTERLEAVED && kind != MSPLAT_BANDS_BLOCK_INTERLEAVED &&
        kind != MSPLAT_BANDS_ROOT_WEIGHTED)
        return gfail(g, MSPLAT_ERR_INVALID_ARG, "msplat_group_set_layout: unknown kind %d", kind);
    if ((kind == MSPLAT_BANDS_BLOCK_INTERLEAVED || kind == MSPLAT_BANDS_ROOT_WEIGHTED) && block_rows < 1)
        return gfail(g, MSPLAT_ERR_INVALID_ARG, "msplat_group_set_layout: block_rows (rows per block / root weight in percent) must be >= 1");
    g->kind = kind;
    g->block_rows = std::max(1, block_rows);
    g->planned_rows = -1;
    return MSPLAT_OK;
}

int msplat_group_set_band_cull(msplat_group* g, int enable)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    g->band_cull = enable != 0;
    g->planned_rows = -1;
    return MSPLAT_OK;
}

int msplat_group_sort(msplat_group* g, const float cameraMat[16], const float projMat[16], const float viewport[4],
                      const float nearFar[2])
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    int rc = plan_rows(g, viewport);
    if (rc) return rc;
    return for_all(g, [&](uint32_t i) { return msplat_sort(g->ctx[i], cameraMat, projMat, viewport, nearFar); });
}

int msplat_group_render(msplat_group* g, const float cameraMat[16], const float projMat[16], const float viewport[4],
                        const float nearFar[2], void* rgba, uint64_t pitch_bytes, int out_is_device)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    if (!rgba) return gfail(g, MSPLAT_ERR_INVALID_ARG, "msplat_group_render: rgba is NULL");
    int rc = plan_rows(g, viewport);
    if (rc) return rc;
    const uint32_t n = (uint32_t)g->ctx.size();
    const int W = (int)viewport[2], H = (int)viewport[3];
    const size_t bpp = g->fb_format == MSPLAT_FB_RGBA16F ? 8 : 16;
    const size_t tight = (size_t)W * bpp;
    if (pitch_bytes == 0) pitch_bytes = tight;
    if (W < 1 || H < 1 || pitch_bytes < tight || pitch_bytes % bpp != 0)
        return gfail(g, MSPLAT_ERR_INVALID_ARG, "msplat_group_render: pitch %llu too small / misaligned for a %dx%d target of %zu bytes per pixel",
                     (unsigned long long)pitch_bytes, W, H, bpp);
    if (!out_is_device)         // host image: every context copies its own rows into it (msplat_render's band rule)
        return for_all(g, [&](uint32_t i) { return msplat_render(g->ctx[i], cameraMat, projMat, viewport, nearFar, rgba, pitch_bytes, 0); });

    // The frame inherits context 0's stream order on EVERY rank: whatever the caller queued on that stream before
    // this call -- typically the consumer of the previous frame in the same framebuffer -- is finished before any rank's
    // compositor or row copy writes into `rgba`, exactly as with a single context.  One event record + n - 1 stream waits.
    void* stream0 = msplat_get_stream(g->ctx[0]);
    if (n > 1) {
        if (hipSetDevice(g->devices[0]) != hipSuccess || hipEventRecord(g->order_ev, (hipStream_t)stream0) != hipSuccess)
            return gfail(g, MSPLAT_ERR_HIP, "msplat_group_render: cannot record the ordering event on device %d", g->devices[0]);
    }
    const bool use_rccl = n > 1 && g->exchange == MSPLAT_EXCHANGE_RCCL && !g->comms.empty();
    g->last_exchange = n == 1 ? g->exchange : (use_rccl ? MSPLAT_EXCHANGE_RCCL : MSPLAT_EXCHANGE_COPY);
    for (uint32_t i = 1; i < n && !use_rccl; ++i)
        if (g->peer_store[i]) g->last_exchange = MSPLAT_EXCHANGE_PEER_STORE;
    if (use_rccl) {
        // every rank renders its rows into an image of its own with the target's pitch (a run of rows is then one contiguous
        // range on both sides); the caller's thread then issues ALL sends and receives as one group: rank i's sends on its
        // stream behind its compositor, device 0's receives on context 0's stream -- synchronising that stream = frame complete
        rc = for_all(g, [&](uint32_t i) -> int {
            msplat_ctx* c = g->ctx[i];
            if (i == 0) return msplat_render(c, cameraMat, projMat, viewport, nearFar, rgba, pitch_bytes, 1);
            const int j = msplat_wait_event(c, g->order_ev);
            if (j) return j;
            if (hipSetDevice(g->devices[i]) != hipSuccess) return rfail(g, i, MSPLAT_ERR_HIP, "hipSetDevice(%d) failed", g->devices[i]);
            const size_t need = (size_t)pitch_bytes * (size_t)H;
            if (g->stage_bytes[i] < need) {
                (void)msplat_synchronize(c);
                if (g->stage[i]) (void)hipFree(g->stage[i]);
                g->stage[i] = nullptr;
                g->stage_bytes[i] = 0;
                if (hipMalloc(&g->stage[i], need) != hipSuccess) return rfail(g, i, MSPLAT_ERR_HIP, "staging framebuffer: out of device memory");
                g->stage_bytes[i] = need;
            }
            return msplat_render(c, cameraMat, projMat, viewport, nearFar, g->stage[i], pitch_bytes, 1);
        });
        if (rc && rc != MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) return rc;
        // (contexts created with async_submit: every rank's worker has issued its render before the sends are queued behind it)
        for (uint32_t i = 0; i < n; ++i) {
            const int w = msplat_stream_wait(g->ctx[i], msplat_get_stream(g->ctx[i]));
            if (w != MSPLAT_OK && w != MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) return gfail(g, w, "msplat_group_render: %s", msplat_last_error(g->ctx[i]));
        }
        const Rccl& R = rccl();
        int nrc = R.GroupStart();
        for (uint32_t i = 1; i < n && nrc == 0; ++i) {
            hipStream_t si = (hipStream_t)msplat_get_stream(g->ctx[i]);
            const int prc = for_each_run(g->kind, g->block_rows, (int32_t)n, (int32_t)i, H, [&](int y0, int nrows) -> int {
                const size_t off = (size_t)y0 * pitch_bytes, bytes = (size_t)(nrows - 1) * pitch_bytes + tight;
                int e = R.Send((const char*)g->stage[i] + off, bytes, kNcclUint8, 0, g->comms[i], si);
                if (e == 0) e = R.Recv((char*)rgba + off, bytes, kNcclUint8, (int)i, g->comms[0], (hipStream_t)stream0);
                if (e != 0) nrc = e;
                return e != 0 ? MSPLAT_ERR_HIP : MSPLAT_OK;
            });
            if (prc && nrc == 0) { (void)R.GroupEnd(); return gfail(g, prc, "msplat_group_render: %s", msplat_last_error(nullptr)); }
        }
        const int erc = R.GroupEnd();
        if (nrc == 0) nrc = erc;
        if (nrc != 0) return gfail(g, MSPLAT_ERR_HIP, "msplat_group_render: RCCL row exchange failed: %s", R.GetErrorString(nrc));
        return rc;
    }
    rc = for_all(g, [&](uint32_t i) -> int {
        msplat_ctx* c = g->ctx[i];
        if (i != 0) { const int j = msplat_wait_event(c, g->order_ev); if (j) return j; }
        if (i == 0 || g->peer_store[i]) {
            // zero-copy: this device's compositor writes its rows where they belong in device 0's framebuffer
            int r = msplat_render(c, cameraMat, projMat, viewport, nearFar, rgba, pitch_bytes, 1);
            if (r && r != MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) return r;
            if (i != 0) { const int j = msplat_stream_wait(c, stream0); if (j) return j; }
            return r;
        }
        // no peer mapping: render into a local image, then one copy per run of consecutive owned rows
        if (hipSetDevice(g->devices[i]) != hipSuccess) return rfail(g, i, MSPLAT_ERR_HIP, "hipSetDevice(%d) failed", g->devices[i]);
        const size_t need = tight * (size_t)H;
        if (g->stage_bytes[i] < need) {
            (void)msplat_synchronize(c);
            if (g->stage[i]) (void)hipFree(g->stage[i]);
            g->stage[i] = nullptr;
            g->stage_bytes[i] = 0;
            if (hipMalloc(&g->stage[i], need) != hipSuccess) return rfail(g, i, MSPLAT_ERR_HIP, "staging framebuffer: out of device memory");
            g->stage_bytes[i] = need;
        }
        int r = msplat_render(c, cameraMat, projMat, viewport, nearFar, g->stage[i], tight, 1);
        if (r && r != MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) return r;
        hipStream_t s = (hipStream_t)msplat_get_stream(c);
        const int pr = for_each_run(g->kind, g->block_rows, (int32_t)n, (int32_t)i, H, [&](int y0, int nrows) -> int {
            if (hipMemcpy2DAsync((char*)rgba + (size_t)y0 * pitch_bytes, pitch_bytes, (const char*)g->stage[i] + (size_t)y0 * tight,
                                 tight, tight, (size_t)nrows, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return rfail(g, i, MSPLAT_ERR_HIP, "hipMemcpy2DAsync (device %d -> %d) failed", g->devices[i], g->devices[0]);
            return MSPLAT_OK;
        });
        if (pr) { if (g->rank_err[i].empty()) rfail(g, i, pr, "%s", msplat_last_error(nullptr)); return pr; }
        const int j = msplat_stream_wait(c, stream0);
        return j ? j : r;
    });
    return rc;
}

// ---- the exchange for one process per GPU --------------------------------------------------------------------------------------
static int band_exchange_impl(msplat_ctx* ctx, void* comm, int32_t rank, int32_t world, int32_t root, int32_t kind, int32_t block_rows,
                              const void* src, void* dst, uint64_t pitch_bytes, int32_t width, int32_t height, int32_t flags, bool loopback)
{
    if (!ctx) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_band_exchange: ctx is NULL");
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_band_exchange: rank %d / root %d of %d", rank, root, world);
    if (world == 1 && !loopback) return MSPLAT_OK;                     // the whole image is this rank's
    if (!comm || !src || !dst || width < 1 || height < 1)
        return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_band_exchange: NULL communicator / framebuffer or empty image");
    const bool wire16 = (flags & MSPLAT_EXCHANGE_WIRE_FP16) != 0;
    // the pixel size is the CONTEXT's target format (it used to be inferred from the pitch)
    const size_t bpp = msplat_get_fb_format(ctx) == MSPLAT_FB_RGBA16F ? 8u : 16u;
    const size_t tight = (size_t)width * bpp;
    if (wire16 && bpp != 16u)
        return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_band_exchange: MSPLAT_EXCHANGE_WIRE_FP16 needs an RGBA32F target");
    if (pitch_bytes < tight || pitch_bytes % bpp != 0)
        return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "msplat_band_exchange: pitch %llu too small / misaligned for width %d of %zu-byte pixels",
                     (unsigned long long)pitch_bytes, width, bpp);
    const Rccl& R = rccl();
    if (!R.ok()) return gfail(nullptr, MSPLAT_ERR_UNSUPPORTED, "msplat_band_exchange: %s", R.why.c_str());
    hipStream_t s = (hipStream_t)msplat_get_stream(ctx);
    // an async_submit context: its worker thread must have ISSUED the frame before the exchange is queued behind it (a wait on the
    // context's own stream is just that; it may also hand over a queued call's overflow warning, which is passed on at the end)
    const int wrc = msplat_stream_wait(ctx, s);
    if (wrc != MSPLAT_OK && wrc != MSPLAT_ERR_PAIR_OVERFLOW_EARLIER) return gfail(nullptr, wrc, "msplat_band_exchange: %s", msplat_last_error(ctx));

    // the runs this call sends and the runs it receives: (peer, first pixel row, rows)
    struct Run { int peer, y0, nrows; };
    std::vector<Run> sends, recvs;
    int prc = MSPLAT_OK;
    if (loopback) {
        // (tests, one rank: this rank plays owner and root at once -- its runs travel from src to dst through ncclSend / ncclRecv to itself)
        prc = for_each_run(kind, block_rows, world, rank, height, [&](int y0, int nrows) -> int {
            sends.push_back(Run{0, y0, nrows}); recvs.push_back(Run{0, y0, nrows}); return MSPLAT_OK; });
    } else if (rank == root) {
        for (int32_t r = 0; r < world && prc == MSPLAT_OK; ++r)
            if (r != root) prc = for_each_run(kind, block_rows, world, r, height, [&](int y0, int nrows) -> int { recvs.push_back(Run{r, y0, nrows}); return MSPLAT_OK; });
    } else {
        prc = for_each_run(kind, block_rows, world, rank, height, [&](int y0, int nrows) -> int { sends.push_back(Run{root, y0, nrows}); return MSPLAT_OK; });
    }
    if (prc) return gfail(nullptr, prc, "msplat_band_exchange: %s", msplat_last_error(nullptr));

    // fp16 on the wire: a run is packed into (and received into) stream-ordered scratch; half the bytes cross the link
    char* scratch = nullptr;
    std::vector<size_t> soff(sends.size()), roff(recvs.size());
    if (wire16) {
        size_t total = 0;
        for (size_t i = 0; i < sends.size(); ++i) { soff[i] = total; total += (size_t)sends[i].nrows * width * 8; }
        for (size_t i = 0; i < recvs.size(); ++i) { roff[i] = total; total += (size_t)recvs[i].nrows * width * 8; }
        if (total != 0 && hipMallocAsync((void**)&scratch, total, s) != hipSuccess) {
            (void)hipGetLastError();
            return gfail(nullptr, MSPLAT_ERR_HIP, "msplat_band_exchange: no memory for %zu bytes of fp16 rows", total);
        }
        for (size_t i = 0; i < sends.size(); ++i) {
            const size_t px = (size_t)sends[i].nrows * width;
            hipLaunchKernelGGL(pack_rows_f16, dim3((unsigned)std::min<size_t>((px + 255) / 256, 4096)), dim3(256), 0, s,
                               (const char*)src + (size_t)sends[i].y0 * pitch_bytes, (size_t)pitch_bytes, width, sends[i].nrows,
                               (uint2*)(scratch + soff[i]));
        }
    }
    // a run travels as ONE message when the rows are tight (pitch == width x pixel size: the usual case), else row by row inside
    // the same group: nothing outside the target's own pixels is read or written -- a target that is a window of a wider
    // surface keeps its neighbours, and the last row need not be followed by a full pitch; or as tight fp16 rows
    const bool tight_rows = (size_t)pitch_bytes == tight;
    auto post = [&](bool send, const Run& r, size_t scratch_off) -> int {
        if (wire16)
            return send ? R.Send(scratch + scratch_off, (size_t)r.nrows * width * 8, kNcclUint8, r.peer, comm, s)
                        : R.Recv(scratch + scratch_off, (size_t)r.nrows * width * 8, kNcclUint8, r.peer, comm, s);
        const int ops = tight_rows ? 1 : r.nrows;
        const size_t bytes = tight_rows ? (size_t)r.nrows * tight : tight;
        for (int k = 0; k < ops; ++k) {
            const size_t off = (size_t)(r.y0 + k) * (size_t)pitch_bytes;
            const int e = send ? R.Send((const char*)src + off, bytes, kNcclUint8, r.peer, comm, s)
                               : R.Recv((char*)dst + off, bytes, kNcclUint8, r.peer, comm, s);
            if (e != 0) return e;
        }
        return 0;
    };
    int nrc = R.GroupStart();
    if (nrc == 0) {
        for (size_t i = 0; i < sends.size() && nrc == 0; ++i) nrc = post(true, sends[i], wire16 ? soff[i] : 0);
        for (size_t i = 0; i < recvs.size() && nrc == 0; ++i) nrc = post(false, recvs[i], wire16 ? roff[i] : 0);
        const int erc = R.GroupEnd();
        if (nrc == 0) nrc = erc;
    }
    if (wire16) {
        if (nrc == 0)
            for (size_t i = 0; i < recvs.size(); ++i) {
                const size_t px = (size_t)recvs[i].nrows * width;
                hipLaunchKernelGGL(unpack_rows_f16, dim3((unsigned)std::min<size_t>((px + 255) / 256, 4096)), dim3(256), 0, s,
                                   (const uint2*)(scratch + roff[i]), (char*)dst + (size_t)recvs[i].y0 * pitch_bytes, (size_t)pitch_bytes,
                                   width, recvs[i].nrows);
            }
        if (scratch) (void)hipFreeAsync(scratch, s);
        if (hipGetLastError() != hipSuccess) return gfail(nullptr, MSPLAT_ERR_HIP, "msplat_band_exchange: a pack / unpack launch failed");
    }
    if (nrc != 0) return gfail(nullptr, MSPLAT_ERR_HIP, "msplat_band_exchange: RCCL: %s", R.GetErrorString(nrc));
    if (wrc != MSPLAT_OK) return gfail(nullptr, wrc, "%s", msplat_last_error(ctx));
    return MSPLAT_OK;
}

int msplat_band_exchange(msplat_ctx* ctx, void* comm, int32_t rank, int32_t world, int32_t root, int32_t kind, int32_t block_rows,
                         void* rgba, uint64_t pitch_bytes, int32_t width, int32_t height, int32_t flags)
{
    return band_exchange_impl(ctx, comm, rank, world, root, kind, block_rows, rgba, rgba, pitch_bytes, width, height, flags, false);
}

int msplat_debug_band_exchange_loopback(msplat_ctx* ctx, void* comm, int32_t kind, int32_t block_rows, int32_t world, int32_t rank,
                                        const void* src, void* dst, uint64_t pitch_bytes, int32_t width, int32_t height, int32_t flags)
{
    return band_exchange_impl(ctx, comm, rank, world, 0, kind, block_rows, src, dst, pitch_bytes, width, height, flags, true);
}

int msplat_group_synchronize(msplat_group* g)
{
    if (!g) return gfail(nullptr, MSPLAT_ERR_INVALID_ARG, "group is NULL");
    return for_all(g, [&](uint32_t i) { return msplat_synchronize(g->ctx[i]); });
}

}  // extern "C"
