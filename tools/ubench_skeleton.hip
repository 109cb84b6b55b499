// ubench_skeleton.hip -- a frame of config 2 as a SKELETON: the same chain of launches (grid, workgroup size, LDS and VGPR footprint
// of every kernel), but every workgroup only WAITS (s_sleep) for the lifetime its real counterpart has when its frame is alone on the
// GPU (profiles/r06_stamp_timeline_cfg2.txt).  Nothing is computed and nothing is read: what P such chains on P streams reach per
// frame is what the launch path and the residency rules alone allow.  (r6; EXPERIMENTS.md "frame skeleton")
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_skeleton.hip -o tools/bin/ubench_skeleton && tools/bin/ubench_skeleton [variant ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// VG: the kernel's VGPR allocation (16, 40, 64, 80, 88, 96, 104, 112 or 128): an asm clobber of the highest register makes the
// allocation; the registers are never written
template <int VG> __device__ __forceinline__ void clobber();
#define CL(N, R) template <> __device__ __forceinline__ void clobber<N>() { asm volatile("" ::: R); }
CL(16, "v15") CL(40, "v39") CL(64, "v63") CL(80, "v79") CL(88, "v87") CL(96, "v95") CL(104, "v103") CL(112, "v111") CL(128, "v127")
#undef CL

template <int THREADS, int VG>
__global__ __launch_bounds__(THREADS) void wait_kernel(unsigned ticks, unsigned* sink)
{
    extern __shared__ unsigned s_dyn[];
    if (threadIdx.x == 0) s_dyn[0] = blockIdx.x;
    clobber<VG>();
    const unsigned long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (s_dyn[0] == 0xFFFFFFFFu) sink[0] = 1;              // keeps the LDS allocation alive
}

struct K { const char* name; int wgs, threads, lds, vg; float life_us; };

static void launch(const K& k, hipStream_t s, unsigned* sink)
{
    const unsigned ticks = (unsigned)(k.life_us * 100.0f);
#define L(T, V) hipLaunchKernelGGL((wait_kernel<T, V>), dim3(k.wgs), dim3(T), k.lds, s, ticks, sink)
#define LV(T) do { switch (k.vg) { case 16: L(T, 16); break; case 40: L(T, 40); break; case 64: L(T, 64); break; case 80: L(T, 80); break; case 88: L(T, 88); break; \
                                  case 96: L(T, 96); break; case 104: L(T, 104); break; case 112: L(T, 112); break; case 128: L(T, 128); break; \
                                  default: printf("no VGPR class %d\n", k.vg); exit(1); } } while (0)
    if (k.threads == 64) LV(64); else if (k.threads == 128) LV(128); else if (k.threads == 256) LV(256); else if (k.threads == 512) LV(512); else LV(1024);
#undef LV
#undef L
}

// the in-flight kernel selection of config 2 (1 M splats, 1920x1080): workgroups, threads, LDS bytes, VGPR class, solo lifetime (us)
static std::vector<K> frame_product()
{
    return {
        {"ws_upsweep<cull>", 490, 256, 9296, 88, 6.9f},   {"ws_downsweep<cull>", 490, 256, 41024, 112, 9.0f},
        {"ws_upsweep", 482, 256, 8192, 40, 1.7f},         {"ws_downsweep", 482, 256, 41024, 104, 7.1f},
        {"ws_upsweep", 482, 256, 8192, 40, 1.7f},         {"ws_downsweep", 482, 256, 41024, 104, 7.1f},
        {"project_kernel", 15406, 64, 17408, 104, 5.85f},
        {"bin1_upsweep", 963, 256, 1056, 40, 2.35f},      {"bin1_downsweep", 963, 256, 29728, 104, 13.5f},
        {"radix_upsweep<pair>", 1609, 256, 6176, 64, 2.9f}, {"radix_downsweep<pair>", 1609, 256, 27664, 128, 4.8f},
        {"tile_start_kernel", 480, 256, 1056, 16, 3.0f},
        {"composite_kernel", 1280, 64, 3120, 80, 100.0f},
    };
}
enum { PROJ = 6, COMP = 12 };

static float run(const std::vector<K>& fr, int P, int frames, unsigned* sink, std::vector<hipStream_t>& st)
{
    for (int f = 0; f < 4 * P; ++f) for (const K& k : fr) launch(k, st[f % P], sink);
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) for (const K& k : fr) launch(k, st[f % P], sink);
    CHECK(hipDeviceSynchronize());
    return 1e6f * std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count() / frames;
}

int main(int argc, char** argv)
{
    unsigned* sink;
    CHECK(hipMalloc(&sink, 64));
    std::vector<hipStream_t> st(8);
    // argv[1] = "halves": stream k runs on the even (k even) / odd (k odd) CU positions of every XCD (hipExtStreamCreateWithCUMask:
    // mask bit 8 c + x = position c of XCD x) -- msplat_config.cu_partition's partition; "quarters": position c % 4 == k % 4
    const int groups = argc > 1 && !strcmp(argv[1], "halves") ? 2 : argc > 1 && !strcmp(argv[1], "quarters") ? 4 : 1;
    for (size_t k = 0; k < st.size(); ++k) {
        if (groups == 1) { CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking)); continue; }
        unsigned mask[8] = {0};
        for (int i = 0; i < 256; ++i) if ((i / 8) % groups == (int)k % groups) mask[i / 32] |= 1u << (i % 32);
        CHECK(hipExtStreamCreateWithCUMask(&st[k], 8, mask));
    }
    printf("# streams: %s\n", groups == 1 ? "every CU" : groups == 2 ? "halves of the CU positions (even / odd)" : "quarters of the CU positions");
#define OPT1(T, V) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wait_kernel<T, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536))
#define OPT(T) do { OPT1(T, 16); OPT1(T, 40); OPT1(T, 64); OPT1(T, 80); OPT1(T, 88); OPT1(T, 96); OPT1(T, 104); OPT1(T, 112); OPT1(T, 128); } while (0)
    OPT(64); OPT(128); OPT(256); OPT(512); OPT(1024);
    auto show = [&](const char* label, const std::vector<K>& fr) {
        float solo = 0.0f;
        printf("%-64s", label);
        for (int P = 1; P <= 4; ++P) {
            const float us = run(fr, P, 240, sink, st);
            if (P == 1) solo = us;
            printf("  P=%d %7.1f us/frame", P, us);
        }
        printf("   (x%.2f)\n", solo / run(fr, 4, 240, sink, st));
    };
    // every kernel of the chain alone: launch-to-end on one stream, for the table
    {
        const auto fr = frame_product();
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float sum = 0.0f;
        for (const K& k : fr) {
            std::vector<float> t;
            for (int r = 0; r < 20; ++r) {
                CHECK(hipEventRecord(a, st[0])); launch(k, st[0], sink); CHECK(hipEventRecord(b, st[0])); CHECK(hipStreamSynchronize(st[0]));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); t.push_back(1e3f * ms);
            }
            std::sort(t.begin(), t.end());
            printf("  %-24s %6d workgroups x %4d threads, %6d B LDS, VGPR class %d, lifetime %6.2f us: alone %7.1f us\n", k.name, k.wgs, k.threads, k.lds, k.vg, k.life_us, t[t.size() / 2]);
            sum += t[t.size() / 2];
        }
        printf("  sum of the kernels alone: %.1f us\n", sum);
    }
    show("product shapes", frame_product());
    typedef std::vector<K> F;
    auto var = [&](const char* label, void (*edit)(F&)) { F fr = frame_product(); edit(fr); show(label, fr); };
    auto chain = [](const K& k) { return strcmp(k.name, "project_kernel") != 0 && strcmp(k.name, "composite_kernel") != 0; };
    (void)chain;
    // what is each kernel's footprint worth?
    var("no compositor (chain only)", [](F& f) { f.pop_back(); });
    var("no projection", [](F& f) { f.erase(f.begin() + PROJ); });
    var("all kernels: 16 VGPRs", [](F& f) { for (auto& k : f) k.vg = 16; });
    var("all kernels: <= 1 KB of LDS", [](F& f) { for (auto& k : f) k.lds = std::min(k.lds, 1024); });
    var("all kernels: 16 VGPRs and <= 1 KB of LDS", [](F& f) { for (auto& k : f) { k.vg = 16; k.lds = std::min(k.lds, 1024); } });
    var("sort + binning kernels: 64 VGPRs", [](F& f) { for (auto& k : f) if (k.threads == 256) k.vg = std::min(k.vg, 64); });
    var("sort + binning kernels: 80 VGPRs", [](F& f) { for (auto& k : f) if (k.threads == 256) k.vg = std::min(k.vg, 80); });
    var("sort + binning kernels: 96 VGPRs", [](F& f) { for (auto& k : f) if (k.threads == 256) k.vg = std::min(k.vg, 96); });
    var("sort + binning kernels: half the LDS", [](F& f) { for (auto& k : f) if (k.threads == 256) k.lds /= 2; });
    var("sort + binning kernels: 80 VGPRs, half the LDS", [](F& f) { for (auto& k : f) if (k.threads == 256) { k.vg = std::min(k.vg, 80); k.lds /= 2; } });
    var("projection: 64 VGPRs", [](F& f) { f[PROJ].vg = 64; });
    var("projection: 80 VGPRs", [](F& f) { f[PROJ].vg = 80; });
    var("projection: 9 KB of LDS", [](F& f) { f[PROJ].lds = 9216; });
    var("projection: 5 KB of LDS", [](F& f) { f[PROJ].lds = 5120; });
    var("projection: 5 KB of LDS, 80 VGPRs", [](F& f) { f[PROJ].lds = 5120; f[PROJ].vg = 80; });
    var("projection: 5 KB of LDS, 128 VGPRs", [](F& f) { f[PROJ].lds = 5120; f[PROJ].vg = 128; });
    var("projection as 4-wave workgroups", [](F& f) { f[PROJ] = {"project_kernel", 3852, 256, 4 * 17408, 104, 5.85f}; });
    var("projection: half the workgroups, twice the lifetime", [](F& f) { f[PROJ].wgs /= 2; f[PROJ].life_us *= 2; });
    var("projection as 2304 persistent one-wave workgroups", [](F& f) { f[PROJ].wgs = 2304; f[PROJ].life_us = 39.1f; });
    var("compositor: 64 VGPRs", [](F& f) { f[COMP].vg = 64; });
    var("compositor: 128 VGPRs", [](F& f) { f[COMP].vg = 128; });
    var("compositor pool 2560 x 50 us", [](F& f) { f[COMP].wgs = 2560; f[COMP].life_us = 50.0f; });
    var("compositor pool 640 x 200 us", [](F& f) { f[COMP].wgs = 640; f[COMP].life_us = 200.0f; });
    var("compositor as 320 four-wave workgroups", [](F& f) { f[COMP] = {"composite_kernel", 320, 256, 4 * 3120, 80, 100.0f}; });
    var("sort as 8-wave workgroups (245)", [](F& f) { for (auto& k : f) if (k.threads == 256 && k.wgs < 600 && k.vg != 16) { k.wgs = (k.wgs + 1) / 2; k.threads = 512; k.lds = std::min(2 * k.lds, 65536); } });
    var("sort + binning as 2-wave workgroups (twice as many)", [](F& f) { for (auto& k : f) if (k.threads == 256) { k.wgs *= 2; k.threads = 128; k.lds /= 2; } });
    var("everything as one-wave workgroups", [](F& f) { for (auto& k : f) if (k.threads == 256) { k.wgs *= 4; k.threads = 64; k.lds /= 4; } });
    var("no tile_start launch", [](F& f) { f.erase(f.begin() + 11); });
    var("each sort pass as ONE launch (upsweep + downsweep lifetimes)", [](F& f) { F g; for (size_t i = 0; i < f.size(); ++i) { if (i < 6 && (i & 1) == 0) continue; K k = f[i]; if (i < 6) k.life_us += f[i - 1].life_us; g.push_back(k); } f = g; });
    return 0;
}
