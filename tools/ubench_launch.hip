// ubench_launch.hip -- the workgroup launch path of an MI355X, in isolation (r6).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/bin/ubench_launch && tools/bin/ubench_launch
// The frame's kernels with four frames in flight run at their solo speed once launched but START late.  This program
// reproduces the situation with dummy kernels whose workgroups only WAIT (s_sleep for a given time), so that nothing but the launch
// path and the slot bookkeeping is involved:
//   A: many one-wave workgroups with 17 KB of LDS and a ~5 us lifetime   (the shape of project_kernel: 15.6 k workgroups)
//   B: 490 four-wave workgroups with 40 KB of LDS and a ~8 us lifetime   (the shape of ws_downsweep with frames in flight)
// measured alone and together on two streams: launch rate of A, duration of B alone, duration of B while A is being dispatched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void wait_kernel(unsigned long long ticks, unsigned* sink)
{
    extern __shared__ unsigned s_dyn[];
    if (threadIdx.x == 0) s_dyn[0] = blockIdx.x;
    const unsigned long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (s_dyn[0] == 0xFFFFFFFFu) sink[0] = 1;              // keeps the LDS allocation alive
}

static float ms_between(hipEvent_t a, hipEvent_t b) { float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main()
{
    unsigned* sink;
    CHECK(hipMalloc(&sink, 64));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wait_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int A_WGS = 15625, B_WGS = 490;
    const unsigned long long A_T = 500, B_T = 800;          // 5 us, 8 us
    auto launchA = [&](hipStream_t s, int lds) { hipLaunchKernelGGL(wait_kernel<64>, dim3(A_WGS), dim3(64), lds, s, A_T, sink); };
    auto launchB = [&](hipStream_t s, int threads, int lds) {
        if (threads == 256) hipLaunchKernelGGL(wait_kernel<256>, dim3(B_WGS), dim3(256), lds, s, B_T, sink);
        else hipLaunchKernelGGL(wait_kernel<64>, dim3(B_WGS * 4), dim3(64), lds / 4, s, B_T, sink);
    };
    for (int i = 0; i < 20; ++i) { launchA(sa, 17408); launchB(sb, 256, 40960); }
    CHECK(hipDeviceSynchronize());

    auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    // 1. A alone: launch rate and implied residency
    for (int lds : {0, 4352, 9216, 17408}) {
        std::vector<float> t;
        for (int r = 0; r < 30; ++r) {
            CHECK(hipEventRecord(a0, sa)); launchA(sa, lds); CHECK(hipEventRecord(a1, sa)); CHECK(hipStreamSynchronize(sa));
            t.push_back(ms_between(a0, a1));
        }
        const float ms = med(t);
        printf("A alone (%5d one-wave workgroups, %5d B LDS, 5 us each): %7.1f us  = %6.1f workgroups / us, ~%4.0f resident\n", A_WGS, lds, 1e3f * ms,
               A_WGS / (1e3f * ms), A_WGS / (1e3f * ms) * 5.0f);
    }
    // 2. B alone, as 4-wave workgroups and as the same waves in one-wave workgroups
    for (int threads : {256, 64}) {
        std::vector<float> t;
        for (int r = 0; r < 30; ++r) {
            CHECK(hipEventRecord(b0, sb)); launchB(sb, threads, 40960); CHECK(hipEventRecord(b1, sb)); CHECK(hipStreamSynchronize(sb));
            t.push_back(ms_between(b0, b1));
        }
        printf("B alone (%d workgroups of %3d threads, %5d B LDS, 8 us each): %7.1f us\n", threads == 256 ? B_WGS : 4 * B_WGS, threads,
               threads == 256 ? 40960 : 10240, 1e3f * med(t));
    }
    // 3. B while A is being dispatched on another stream (A launched first, B 10 us later by stream order of the host)
    for (int threads : {256, 64}) {
        for (int ldsA : {0, 4352, 9216, 17408}) {
            std::vector<float> tb, ta;
            for (int r = 0; r < 30; ++r) {
                CHECK(hipEventRecord(a0, sa)); launchA(sa, ldsA); CHECK(hipEventRecord(a1, sa));
                CHECK(hipEventRecord(b0, sb)); launchB(sb, threads, 40960); CHECK(hipEventRecord(b1, sb));
                CHECK(hipStreamSynchronize(sa)); CHECK(hipStreamSynchronize(sb));
                tb.push_back(ms_between(b0, b1)); ta.push_back(ms_between(a0, a1));
            }
            printf("B (%3d-thread workgroups) while A (%5d B LDS) dispatches: B %7.1f us, A %7.1f us\n", threads, ldsA, 1e3f * med(tb), 1e3f * med(ta));
        }
    }
    return 0;
}
