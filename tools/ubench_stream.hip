// ubench_stream.hip -- what does the MI355X deliver for project_kernel's traffic shape? (r6)
// project_kernel moves 256 B in and 56 B out per splat in one-wave workgroups of 64 splats (16 dwordx4 loads per lane, all issued
// first) and reaches 5.1 TB/s.  This program moves the same bytes with nothing in between, in several shapes:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/bin/ubench_stream && tools/bin/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave = 64 records of 256 B: lane l of the wave reads parts (it * 4 + l / 16) ... exactly project_block's pattern (16 lanes per
// record), sums them (so that the loads are used) and writes 48 + 8 bytes per lane.  LDS: bytes of dynamic LDS only to cap the residency.
template <int THREADS, bool PERSISTENT>
__global__ __launch_bounds__(THREADS) void stream_kernel(const float4* __restrict__ recs, float4* __restrict__ out, uint2* __restrict__ out2,
                                                          unsigned n_waves, const unsigned* __restrict__ perm)
{
    extern __shared__ unsigned s_dyn[];
    const int lane = threadIdx.x & 63;
    const unsigned wave0 = blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
    const unsigned stride = PERSISTENT ? gridDim.x * (THREADS / 64) : n_waves;
    for (unsigned w = wave0; w < n_waves; w += stride) {
        const unsigned base = w * 64u;
        float4 t[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            unsigned rec = base + it * 4 + lane / 16;
            if (perm) rec = perm[rec];
            t[it] = recs[(size_t)rec * 16 + (lane & 15)];
        }
        float4 a = t[0];
#pragma unroll
        for (int it = 1; it < 16; ++it) { a.x += t[it].x; a.y += t[it].y; a.z += t[it].z; a.w += t[it].w; }
        const size_t r = (size_t)base + lane;
        out[r * 3 + 0] = a; out[r * 3 + 1] = a; out[r * 3 + 2] = a;
        out2[r] = make_uint2(__float_as_uint(a.x), __float_as_uint(a.y));
    }
    if (s_dyn[0] == 0xFFFFFFFFu && threadIdx.x == 12345) out2[0] = make_uint2(0, 0);
}

// plain streams for the ceiling: every thread float4 loads, UNROLL in flight, grid-stride; COPY also writes them
template <int UNROLL, bool COPY>
__global__ __launch_bounds__(256) void plain_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, float4* __restrict__ sink)
{
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 t[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) t[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (COPY) dst[i + u * stride] = t[u];
            else { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
        }
    }
    if (!COPY && acc.x == 12345.678f) sink[0] = acc;
}

int main()
{
    const unsigned N = 985u * 1024u;             // visible splats of config 2 (a multiple of 64)
    const unsigned n_waves = N / 64;
    float4 *recs, *out; uint2* out2; unsigned* perm;
    CHECK(hipMalloc(&recs, (size_t)N * 256)); CHECK(hipMalloc(&out, (size_t)N * 48)); CHECK(hipMalloc(&out2, (size_t)N * 8));
    CHECK(hipMalloc(&perm, (size_t)N * 4));
    CHECK(hipMemset(recs, 0, (size_t)N * 256));
    {   // a depth-order-like permutation: a random shuffle (worst case for the gather)
        std::vector<unsigned> p(N);
        for (unsigned i = 0; i < N; ++i) p[i] = i;
        unsigned long long s = 88172645463325252ull;
        for (unsigned i = N - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(p[i], p[(unsigned)(s % (i + 1))]); }
        CHECK(hipMemcpy(perm, p.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const double bytes = (double)N * (256 + 56);
    auto time = [&](const char* label, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        std::vector<float> t;
        for (int r = 0; r < 30; ++r) {
            CHECK(hipEventRecord(a, st)); launch(); CHECK(hipEventRecord(b, st)); CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-88s %7.1f us  %5.2f TB/s\n", label, 1e3 * t[t.size() / 2], bytes / (1e-3 * t[t.size() / 2]) * 1e-12);
    };
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    printf("# %u records of 256 B in, 56 B out per record (%.0f MB per launch)\n", N, bytes * 1e-6);
    for (int lds : {0, 5120, 17408}) {
        char l[160];
        snprintf(l, sizeof l, "one-wave workgroups, one per 64 records, storage order, %5d B LDS", lds);
        time(l, [&] { hipLaunchKernelGGL((stream_kernel<64, false>), dim3(n_waves), dim3(64), lds, st, recs, out, out2, n_waves, (const unsigned*)nullptr); });
        snprintf(l, sizeof l, "one-wave workgroups, one per 64 records, shuffled order, %5d B LDS", lds);
        time(l, [&] { hipLaunchKernelGGL((stream_kernel<64, false>), dim3(n_waves), dim3(64), lds, st, recs, out, out2, n_waves, (const unsigned*)perm); });
    }
    time("four-wave workgroups, one per 256 records, storage order", [&] { hipLaunchKernelGGL((stream_kernel<256, false>), dim3(n_waves / 4), dim3(256), 0, st, recs, out, out2, n_waves, (const unsigned*)nullptr); });
    time("four-wave workgroups, one per 256 records, shuffled order", [&] { hipLaunchKernelGGL((stream_kernel<256, false>), dim3(n_waves / 4), dim3(256), 0, st, recs, out, out2, n_waves, (const unsigned*)perm); });
    for (int g : {1024, 2048, 4096, 8192}) {
        char l[160];
        snprintf(l, sizeof l, "%d persistent one-wave workgroups, storage order", g);
        time(l, [&] { hipLaunchKernelGGL((stream_kernel<64, true>), dim3(g), dim3(64), 0, st, recs, out, out2, n_waves, (const unsigned*)nullptr); });
        snprintf(l, sizeof l, "%d persistent one-wave workgroups, shuffled order", g);
        time(l, [&] { hipLaunchKernelGGL((stream_kernel<64, true>), dim3(g), dim3(64), 0, st, recs, out, out2, n_waves, (const unsigned*)perm); });
    }
    for (int g : {512, 1024, 2048}) {
        char l[160];
        snprintf(l, sizeof l, "%d persistent four-wave workgroups, storage order", g);
        time(l, [&] { hipLaunchKernelGGL((stream_kernel<256, true>), dim3(g), dim3(256), 0, st, recs, out, out2, n_waves, (const unsigned*)nullptr); });
    }
    {
        const size_t big = (size_t)1 << 30;      // 1 GiB: beyond the 256 MB of MALL
        float4 *src, *dst;
        CHECK(hipMalloc(&src, big)); CHECK(hipMalloc(&dst, big));
        CHECK(hipMemset(src, 0, big)); CHECK(hipMemset(dst, 0, big));
        const size_t n4 = big / 16;
        auto timeb = [&](const char* label, double moved, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            std::vector<float> t;
            for (int r = 0; r < 15; ++r) {
                CHECK(hipEventRecord(a, st)); launch(); CHECK(hipEventRecord(b, st)); CHECK(hipStreamSynchronize(st));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("%-88s %7.1f us  %5.2f TB/s\n", label, 1e3 * t[t.size() / 2], moved / (1e-3 * t[t.size() / 2]) * 1e-12);
        };
        for (int g : {2048, 8192, 32768}) {
            char l[160];
            snprintf(l, sizeof l, "plain read of 1 GiB, %d workgroups x 256, 4 float4 in flight per thread", g);
            timeb(l, (double)big, [&] { hipLaunchKernelGGL((plain_kernel<4, false>), dim3(g), dim3(256), 0, st, (const float4*)src, dst, n4, dst); });
            snprintf(l, sizeof l, "plain read of 1 GiB, %d workgroups x 256, 8 float4 in flight per thread", g);
            timeb(l, (double)big, [&] { hipLaunchKernelGGL((plain_kernel<8, false>), dim3(g), dim3(256), 0, st, (const float4*)src, dst, n4, dst); });
            snprintf(l, sizeof l, "plain copy of 1 GiB (2 GiB moved), %d workgroups x 256, 4 float4 in flight per thread", g);
            timeb(l, 2.0 * big, [&] { hipLaunchKernelGGL((plain_kernel<4, true>), dim3(g), dim3(256), 0, st, (const float4*)src, dst, n4, dst); });
        }
        timeb("hipMemcpyAsync device to device, 1 GiB (2 GiB moved)", 2.0 * big, [&] { CHECK(hipMemcpyAsync(dst, src, big, hipMemcpyDeviceToDevice, st)); });
        const size_t small = (size_t)128 << 20;   // 128 MiB: inside MALL
        timeb("plain read of 128 MiB (fits the MALL), 8192 workgroups, 8 in flight", (double)small, [&] { hipLaunchKernelGGL((plain_kernel<8, false>), dim3(8192), dim3(256), 0, st, (const float4*)src, dst, small / 16, dst); });
    }
    return 0;
}
