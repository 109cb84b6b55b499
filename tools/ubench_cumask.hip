// ubench_cumask.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask get on an MI355X (8 XCDs x 32 CUs)? (r6)
// Every workgroup records its XCC_ID and HW_ID; per mask pattern: workgroups seen per XCD and distinct (xcc, se, cu) ids.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_cumask.hip -o tools/bin/ubench_cumask && tools/bin/ubench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void where_kernel(unsigned* out)
{
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(8);                  // 3 us: the grid spreads over everything it may use
}

int main()
{
    const int WGS = 16384;
    unsigned* out; CHECK(hipMalloc(&out, WGS * 8));
    std::vector<unsigned> h(2 * WGS);
    auto probe = [&](const char* label, const std::vector<unsigned>& mask) {
        hipStream_t s;
        if (mask.empty()) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        else CHECK(hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()));
        hipLaunchKernelGGL(where_kernel, dim3(WGS), dim3(64), 0, s, out);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemcpy(h.data(), out, WGS * 8, hipMemcpyDeviceToHost));
        int per_xcc[16] = {0};
        std::set<unsigned> cus;
        for (int i = 0; i < WGS; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15u;
            per_xcc[xcc]++;
            cus.insert((xcc << 16) | (hw & 0xFF00u) | ((hw >> 13) & 7u) << 4 | 0);       // (xcc, cu_id [11:8], sh [12], se [15:13])
            (void)hw;
        }
        printf("%-44s distinct (xcc, se, sh, cu): %3zu   workgroups per XCD:", label, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %5d", per_xcc[x]);
        printf("\n");
        CHECK(hipStreamDestroy(s));
    };
    auto bits = [](auto pred) { std::vector<unsigned> m(8, 0u); for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    probe("no mask", {});
    probe("all 256 bits", bits([](int) { return true; }));
    probe("bits 0..63", bits([](int i) { return i < 64; }));
    probe("bits 64..127", bits([](int i) { return i >= 64 && i < 128; }));
    probe("bits 0..127", bits([](int i) { return i < 128; }));
    probe("bits with i % 8 < 2", bits([](int i) { return i % 8 < 2; }));
    probe("bits with i % 8 == 0", bits([](int i) { return i % 8 == 0; }));
    probe("bits with i % 4 == 0", bits([](int i) { return i % 4 == 0; }));
    probe("bits with (i / 8) % 4 == 0", bits([](int i) { return (i / 8) % 4 == 0; }));
    probe("bits with (i / 32) % 4 == 0", bits([](int i) { return (i / 32) % 4 == 0; }));
    probe("bits with i % 2 == 0", bits([](int i) { return i % 2 == 0; }));
    return 0;
}
