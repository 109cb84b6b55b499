// Layout probe for v_mfma_f32_4x4x1_16b_f32 (r4 compositor experiment): prints, for a few lanes, which lanes' A and B values
// end up in the four result registers.  A(lane) = lane, B(lane) = 1000 * lane + 1:  D[r] = A(src_a) * B(src_b).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma4x4.hip -o /tmp/ubench_mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(float* out)
{
    const int lane = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.0f, c, 0, 0, 0);        // D[r] = A(src lane) -> which lane?
    v4f e = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(lane + 1), c, 0, 0, 0);        // D[r] = B(src lane)
    for (int r = 0; r < 4; ++r) { out[lane * 8 + r] = d[r]; out[lane * 8 + 4 + r] = e[r]; }
}
int main()
{
    float* d; float h[64 * 8];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int a_src = (int)h[l * 8 + r] - 1, b_src = (int)h[l * 8 + 4 + r] - 1;
            if (a_src != (l / 4) * 4 + r || b_src != l) ok = 0;
            if (l < 6 || l == 63) printf("lane %2d reg %d: A from lane %2d, B from lane %2d\n", l, r, a_src, b_src);
        }
    printf("layout assumed by composite_kernel<., true> (A from lane 4 * (l / 4) + r, B from the lane itself): %s\n", ok ? "CONFIRMED" : "WRONG");
    return ok ? 0 : 1;
}
