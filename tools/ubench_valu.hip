// ubench_valu.hip -- issue-rate microbenchmark for the VALU instructions of the compositor's inner loop
// (MI355X / gfx950).  Prints SIMD-level throughput in shader cycles per wave64 instruction for 1, 2, 4, 8
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_valu tools/ubench_valu.hip
// Used to decide between packed (v_pk_*_f32, 2 pixels per lane) and plain (1 pixel per lane) compositor
// layouts -- DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

enum { K_FMA, K_PKFMA, K_EXP, K_CMPCND, K_MUL, K_PKMUL, K_PKADD, K_FMA_S, K_LDSB128, K_RCP, K_MAX, K_CND,
       K_MIX, K_LDSB32, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cmp_gt+v_cndmask", "v_mul_f32",
                                      "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f32 (sgpr src)", "ds_read_b128 bcast",
                                      "v_rcp_f32", "v_max_f32", "v_cndmask_b32", "inner-loop mix (36)", "ds_read_b32 bcast"};
static const int kInstrPerIter[K_COUNT] = {64, 64, 64, 128, 64, 64, 64, 64, 16, 64, 64, 64, 36 * 4, 16};

template <int KIND>
__global__ __launch_bounds__(256) void bench(float* out, unsigned long long* cyc, int iters, float seed)
{
    __shared__ float4 s_lds[64];
    if (threadIdx.x < 64) s_lds[threadIdx.x] = make_float4(seed, seed * 0.5f, 1.0f, 0.25f);
    __syncthreads();
    float a[8];
    v2f p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = seed + (float)k * 0.001f + threadIdx.x * 1e-6f; p[k] = (v2f){a[k], a[k] + 0.5f}; }
    const float m = 0.999f, c = 1e-6f;
    const v2f pm = (v2f){m, m}, pc = (v2f){c, c};
    float sm;
    asm volatile("s_mov_b32 %0, 0x3f7fbe77" : "=s"(sm));
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));)
        } else if constexpr (KIND == K_PKFMA) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pc));)
        } else if constexpr (KIND == K_EXP) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));)
        } else if constexpr (KIND == K_CMPCND) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc" : "+v"(a[k]) : "v"(c), "v"(m) : "vcc");)
        } else if constexpr (KIND == K_MUL) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));)
        } else if constexpr (KIND == K_PKMUL) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm));)
        } else if constexpr (KIND == K_PKADD) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));)
        } else if constexpr (KIND == K_FMA_S) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(sm), "v"(c));)
        } else if constexpr (KIND == K_LDSB128) {
            // 16 broadcast 16-byte reads (all lanes the same address), consumed so they cannot be dropped
            float4 r[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = s_lds[(it + k + h) & 63];
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += r[k].x;
            }
        } else if constexpr (KIND == K_LDSB32) {
            float r[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = s_lds[(it + k + h) & 63].y;
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += r[k];
            }
        } else if constexpr (KIND == K_RCP) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));)
        } else if constexpr (KIND == K_MAX) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));)
        } else if constexpr (KIND == K_CND) {
            REP8(_Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[k]) : "v"(m) : "vcc");)
        } else if constexpr (KIND == K_MIX) {
            // the compositor's per-record instruction mix for 4 pixels per lane (two packed strip pairs):
            // 3 plain + per pair {8 packed, 2 cmp, 2 cndmask, 2 exp}; four records per iteration
#pragma unroll
            for (int rcd = 0; rcd < 4; ++rcd) {
                asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %2, %2, %0\n v_fma_f32 %3, %2, %0, %1"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f e = p[h], w;
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(e) : "v"(pc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(e) : "v"(pm), "v"(pc));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(e) : "v"(pm), "v"(pc));
                    float ex = e.x, ey = e.y;
                    asm volatile("v_exp_f32 %0, %0" : "+v"(ex));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(ey));
                    asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc" : "+v"(ex) : "v"(c), "v"(m) : "vcc");
                    asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc" : "+v"(ey) : "v"(c), "v"(m) : "vcc");
                    w = (v2f){ex, ey};
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w) : "v"(p[4 + h]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 + h]) : "v"(w), "v"(pm));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[6 + h]) : "v"(w), "v"(pm));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[h]) : "v"(w), "v"(pc));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[4 + h]) : "v"(w));
                }
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(float* d_out, unsigned long long* d_cyc, int iters)
{
    printf("%-24s", kNames[KIND]);
    for (int wps : {1, 2, 4, 8}) {
        // 256 CUs, wps waves per SIMD: blocks of 256 threads = 1 wave per SIMD; wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 16, 1.0f);     // warm-up
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= h.size();
        const double n_instr = (double)iters * kInstrPerIter[KIND];
        // per-wave cycles per instruction, and SIMD-level throughput (cycles per instruction when wps waves share it)
        printf("  wps%-1d: %6.2f/wave %5.2f/simd (%.0f us)", wps, mean / n_instr, mean / n_instr / wps, ms * 1e3);
    }
    printf("\n");
}

int main()
{
    float* d_out; unsigned long long* d_cyc;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    (void)hipMalloc(&d_cyc, 256 * 8 * 4 * 8);
    const int iters = 2000;
    printf("shader cycles (clock64) per wave64 instruction; /simd = per-wave value / waves per SIMD (throughput)\n");
    run<K_FMA>(d_out, d_cyc, iters);
    run<K_PKFMA>(d_out, d_cyc, iters);
    run<K_MUL>(d_out, d_cyc, iters);
    run<K_PKMUL>(d_out, d_cyc, iters);
    run<K_PKADD>(d_out, d_cyc, iters);
    run<K_FMA_S>(d_out, d_cyc, iters);
    run<K_EXP>(d_out, d_cyc, iters);
    run<K_RCP>(d_out, d_cyc, iters);
    run<K_CMPCND>(d_out, d_cyc, iters);
    run<K_CND>(d_out, d_cyc, iters);
    run<K_MAX>(d_out, d_cyc, iters);
    run<K_LDSB128>(d_out, d_cyc, iters);
    run<K_LDSB32>(d_out, d_cyc, iters);
    run<K_MIX>(d_out, d_cyc, iters);
    return 0;
}
