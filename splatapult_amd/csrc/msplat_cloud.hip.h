// msplat_cloud.hip.h -- upload-time kernels: GPU PLY ingest (GaussianCloud::ImportPly's per-vertex math) and the spatial storage order
// (moments, Morton codes, gather, cull boxes)
// (one of the parts of msplat_kernels.hip.h)
#pragma once

#include "msplat_common.hip.h"

#pragma clang fp contract(off)

namespace msplat {

// ------------------------------------------------------------------------------------------
// GPU ingest: GaussianCloud::ImportPly's per-vertex lambda (gaussiancloud.cpp:254-361)
//   alpha = 1/(1+exp(-opacity)), scale = exp(log scale), Sigma = R S S^T R^T from the normalised quaternion,
//   SH repack -- written straight into the renderer's device layout (pos4 + padded records).
// One wave per 64 vertices: their bytes are contiguous in the PLY vertex block, so the wave copies the span
// with coalesced 16-byte loads into LDS and every lane then picks its properties out of its own vertex.
// Same operation order as the host code (splatapult_amd/host/gaussian_scene.cpp), contraction off.
// ------------------------------------------------------------------------------------------
struct PlyLayout {             // mirrors msplat_ply_layout (include/msplat.h)
    uint32_t vertex_size;
    int32_t x, y, z;
    int32_t f_dc[3];
    int32_t f_rest[45];
    int32_t opacity;
    int32_t scale[3];
    int32_t rot[4];
};

template <bool FULL_SH>
__global__ __launch_bounds__(64) void ingest_kernel(const char* __restrict__ raw, uint64_t n, PlyLayout L,
                                                    float4* __restrict__ pos4, float4* __restrict__ recs)
{
    extern __shared__ __attribute__((aligned(16))) char s_raw[];
    constexpr int F4 = FULL_SH ? 16 : 8;
    const int lane = threadIdx.x;
    const uint32_t vs = L.vertex_size;
    const uint64_t v0 = (uint64_t)blockIdx.x * 64u;
    const uint64_t byte0 = v0 * vs;
    const uint64_t total = n * (uint64_t)vs;
    const uint32_t span = (uint32_t)min((uint64_t)64u * vs, total - byte0);      // multiple of 4
    for (uint32_t off = lane * 16u; off < span; off += 64u * 16u) {
        if (off + 16u <= span) {
            *reinterpret_cast<float4*>(s_raw + off) = *reinterpret_cast<const float4*>(raw + byte0 + off);
        } else {
            for (uint32_t o = off; o < span; o += 4u)
                *reinterpret_cast<float*>(s_raw + o) = *reinterpret_cast<const float*>(raw + byte0 + o);
        }
    }
    __syncthreads();
    const uint64_t i = v0 + lane;
    if (i >= n) return;
    const char* v = s_raw + (size_t)lane * vs;
    auto rd = [&](int32_t off) -> float { return off >= 0 ? *reinterpret_cast<const float*>(v + off) : 0.0f; };

    float f[F4 * 4];
#pragma unroll
    for (int k = 0; k < F4 * 4; ++k) f[k] = 0.0f;
    f[0] = rd(L.x); f[1] = rd(L.y); f[2] = rd(L.z);
    f[3] = 1.0f / (1.0f + expf(-rd(L.opacity)));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f[4 + 4 * c] = rd(L.f_dc[c]);
        if constexpr (FULL_SH) {
#pragma unroll
            for (int k = 1; k < 4; ++k) f[4 + 4 * c + k] = rd(L.f_rest[c * 15 + k - 1]);
#pragma unroll
            for (int k = 4; k < 16; ++k) f[25 + 12 * c + (k - 4)] = rd(L.f_rest[c * 15 + k - 1]);
        }
    }
    const float s0 = expf(rd(L.scale[0])), s1 = expf(rd(L.scale[1])), s2 = expf(rd(L.scale[2]));
    float w = rd(L.rot[0]), x = rd(L.rot[1]), y = rd(L.rot[2]), z = rd(L.rot[3]);
    const float len = sqrtf((w * w + x * x) + (y * y + z * z));
    if (len <= 0.0f) { w = 1.0f; x = 0.0f; y = 0.0f; z = 0.0f; }
    else { const float inv = 1.0f / len; w *= inv; x *= inv; y *= inv; z *= inv; }
    const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    // R[c][r], column-major like the host code
    const float R[9] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz),        2.0f * (xz - wy),
                        2.0f * (xy - wz),        1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx),
                        2.0f * (xz + wy),        2.0f * (yz - wx),        1.0f - 2.0f * (xx + yy)};
    const float sc[3] = {s0, s1, s2};
    float B[9];      // (R S) S^T : column c scaled by s_c twice (the zero terms of the 3x3 products add exactly)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) B[c * 3 + r] = (R[c * 3 + r] * sc[c]) * sc[c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // V[c][r] = B[0][r]*Rt[c][0] + B[1][r]*Rt[c][1] + B[2][r]*Rt[c][2],  Rt[c][k] = R[k][c]
            float s = B[0 * 3 + r] * R[0 * 3 + c];
            s = s + B[1 * 3 + r] * R[1 * 3 + c];
            s = s + B[2 * 3 + r] * R[2 * 3 + c];
            f[16 + c * 3 + r] = s;
        }
    pos4[i] = make_float4(f[0], f[1], f[2], footprint_bound(&f[16], f[3]));      // .w: world-space footprint bound for the band cull
#pragma unroll
    for (int k = 0; k < F4; ++k) recs[i * F4 + k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
}

// ------------------------------------------------------------------------------------------
// Spatial storage order (round 4; see box_live above).  Upload-time only: the moments of the positions and of the footprint
// bounds, a 32-bit code per splat (2 bits of size class above a 30-bit Morton code: 10 bits per axis over mean +- 3 sigma,
// outliers clamped to the border cells), a stable sort of the codes with the 8-bit radix passes above (ties keep upload
// order), a gather of the cloud into that order and one bounding box per kBoxSplats stored splats.  Draw order is by depth key,
// ties by STORAGE order.
// ------------------------------------------------------------------------------------------
// (two stages with a fixed summation order and no atomics: every device of a group, and every run, must arrive at the same
//  storage order bit for bit -- tie order is part of the frame)
constexpr int kMoments = 10;
__global__ __launch_bounds__(kThreads) void cloud_moments_kernel(const float4* __restrict__ pos, uint32_t n,
                                                                 double* __restrict__ part /* [gridDim.x][kMoments] */)
{
    __shared__ double s_w[kThreads / 64][kMoments];
    // sum xyz, sum of squares xyz, count of finite positions; sum, sum of squares, count of log2(footprint bound) where it is > 0
    double s[kMoments] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const float4 p = pos[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            s[0] += p.x; s[1] += p.y; s[2] += p.z;
            s[3] += (double)p.x * p.x; s[4] += (double)p.y * p.y; s[5] += (double)p.z * p.z;
            s[6] += 1.0;
            if (p.w > 0.0f && isfinite(p.w)) {
                const double l = (double)log2f(p.w);
                s[7] += l; s[8] += l * l; s[9] += 1.0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kMoments; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[k] += __shfl_xor(s[k], d, 64);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < kMoments) {
        double t = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) t += s_w[w][threadIdx.x];
        part[(size_t)blockIdx.x * kMoments + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(kThreads) void cloud_moments_finish(const double* __restrict__ part, uint32_t rows,
                                                                 double* __restrict__ acc /* kMoments */)
{
    __shared__ double s_t[kThreads][kMoments];
    double s[kMoments] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t r = threadIdx.x; r < rows; r += kThreads)
#pragma unroll
        for (int k = 0; k < kMoments; ++k) s[k] += part[(size_t)r * kMoments + k];
#pragma unroll
    for (int k = 0; k < kMoments; ++k) s_t[threadIdx.x][k] = s[k];
    __syncthreads();
    for (int half = kThreads / 2; half >= 1; half >>= 1) {
        if ((int)threadIdx.x < half)
#pragma unroll
            for (int k = 0; k < kMoments; ++k) s_t[threadIdx.x][k] += s_t[threadIdx.x + half][k];
        __syncthreads();
    }
    if (threadIdx.x < kMoments) acc[threadIdx.x] = s_t[0][threadIdx.x];
}

__device__ __forceinline__ uint32_t morton_spread10(uint32_t v)      // 10 bits -> every third bit
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(kThreads) void morton_kernel(const float4* __restrict__ pos, uint32_t n,
                                                          const double* __restrict__ acc, uint32_t* __restrict__ code,
                                                          uint32_t* __restrict__ index)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const double cnt = acc[6] > 0.0 ? acc[6] : 1.0;
    float q[3];
    const float4 p = pos[i];
    const float c[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double mean = acc[k] / cnt;
        const double var = acc[3 + k] / cnt - mean * mean;
        const float sd = (float)sqrt(var > 1e-30 ? var : 1e-30);
        const float t = (c[k] - (float)mean) / (6.0f * sd) + 0.5f;                // mean +- 3 sigma -> [0, 1]
        q[k] = isfinite(t) ? fminf(fmaxf(t, 0.0f), 1.0f) * 1023.0f : 0.0f;
    }
    // Size class in the two top bits: a box's reach on screen is its extent plus its LARGEST footprint, and the largest of
    // 256 log-normal sizes is several times the typical one -- splats are therefore grouped by footprint bound first (z = deviation
    // of log2(bound) from its mean in sigmas: <= 0.5 | <= 1.25 | <= 2 | the rest, ~69 / 20 / 9 / 2 %), by position inside a class.
    uint32_t cls = 3u;
    if (p.w > 0.0f && isfinite(p.w) && acc[9] > 0.0) {
        const double lm = acc[7] / acc[9], lv = acc[8] / acc[9] - lm * lm;
        const float z = (log2f(p.w) - (float)lm) / (float)sqrt(lv > 1e-12 ? lv : 1e-12);
        cls = z <= 0.5f ? 0u : (z <= 1.25f ? 1u : (z <= 2.0f ? 2u : 3u));
    }
    code[i] = (cls << 30) | morton_spread10((uint32_t)q[0]) | (morton_spread10((uint32_t)q[1]) << 1) | (morton_spread10((uint32_t)q[2]) << 2);
    index[i] = i;
}

// stored slot j <- uploaded splat order[j]: one wave moves 64 / F4 records per step with coalesced 16-byte accesses
__global__ __launch_bounds__(kThreads) void gather_cloud_kernel(const uint32_t* __restrict__ order, uint32_t n, int F4,
                                                                const float4* __restrict__ pos_in,
                                                                const float4* __restrict__ recs_in,
                                                                float4* __restrict__ pos_out, float4* __restrict__ recs_out)
{
    const uint64_t total = (uint64_t)n * (uint32_t)F4;
    for (uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (uint64_t)gridDim.x * kThreads) {
        const uint32_t j = (uint32_t)(e / (uint32_t)F4), sub = (uint32_t)(e - (uint64_t)j * (uint32_t)F4);
        const uint32_t src = order[j];
        recs_out[e] = recs_in[(size_t)src * F4 + sub];
        if (sub == 0u) pos_out[j] = pos_in[src];
    }
}

// one workgroup per box of kBoxSplats stored splats
__global__ __launch_bounds__(kThreads) void cull_boxes_kernel(const float4* __restrict__ pos, uint32_t n,
                                                              CullBox* __restrict__ boxes)
{
    __shared__ float s_red[7][kThreads / 64];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, wmax = 0.0f;
    const uint32_t base = blockIdx.x * kBoxSplats;
    for (uint32_t k = threadIdx.x; k < (uint32_t)kBoxSplats && base + k < n; k += kThreads) {
        const float4 p = pos[base + k];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
            if (p.w > wmax) wmax = p.w;                                     // (NaN never wins; inf does, and then nothing is band-culled)
        }
    }
    float r[7] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], wmax};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float o = __shfl_xor(r[k], d, 64);
            r[k] = k < 3 ? fminf(r[k], o) : fmaxf(r[k], o);
        }
        if ((threadIdx.x & 63) == 0) s_red[k][threadIdx.x >> 6] = r[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
            for (int w = 1; w < kThreads / 64; ++w) r[k] = k < 3 ? fminf(r[k], s_red[k][w]) : fmaxf(r[k], s_red[k][w]);
        CullBox b;
        b.lo = make_float4(r[0], r[1], r[2], r[6]);
        b.hi = make_float4(r[3], r[4], r[5], 0.0f);
        boxes[blockIdx.x] = b;
    }
}

}  // namespace msplat
