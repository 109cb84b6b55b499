// msplat_points.hip.h -- point-cloud renderer (shader/point_*.glsl): sprite projection and draw-order compositor
// (one of the parts of msplat_kernels.hip.h)
#pragma once

#include "msplat_common.hip.h"
#include "msplat_project.hip.h"
#include "msplat_composite.hip.h"

#pragma clang fp contract(off)

namespace msplat {

// ------------------------------------------------------------------------------------------
// point-cloud renderer: PointRenderer::Render (pointrenderer.cpp:113-196) after the shared
// presort + sort.  point_vert.glsl: clip = proj * view * position.  point_geom.glsl:22-46: a quad of
// +-(pointSize * invAspectRatio, pointSize) added IN CLIP SPACE (so it shrinks with 1/w), uv 0..1 across it.
// point_frag.glsl:20-25: out = (a * rgb * tex.rgb, a * tex.a), blended GL_ONE / GL_ONE_MINUS_SRC_ALPHA in
// draw order (far to near).  Texture: LinearMipmapLinear / Linear / ClampToEdge (pointrenderer.cpp:62-63).
// ------------------------------------------------------------------------------------------
struct SpriteParams {
    int w, h, levels;
    uint32_t off[14];          // texel offset of every mip level inside the float4 chain
};

constexpr float kPointSize = 0.02f;       // pointrenderer.cpp:176 ("in ndc space?!?": it is clip space)

__global__ __launch_bounds__(kProjThreads) void point_project_kernel(const uint32_t* __restrict__ sorted_idx,
                                                                     const uint32_t* __restrict__ d_V,
                                                                     const float4* __restrict__ pos4,
                                                                     const float4* __restrict__ colors,
                                                                     FrameParams fp, SpriteParams sp,
                                                                     float4* __restrict__ out_rec,
                                                                     uint32_t* __restrict__ out_rect,
                                                                     uint32_t* __restrict__ out_zq)
{
    const uint32_t V = *d_V;
    const uint32_t r = blockIdx.x * kProjThreads + threadIdx.x;
    if (r >= V) return;
    const uint32_t i = sorted_idx[r];
    const float4 P = pos4[i];
    const float4 col = colors[i];
    const float* vm = fp.view;
    const float* pm = fp.proj;
    float t[4], p4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        t[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(vm[c], P.x), __fmul_rn(vm[4 + c], P.y)), __fmul_rn(vm[8 + c], P.z)), vm[12 + c]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        p4[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pm[c], t[0]), __fmul_rn(pm[4 + c], t[1])), __fmul_rn(pm[8 + c], t[2])), __fmul_rn(pm[12 + c], t[3]));
    const float w = p4[3];
    // all four vertices share z and w: the near/far clip keeps or drops the whole quad
    bool reject = !(w > 0.0f) || !(p4[2] >= -w) || !(p4[2] <= w);
    const float ndcx = __fdiv_rn(p4[0], w), ndcy = __fdiv_rn(p4[1], w), ndcz = __fdiv_rn(p4[2], w);
    const float WIDTH = fp.W, HEIGHT = fp.H;
    // GL viewport transform with the viewport origin at the image origin
    const float cx = __fmul_rn(__fadd_rn(ndcx, 1.0f), __fmul_rn(0.5f, WIDTH));
    const float cy = __fmul_rn(__fadd_rn(ndcy, 1.0f), __fmul_rn(0.5f, HEIGHT));
    const float invAspect = __fdiv_rn(1.0f, __fdiv_rn(WIDTH, HEIGHT));        // pointrenderer.cpp:170-177
    const float hx = __fmul_rn(__fdiv_rn(__fmul_rn(kPointSize, invAspect), w), __fmul_rn(0.5f, WIDTH));
    const float hy = __fmul_rn(__fdiv_rn(kPointSize, w), __fmul_rn(0.5f, HEIGHT));
    if (!(hx > 0.0f) || !(hy > 0.0f) || !(cx == cx) || !(cy == cy)) reject = true;
    // isotropic level of detail: texels per pixel along the denser axis
    const float rho = fmaxf(__fdiv_rn((float)sp.w, __fmul_rn(2.0f, hx)), __fdiv_rn((float)sp.h, __fmul_rn(2.0f, hy)));
    const float lambda = log2f(rho);
    uint32_t rect = kRectEmpty;
    if (!reject) {
        // pixels whose centre lies in [c - h, c + h)
        float x0f = ceilf(cx - hx - 0.5f), x1f = ceilf(cx + hx - 0.5f) - 1.0f;
        float y0f = ceilf(cy - hy - 0.5f), y1f = ceilf(cy + hy - 0.5f) - 1.0f;
        x0f = fmaxf(x0f, 0.0f);
        y0f = fmaxf(y0f, 0.0f);
        x1f = fminf(x1f, (float)(fp.width - 1));
        y1f = fminf(y1f, (float)(fp.height - 1));
        if (x0f <= x1f && y0f <= y1f) {
            const int tx0 = (int)x0f / kBin, tx1 = (int)x1f / kBin;
            int ty0 = (int)y0f / kBin, ty1 = (int)y1f / kBin;
            if (fp.banded) {
                const int v0 = band_first_owned_from(fp, ty0), v1 = min(band_last_owned_upto(fp, ty1), fp.tiles_y - 1);
                ty0 = v0;
                ty1 = v1;
            }
            if (ty0 <= ty1) rect = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
        }
    }
    out_rec[(size_t)r * 3 + 0] = make_float4(cx, cy, hx, hy);
    out_rec[(size_t)r * 3 + 1] = col;
    out_rec[(size_t)r * 3 + 2] = make_float4(lambda, 0.0f, 0.0f, 0.0f);
    out_rect[r] = rect;
    if (out_zq != nullptr) out_zq[r] = quantise_depth(ndcz, fp.depth_bits);
}

// bilinear tap of one mip level, ClampToEdge
__device__ __forceinline__ float4 sprite_tap(const float4* __restrict__ tex, uint32_t off, int sw, int sh, float u, float v)
{
    const float x = u * (float)sw - 0.5f, y = v * (float)sh - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float ax = x - xf, ay = y - yf;
    const int i0 = min(max((int)xf, 0), sw - 1), i1 = min(max((int)xf + 1, 0), sw - 1);
    const int j0 = min(max((int)yf, 0), sh - 1), j1 = min(max((int)yf + 1, 0), sh - 1);
    const float4 t00 = tex[off + j0 * sw + i0], t10 = tex[off + j0 * sw + i1];
    const float4 t01 = tex[off + j1 * sw + i0], t11 = tex[off + j1 * sw + i1];
    const float bx = 1.0f - ax, by = 1.0f - ay;
    float4 o;
    o.x = (t00.x * bx + t10.x * ax) * by + (t01.x * bx + t11.x * ax) * ay;
    o.y = (t00.y * bx + t10.y * ax) * by + (t01.y * bx + t11.y * ax) * ay;
    o.z = (t00.z * bx + t10.z * ax) * by + (t01.z * bx + t11.z * ax) * ay;
    o.w = (t00.w * bx + t10.w * ax) * by + (t01.w * bx + t11.w * ax) * ay;
    return o;
}

__device__ __forceinline__ float4 sprite_sample(const float4* __restrict__ tex, const SpriteParams& sp, float u, float v,
                                                float lambda)
{
    if (!(lambda > 0.0f)) return sprite_tap(tex, sp.off[0], sp.w, sp.h, u, v);       // magnification: Linear
    const float lf = floorf(lambda);
    const int l0 = min((int)lf, sp.levels - 1), l1 = min(l0 + 1, sp.levels - 1);
    const float4 a = sprite_tap(tex, sp.off[l0], max(sp.w >> l0, 1), max(sp.h >> l0, 1), u, v);
    if (l1 == l0) return a;
    const float4 b = sprite_tap(tex, sp.off[l1], max(sp.w >> l1, 1), max(sp.h >> l1, 1), u, v);
    const float f = lambda - lf, g = 1.0f - f;
    return make_float4(a.x * g + b.x * f, a.y * g + b.y * f, a.z * g + b.z * f, a.w * g + b.w * f);
}

// draw-order walk like composite_depth_kernel (optional emulated depth test: zq == nullptr -> colour only)
template <bool HALF>
__global__ __launch_bounds__(kCompThreads) void composite_points_kernel(const uint32_t* __restrict__ tile_start,
                                                                        const uint32_t* __restrict__ pairs,
                                                                        const float4* __restrict__ rec,
                                                                        const uint32_t* __restrict__ zq,
                                                                        const float4* __restrict__ tex, SpriteParams sp,
                                                                        void* __restrict__ out, size_t pitch_bytes,
                                                                        FrameParams fp, uint32_t cap,
                                                                        const uint32_t* __restrict__ order,
                                                                        uint32_t* __restrict__ queue, uint32_t ntiles)
{
    __shared__ float4 s_rec[kCompThreads * 3];
    __shared__ uint32_t s_z[kCompThreads];
    const int lane = threadIdx.x;
    const int lx = lane & 15, ly = lane >> 4;
    for (uint32_t qpos = blockIdx.x; qpos < ntiles;) {
        const int bin = (int)order[qpos >> 2];
        const int quad = (int)(qpos & 3u);
        const int bvy = bin / fp.tiles_x;
        const int tx = (bin - bvy * fp.tiles_x) * 2 + (quad & 1);
        const int ty = band_real_row(fp, bvy) * 2 + (quad >> 1);
        if (tx * kTile < fp.width && ty * kTile < fp.height) {
            const int x = tx * kTile + lx, ybase = ty * kTile + ly;
            const float fx = (float)x + 0.5f;
            uint32_t start = tile_start[bin], end = tile_start[bin + 1];
            if (start > cap) start = cap;
            if (end > cap) end = cap;
            float cr[4], cg[4], cb[4];
            uint32_t zbuf[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { cr[k] = 0.0f; cg[k] = 0.0f; cb[k] = 0.0f; zbuf[k] = 0xFFFFFFFFu; }
            const float X0 = (float)(tx * kTile) + 0.5f, X1 = X0 + (float)(kTile - 1);
            const float Y0 = (float)(ty * kTile) + 0.5f, Y1 = Y0 + (float)(kTile - 1);
            for (uint32_t base = start; base < end; base += kCompThreads) {
                const uint32_t cnt = min((uint32_t)kCompThreads, end - base);
                float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
                uint32_t z = 0;
                bool rel = false;
                if (lane < (int)cnt) {
                    uint32_t rank = pairs[base + lane] & kRankMask;
                    asm volatile("" : "+v"(rank));          // see composite_depth_kernel (hipcc mask/mad folding)
                    const float4* src = rec + (size_t)rank * 3;
                    p0 = src[0]; p1 = src[1]; p2 = src[2];
                    if (zq != nullptr) z = zq[rank];
                    // some pixel centre of the tile inside [c - h, c + h) on both axes
                    rel = (p0.x - p0.z <= X1) && (p0.x + p0.z > X0) && (p0.y - p0.w <= Y1) && (p0.y + p0.w > Y0);
                }
                const uint64_t relmask = __ballot(rel);
                const uint32_t n = (uint32_t)__popcll(relmask);
                if (rel) {
                    const int slot = __popcll(relmask & ((1ull << lane) - 1ull));
                    s_rec[slot * 3 + 0] = p0;
                    s_rec[slot * 3 + 1] = p1;
                    s_rec[slot * 3 + 2] = p2;
                    s_z[slot] = z;
                }
                __syncthreads();
                for (uint32_t j = 0; j < n; ++j) {
                    const float4 q = s_rec[j * 3 + 0];      // cx, cy, hx, hy
                    const float4 col = s_rec[j * 3 + 1];
                    const float lambda = s_rec[j * 3 + 2].x;
                    const uint32_t zj = s_z[j];
                    const float xlo = q.x - q.z, xhi = q.x + q.z;
                    if (!(fx >= xlo && fx < xhi)) continue;
                    const float u = (fx - xlo) / (2.0f * q.z);
                    const float ylo = q.y - q.w, yhi = q.y + q.w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float fy = (float)(ybase + 4 * k) + 0.5f;
                        if (fy >= ylo && fy < yhi && (zq == nullptr || zj < zbuf[k])) {
                            const float v = (fy - ylo) / (2.0f * q.w);
                            const float4 tx4 = sprite_sample(tex, sp, u, v, lambda);
                            const float sa = col.w * tx4.w;                  // point_frag.glsl:24
                            const float oma = 1.0f - sa;
                            cr[k] = ((col.w * col.x) * tx4.x) + oma * cr[k];     // point_frag.glsl:23, GL_ONE / 1 - src.a
                            cg[k] = ((col.w * col.y) * tx4.y) + oma * cg[k];
                            cb[k] = ((col.w * col.z) * tx4.z) + oma * cb[k];
                            zbuf[k] = zj;                                    // no discard in point_frag: always writes depth
                        }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (x < fp.width && ybase + 4 * k < fp.height) {
                    char* row = (char*)out + (size_t)(ybase + 4 * k) * pitch_bytes;
                    if (HALF) {
                        union { _Float16 h[4]; uint2 u; } pk;
                        pk.h[0] = (_Float16)cr[k]; pk.h[1] = (_Float16)cg[k]; pk.h[2] = (_Float16)cb[k]; pk.h[3] = (_Float16)1.0f;
                        ((uint2*)row)[x] = pk.u;
                    } else {
                        ((float4*)row)[x] = make_float4(cr[k], cg[k], cb[k], 1.0f);
                    }
                }
            }
        }
        uint32_t nq = 0;
        if (threadIdx.x == 0) nq = atomicAdd(queue, 1u);
        qpos = gridDim.x + __builtin_amdgcn_readfirstlane(nq);
    }
}

}  // namespace msplat
