// msplat_occlusion.hip.h -- a frame in two passes with occlusion feedback (round 4).
// (one of the parts of msplat_kernels.hip.h)
//
// The compositor walks every bin's list nearest-first and stops when the pixels are saturated; the stages in front of it
// do not know that and project / bin every visible splat: of the (splat, bin) pairs they write, the walk reads 24 % at BASELINE
// config 2, 13 % at config 4 and 2 % on a scene-like cloud (tools/occlusion_potential.py).  The two-pass frame uses the
// compositor's own verdict, and its pixels are BIT FOR BIT those of the single pass:
//   pass 1   the nearest R1 splats (ranks >= cut = V - R1; the reference's draw order is far to near, the lists are walked from
//            their end): projected, binned, composited in WHOLE batches of 64 entries.  Batches count from the list's end and
//            this list is a suffix of the complete one, so every batch is a batch of the single pass.  A tile saturated after one
//            of them is final; any other stops in front of its first incomplete batch and leaves its accumulators (r, g, b, T per
//            pixel, fp32) and the number of entries it composited.
//   gate     the unfinished bins form a summed-area table (any block of bins is tested with four reads).  A splat of pass 1
//            keeps its rectangle only if it touches an unfinished bin; a splat behind the cut is tested BEFORE its 256-byte
//            record is fetched -- first its bounding box of 256 stored splats (spatially ordered clouds), then a conservative
//            screen box from its centre and its footprint bound (pos4.w, the bound of the band-restricted cull) -- and is
//            projected only if that box touches one.
//   pass 2   the surviving rectangles of ALL ranks are binned again: an unfinished bin gets its complete list (a splat that
//            touches it passes the gate by construction), whose last entries are the ones pass 1 composited: its unfinished
//            tiles RESUME behind them -- at a batch boundary of the single pass; finished tiles are skipped.
// Any R1 is correct; it only decides how much work is left.  The host steers it from the pair counts of an earlier frame
// (host-mapped words, never waited for).  Contract: src/splatrenderer.cpp:315-343 + shader/splat_*.glsl + the blend state of
// src/app.cpp:144-164 -- the image of Render() is unchanged.
#pragma once

#include "msplat_common.hip.h"

#pragma clang fp contract(off)

namespace msplat {

// Summed-area table of the unfinished bins: sat[(row + 1) * (tiles_x + 1) + col + 1] = unfinished bins in rows [0, row] x columns
// [0, col] (virtual rows in band mode, like the rectangles); row 0 and column 0 are zero.  uint16: at most kOccSatMax entries.
constexpr int kOccSatMax = 24576;             // (tiles_x + 1) (tiles_y + 1): 4096 x 4096 pixels need 129 x 129

// occ[0] = cut (first rank of pass 1, a multiple of 1024 = kBinChunk), occ[1] = splats behind the cut that pass the gate, occ[2] =
// unfinished bins, occ[4] = ranks the second binning chain walks (occ_mask_kernel).  The cut is a pure function of V and the
// share (occ_cut, msplat_common.hip.h), evaluated by project_kernel's first pass itself (its first workgroup also publishes it
// and clears occ[1], occ[2]).

// ONE workgroup: fin[bin * 4 + quadrant] (composite_kernel, pass 1) -> the summed-area table of the bins with a tile to resume
// (built in LDS: a row prefix per thread, then a column prefix per thread), their number
constexpr int kOccMaskThreads = 1024;
__global__ __launch_bounds__(kOccMaskThreads) void occ_mask_kernel(const uint32_t* __restrict__ fin, int tiles_x, int tiles_y,
                                                            uint16_t* __restrict__ sat, uint32_t* __restrict__ occ,
                                                            const uint32_t* __restrict__ d_V, uint32_t* __restrict__ unf_bins)
{
    // also: unf_bins[0 .. occ[2]) = the unfinished bins (any order: the compositor's second launch walks these alone), and
    // occ[4] = the rank count the second binning chain walks: V, or 0 when no bin is left (its kernels then find no work)
    __shared__ uint16_t s_sat[kOccSatMax];
    __shared__ uint32_t s_cnt;
    const int stride = tiles_x + 1, nbins = tiles_x * tiles_y, nsat = stride * (tiles_y + 1);
    if (threadIdx.x == 0) s_cnt = 0u;
    for (int i = threadIdx.x; i < nsat; i += kOccMaskThreads) s_sat[i] = 0;
    __syncthreads();
    for (int bin = threadIdx.x; bin < nbins; bin += kOccMaskThreads) {
        const uint4 f = *reinterpret_cast<const uint4*>(fin + (size_t)bin * 4);
        const bool unfinished = (f.x & f.y & f.z & f.w) != 0xFFFFFFFFu;      // a tile that is not final (composite_kernel, occ_pass 1)
        if (unfinished) {
            const int row = bin / tiles_x, col = bin - row * tiles_x;
            s_sat[(row + 1) * stride + col + 1] = 1;
            unf_bins[atomicAdd(&s_cnt, 1u)] = (uint32_t)bin;
        }
    }
    __syncthreads();
    for (int row = 1 + (int)threadIdx.x; row <= tiles_y; row += kOccMaskThreads) {
        uint32_t run = 0;
        for (int c = 1; c <= tiles_x; ++c) {
            run += s_sat[row * stride + c];
            s_sat[row * stride + c] = (uint16_t)run;
        }
    }
    __syncthreads();
    for (int c = 1 + (int)threadIdx.x; c <= tiles_x; c += kOccMaskThreads) {
        uint32_t run = 0;
        for (int row = 1; row <= tiles_y; ++row) {
            run += s_sat[row * stride + c];
            s_sat[row * stride + c] = (uint16_t)run;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nsat; i += kOccMaskThreads) sat[i] = s_sat[i];
    if (threadIdx.x == 0) {
        occ[2] = s_cnt;
        occ[4] = s_cnt != 0u ? *d_V : 0u;
    }
}

// does the block of bins [tx0, tx1] x [ty0, ty1] (virtual rows) contain an unfinished bin?  Four reads of the summed-area table
// (a loop over a bit mask cost a wave as many iterations as its largest rectangle has columns: 200 us at 2.4 M splats)
__device__ __forceinline__ bool occ_touches(const uint16_t* __restrict__ sat, uint32_t stride, uint32_t tx0, uint32_t tx1, uint32_t ty0,
                                            uint32_t ty1)
{
    const uint32_t a = sat[(ty1 + 1u) * stride + tx1 + 1u], b = sat[ty0 * stride + tx1 + 1u];
    const uint32_t c = sat[(ty1 + 1u) * stride + tx0], d = sat[ty0 * stride + tx0];
    return ((a - b - c + d) & 0xFFFFu) != 0u;
}

// One thread per bounding box of the spatially ordered cloud (kBoxSplats stored splats, msplat_common.hip.h): can any splat of the
// box touch an unfinished bin?  Conservative screen box of the whole box -- the extremes of x / w and y / w over its corners (the
// box must lie in front of the camera: monotone along every edge only there) widened by the largest footprint any of its splats
// can have (the bound of box_live's band test, on both axes).  dead bit set: the gate drops the box's splats without reading
// their centres -- the 16-byte gathers by sorted index cost a cache line each and made the gate as expensive as a third of the
// projection it saves.
__global__ __launch_bounds__(kThreads) void occ_box_kernel(const CullBox* __restrict__ boxes, uint32_t nboxes, FrameParams fp,
                                                           const uint16_t* __restrict__ sat, uint32_t* __restrict__ boxdead)
{
    const uint32_t stride = (uint32_t)fp.tiles_x + 1u;
    const uint32_t bi = blockIdx.x * kThreads + threadIdx.x;
    bool dead = false;
    if (bi < nboxes) {
        const CullBox b = boxes[bi];
        if (!(b.lo.x <= b.hi.x) || !(b.lo.w > 0.0f)) {
            dead = true;                               // nothing finite inside / every splat has alpha <= 1/256: nothing is drawn
        } else {
            const float* m = fp.mvp;
            const float* v = fp.view;
            float w_min = INFINITY, w_mag = 0.0f, xx_min = INFINITY, xx_max = -INFINITY, yy_min = INFINITY, yy_max = -INFINITY;
            float tz_max = -INFINITY, tx_abs = 0.0f, ty_abs = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = (c & 1) ? b.hi.x : b.lo.x, y = (c & 2) ? b.hi.y : b.lo.y, z = (c & 4) ? b.hi.z : b.lo.z;
                const float px = m[0] * x + m[4] * y + m[8] * z + m[12];
                const float py = m[1] * x + m[5] * y + m[9] * z + m[13];
                const float pw = m[3] * x + m[7] * y + m[11] * z + m[15];
                w_mag = fmaxf(w_mag, fabsf(m[3] * x) + fabsf(m[7] * y) + fabsf(m[11] * z) + fabsf(m[15]));
                w_min = fminf(w_min, pw);
                xx_min = fminf(xx_min, px / pw); xx_max = fmaxf(xx_max, px / pw);
                yy_min = fminf(yy_min, py / pw); yy_max = fmaxf(yy_max, py / pw);
                tz_max = fmaxf(tz_max, v[2] * x + v[6] * y + v[10] * z + v[14]);
                tx_abs = fmaxf(tx_abs, fabsf(v[0] * x + v[4] * y + v[8] * z + v[12]));
                ty_abs = fmaxf(ty_abs, fabsf(v[1] * x + v[5] * y + v[9] * z + v[13]));
            }
            if (w_min > 1e-5f * w_mag && tz_max < 0.0f) {        // the whole box in front of the camera; otherwise: alive
                const float rtz = 1.0f / -tz_max;                // largest 1 / |tz| in the box
                const float jsx = 0.5f * fabsf(fp.proj[0]) * fp.W * rtz, jsy = 0.5f * fabsf(fp.proj[5]) * fp.H * rtz;
                const float trx = tx_abs * rtz, try_ = ty_abs * rtz;
                const float ex = sqrtf(jsx * jsx * (1.0f + trx * trx) * fp.view_scale2 * b.lo.w + 3.4f) * 1.004f + 2.5f;
                const float ey = sqrtf(jsy * jsy * (1.0f + try_ * try_) * fp.view_scale2 * b.lo.w + 3.4f) * 1.004f + 2.5f;
                const float cx0 = 0.5f * (fp.W + xx_min * fp.W) + fp.X0, cx1 = 0.5f * (fp.W + xx_max * fp.W) + fp.X0;
                const float cy0 = 0.5f * (fp.H + yy_min * fp.H) + fp.Y0, cy1 = 0.5f * (fp.H + yy_max * fp.H) + fp.Y0;
                const float sx = 1e-4f * (fabsf(cx0) + fabsf(cx1) + fp.W), sy = 1e-4f * (fabsf(cy0) + fabsf(cy1) + fp.H);
                const float x0 = fmaxf(cx0 - ex - sx, 0.0f), x1 = fminf(cx1 + ex + sx, fp.W - 1.0f);
                const float y0 = fmaxf(cy0 - ey - sy, 0.0f), y1 = fminf(cy1 + ey + sy, fp.H - 1.0f);
                const bool finite = (x0 == x0) && (x1 == x1) && (y0 == y0) && (y1 == y1);
                if (finite) {
                    dead = true;
                    if (x0 <= x1 && y0 <= y1) {
                        const int c0 = (int)x0 / kBin, c1 = min((int)x1 / kBin, fp.tiles_x - 1);
                        const int r0 = (int)y0 / kBin, r1 = (int)y1 / kBin;
                        int v0 = r0, v1 = r1;
                        if (fp.banded) {
                            v0 = band_first_owned_from(fp, r0);
                            v1 = band_last_owned_upto(fp, r1);
                        }
                        v1 = min(v1, fp.tiles_y - 1);
                        if (c0 <= c1 && v0 <= v1) dead = !occ_touches(sat, stride, (uint32_t)c0, (uint32_t)c1, (uint32_t)v0, (uint32_t)v1);
                    }
                }
            }
        }
    }
    const unsigned long long d = __ballot(dead);
    const int lane = threadIdx.x & 63;
    const uint32_t word = (blockIdx.x * kThreads + (threadIdx.x & ~63u)) >> 5;      // this wave's first of two words
    if (lane == 0) boxdead[word] = (uint32_t)d;
    if (lane == 32) boxdead[word + 1u] = (uint32_t)(d >> 32);
}

// kOccGateRanks ranks per workgroup.  Ranks of pass 1 (>= cut): the exact rectangle stays only if it touches an unfinished bin.
// Ranks behind the cut: conservative screen box of the footprint (the bound of cull_key's band test, on both axes); when it
// touches an unfinished bin the rank goes on the list project_kernel walks next (order irrelevant: records and rectangles are
// stored by rank), otherwise its rectangle stays empty (project_kernel wrote it in pass 1).  The list is gathered in LDS and
// appended with ONE global atomic per workgroup (same-address atomics are served one per ~10 ns: an atomic per wave made this
// kernel 137 us at 1 M splats).
constexpr int kOccGateItems = 8;          // (16: 168 VGPRs, the gate 8 us slower at 6 M splats; 4: no better)
constexpr int kOccGateRanks = kThreads * kOccGateItems;
__global__ __launch_bounds__(kThreads) void occ_gate_kernel(const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ d_V,
                                                            uint32_t* __restrict__ occ, const float4* __restrict__ pos4,
                                                            uint32_t* __restrict__ rect, FrameParams fp,
                                                            const uint16_t* __restrict__ sat, uint32_t* __restrict__ live_list,
                                                            uint32_t* __restrict__ host_words, const uint32_t* __restrict__ d_D,
                                                            uint32_t seq, const uint32_t* __restrict__ boxdead, uint32_t boxwords)
{
    // boxdead (spatially ordered clouds, occ_box_kernel): bit b set = no splat of box b can touch an unfinished bin
    __shared__ uint32_t s_list[kOccGateRanks];
    __shared__ uint32_t s_dead[2048];                 // 65 536 boxes = 2^24 splats
    __shared__ uint32_t s_n, s_base;
    const uint32_t stride = (uint32_t)fp.tiles_x + 1u;
    if (boxdead != nullptr)
        for (uint32_t i = threadIdx.x; i < min(boxwords, 2048u); i += kThreads) s_dead[i] = boxdead[i];
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const uint32_t V = *d_V, cut = occ[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && host_words != nullptr) {
        // feedback for the host's choice of the share in a LATER frame: pairs of pass 1, unfinished bins, splats in pass 1, and
        // which frame this is about
        __hip_atomic_store(host_words + 4, *d_D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 5, occ[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 6, V - cut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 9, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t base = blockIdx.x * (uint32_t)kOccGateRanks;
    if (base >= V) return;
    const int lane = threadIdx.x & 63;
    // loads first, all in flight together: the index and the centre of a rank behind the cut, the rectangle of a rank of pass 1
    uint32_t idx[kOccGateItems], rc[kOccGateItems];
#pragma unroll
    for (int k = 0; k < kOccGateItems; ++k) {
        const uint32_t r = base + (uint32_t)k * kThreads + threadIdx.x;
        const bool in = r < V, behind = in && r < cut;
        idx[k] = behind ? sorted_idx[r] : 0xFFFFFFFFu;
        rc[k] = (in && !behind) ? rect[r] : kRectEmpty;
    }
    // a splat behind the cut whose box is dead is dead: its centre is not fetched (w = 0 reads as "never drawn" below)
    float4 pp[kOccGateItems];
#pragma unroll
    for (int k = 0; k < kOccGateItems; ++k) {
        bool fetch = idx[k] != 0xFFFFFFFFu;
        if (fetch && boxdead != nullptr) {
            const uint32_t bx = idx[k] / (uint32_t)kBoxSplats;
            if ((bx >> 5) < min(boxwords, 2048u) && ((s_dead[bx >> 5] >> (bx & 31u)) & 1u)) fetch = false;
        }
        pp[k] = fetch ? pos4[idx[k]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int k = 0; k < kOccGateItems; ++k) {
        const uint32_t r = base + (uint32_t)k * kThreads + threadIdx.x;
        bool live = false;
        if (idx[k] == 0xFFFFFFFFu) {
            const uint32_t tx0 = rc[k] & 255u, ty0 = (rc[k] >> 8) & 255u, tx1 = (rc[k] >> 16) & 255u, ty1 = rc[k] >> 24;
            if (tx0 <= tx1 && !occ_touches(sat, stride, tx0, tx1, ty0, ty1)) rect[r] = kRectEmpty;
        } else {
            const float4 p = pp[k];
            if (p.w > 0.0f) {               // (0: alpha <= 1/256, never drawn)
                const float* m = fp.mvp;
                const float* v = fp.view;
                const float px = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
                const float py = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
                const float pw = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
                const float tx = v[0] * p.x + v[4] * p.y + v[8] * p.z + v[12];
                const float ty = v[1] * p.x + v[5] * p.y + v[9] * p.z + v[13];
                const float tz = v[2] * p.x + v[6] * p.y + v[10] * p.z + v[14];
                const float rw = 1.0f / pw, rtz = 1.0f / tz;
                // e^2 = rho^2 (M Sigma M^T + 0.3) <= |J_row|^2 |W|^2 rho^2 lambda_max(Sigma) + 0.3 rho^2_max  (see cull_key)
                const float jsx = 0.5f * fp.proj[0] * fp.W * rtz, jsy = 0.5f * fp.proj[5] * fp.H * rtz;
                const float trx = tx * rtz, try_ = ty * rtz;
                const float ex = sqrtf(jsx * jsx * (1.0f + trx * trx) * fp.view_scale2 * p.w + 3.4f) * 1.002f + 1.5f;
                const float ey = sqrtf(jsy * jsy * (1.0f + try_ * try_) * fp.view_scale2 * p.w + 3.4f) * 1.002f + 1.5f;
                const float cx = 0.5f * (fp.W + px * rw * fp.W) + fp.X0, cy = 0.5f * (fp.H + py * rw * fp.H) + fp.Y0;
                const float x0 = fmaxf(cx - ex, 0.0f), x1 = fminf(cx + ex, fp.W - 1.0f);
                const float y0 = fmaxf(cy - ey, 0.0f), y1 = fminf(cy + ey, fp.H - 1.0f);
                const bool finite = (x0 == x0) && (x1 == x1) && (y0 == y0) && (y1 == y1);
                if (!finite) {
                    live = true;            // (whatever the projection makes of it)
                } else if (x0 <= x1 && y0 <= y1) {
                    const int c0 = (int)x0 / kBin, c1 = min((int)x1 / kBin, fp.tiles_x - 1);
                    const int r0 = (int)y0 / kBin, r1 = (int)y1 / kBin;
                    int v0 = r0, v1 = r1;
                    if (fp.banded) {
                        v0 = band_first_owned_from(fp, r0);
                        v1 = band_last_owned_upto(fp, r1);
                    }
                    v1 = min(v1, fp.tiles_y - 1);
                    if (c0 <= c1 && v0 <= v1) live = occ_touches(sat, stride, (uint32_t)c0, (uint32_t)c1, (uint32_t)v0, (uint32_t)v1);
                }
            }
        }
        const unsigned long long b = __ballot(live);
        if (b != 0ull) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&s_n, (uint32_t)__popcll(b));
            at = __shfl(at, 0, 64);
            if (live) s_list[at + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = r;
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    if (n == 0u) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&occ[1], n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) live_list[s_base + i] = s_list[i];
}

}  // namespace msplat
