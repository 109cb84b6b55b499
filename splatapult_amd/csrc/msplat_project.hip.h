// msplat_project.hip.h -- project_kernel: the reference's vertex + geometry stage (shader/splat_vert.glsl, shader/splat_geom.glsl) per draw-order rank
// (one of the parts of msplat_kernels.hip.h)
#pragma once

#include "msplat_common.hip.h"

#pragma clang fp contract(off)

namespace msplat {

// ------------------------------------------------------------------------------------------
// project: vertex + geometry stage for the splats in draw order (one thread per rank)
//   splat_vert.glsl:153-222 (+ SH :51-127, sRGB :129-151), splat_geom.glsl:22-54
// Writes a 48-byte record per rank, a packed tile rectangle, and counts pairs per tile.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float srgb_to_linear(float s)
{
    if (s <= 0.04045f) return s / 12.92f;
    return powf((s + 0.055f) / 1.055f, 2.4f);
}

// Window-space depth as an order-preserving uint32: 24-bit unorm like the default back buffer
// (sdl_main.cpp:79), or the raw bits of the non-negative float for a 32F depth attachment.
__device__ __forceinline__ uint32_t quantise_depth(float ndcz, int depth_bits)
{
    const float zw = __fadd_rn(__fmul_rn(0.5f, ndcz), 0.5f);
    if (!(zw >= 0.0f)) return 0u;
    if (depth_bits == 24) {
        const double q = floor((double)zw * 16777215.0 + 0.5);
        return q >= 16777215.0 ? 16777215u : (uint32_t)q;
    }
    return __float_as_uint(zw);
}

constexpr int kProjThreads = 64;          // one wave per workgroup: wave-private LDS staging, no block barriers

// one wave-block of 64 ranks: cooperative gather of the records, then the vertex + geometry stage per lane (rank r, rank rl inside
// its view; lanes with !valid take part in the gather only).  s_stage: 64 * STRIDE floats of wave-private LDS.
// vm / pm / eye: the view's matrices (kernarg words: they stay in SGPRs); SECOND: the second view of a two-view chain.
template <bool FULL_SH, bool SECOND>
__device__ __forceinline__ void project_block(const uint32_t r, const uint32_t rl, const bool valid, const int lane,
                                              const uint32_t* __restrict__ sorted_idx, const float4* __restrict__ recs,
                                              const ProjParams& fp, const float* vm, const float* pm, const float* eye,
                                              float4* __restrict__ out_rec, uint32_t* __restrict__ out_rect,
                                              uint32_t* __restrict__ out_zq, float* s_stage)
{
    constexpr int F4 = FULL_SH ? 16 : 8;
    constexpr int RPI = 64 / F4;              // records fetched per wave-wide load instruction
    constexpr int STRIDE = F4 * 4 + 4;        // dwords
    const uint32_t i = valid ? sorted_idx[rl] : 0u;
    {
        const int sub = lane % F4;
        float4 tmp[F4];
#pragma unroll
        for (int it = 0; it < F4; ++it) {
            const int owner = it * RPI + lane / F4;
            const uint32_t oi = __shfl(i, owner, 64);
            tmp[it] = recs[(size_t)oi * F4 + sub];
        }
#pragma unroll
        for (int it = 0; it < F4; ++it) {
            const int owner = it * RPI + lane / F4;
            *reinterpret_cast<float4*>(&s_stage[owner * STRIDE + sub * 4]) = tmp[it];
        }
    }
    __syncthreads();
    float f[F4 * 4];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(&s_stage[lane * STRIDE + k * 4]);
        f[4 * k + 0] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w;
    }
    if (!valid) return;
    const float x = f[0], y = f[1], z = f[2], alpha = f[3];

    // t = viewMat * vec4(pos, 1)   -- same op order as the oracle (reject tests must not flip)
    float t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        t[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(vm[c], x), __fmul_rn(vm[4 + c], y)), __fmul_rn(vm[8 + c], z)), vm[12 + c]);
    float p4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        p4[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pm[c], t[0]), __fmul_rn(pm[4 + c], t[1])), __fmul_rn(pm[8 + c], t[2])), __fmul_rn(pm[12 + c], t[3]));
    const float ndcx = __fdiv_rn(p4[0], p4[3]);
    const float ndcy = __fdiv_rn(p4[1], p4[3]);
    const float ndcz = __fdiv_rn(p4[2], p4[3]);

    bool reject = (ndcz < 0.25f) || (ndcx > 2.0f) || (ndcx < -2.0f) || (ndcy > 2.0f) || (ndcy < -2.0f);
    if (!(ndcz <= 1.0f)) reject = true;     // far-plane clip of the whole quad / NaN
    if (!(p4[3] > 0.0f)) reject = true;

    const float WIDTH = fp.W, HEIGHT = fp.H;
    const float px = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(WIDTH, __fmul_rn(ndcx, WIDTH)), __fmul_rn(2.0f, fp.X0)));
    const float py = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(HEIGHT, __fmul_rn(ndcy, HEIGHT)), __fmul_rn(2.0f, fp.Y0)));

    // Jacobian rows (splat_vert.glsl:170-181); third row only feeds dropped terms
    const float SX = pm[0], SY = pm[5];
    const float tz = t[2];
    const float tzSq = tz * tz;
    const float jsx = -(SX * WIDTH) / (2.0f * tz);
    const float jsy = -(SY * HEIGHT) / (2.0f * tz);
    const float jtx = (SX * t[0] * WIDTH) / (2.0f * tzSq);
    const float jty = (SY * t[1] * HEIGHT) / (2.0f * tzSq);
    // M = [J0;J1] * mat3(viewMat):  M[r][k] = J[r][0]*W[0][k] + J[r][1]*W[1][k] + J[r][2]*W[2][k]
    // with W[row][col] = vm[col*4 + row]
    float M0[3], M1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        M0[k] = jsx * vm[k * 4 + 0] + jtx * vm[k * 4 + 2];
        M1[k] = jsy * vm[k * 4 + 1] + jty * vm[k * 4 + 2];
    }
    // Sigma columns: col0 = f[16..18], col1 = f[19..21], col2 = f[22..24];  S[row][col] = f[16 + col*3 + row]
    float A0[3], A1[3];   // A = M * Sigma
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A0[c] = M0[0] * f[16 + c * 3 + 0] + M0[1] * f[16 + c * 3 + 1] + M0[2] * f[16 + c * 3 + 2];
        A1[c] = M1[0] * f[16 + c * 3 + 0] + M1[1] * f[16 + c * 3 + 1] + M1[2] * f[16 + c * 3 + 2];
    }
    const float m00 = (A0[0] * M0[0] + A0[1] * M0[1] + A0[2] * M0[2]) + 0.3f;
    const float m10 = (A0[0] * M1[0] + A0[1] * M1[1] + A0[2] * M1[2]);   // row 0, col 1
    const float m01 = (A1[0] * M0[0] + A1[1] * M0[1] + A1[2] * M0[2]);   // row 1, col 0
    const float m11 = (A1[0] * M1[0] + A1[1] * M1[1] + A1[2] * M1[2]) + 0.3f;
    const float det = m00 * m11 - m01 * m10;
    const float i00 = m11 / det;
    const float i01 = -m01 / det;
    const float i10 = -m10 / det;
    const float i11 = m00 / det;

    // colour: 0.5 + SH(v), no clamp (splat_vert.glsl:51-127,206-207)
    const float dx = x - eye[0], dy = y - eye[1], dz = z - eye[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float vx = dx / len, vy = dy / len, vz = dz / len;
    float b[FULL_SH ? 16 : 4];
    b[0] = 0.28209479177387814f;
    const float k1 = 0.4886025119029199f;
    b[1] = -k1 * vy;
    b[2] = k1 * vz;
    b[3] = -k1 * vx;
    float rgb[3];
    if constexpr (FULL_SH) {
        const float vx2 = vx * vx, vy2 = vy * vy, vz2 = vz * vz;
        const float k2 = 1.0925484305920792f, k3 = 0.31539156525252005f, k4 = 0.5462742152960396f;
        b[4] = k2 * vy * vx;
        b[5] = -k2 * vy * vz;
        b[6] = k3 * (3.0f * vz2 - 1.0f);
        b[7] = -k2 * vx * vz;
        b[8] = k4 * (vx2 - vy2);
        const float k5 = 0.5900435899266435f, k6 = 2.8906114426405543f, k7 = 0.4570457994644658f;
        const float k8 = 0.37317633259011546f, k9 = 1.4453057213202771f;
        b[9] = -k5 * vy * (3.0f * vx2 - vy2);
        b[10] = k6 * vy * vx * vz;
        b[11] = -k7 * vy * (5.0f * vz2 - 1.0f);
        b[12] = k8 * vz * (5.0f * vz2 - 3.0f);
        b[13] = -k7 * vx * (5.0f * vz2 - 1.0f);
        b[14] = k9 * vz * (vx2 - vy2);
        b[15] = -k5 * vx * (vx2 - 3.0f * vy2);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // channel c: sh0 at floats 4+4c.., sh1..3 at floats 25+12c..
            float s = b[0] * f[4 + 4 * c];
#pragma unroll
            for (int k = 1; k < 4; ++k) s = s + b[k] * f[4 + 4 * c + k];
#pragma unroll
            for (int k = 4; k < 16; ++k) s = s + b[k] * f[25 + 12 * c + (k - 4)];
            rgb[c] = 0.5f + s;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = b[0] * f[4 + 4 * c];
#pragma unroll
            for (int k = 1; k < 4; ++k) s = s + b[k] * f[4 + 4 * c + k];
            rgb[c] = 0.5f + s;
        }
    }
    if (fp.srgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = srgb_to_linear(rgb[c]);
    }

    // footprint: w = alpha*exp(-q/2) > 1/256  <=>  q < 2 ln(256 alpha) =: rho2 (splat_frag.glsl:37-40).
    // The 3.5-sigma quad of splat_geom.glsl:56-106 always contains it (rho <= 3.33), so the discard
    // test alone defines coverage.
    uint32_t rect = kRectEmpty;
    const float rho2 = 2.0f * logf(256.0f * alpha);
    if (!(rho2 > 0.0f)) reject = true;                      // alpha <= 1/256 (or NaN): never visible
    if (!(det > 0.0f) || !(m00 > 0.0f) || !(m11 > 0.0f)) reject = true;   // degenerate/NaN covariance
    float ex = 0.0f, ey = 0.0f;       // conservative half extents of the footprint (pixels)
    if (!reject) {
        ex = sqrtf(rho2 * m00) * 1.0001f + 0.01f;
        ey = sqrtf(rho2 * m11) * 1.0001f + 0.01f;
        float x0f = ceilf(px - ex - 0.5f), x1f = floorf(px + ex - 0.5f);
        float y0f = ceilf(py - ey - 0.5f), y1f = floorf(py + ey - 0.5f);
        x0f = fmaxf(x0f, 0.0f);
        y0f = fmaxf(y0f, 0.0f);
        x1f = fminf(x1f, (float)(fp.width - 1));
        y1f = fminf(y1f, (float)(fp.height - 1));
        if (x0f <= x1f && y0f <= y1f) {
            const int tx0 = (int)x0f / kBin, tx1 = (int)x1f / kBin;
            int ty0 = (int)y0f / kBin, ty1 = (int)y1f / kBin;
            // band mode: keep only the owned bin rows, as their virtual numbers (a contiguous range: vy ascends with the row)
            if (fp.banded) {
                const int v0 = band_first_owned_from(fp, ty0), v1 = min(band_last_owned_upto(fp, ty1), fp.tiles_y - 1);
                ty0 = v0;
                ty1 = v1;
            }
            if (SECOND) { ty0 += fp.rows_view; ty1 += fp.rows_view; }         // the second view's bins follow the first's
            if (ty0 <= ty1) {
                rect = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
            }
        }
    }

    // record: w(dx,dy) = exp2(A dx^2 + B dx dy + C dy^2 + log2 alpha)
    const float kk = -0.5f * 1.44269504088896340736f;
    float4 r0, r1, r2;
    r0.x = px; r0.y = py; r0.z = kk * i00; r0.w = kk * (i01 + i10);
    r1.x = kk * i11; r1.y = log2f(alpha); r1.z = rgb[0]; r1.w = rgb[1];
    r2.x = rgb[2]; r2.y = alpha; r2.z = ex; r2.w = ey;
    out_rec[(size_t)r * 3 + 0] = r0;
    out_rec[(size_t)r * 3 + 1] = r1;
    out_rec[(size_t)r * 3 + 2] = r2;
    out_rect[r] = rect;
    // depth-buffer emulation (composite_depth_kernel): the quad's fragments all carry the centre's depth
    // (splat_geom.glsl:93-101 offsets only x and y); window z = 0.5 ndc.z + 0.5 (default glDepthRange)
    if (out_zq != nullptr) out_zq[r] = quantise_depth(ndcz, fp.depth_bits);
}

// MODE (r5: template parameters instead of r4's run-time arguments, which cost the plain frame 149 VGPRs / 36 SGPR spills against
// r3's 98 / 0): PROJ_PLAIN one view, one pass; PROJ_PASS1 / PROJ_LISTED the two projections of a two-pass frame
// (msplat_occlusion.hip.h); PROJ_TWO_VIEWS both eyes in one chain.
// (plain ints, and the second view's type through a named trait: an unnamed enum or a conditional on MODE in the kernel's signature
//  mangles differently in the host and the device compilation -- "Cannot find Symbol" at the first launch)
constexpr int PROJ_PLAIN = 0, PROJ_PASS1 = 1, PROJ_LISTED = 2, PROJ_TWO_VIEWS = 3;
template <int MODE> struct ProjSecondView { using type = ProjNoView1; };
template <> struct ProjSecondView<3> { using type = ProjView1; };
struct ProjExtra {
    uint32_t* d_Veff;            // PROJ_TWO_VIEWS: ranks the binning walks (V1 + V); PROJ_LISTED: optional host-mapped count of listed ranks
    uint32_t* d_cut;             // PROJ_PASS1: occ[0 .. 2]
    const uint32_t* rank_list;   // PROJ_LISTED: the *d_V ranks to project (any order)
    float occ_share;             // PROJ_PASS1: share of the visible splats in pass 1
};

template <bool FULL_SH, int MODE>
__global__ __launch_bounds__(kProjThreads) void project_kernel(const uint32_t* __restrict__ sorted_idx,
                                                               const uint32_t* __restrict__ d_V,
                                                               const float4* __restrict__ recs,
                                                               ProjParams fp,
                                                               float4* __restrict__ out_rec,
                                                               uint32_t* __restrict__ out_rect,
                                                               uint32_t* __restrict__ out_zq, ProjExtra ex,
                                                               typename ProjSecondView<MODE>::type v1)
{
    // PROJ_PASS1: ranks below cut = occ_cut(V, occ_share) only get an empty rectangle, their records are not fetched; the cut is
    // left in d_cut[0] (= occ[0]).  PROJ_LISTED: the *d_V ranks to project are listed (any order); records and rectangles are
    // stored by rank as always.
    // Records are 256 B (full SH) or 128 B (base) and line aligned.  The gather by sorted index is
    // done cooperatively (project_block): F4 consecutive lanes fetch one whole record (coalesced 256/128 B), the wave
    // stages 64 records in LDS, then every lane reads its own record back (stride 68/36 dwords keeps
    // the ds_read_b128 accesses conflict free).
    __builtin_amdgcn_s_setprio(kProjPrio);
    MSPLAT_STAMP(KID_PROJECT);
    constexpr int F4 = FULL_SH ? 16 : 8;
    constexpr int STRIDE = F4 * 4 + 4;        // dwords
    __shared__ __attribute__((aligned(16))) float s_stage[64 * STRIDE];
    const uint32_t V = *d_V;
    const int lane = threadIdx.x;
    if constexpr (MODE == PROJ_LISTED) {
        // pass 2 of a two-pass frame: the listed ranks, grid-stride (the list is short: a grid sized for the cloud would be ~10^5
        // workgroups that find nothing -- 35 us at 6 M splats)
        if (ex.d_Veff != nullptr && blockIdx.x == 0 && lane == 0) *ex.d_Veff = V;
        for (uint32_t s0 = blockIdx.x * kProjThreads; s0 < V; s0 += gridDim.x * kProjThreads) {
            const bool ok = s0 + lane < V;
            const uint32_t rk = ok ? ex.rank_list[s0 + lane] : 0u;
            project_block<FULL_SH, false>(rk, rk, ok, lane, sorted_idx, recs, fp, fp.view, fp.proj, fp.eye, out_rec, out_rect, out_zq, s_stage);
            __syncthreads();              // s_stage is reused
        }
    } else if constexpr (MODE == PROJ_PASS1) {
        // pass 1 of a two-pass frame: ranks [cut, V) are projected (grid-stride), the ranks behind the cut get empty rectangles
        const uint32_t cut = occ_cut(V, ex.occ_share);
        if (blockIdx.x == 0 && lane == 0) { ex.d_cut[0] = cut; ex.d_cut[1] = 0u; ex.d_cut[2] = 0u; }      // occ[0 .. 2] for the kernels that follow
        for (uint32_t i = blockIdx.x * kProjThreads + lane; i < cut; i += gridDim.x * kProjThreads) out_rect[i] = kRectEmpty;
        for (uint32_t r0 = cut + blockIdx.x * kProjThreads; r0 < V; r0 += gridDim.x * kProjThreads) {
            project_block<FULL_SH, false>(r0 + lane, r0 + lane, r0 + lane < V, lane, sorted_idx, recs, fp, fp.view, fp.proj, fp.eye, out_rec, out_rect, out_zq, s_stage);
            __syncthreads();
        }
    } else if constexpr (MODE == PROJ_TWO_VIEWS) {
        // ranks [0, V) are view 0, [V1, V1 + V) view 1 (V1 = V rounded up to 64: a wave never straddles the views); the gap gets
        // empty rectangles
        const uint32_t V1 = (V + 63u) & ~63u;
        const uint32_t total = V1 + V;
        if (blockIdx.x == 0 && lane == 0) *ex.d_Veff = total;       // what the binning passes walk
        if (blockIdx.x * kProjThreads >= total) return;
        const uint32_t r = blockIdx.x * kProjThreads + lane;
        if (blockIdx.x * kProjThreads >= V1) {                        // wave-uniform
            const uint32_t rl = r - V1;                               // rank inside the view
            project_block<FULL_SH, true>(r, rl, rl < V, lane, sorted_idx, recs, fp, v1.view, v1.proj, v1.eye, out_rec, out_rect, out_zq, s_stage);
        } else {
            const bool valid = r < V;
            if (!valid) out_rect[r] = kRectEmpty;                     // (the gap between the views, and nothing else)
            project_block<FULL_SH, false>(r, r, valid, lane, sorted_idx, recs, fp, fp.view, fp.proj, fp.eye, out_rec, out_rect, out_zq, s_stage);
        }
    } else {
        if (blockIdx.x * kProjThreads >= V) return;
        const uint32_t r = blockIdx.x * kProjThreads + lane;
        project_block<FULL_SH, false>(r, r, r < V, lane, sorted_idx, recs, fp, fp.view, fp.proj, fp.eye, out_rec, out_rect, out_zq, s_stage);
    }
}

__device__ __forceinline__ uint32_t rect_width(uint32_t rc)
{
    const uint32_t tx0 = rc & 255u, tx1 = (rc >> 16) & 255u;
    return tx0 <= tx1 ? tx1 - tx0 + 1u : 0u;
}

// statistics only (msplat_get_stats): number of splats with a non-empty rectangle, and the number of
// (splat, 16x16 tile) pairs their footprints cover (the "D" of the algorithmic byte count)
__global__ __launch_bounds__(kThreads) void count_drawn_kernel(const uint32_t* __restrict__ rect,
                                                               const float4* __restrict__ rec,
                                                               const uint32_t* __restrict__ d_V, FrameParams fp,
                                                               uint32_t* __restrict__ d_drawn,
                                                               unsigned long long* __restrict__ d_pairs16)
{
    const uint32_t V = *d_V;
    uint32_t c = 0;
    unsigned long long p16 = 0;
    for (uint32_t r = blockIdx.x * kThreads + threadIdx.x; r < V; r += gridDim.x * kThreads) {
        if (rect_width(rect[r]) == 0u) continue;
        ++c;
        const float4 a = rec[(size_t)r * 3 + 0], q = rec[(size_t)r * 3 + 2];      // px, py ... ex, ey
        const float x0 = fmaxf(ceilf(a.x - q.z - 0.5f), 0.0f), x1 = fminf(floorf(a.x + q.z - 0.5f), (float)(fp.width - 1));
        const float y0 = fmaxf(ceilf(a.y - q.w - 0.5f), 0.0f), y1 = fminf(floorf(a.y + q.w - 0.5f), (float)(fp.height - 1));
        if (x0 <= x1 && y0 <= y1) {
            int ty0 = (int)y0 / kTile, ty1 = (int)y1 / kTile, rows = 0;
            for (int ty = ty0; ty <= ty1; ++ty) {
                const int br = ty / (kBin / kTile), v = band_first_owned_from(fp, br);
                rows += (v < fp.tiles_y && band_real_row(fp, v) == br) ? 1 : 0;
            }
            p16 += (unsigned long long)((int)x1 / kTile - (int)x0 / kTile + 1) * (unsigned long long)rows;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        c += __shfl_down(c, d, 64);
        p16 += __shfl_down(p16, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && c) {
        atomicAdd(d_drawn, c);
        atomicAdd(d_pairs16, p16);
    }
}

}  // namespace msplat
