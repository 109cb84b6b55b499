// msplat_kernels.hip.h -- hand-written CDNA4 (gfx950, wave64) kernels of the splat hot path.
//
// Replaces (not ports) the reference's GL pipeline:
//   shader/presort_compute.glsl + shader/multi_radixsort*.glsl  -> radix_* (cull fused in pass 0)
//   shader/splat_vert.glsl + shader/splat_geom.glsl             -> project_kernel
//   GL rasteriser + shader/splat_frag.glsl + ROP blend          -> bin1_*/radix_*<MODE_PAIR> + composite_kernel
//
// Design notes: everything is HBM/LDS/VALU work -- no MFMA anywhere.
// Compiled with -ffp-contract=off: an FMA happens only where __builtin_fmaf is written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The key / reject arithmetic must execute exactly as written (bit-parity with the oracle):
// no implicit FMA contraction anywhere in this file (the build also passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace msplat {

constexpr int kThreads = 256;            // 4 wave64 per workgroup
// keys per thread per chunk of the sort passes: 8 (2048-key chunks) up to 2 M splats -- 4 measured no faster (r1), 16 slower
// at 1 M (245 workgroups for 256 CUs: sort 76 -> 91 us) -- and 16 (4096-key chunks) beyond: digit runs twice as long make
// the scattered write-out cheaper (6 M splats: sort 234 -> 212 us, r2)
constexpr int kSortItems = 8;
constexpr int kSortItemsLarge = 16;
constexpr int kSortChunk = kThreads * kSortItems;   // 2048 keys per chunk
constexpr int kPairItems = 16;           // binning pass 2 moves 7x more words: bigger chunks, longer runs
constexpr int kPairChunk = kThreads * kPairItems;   // 4096 words per chunk
template <int MODE, int SORT_ITEMS> struct RadixCfg { static constexpr int ITEMS = (MODE == 2) ? kPairItems : SORT_ITEMS; static constexpr int CHUNK = kThreads * ITEMS; };
constexpr int kBinChunk = 1024;          // draw-order ranks per chunk in the tile-column partition (512 / 2048 measured at 1 M: binning 64 -> 70 us)
constexpr uint32_t kHeavyPairs = 49152;  // a column-pass chunk with more pairs than this is split over kHeavyParts workgroups
constexpr uint32_t kHeavyCap = 128;      // at most this many split chunks per frame (the rest run unsplit: correct, slower)
constexpr uint32_t kHeavyParts = 8;      // column blocks per split chunk
constexpr int kTile = 16;                // one compositor wave owns a 16x16 pixel tile ...
constexpr int kBin = 32;                 // ... binning works on 32x32 bins (4 tiles share one list, each
                                         // wave filters it for its own quadrant): 2.2-2.9x fewer pairs
// the compositors' sharded work queue (queue_next): 32 heads, one 64-byte line each
constexpr uint32_t kQueueShards = 32;
constexpr uint32_t kQueueStride = 16;          // words between heads
constexpr uint32_t kRectEmpty = 0x000000FFu;   // tx0=255 > tx1=0
constexpr uint32_t kRankMask = 0x00FFFFFFu;

enum { MODE_KEYS = 0, MODE_CULL = 1, MODE_PAIR = 2 };

// Per-frame constants, passed by value (lives in SGPRs / kernarg segment).
struct FrameParams {
    float mvp[16];     // projMat * inverse(cameraMat)            (splatrenderer.cpp:161,175)
    float view[16];    // inverse(cameraMat)                      (splatrenderer.cpp:327)
    float proj[16];
    float eye[3];      // cameraMat[3].xyz                        (splatrenderer.cpp:328)
    float W, H, X0, Y0, zn, zf;
    float t_eps;
    int width, height;
    int tiles_x, tiles_y;       // tiles_y = number of OWNED bin rows (band mode) else ceil(H / 32)
    // Band (multi-GPU): the owned bin rows are blocks of band_block consecutive rows that start at
    // band_first, band_first + band_stride, ...; they are numbered vy = 0 .. tiles_y - 1 in ascending order ("virtual rows":
    // what the pair words, the bin lists and the compositor's work items carry).  banded == 0: every row, vy == row.
    int banded, band_first, band_block, band_stride;
    float band_inv_stride;      // 1 / band_stride: row numbers are below 256, so their quotients are taken in float (exact)
    int full_sh, srgb;
    int band_cull;              // multi-GPU only: Sort also drops splats that cannot reach an owned bin row
    float view_scale2;          // largest squared column norm of mat3(view) (1 for a rigid camera)
    int depth_bits;             // 0 = colour-only target (no depth test); 24 / 32 = emulated depth buffer
    int rop;                    // 0 = float accumulation; 1 = RGBA8, 2 = RGBA16F render-target rounding after every blend
    // Two views in ONE render chain (msplat_render_stereo, r4): the second view's matrices; its splats are the draw-order ranks
    // [V1, V1 + V) with V1 = V rounded up to 64 (a projection wave never straddles the views), its bin rows follow the first
    // view's: rows_view .. 2 rows_view - 1.  views == 1: everything above describes the only view.
    int views, rows_view;
    float view1[16], proj1[16], eye1[3];
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// band geometry (see FrameParams): real bin row of virtual row vy
__host__ __device__ __forceinline__ int band_real_row(const FrameParams& fp, int vy)
{
    if (!fp.banded) return vy;
    const int k = vy / fp.band_block;
    return fp.band_first + k * fp.band_stride + (vy - k * fp.band_block);
}
// d / band_stride for 0 <= d < 65536 without an integer division (~30 instructions on this hardware, twice per splat in the
// band-culled sort and in project_kernel): (d + 0.5) / s lies at least 0.5 / s away from every integer, far more than the
// float error of the product
__host__ __device__ __forceinline__ int band_quot(const FrameParams& fp, int d)
{
    return (int)(((float)d + 0.5f) * fp.band_inv_stride);
}
// virtual index of the first owned row >= t (>= tiles_y: there is none)
__host__ __device__ __forceinline__ int band_first_owned_from(const FrameParams& fp, int t)
{
    if (!fp.banded) return t < 0 ? 0 : t;
    if (t <= fp.band_first) return 0;
    const int d = t - fp.band_first, k = band_quot(fp, d), j = d - k * fp.band_stride;
    return j < fp.band_block ? k * fp.band_block + j : (k + 1) * fp.band_block;
}
// virtual index of the last owned row <= t (-1: there is none; may be >= tiles_y: clamp)
__host__ __device__ __forceinline__ int band_last_owned_upto(const FrameParams& fp, int t)
{
    if (!fp.banded) return t;
    if (t < fp.band_first) return -1;
    const int d = t - fp.band_first, k = band_quot(fp, d), j = d - k * fp.band_stride;
    return k * fp.band_block + (j < fp.band_block ? j : fp.band_block - 1);
}

// inclusive scan of one uint32 per thread across a 256-thread workgroup.
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* s_tmp4, uint32_t& total)
{
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_tmp4[w] = v;
    __syncthreads();
    uint32_t off = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t s = s_tmp4[k];
        if (k < w) off += s;
        total += s;
    }
    __syncthreads();
    return v + off;
}

// Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md; speed only, never correctness).
// Remap so that each XCD processes a CONTIGUOUS range of chunks.  Measured r1: using it for the
// scatter kernels (radix/bin1 downsweep) was 5-15 % SLOWER than the plain round-robin mapping, so it
// is currently unused there.
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, xcd = b & 7u, idx = b >> 3;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}

// World-space footprint bound of a splat, kept in pos4.w for the band-restricted cull: rho^2 * lambda_max(Sigma) with
// rho^2 = 2 ln(256 alpha) (the fragment shader's discard radius) -- every projected variance M Sigma M^T is at most |M|^2 times
// the largest eigenvalue.  (r1-r3 used trace(Sigma), 1.7x the radius of an isotropic splat: a rank of an 8-way row-sharded
// frame then kept 17 % of the visible splats for 12.5 % of the rows.)  Closed form for a symmetric 3x3, in double, + 1e-5.
// S = Sigma as stored: column-major 3x3 (S[3c + r]).  0 when alpha <= 1/256 (the splat can never pass the discard test).
__host__ __device__ inline float footprint_bound(const float* S, float alpha)
{
    const float rho2 = 2.0f * logf(256.0f * alpha);
    if (!(rho2 > 0.0f)) return 0.0f;
    const double a00 = S[0], a11 = S[4], a22 = S[8];
    const double a01 = 0.5 * ((double)S[1] + S[3]), a02 = 0.5 * ((double)S[2] + S[6]), a12 = 0.5 * ((double)S[5] + S[7]);
    const double tr = a00 + a11 + a22, q = tr / 3.0;
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double p2 = (a00 - q) * (a00 - q) + (a11 - q) * (a11 - q) + (a22 - q) * (a22 - q) + 2.0 * p1;
    double lmax = q;
    if (p2 > 0.0) {
        const double p = sqrt(p2 / 6.0);
        const double b00 = (a00 - q) / p, b11 = (a11 - q) / p, b22 = (a22 - q) / p, b01 = a01 / p, b02 = a02 / p, b12 = a12 / p;
        double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
        r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        lmax = q + 2.0 * p * cos(acos(r) / 3.0);
    }
    if (!(lmax <= tr)) lmax = tr;                       // NaN / negative-eigenvalue junk: fall back to the trace (also a bound for PSD)
    if (!(lmax >= 0.0)) lmax = 0.0;
    return (float)((double)rho2 * lmax * (1.0 + 1e-5));
}

// Milder form: only GROUPS of g consecutive chunks share an XCD (workgroups p and p + 8 of every block of 8 g, which are
// dispatched right after one another): the partial cache lines at the seams between the runs that g neighbouring chunks write
// next to each other are then merged in that XCD's L2 before they go to HBM, without giving each XCD one long region of the
// output.  Identity on the last, incomplete block.
__device__ __forceinline__ uint32_t xcd_grouped(uint32_t b, uint32_t n, uint32_t g)
{
    const uint32_t blk = 8u * g, p = b % blk, base = b - p;
    if (base + blk > n) return b;
    return base + (p & 7u) * g + (p >> 3);
}

// presort_compute.glsl:38-55.  Operation order identical to oracle/msplat_oracle.c (orc_cull_key)
// so that keys and the visible set are bit-exact.
__device__ __forceinline__ bool cull_key(const float4 p, const FrameParams& fp, uint32_t& key)
{
    const float* m = fp.mvp;
    float px = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], p.x), __fmul_rn(m[4], p.y)), __fmul_rn(m[8], p.z)), m[12]);
    float py = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[1], p.x), __fmul_rn(m[5], p.y)), __fmul_rn(m[9], p.z)), m[13]);
    float pw = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[3], p.x), __fmul_rn(m[7], p.y)), __fmul_rn(m[11], p.z)), m[15]);
    float depth = pw;
    float xx = __fdiv_rn(px, depth);
    float yy = __fdiv_rn(py, depth);
    const float CLIP = 1.5f;
    if (depth > 0.0f && xx < CLIP && xx > -CLIP && yy < CLIP && yy > -CLIP) {
        if (fp.band_cull) {
            // Band-restricted cull (never active on a single GPU, where the reference's cull
            // must be reproduced exactly).  Conservative bound on the footprint's y half-extent:
            //   ey^2 = rho^2 (M1 Sigma M1^T + 0.3) <= |J1|^2 |W|^2 * (rho^2 lambda_max(Sigma)) + 0.3 rho^2_max,
            // p.w = rho^2 lambda_max(Sigma) precomputed at upload (footprint_bound; 0 when alpha <= 1/256: never visible).
            if (!(p.w > 0.0f)) return false;
            // (a bound, not parity arithmetic: v_rcp_f32 instead of IEEE divisions, the 1 ulp is inside the 0.2 % + 1.5 px margin)
            const float* v = fp.view;
            const float ty = v[1] * p.x + v[5] * p.y + v[9] * p.z + v[13];
            const float tz = v[2] * p.x + v[6] * p.y + v[10] * p.z + v[14];
            const float rtz = __builtin_amdgcn_rcpf(tz);
            const float jsy = 0.5f * fp.proj[5] * fp.H * rtz;
            const float tr = ty * rtz;
            const float j2 = jsy * jsy * (1.0f + tr * tr);
            const float ey = __builtin_amdgcn_sqrtf(j2 * fp.view_scale2 * p.w + 3.4f) * 1.002f + 1.5f;
            const float cy = 0.5f * (fp.H + yy * fp.H) + fp.Y0;
            const float y0 = fmaxf(cy - ey, 0.0f), y1 = fminf(cy + ey, fp.H - 1.0f);
            if (!(y0 <= y1)) return false;
            const int r0 = (int)y0 / kBin, r1 = (int)y1 / kBin;
            const int v0 = band_first_owned_from(fp, r0), v1 = min(band_last_owned_upto(fp, r1), fp.tiles_y - 1);
            if (v0 > v1) return false;                 // no owned row in [r0, r1]
        }
        float f = __fmul_rn(__fdiv_rn(depth, fp.zf), 4294967296.0f);
        uint32_t q = (f >= 4294967296.0f) ? 0xFFFFFFFFu : (uint32_t)f;
        key = 0xFFFFFFFFu - q;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// Chunk-level cull over a spatially ordered cloud (round 4).
// The reference culls per splat over the whole cloud (presort_compute.glsl:31-57, dispatched over N at splatrenderer.cpp:188-189)
// and so does pass 0 of the sort -- which is all of its work when most of the cloud cannot be seen: a rank of a row-sharded
// frame keeps 17 % of the splats, a camera inside a scene 40 %.  Large clouds are therefore STORED in Morton order of their
// positions (msplat_device.hip, spatial_reorder; the storage order is the library's business: sorted indices are reported in
// upload numbering and msplat_get_storage_order exposes the permutation) and every kBoxSplats consecutive stored splats carry a
// bounding box.  When an earlier frame saw less than 70 % of the cloud, Sort starts with box_cull_kernel: one thread per box,
// the live boxes listed in storage order; pass 0 then runs over the LISTED boxes only -- its chunks are made of live boxes, so
// its work is proportional to what can be seen, not to N.  (First attempt, r4: a per-chunk mask of live 1024-splat boxes tested
// inside pass 0 -- exact, but no faster: 40-64 % of such boxes are live for a rank that sees 17 % of the splats, every chunk
// still ran all its phases, and the box test sat on each chunk's critical path.)
// box_live is CONSERVATIVE: it returns false only if cull_key returns false for every splat the box can contain, so the
// visible set and the keys are exactly those of the per-splat test, and the listed boxes keep storage order, so ties do too.
// ------------------------------------------------------------------------------------------
constexpr int kBoxSplats = 256;          // stored splats per bounding box (four wave rows): 8 / 16 / 32 boxes per pass-0 chunk
constexpr int kBoxGroup = 256;           // boxes per workgroup of box_cull_kernel = entries per segment of the live list
struct CullBox {                         // 32 bytes
    float4 lo;                           // min x, y, z of the finite positions; .w = max footprint bound (pos4.w) of the box
    float4 hi;                           // max x, y, z; .w unused.  lo.x > hi.x: no finite position in the box
};

__device__ __forceinline__ bool box_live(const CullBox& b, const FrameParams& fp)
{
    if (!(b.lo.x <= b.hi.x)) return false;            // nothing finite inside: cull_key rejects NaN / inf positions (comparisons false)
    const float* m = fp.mvp;
    const float* v = fp.view;
    // clip coordinates are affine in the position: over the box every plane function takes its extremes at the corners
    float w_max = -INFINITY, w_min = INFINITY, w_mag = 0.0f;
    float xr_min = INFINITY, xl_max = -INFINITY, yt_min = INFINITY, yb_max = -INFINITY, xy_mag = 0.0f;
    float yy_min = INFINITY, yy_max = -INFINITY, tz_max = -INFINITY, ty_abs = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float x = (c & 1) ? b.hi.x : b.lo.x, y = (c & 2) ? b.hi.y : b.lo.y, z = (c & 4) ? b.hi.z : b.lo.z;
        const float px = m[0] * x + m[4] * y + m[8] * z + m[12];
        const float py = m[1] * x + m[5] * y + m[9] * z + m[13];
        const float pw = m[3] * x + m[7] * y + m[11] * z + m[15];
        const float aw = fabsf(m[3] * x) + fabsf(m[7] * y) + fabsf(m[11] * z) + fabsf(m[15]);
        const float ax = fabsf(m[0] * x) + fabsf(m[4] * y) + fabsf(m[8] * z) + fabsf(m[12]);
        const float ay = fabsf(m[1] * x) + fabsf(m[5] * y) + fabsf(m[9] * z) + fabsf(m[13]);
        w_max = fmaxf(w_max, pw); w_min = fminf(w_min, pw); w_mag = fmaxf(w_mag, aw);
        xy_mag = fmaxf(xy_mag, fmaxf(ax, ay) + 1.5f * aw);
        xr_min = fminf(xr_min, px - 1.5f * pw);       // visible needs px / pw <  1.5  <=>  px - 1.5 pw < 0  (pw > 0)
        xl_max = fmaxf(xl_max, px + 1.5f * pw);       //                px / pw > -1.5  <=>  px + 1.5 pw > 0
        yt_min = fminf(yt_min, py - 1.5f * pw);
        yb_max = fmaxf(yb_max, py + 1.5f * pw);
        if (fp.band_cull) {
            yy_min = fminf(yy_min, py / pw); yy_max = fmaxf(yy_max, py / pw);       // meaningful only when w_min > 0 (below)
            tz_max = fmaxf(tz_max, v[2] * x + v[6] * y + v[10] * z + v[14]);
            ty_abs = fmaxf(ty_abs, fabsf(v[1] * x + v[5] * y + v[9] * z + v[13]));
        }
    }
    // margins: the per-splat test evaluates the same sums in fp32 in another order (a few ulp of the sum of magnitudes)
    const float ew = 1e-5f * w_mag, exy = 1e-5f * xy_mag;
    if (!(w_max > -ew)) return false;                 // every splat has depth <= 0 (or the box is NaN: then nothing passes either)
    if (xr_min > exy || xl_max < -exy || yt_min > exy || yb_max < -exy) return false;
    if (fp.band_cull) {
        if (!(b.lo.w > 0.0f)) return false;           // every splat has alpha <= 1/256: cull_key drops them under the band cull
        // the band test needs the whole box in front of the camera (y / w is monotone along every edge only there)
        if (w_min > ew && tz_max < 0.0f) {
            const float rtz = 1.0f / -tz_max;         // largest 1 / |tz| in the box
            const float jsy = 0.5f * fabsf(fp.proj[5]) * fp.H * rtz;
            const float tr = ty_abs * rtz;
            const float j2 = jsy * jsy * (1.0f + tr * tr);
            // cull_key: ey = sqrt(j2 view_scale2 p.w + 3.4) * 1.002 + 1.5 with 1-ulp rcp / sqrt: 0.2 % + 1 px on top
            const float ey = sqrtf(j2 * fp.view_scale2 * b.lo.w + 3.4f) * 1.004f + 2.5f;
            const float cy0 = 0.5f * (fp.H + yy_min * fp.H) + fp.Y0, cy1 = 0.5f * (fp.H + yy_max * fp.H) + fp.Y0;
            const float slack = 1e-4f * (fabsf(cy0) + fabsf(cy1) + fp.H);
            const float y0 = fmaxf(cy0 - ey - slack, 0.0f), y1 = fminf(cy1 + ey + slack, fp.H - 1.0f);
            if (!(y0 <= y1)) return false;
            const int r0 = (int)y0 / kBin, r1 = (int)y1 / kBin;
            const int v0 = band_first_owned_from(fp, r0), v1 = min(band_last_owned_upto(fp, r1), fp.tiles_y - 1);
            if (v0 > v1) return false;                // no owned bin row between the box's lowest and highest reach
        }
    }
    return true;
}

// The live boxes of the current Sort: workgroup g of box_cull_kernel leaves the live ones of its kBoxGroup boxes, ascending, in
// list[g * kBoxGroup ...] and their number in cnt[g] (g < wgs <= 256: up to 2^24 splats).  list == nullptr: no list, pass 0
// walks the whole cloud.
struct LiveBoxes {
    const uint32_t* list;
    const uint32_t* cnt;
    uint32_t wgs;
    uint32_t n_storage;                  // splats in the cloud (the last box may be partial)
};

__global__ __launch_bounds__(kBoxGroup) void box_cull_kernel(const CullBox* __restrict__ boxes, uint32_t nboxes, FrameParams fp,
                                                             uint32_t* __restrict__ list, uint32_t* __restrict__ cnt)
{
    __shared__ uint32_t s_w[kBoxGroup / 64];
    const uint32_t b = blockIdx.x * kBoxGroup + threadIdx.x;
    const bool l = b < nboxes && box_live(boxes[b], fp);
    const unsigned long long m = __ballot(l);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kBoxGroup / 64; ++k) {
        if (k < w) off += s_w[k];
        total += s_w[k];
    }
    if (l) list[blockIdx.x * kBoxGroup + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = b;
    if (threadIdx.x == 0) cnt[blockIdx.x] = total;
}

// exclusive prefix of the segment counts: s_lpre[g] = live boxes before segment g, s_lpre[256] = all of them.  Every thread of a
// workgroup of >= 256 threads must call it (WAVES = its waves; s_tmp: WAVES words).  Ends with a barrier.
template <int WAVES>
__device__ __forceinline__ void live_prefix(const LiveBoxes& lb, uint32_t* s_lpre, uint32_t* s_tmp)
{
    const uint32_t t = threadIdx.x;
    uint32_t v = (t < 256u && t < lb.wgs) ? lb.cnt[t] : 0u;
    const uint32_t c = v;
    const int lane = t & 63, w = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(v, d, 64);
        if (lane >= d) v += x;
    }
    if (lane == 63) s_tmp[w] = v;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) {
        const uint32_t x = s_tmp[k];
        if (k < w) off += x;
        total += x;
    }
    if (t < 256u) s_lpre[t] = v + off - c;
    if (t == 0u) s_lpre[256] = total;
    __syncthreads();
}

// storage box number of the vb-th live box (0xFFFFFFFF beyond the last)
__device__ __forceinline__ uint32_t live_box_at(const LiveBoxes& lb, const uint32_t* s_lpre, uint32_t vb)
{
    if (vb >= s_lpre[256]) return 0xFFFFFFFFu;
    uint32_t lo = 0, hi = 255;                  // last segment g with s_lpre[g] <= vb
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (s_lpre[mid] <= vb) lo = mid; else hi = mid - 1u;
    }
    return lb.list[lo * kBoxGroup + (vb - s_lpre[lo])];
}

template <int MODE>
__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift)
{
    if (MODE == MODE_PAIR) return key >> 24;
    return (key >> shift) & 255u;
}

// Self-test for the ATOMIC_RANK paths: every lane adds 1 to a per-wave LDS counter selected by a
// pseudo-random digit; bad[0] counts lanes whose returned value is not "number of lower lanes (and
// earlier rounds) with the same digit".  Run once per context; a non-zero result selects the ballot paths.
__global__ __launch_bounds__(kThreads) void lds_atomic_order_probe(uint32_t* __restrict__ bad)
{
    __shared__ uint32_t s_c[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t errs = 0;
    for (int mod = 1; mod <= 256; mod = mod * 3 + 1) {          // 1, 4, 13, 40, 121 distinct digits
        for (int q = 0; q < 4; ++q) s_c[q][threadIdx.x] = 0;
        __syncthreads();
        uint32_t expect_base[1];
        (void)expect_base;
        for (int r = 0; r < 8; ++r) {
            uint32_t h = (uint32_t)(threadIdx.x * 2654435761u) ^ (uint32_t)(r * 40503u + blockIdx.x * 977u + mod);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const uint32_t d = (h % (uint32_t)mod) * (mod == 13 ? 32u : 1u) % 256u;    // mod 13: same-bank strides
            uint64_t m = ~0ull;
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t before = s_c[w][d];
            __builtin_amdgcn_wave_barrier();
            const uint32_t got = atomicAdd(&s_c[w][d], 1u);
            __builtin_amdgcn_wave_barrier();
            if (got != before + (uint32_t)__popcll(m & lt)) ++errs;
        }
        __syncthreads();
    }
    if (errs) atomicAdd(bad, errs);
}

// ------------------------------------------------------------------------------------------
// 8-bit-digit stable LSD radix pass: upsweep (per-chunk histograms), scan, downsweep (rank+scatter)
//   MODE_KEYS : keys from a buffer, n = *d_n
//   MODE_CULL : pass 0 -- keys computed on the fly from positions (fused presort), value = index,
//               culled splats are neither counted nor scattered (ordered compaction for free)
//   MODE_PAIR : key-only words (ty<<24 | rank), digit = top byte
// hist layout: chunk-major, hist[chunk * 256 + digit] (hist_stride = number of chunk rows allocated): the
// per-chunk kernels write / read one coalesced 1 KB row; only radix_scan walks it with a 1 KB stride, out of L2
// (digit-major rows made every upsweep write and downsweep read a 4-byte access to its own 32-byte sector:
// 8x amplification, ~40 MB of HBM traffic per frame)
// ------------------------------------------------------------------------------------------

// Group tables (the scan-free path).  A dependent launch costs ~1.5-2 us on this part and the 1 M-splat frame
// is a chain of ~5-10 us kernels, so the separate scan launch between upsweep and downsweep is dropped:
// the upsweep also adds each chunk's histogram row into the row of its GROUP of 32 chunks with global atomics
// (no return value, <= 32 adds per address: nothing serialises), and the downsweep rebuilds its chunk's exclusive
// prefix as  sum(group rows before its group) + sum(chunk rows before it inside the group)  from L2
// (<= nchunks/32 + 31 coalesced 1 KB rows).  No inter-workgroup communication inside a kernel: every table is
// complete at a kernel boundary.  A table must be zero before its upsweep: each upsweep zeroes the table its
// SUCCESSOR pass will accumulate into (gsum_zero), whose previous consumer finished one launch earlier.
// The prefix work grows with nchunks^2/32, so beyond a few thousand chunks the host picks the 3-kernel path
// (radix_scan*) instead; both are correct at any size.
constexpr int kGroupShift = 5;
// r3: a second level.  With one level a downsweep summed nchunks / 32 + 31 rows, which grows past a few thousand chunks (6 M
// splats: the column pass's 5860 rows and the row pass's 10 k rows fell back to a radix_scan launch of 30-50 us each).  Every
// table now starts with `gsup` rows of SUPERGROUP sums (128 chunks = 4 groups each), the group rows follow: an exclusive
// prefix is
//   sum(supergroup rows before the chunk's supergroup) + sum(group rows inside it before the chunk's group) + sum(chunk rows
//   inside the group before the chunk)   <=  nchunks / 128 + 3 + 31 rows,
// and the digit totals are the sum of the supergroup rows alone.  Costs the upsweep one more row of no-return atomics.  The
// supergroup must stay small: every chunk of it adds to the same row, and same-address atomics are served one per ~10 ns --
// supergroups of 1024 chunks (first attempt) put a 10 us chain on every address and cost the two binning upsweeps 30 us.
constexpr int kSuperShift = 7;

template <int MODE, int SORT_ITEMS = kSortItems>
__global__ __launch_bounds__(kThreads) void radix_upsweep(const uint32_t* __restrict__ keys,
                                                          const float4* __restrict__ pos,
                                                          const uint32_t* __restrict__ d_n, uint32_t n_static,
                                                          uint32_t n_cap, int shift,
                                                          uint32_t* __restrict__ hist, uint32_t hist_stride,
                                                          uint32_t* __restrict__ gsum_acc,
                                                          uint32_t* __restrict__ gsum_zero, uint32_t gsum_zero_rows,
                                                          FrameParams fp,
                                                          const uint32_t* __restrict__ col_totals = nullptr,
                                                          uint32_t* __restrict__ bincnt = nullptr, uint32_t gsup = 0u,
                                                          LiveBoxes lb = LiveBoxes{nullptr, nullptr, 0u, 0u})
{
    // MODE_CULL with lb.list != nullptr: pass 0 over the listed live boxes only (virtual positions), see ws_upsweep / box_live
    // MODE_PAIR with bincnt != nullptr (r3): the input is ordered by (column, rank) and carries the row in its top byte, so
    // counting the words per (row, column) here gives every bin's list length before the partition has run: the
    // downsweep's extra workgroup turns the counts into the bins' list offsets (tile_table_role) and the two launches
    // that used to derive them from the partitioned array (tile_start_kernel's searches, tile_order_kernel) are gone.
    // A chunk of 4096 words lies inside one or two columns: counts go to an LDS table of the first kPairCols columns the
    // chunk touches (one LDS atomic per word, as before) and leave the workgroup as one global atomic per non-empty
    // (row, column); words further right (tiny scenes: columns shorter than a chunk) use a global atomic each.
    constexpr int ITEMS = RadixCfg<MODE, SORT_ITEMS>::ITEMS;
    constexpr int CHUNK = RadixCfg<MODE, SORT_ITEMS>::CHUNK;
    constexpr int kPairCols = 4;
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_bin[MODE == MODE_PAIR ? kPairCols * 256 : 1];
    __shared__ uint32_t s_col[MODE == MODE_PAIR ? 257 : 1];       // first input position of each column
    __shared__ uint32_t s_tmp4[4];
    constexpr int BPC = CHUNK / kBoxSplats;
    __shared__ uint32_t s_lpre[MODE == MODE_CULL ? 257 : 1], s_box[MODE == MODE_CULL ? BPC : 1];
    const bool compact = MODE == MODE_CULL && lb.list != nullptr;
    if (gsum_zero != nullptr)
        for (uint32_t row = blockIdx.x; row < gsum_zero_rows; row += gridDim.x) gsum_zero[(size_t)row * 256 + threadIdx.x] = 0u;
    uint32_t n = d_n ? *d_n : n_static;
    if (n > n_cap) n = n_cap;
    if (compact) {
        live_prefix<kThreads / 64>(lb, s_lpre, s_tmp4);
        n = s_lpre[256] * (uint32_t)kBoxSplats;
    }
    const uint32_t nchunks = (n + CHUNK - 1) / CHUNK;
    const bool count_bins = MODE == MODE_PAIR && bincnt != nullptr;
    if (count_bins) {
        const uint32_t t = col_totals[threadIdx.x];
        uint32_t tot;
        const uint32_t incl = block_incl_scan(t, s_tmp4, tot);
        s_col[threadIdx.x] = incl - t;
        if (threadIdx.x == 255) s_col[256] = 0xFFFFFFFFu;
        __syncthreads();
    }
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (compact && threadIdx.x < (uint32_t)BPC) s_box[threadIdx.x] = live_box_at(lb, s_lpre, chunk * BPC + threadIdx.x);
        s_hist[threadIdx.x] = 0;
        if (count_bins)
#pragma unroll
            for (int j = 0; j < kPairCols; ++j) s_bin[j * 256 + threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t base = chunk * CHUNK;
        uint32_t c0 = 0;
        bool one_col = false;      // the whole chunk lies in column c0 (almost every chunk: a column holds ~D / tiles_x words)
        if (count_bins) {          // columns of the chunk's first and last word: last c with s_col[c] <= position
            const uint32_t last = min(base + (uint32_t)CHUNK, n) - 1u;
            uint32_t lo = 0, hi = 255, lo1 = 0, hi1 = 255;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint32_t mid = (lo + hi + 1u) >> 1, mid1 = (lo1 + hi1 + 1u) >> 1;
                if (s_col[mid] <= base) lo = mid; else hi = mid - 1u;
                if (s_col[mid1] <= last) lo1 = mid1; else hi1 = mid1 - 1u;
            }
            c0 = lo;
            one_col = lo1 == lo;
        }
        if (count_bins && !one_col) {
            // general form (a chunk that spans columns: tiny scenes, column boundaries): the positions of a thread ascend
            // with r, so its column only moves right
            uint32_t cw = c0;
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const uint32_t i = base + r * kThreads + threadIdx.x;
                if (i < n) {
                    const uint32_t key = keys[i];
                    while (s_col[cw + 1u] <= i) ++cw;          // s_col[256] is a sentinel
                    const uint32_t row = key >> 24, j = cw - c0;
                    if (j < (uint32_t)kPairCols) {
                        atomicAdd(&s_bin[j * 256u + row], 1u);
                    } else {
                        atomicAdd(&s_hist[row], 1u);
                        (void)__hip_atomic_fetch_add(&bincnt[cw * 256u + row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        } else {
            // straight-line form: every load of the chunk is in flight before the first LDS atomic (one_col: the counts land
            // in s_bin[0][row], i.e. column c0)
            if (MODE == MODE_CULL) {
                float4 pp[MODE == MODE_CULL ? ITEMS : 1];          // unconditional loads, all in flight together (see below)
                // storage index of virtual position i (== i without a list); `in`: the position holds a splat
                auto locate = [&](uint32_t i, bool& in) -> uint32_t {
                    if (!compact) { in = i < n; return i; }
                    const uint32_t e = i - base, bx = s_box[e / kBoxSplats], st = bx * kBoxSplats + (e % kBoxSplats);
                    in = bx != 0xFFFFFFFFu && st < lb.n_storage;
                    return st;
                };
#pragma unroll
                for (int r = 0; r < ITEMS; ++r) {
                    bool in;
                    const uint32_t st = locate(base + r * kThreads + threadIdx.x, in);
                    pp[r] = pos[in ? st : 0u];
                }
#pragma unroll
                for (int r = 0; r < ITEMS; ++r) {
                    bool in;
                    (void)locate(base + r * kThreads + threadIdx.x, in);
                    uint32_t key;
                    if (in && cull_key(pp[r], fp, key)) atomicAdd(&s_hist[digit_of<MODE>(key, shift)], 1u);
                }
            } else {
                // unconditional (clamped) loads first: under `if (i < n)` the compiler waits for every load before it
                // issues the next one (seen in the ISA: global_load, s_waitcnt vmcnt(0), ds_add, 16 times in a row)
                uint32_t kk[ITEMS];
#pragma unroll
                for (int r = 0; r < ITEMS; ++r) kk[r] = keys[min(base + r * kThreads + threadIdx.x, n - 1u)];
                if (MODE == MODE_PAIR && count_bins) {
#pragma unroll
                    for (int r = 0; r < ITEMS; ++r)
                        if (base + r * kThreads + threadIdx.x < n) atomicAdd(&s_bin[kk[r] >> 24], 1u);
                } else {
#pragma unroll
                    for (int r = 0; r < ITEMS; ++r)
                        if (base + r * kThreads + threadIdx.x < n) atomicAdd(&s_hist[digit_of<MODE>(kk[r], shift)], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t c = s_hist[threadIdx.x];
        if (count_bins) {
#pragma unroll
            for (int j = 0; j < kPairCols; ++j) {
                const uint32_t v = s_bin[j * 256 + threadIdx.x];
                c += v;
                // bincnt is [column][row] (256 rows per column): the rows of one column are consecutive words, so a wave's
                // adds touch one or two cache lines (with [row][column] every lane hit its own line: 15 us instead of 7)
                if (v != 0u)
                    (void)__hip_atomic_fetch_add(&bincnt[(c0 + j) * 256u + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        hist[(size_t)chunk * 256 + threadIdx.x] = c;
        if (gsum_acc != nullptr && c != 0u) {
            (void)__hip_atomic_fetch_add(&gsum_acc[(size_t)(gsup + (chunk >> kGroupShift)) * 256 + threadIdx.x], c,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&gsum_acc[(size_t)(chunk >> kSuperShift) * 256 + threadIdx.x], c, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

// Sum of n0 rows starting at rows0 plus n1 rows starting at rows1 (rows of 256 uint32), returned per digit
// (thread d gets digit d).  The whole workgroup cooperates: wave w takes rows w, w+4, ... and every lane loads
// 16 bytes, so one wave-load is one coalesced 1 KB row and a thread issues a quarter of the loads a
// thread-per-digit loop would (that loop cost 3-5 us per downsweep: r2 measurement); partial sums meet in `s_part`
// (256 uint4 of scratch LDS).  Contains two barriers: every thread of the workgroup must call it.
__device__ __forceinline__ uint32_t coop_row_sum(const uint32_t* __restrict__ rows0, uint32_t n0,
                                                 const uint32_t* __restrict__ rows1, uint32_t n1, uint4* s_part,
                                                 const uint32_t* __restrict__ rows2 = nullptr, uint32_t n2 = 0u)
{
    const uint32_t q = threadIdx.x & 63u, rg = threadIdx.x >> 6;
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    // four rows per step with their loads issued together, and a tail of up to three rows loaded together too (r3: the
    // remainder iterations of an unrolled loop compile to load, wait, add, load, wait, ... -- up to six memory latencies
    // on the critical path of every downsweep)
    auto sum_rows = [&](const uint32_t* __restrict__ rows, uint32_t n) {
        const uint32_t* p = rows + q * 4u;
        uint32_t r = rg;
        for (; r + 12u < n; r += 16u) {
            const uint4 x0 = *reinterpret_cast<const uint4*>(p + (size_t)r * 256);
            const uint4 x1 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 4u) * 256);
            const uint4 x2 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 8u) * 256);
            const uint4 x3 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 12u) * 256);
            acc.x += (x0.x + x1.x) + (x2.x + x3.x); acc.y += (x0.y + x1.y) + (x2.y + x3.y);
            acc.z += (x0.z + x1.z) + (x2.z + x3.z); acc.w += (x0.w + x1.w) + (x2.w + x3.w);
        }
        const uint32_t r1 = r + 4u, r2 = r + 8u;
        const bool h1 = r1 < n, h2 = r2 < n;
        if (r < n) {
            const uint4 x0 = *reinterpret_cast<const uint4*>(p + (size_t)r * 256);
            const uint4 x1 = *reinterpret_cast<const uint4*>(p + (size_t)(h1 ? r1 : r) * 256);
            const uint4 x2 = *reinterpret_cast<const uint4*>(p + (size_t)(h2 ? r2 : r) * 256);
            const uint32_t m1 = h1 ? 0xFFFFFFFFu : 0u, m2 = h2 ? 0xFFFFFFFFu : 0u;
            acc.x += x0.x + (x1.x & m1) + (x2.x & m2); acc.y += x0.y + (x1.y & m1) + (x2.y & m2);
            acc.z += x0.z + (x1.z & m1) + (x2.z & m2); acc.w += x0.w + (x1.w & m1) + (x2.w & m2);
        }
    };
    sum_rows(rows0, n0);
    sum_rows(rows1, n1);
    if (n2 != 0u) sum_rows(rows2, n2);
    s_part[rg * 64u + q] = acc;
    __syncthreads();
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(s_part);
    const uint32_t d = threadIdx.x;
    const uint32_t sum = sp[d] + sp[256u + d] + sp[512u + d] + sp[768u + d];
    __syncthreads();
    return sum;
}

// exclusive prefix of chunk `chunk`'s histogram row over the earlier chunks (scan-free path): the group rows before
// its group plus the chunk rows before it inside the group
__device__ __forceinline__ uint32_t group_prefix(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ gsum,
                                                 uint32_t chunk, uint4* s_part, uint32_t gsup)
{
    const uint32_t g = chunk >> kGroupShift, sg = chunk >> kSuperShift, g0 = sg << (kSuperShift - kGroupShift);
    return coop_row_sum(gsum, sg, gsum + (size_t)(gsup + g0) * 256, g - g0, s_part,
                        hist + (size_t)(g << kGroupShift) * 256, chunk - (g << kGroupShift));
}

// digit totals = sum of all group rows
__device__ __forceinline__ uint32_t group_total(const uint32_t* __restrict__ gsum, uint32_t nchunks, uint4* s_part)
{
    const uint32_t ns = (nchunks + (1u << kSuperShift) - 1u) >> kSuperShift;       // the supergroup rows alone
    return coop_row_sum(gsum, ns, gsum, 0u, s_part);
}

// one workgroup per digit: exclusive scan of that digit's row over the active chunks; row total -> totals
__global__ __launch_bounds__(kThreads) void radix_scan(uint32_t* __restrict__ hist, uint32_t hist_stride,
                                                       const uint32_t* __restrict__ d_n, uint32_t n_static,
                                                       uint32_t n_cap, uint32_t chunk_size,
                                                       uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_tmp[4];
    uint32_t n = d_n ? *d_n : n_static;
    if (n > n_cap) n = n_cap;
    const uint32_t nchunks = (n + chunk_size - 1) / chunk_size;
    uint32_t* col = hist + blockIdx.x;            // this digit's column of the chunk-major table
    uint32_t running = 0;
    for (uint32_t base = 0; base < nchunks; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < nchunks) ? col[(size_t)i * 256] : 0u;
        uint32_t total;
        const uint32_t incl = block_incl_scan(v, s_tmp, total);
        if (i < nchunks) col[(size_t)i * 256] = running + incl - v;
        running += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = running;
}

// The same scan for tables of at most a few thousand chunk rows (the 1 M-splat sizes): 32 workgroups, each owning 8
// digits; its 256 threads are 32 chunk ranges x 8 digits, so 8 lanes read one whole 32-byte sector of a 1 KB row
// (the one-workgroup-per-digit version above touches a separate sector per 4-byte value).  Up to 1024 chunk rows a
// thread's range fits in registers: every load is issued before the first is used and nothing is read twice.
constexpr int kScanSmallBlocks = 32;
__global__ __launch_bounds__(kThreads) void radix_scan_small(uint32_t* __restrict__ hist,
                                                             const uint32_t* __restrict__ d_n, uint32_t n_static,
                                                             uint32_t n_cap, uint32_t chunk_size,
                                                             uint32_t* __restrict__ totals)
{
    constexpr int G = 32, DIG = 8, REG = 32;
    __shared__ uint32_t s_part[G][DIG + 1];
    uint32_t n = d_n ? *d_n : n_static;
    if (n > n_cap) n = n_cap;
    const uint32_t nchunks = (n + chunk_size - 1) / chunk_size;
    const int dd = threadIdx.x & (DIG - 1), g = threadIdx.x / DIG;
    const uint32_t digit = blockIdx.x * DIG + dd;
    const uint32_t per = (nchunks + G - 1u) / G;
    const uint32_t c0 = min(nchunks, (uint32_t)g * per), c1 = min(nchunks, c0 + per);
    uint32_t* col = hist + digit;
    const bool in_regs = per <= (uint32_t)REG;          // workgroup-uniform
    uint32_t v[REG];
    uint32_t sum = 0;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < REG; ++k) v[k] = (c0 + k < c1) ? col[(size_t)(c0 + k) * 256] : 0u;
#pragma unroll
        for (int k = 0; k < REG; ++k) sum += v[k];
    } else {
#pragma unroll 8
        for (uint32_t c = c0; c < c1; ++c) sum += col[(size_t)c * 256];
    }
    s_part[g][dd] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const uint32_t p = s_part[k][dd];
        if (k < g) run += p;
        total += p;
    }
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < REG; ++k) {
            if (c0 + k < c1) col[(size_t)(c0 + k) * 256] = run;
            run += v[k];
        }
    } else {
#pragma unroll 8
        for (uint32_t c = c0; c < c1; ++c) {
            const uint32_t x = col[(size_t)c * 256];
            col[(size_t)c * 256] = run;
            run += x;
        }
    }
    if (g == 0) totals[digit] = total;
}

// The bins' list offsets and the compositors' work order from the per-bin pair counts (r3; see radix_upsweep<MODE_PAIR>).
// Run by ONE extra workgroup of the row pass's downsweep, beside the workgroups that move the pairs: the final pair
// array is ordered by (row, column) = bin index, so the offset of a bin's list is the exclusive prefix sum of the
// counts -- no search in the partitioned array -- and the counting sort of the bins by list length (heaviest first,
// what tile_order_kernel did in its own launch) reads the same numbers.  Clears the counts for the next frame and
// resets the compositors' queue heads.  tile_start gets ceil((ntiles + 1) / 1024) * 1024 entries (the tail = D).
__device__ __forceinline__ void tile_table_role(uint32_t* __restrict__ bincnt, int ntiles, int tiles_x,
                                                uint32_t* __restrict__ tile_start, uint32_t* __restrict__ order,
                                                uint32_t* __restrict__ queue, int do_order,
                                                uint32_t* s_cnt256, uint32_t* s_off256, uint32_t* s_tmp4)
{
    if (threadIdx.x < kQueueShards) queue[threadIdx.x * kQueueStride] = 0u;
    const uint32_t nblk = ((uint32_t)ntiles + 1u + 1023u) / 1024u;
    uint32_t running = 0;
    for (uint32_t b = 0; b < nblk; ++b) {
        const uint32_t i0 = b * 1024u + threadIdx.x * 4u;      // bins i0 .. i0 + 3 (bin = row * tiles_x + column)
        uint32_t c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t bin = i0 + (uint32_t)k;
            c[k] = 0u;
            if (bin < (uint32_t)ntiles) {
                const uint32_t row = bin / (uint32_t)tiles_x, col = bin - row * (uint32_t)tiles_x;
                c[k] = bincnt[col * 256u + row];                  // the counts are stored [column][row]
                bincnt[col * 256u + row] = 0u;
            }
        }
        const uint32_t local = c[0] + c[1] + c[2] + c[3];
        uint32_t total;
        const uint32_t e = running + block_incl_scan(local, s_tmp4, total) - local;
        *reinterpret_cast<uint4*>(tile_start + i0) = make_uint4(e, e + c[0], e + c[0] + c[1], e + c[0] + c[1] + c[2]);
        running += total;
    }
    if (!do_order) return;
    // bins by descending list length (counting sort on len / 16): the compositor's waves take them heaviest first
    s_cnt256[threadIdx.x] = 0u;
    __syncthreads();                 // also: this workgroup's tile_start stores are visible to all its threads
    for (int i = threadIdx.x; i < ntiles; i += kThreads) {
        const uint32_t len = tile_start[i + 1] - tile_start[i];
        atomicAdd(&s_cnt256[255u - min(len >> 4, 255u)], 1u);
    }
    __syncthreads();
    {
        const uint32_t c = s_cnt256[threadIdx.x];
        uint32_t total;
        s_off256[threadIdx.x] = block_incl_scan(c, s_tmp4, total) - c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += kThreads) {
        const uint32_t len = tile_start[i + 1] - tile_start[i];
        const uint32_t pos = atomicAdd(&s_off256[255u - min(len >> 4, 255u)], 1u);
        order[pos] = (uint32_t)i;      // order inside a bucket is irrelevant (tiles are independent)
    }
}

template <int MODE, bool HAS_VALUES, bool ATOMIC_RANK, int SORT_ITEMS = kSortItems>
__global__ __launch_bounds__(kThreads, (ATOMIC_RANK && MODE != MODE_PAIR && SORT_ITEMS == kSortItems) ? 5 : 2) void radix_downsweep(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            const float4* __restrict__ pos,
                                                            const uint32_t* __restrict__ d_n, uint32_t n_static,
                                                            uint32_t n_cap, int shift,
                                                            const uint32_t* __restrict__ hist, uint32_t hist_stride,
                                                            const uint32_t* __restrict__ totals,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out,
                                                            uint32_t* __restrict__ d_count_out,
                                                            const uint32_t* __restrict__ col_totals,
                                                            const uint32_t* __restrict__ gsum,
                                                            uint32_t* __restrict__ totals_out,
                                                            FrameParams fp,
                                                            uint32_t* __restrict__ bincnt = nullptr,
                                                            uint32_t* __restrict__ tile_start = nullptr,
                                                            uint32_t* __restrict__ tile_order = nullptr,
                                                            uint32_t* __restrict__ queue = nullptr,
                                                            int ntiles = 0, int do_order = 0, uint32_t gsup = 0u,
                                                            LiveBoxes lb = LiveBoxes{nullptr, nullptr, 0u, 0u})
{
    // lb.list != nullptr (MODE_CULL): pass 0 over the listed live boxes only (virtual positions), see ws_upsweep
    // gsum != nullptr: scan-free path -- hist holds raw per-chunk counts, prefixes come from the group tables;
    // otherwise hist holds exclusive prefixes and totals the digit totals (radix_scan*).
    // totals_out != nullptr: workgroup 0 publishes the digit totals (the row totals tile_start_kernel needs).
    // bincnt != nullptr (MODE_PAIR, r3): workgroup 0 of the grid does not move pairs, it builds the bins' list
    // offsets and work order from the counts the upsweep took (tile_table_role); the others are the workers.
    constexpr int ITEMS = RadixCfg<MODE, SORT_ITEMS>::ITEMS;
    constexpr int CHUNK = RadixCfg<MODE, SORT_ITEMS>::CHUNK;
    __shared__ uint32_t s_col[MODE == MODE_PAIR ? 256 : 1];   // MODE_PAIR: first input position of each column
    __shared__ uint32_t s_cnt[4][256];   // per-wave digit counters, then per-wave scatter bases
    __shared__ uint32_t s_base[256];     // exclusive scan of the digit totals
    __shared__ uint32_t s_gdelta[256];   // global position minus chunk-local position, per digit
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[CHUNK];
    __shared__ uint32_t s_vals[HAS_VALUES ? CHUNK : 1];
    __shared__ uint8_t s_dig[CHUNK];
    __shared__ uint32_t s_tmp[4];

    uint32_t nworkers = gridDim.x, wb = blockIdx.x;      // worker count / this workgroup's worker index
    if (MODE == MODE_PAIR && bincnt != nullptr) {
        if (blockIdx.x == 0u) {                  // workgroup-uniform; dispatched first
            tile_table_role(bincnt, ntiles, fp.tiles_x, tile_start, tile_order, queue, do_order & 1, s_cnt[0], s_base, s_tmp);
            return;
        }
        nworkers = gridDim.x - 1u;
        wb = blockIdx.x - 1u;
    }
    constexpr int BPC = CHUNK / kBoxSplats;
    __shared__ uint32_t s_lpre[MODE == MODE_CULL ? 257 : 1], s_box[MODE == MODE_CULL ? BPC : 1];
    const bool compact = MODE == MODE_CULL && lb.list != nullptr;
    uint32_t n = d_n ? *d_n : n_static;
    if (n > n_cap) n = n_cap;
    if (compact) {
        live_prefix<kThreads / 64>(lb, s_lpre, s_tmp);
        n = s_lpre[256] * (uint32_t)kBoxSplats;
    }
    const uint32_t nchunks = (n + CHUNK - 1) / CHUNK;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    {
        // (s_keys doubles as the 4 KB scratch of the cooperative row sums: it is not live before the local sort)
        const uint32_t t = (gsum != nullptr) ? group_total(gsum, nchunks, reinterpret_cast<uint4*>(s_keys)) : totals[threadIdx.x];
        uint32_t tot;
        const uint32_t incl = block_incl_scan(t, s_tmp, tot);
        s_base[threadIdx.x] = incl - t;
        if (d_count_out != nullptr && wb == 0 && threadIdx.x == 255) *d_count_out = incl;
        if (totals_out != nullptr && wb == 0) totals_out[threadIdx.x] = t;
    }
    if (MODE == MODE_PAIR) {
        const uint32_t t = col_totals[threadIdx.x];
        uint32_t tot;
        const uint32_t incl = block_incl_scan(t, s_tmp, tot);
        s_col[threadIdx.x] = incl - t;
    }
    __syncthreads();

    // do_order bit 1 (MODE_PAIR): XCD-contiguous chunk ranges, see ws_downsweep -- a column's chunks write adjacent runs
    const bool xmap = MODE == MODE_PAIR && (do_order & 2) && (nworkers >= nchunks || (nworkers & 7u) == 0u);
    for (uint32_t cidx = wb; cidx < nchunks; cidx += nworkers) {
        const uint32_t chunk = xmap ? xcd_contiguous(cidx, nchunks) : cidx;
        if (compact && threadIdx.x < (uint32_t)BPC) s_box[threadIdx.x] = live_box_at(lb, s_lpre, chunk * BPC + threadIdx.x);   // (barriers follow)
        // this chunk's exclusive prefix per digit (thread = digit): issued first, consumed after the local ranking
        const uint32_t chunk_pre = (gsum != nullptr) ? group_prefix(hist, gsum, chunk, reinterpret_cast<uint4*>(s_keys), gsup)
                                                     : hist[(size_t)chunk * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 4; ++k) s_cnt[k][threadIdx.x] = 0;
        __syncthreads();

        uint32_t key[ITEMS];
        uint32_t val[ITEMS];
        uint32_t lrank[ITEMS];
        bool valid[ITEMS];
        // wave w owns the contiguous sub-chunk [w*64*ITEMS, (w+1)*64*ITEMS): keeps the sort stable
        const uint32_t base = chunk * CHUNK + (uint32_t)w * (64 * ITEMS);
        // MODE_CULL: clamped position loads, four in flight together (n >= 1 here; all ITEMS at once would cost the kernel its
        // fifth wave per SIMD: 16-byte loads)
        constexpr int kPosBatch = 4;
        float4 pp[MODE == MODE_CULL ? kPosBatch : 1];
        // storage index of virtual position i (== i without a list); `in`: the position holds a splat
        auto locate = [&](uint32_t i, bool& in) -> uint32_t {
            if (!compact) { in = i < n; return i; }
            const uint32_t e = i - chunk * CHUNK, bx = s_box[e / kBoxSplats], st = bx * kBoxSplats + (e % kBoxSplats);
            in = bx != 0xFFFFFFFFu && st < lb.n_storage;
            return st;
        };
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (MODE == MODE_CULL && (r % kPosBatch) == 0) {
#pragma unroll
                for (int k = 0; k < kPosBatch; ++k)
                    if (r + k < ITEMS) {
                        bool in;
                        const uint32_t st = locate(base + (r + k) * 64 + lane, in);
                        pp[k] = pos[in ? st : 0u];
                    }
            }
            const uint32_t i = base + r * 64 + lane;
            valid[r] = i < n;
            key[r] = 0;
            val[r] = 0;
            if (MODE == MODE_CULL) {
                bool in;
                const uint32_t st = locate(i, in);
                valid[r] = in && cull_key(pp[r % kPosBatch], fp, key[r]);
                val[r] = st;
            } else if (valid[r]) {
                {
                    key[r] = keys_in[i];
                    if (HAS_VALUES) val[r] = vals_in[i];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t d = digit_of<MODE>(key[r], shift);
            if (ATOMIC_RANK) {
                // ds_add_rtn_u32 serves the lanes of one wave instruction in ascending lane order and a
                // wave's DS instructions in program order (verified at context creation by
                // lds_atomic_order_probe; if the probe ever fails the ballot path below is used), so the
                // returned value IS the stable local rank: 1 LDS op instead of ~45 VALU ops per key.
                lrank[r] = 0;
                if (valid[r]) lrank[r] = atomicAdd(&s_cnt[w][d], 1u);
            } else {
                uint64_t m = __ballot(valid[r]);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool bit = (d >> b) & 1u;
                    const uint64_t bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                uint32_t prev = 0;
                if (valid[r]) prev = s_cnt[w][d];
                __builtin_amdgcn_wave_barrier();
                const uint32_t rk = __popcll(m & lt_mask);
                const uint32_t cnt = __popcll(m);
                lrank[r] = prev + rk;
                if (valid[r] && rk == 0) s_cnt[w][d] = prev + cnt;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        // Local sort through LDS, then a coalesced write-out: position p of the chunk's digit-sorted
        // order goes to p + s_gdelta[digit], so neighbouring threads write neighbouring words of a
        // digit run (direct scattering cost 1.8-2.4x write amplification in 32-byte partial lines).
        uint32_t chunk_count;
        {
            const int d = threadIdx.x;
            const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
            const uint32_t tot = c0 + c1 + c2 + c3;
            const uint32_t incl = block_incl_scan(tot, s_tmp, chunk_count);
            const uint32_t excl = incl - tot;
            s_cnt[0][d] = excl;
            s_cnt[1][d] = excl + c0;
            s_cnt[2][d] = excl + c0 + c1;
            s_cnt[3][d] = excl + c0 + c1 + c2;
            s_gdelta[d] = s_base[d] + chunk_pre - excl;
        }
        __syncthreads();
        uint32_t wave_col = 0;
        if (MODE == MODE_PAIR) {       // column of the wave's first input position: last c with s_col[c] <= base
            uint32_t lo = 0, hi = 255;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (s_col[mid] <= base) lo = mid; else hi = mid - 1u;
            }
            wave_col = lo;
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (valid[r]) {
                const uint32_t d = digit_of<MODE>(key[r], shift);
                const uint32_t p = s_cnt[w][d] + lrank[r];
                uint32_t kout = key[r];
                if (MODE == MODE_PAIR) {
                    // input is ordered by (column, rank): recover the column from the input position and
                    // store (tx << 24) | rank, so each row of the result is ascending (tile_start_kernel).
                    // The wave's positions are consecutive and a column holds ~D/tiles_x words, so almost
                    // every wave sits inside one column: search once per wave, then walk.
                    const uint32_t i = base + r * 64 + lane;
                    uint32_t c = wave_col;
                    while (c < 255u && s_col[c + 1u] <= i) ++c;       // rarely iterates
                    kout = (c << 24) | (key[r] & kRankMask);
                }
                s_keys[p] = kout;
                s_dig[p] = (uint8_t)d;
                if (HAS_VALUES) s_vals[p] = val[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t p = k * kThreads + threadIdx.x;
            if (p < chunk_count) {
                const uint32_t dst = p + s_gdelta[s_dig[p]];
                keys_out[dst] = s_keys[p];
                if (HAS_VALUES) vals_out[dst] = s_vals[p];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Wide-digit sort (round 3): THREE stable LSD passes over the 32-bit depth key instead of four.
//
// Contract kept: stable ascending 32-bit key, values = splat indices (splatrenderer.cpp:165-169,223-264; the
// reference itself tried and rejected a 24-bit key, :165-167).  What changes is only how the 32 bits are cut:
//   pass 0 sorts key bits [0, 10) -- fused with the presort cull as before -- and while it computes the keys it
//   also takes the minimum key of the visible set (one atomicMin per workgroup).  key = 0xFFFFFFFF - q with
//   q = trunc(depth / far * 2^32), so with B = bit length of the largest q every key has its top 32 - B bits set:
//   only B bits can differ.  Passes 1 and 2 read that word and split the remaining max(B - 10, 16) bits in two
//   digits of 8..11 bits.  A scene whose depths stay below far / 64 (B <= 26: every BASELINE workload) is sorted
//   with digits of 10 + 8 + 8 bits; the general case (depth up to far and beyond: the key saturates at 0) with
//   10 + 11 + 11.  Exact for every input: the ignored bits are provably constant.
// One pass = upsweep + downsweep (scan-free, group tables: see radix_upsweep), so Sort = 6 launches instead of 8.
// Differences from the 8-bit kernels above, all following from the wider digit:
//   * 512 threads and 4096 / 8192-key chunks: a histogram row has up to 2048 entries, so rows must be rarer;
//   * per-wave rank counters are 16-bit halves of packed words (a wave ranks at most 64 * ITEMS <= 1024 keys per
//     digit and a chunk position is < 8192): 8 waves x 2048 digits fit in 32 KB of LDS.  The rank of a key is
//     still the return value of ONE lane-ordered LDS atomic (ds_add_rtn_u32 of 1 or 1 << 16);
//   * the digit is recomputed from the key at write-out (no digit array in LDS);
//   * pass 0's upsweep writes the key and a visibility bit per splat, so the downsweep reads 4 bytes + 1 bit per
//     splat instead of re-reading the 16-byte position and recomputing the cull (r2: 1.48x traffic in pass 0).
// Needs the lane-ordered LDS atomics (probed at msplat_create); without them the 8-bit ballot kernels are used.
// ------------------------------------------------------------------------------------------
constexpr int kWsThreads = 512;              // workgroup size for one frame at a time (8 waves, 72 / 104 KB of LDS)
constexpr int kWsThreadsSmall = 256;         // 4 waves, 40 KB: the form for contexts that share the GPU with other frames
constexpr int kWsBits0 = 10;                 // digit of pass 0: key bits [0, 10)
constexpr int kWsMinBits = 8, kWsMaxBits = 11;
constexpr int kWsMaxBins = 1 << kWsMaxBits;
constexpr int ws_qpt(int threads) { return kWsMaxBins / 4 / threads > 0 ? kWsMaxBins / 4 / threads : 1; }   // quads of digits per thread

// digit of pass `pass`: bits [shift, shift + bits) of the key.  minkey = smallest key of the visible set (pass >= 1)
__device__ __forceinline__ void ws_digit_range(int pass, uint32_t minkey, int& shift, int& bits)
{
    if (pass == 0) { shift = 0; bits = kWsBits0; return; }
    const uint32_t q = ~minkey;                          // largest quantised depth among the visible splats
    const int B = q ? 32 - __clz((int)q) : 0;            // keys differ in their low B bits only
    int rem = B - kWsBits0;
    if (rem < 2 * kWsMinBits) rem = 2 * kWsMinBits;      // at least 8 bits per pass (constant high bits sort trivially)
    const int b1 = (rem + 1) >> 1;                       // <= 11 since B <= 32
    if (pass == 1) { shift = kWsBits0; bits = b1; }
    else { shift = kWsBits0 + b1; bits = rem - b1; }
}

// inclusive scan of one uint32 per thread across a workgroup of WAVES waves (s_tmp: WAVES words); ends with a barrier
template <int WAVES>
__device__ __forceinline__ uint32_t ws_block_incl_scan(uint32_t v, uint32_t* s_tmp, uint32_t& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_tmp[w] = v;
    __syncthreads();
    uint32_t off = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) {
        const uint32_t s = s_tmp[k];
        if (k < w) off += s;
        total += s;
    }
    __syncthreads();
    return v + off;
}

// Sum of n0 rows at rows0 plus n1 rows at rows1 (rows of `nbins` uint32, nbins = 256..2048), as quads: thread t receives in
// out[k] the sums of digits 4 (t + k THREADS) .. + 3 for every quad t + k THREADS < nbins / 4 (QPT = 1 quad per thread with 512
// threads, up to 2 with 256).  A row is nbins / 4 16-byte quads; with fewer quads than threads, thread t loads quad t % Q of
// the rows t / Q, t / Q + THREADS / Q, ... (coalesced) and the partial sums meet in s_part (THREADS x QPT uint4).  Two barriers.
template <int THREADS>
__device__ __forceinline__ void ws_row_sum(const uint32_t* __restrict__ rows0, uint32_t n0,
                                           const uint32_t* __restrict__ rows1, uint32_t n1, uint32_t nbins, int bits,
                                           uint4* s_part, uint4 (&out)[ws_qpt(THREADS)])
{
    constexpr int QPT = ws_qpt(THREADS);
    const uint32_t Q = nbins >> 2;
    const bool wide = Q >= (uint32_t)THREADS;                               // workgroup-uniform
    const uint32_t RL = wide ? 1u : ((uint32_t)THREADS >> (bits - 2));      // row lanes
    const uint32_t q = wide ? threadIdx.x : (threadIdx.x & (Q - 1u)), rl = wide ? 0u : (threadIdx.x >> (bits - 2));
    uint4 acc[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) acc[k] = make_uint4(0u, 0u, 0u, 0u);
    // four rows per step, their loads issued together (a plain `for (r += RL)` loop compiles to load, wait, add, load, ...:
    // one memory latency per row on the critical path of every downsweep)
    auto sum_rows = [&](const uint32_t* __restrict__ rows, uint32_t n, int k) {
        const uint32_t* p = rows + (q + (uint32_t)k * THREADS) * 4u;
        uint4& a = acc[k];
        uint32_t r = rl;
        for (; r + 3u * RL < n; r += 4u * RL) {
            const uint4 x0 = *reinterpret_cast<const uint4*>(p + (size_t)r * nbins);
            const uint4 x1 = *reinterpret_cast<const uint4*>(p + (size_t)(r + RL) * nbins);
            const uint4 x2 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 2u * RL) * nbins);
            const uint4 x3 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 3u * RL) * nbins);
            a.x += (x0.x + x1.x) + (x2.x + x3.x); a.y += (x0.y + x1.y) + (x2.y + x3.y);
            a.z += (x0.z + x1.z) + (x2.z + x3.z); a.w += (x0.w + x1.w) + (x2.w + x3.w);
        }
        // tail: up to three rows, loaded together (the clamped row is added with weight 0)
        const uint32_t r1 = r + RL, r2 = r + 2u * RL;
        const bool h1 = r1 < n, h2 = r2 < n;
        if (r < n) {
            const uint4 x0 = *reinterpret_cast<const uint4*>(p + (size_t)r * nbins);
            const uint4 x1 = *reinterpret_cast<const uint4*>(p + (size_t)(h1 ? r1 : r) * nbins);
            const uint4 x2 = *reinterpret_cast<const uint4*>(p + (size_t)(h2 ? r2 : r) * nbins);
            const uint32_t m1 = h1 ? 0xFFFFFFFFu : 0u, m2 = h2 ? 0xFFFFFFFFu : 0u;
            a.x += x0.x + (x1.x & m1) + (x2.x & m2); a.y += x0.y + (x1.y & m1) + (x2.y & m2);
            a.z += x0.z + (x1.z & m1) + (x2.z & m2); a.w += x0.w + (x1.w & m1) + (x2.w & m2);
        }
    };
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        if (k == 0 || q + (uint32_t)k * THREADS < Q) {
            sum_rows(rows0, n0, k);
            sum_rows(rows1, n1, k);
        }
    }
    if (wide) {               // every thread already holds the complete sums of its own quads
#pragma unroll
        for (int k = 0; k < QPT; ++k) out[k] = acc[k];
        return;
    }
    s_part[threadIdx.x] = acc[0];              // == s_part[rl * Q + q]
    __syncthreads();
    uint4 sum = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < Q)
        for (uint32_t k = 0; k < RL; ++k) {
            const uint4 x = s_part[k * Q + threadIdx.x];
            sum.x += x.x; sum.y += x.y; sum.z += x.z; sum.w += x.w;
        }
    __syncthreads();
    out[0] = sum;
#pragma unroll
    for (int k = 1; k < QPT; ++k) out[k] = make_uint4(0u, 0u, 0u, 0u);
}

// CULL: pass 0.  keys are computed from the positions (presort_compute.glsl:38-55 via cull_key), written to raw_keys
// together with one visibility bit per splat (vmask: one uint64 per 64 splats), and their minimum goes to *minkey_cur.
template <bool CULL, int ITEMS, int THREADS = kWsThreads>
__global__ __launch_bounds__(THREADS) void ws_upsweep(const uint32_t* __restrict__ keys_in,
                                                      const float4* __restrict__ pos,
                                                      uint32_t* __restrict__ raw_keys,
                                                      unsigned long long* __restrict__ vmask,
                                                      const uint32_t* __restrict__ d_n, uint32_t n_static, uint32_t n_cap,
                                                      int pass, uint32_t* __restrict__ minkey_cur,
                                                      uint32_t* __restrict__ minkey_next,
                                                      uint32_t* __restrict__ hist,
                                                      uint32_t* __restrict__ gsum_acc, int gshift,
                                                      uint32_t* __restrict__ gsum_zero, uint32_t gsum_zero_words,
                                                      FrameParams fp, LiveBoxes lb = LiveBoxes{nullptr, nullptr, 0u, 0u})
{
    // lb.list != nullptr (CULL, spatially ordered cloud, box_cull_kernel has run): the pass walks the LISTED boxes only.  A chunk
    // is BPC consecutive live boxes; element e of chunk c is splat box[c * BPC + e / kBoxSplats] * kBoxSplats + e % kBoxSplats;
    // raw_keys / vmask / the histogram rows are indexed by the VIRTUAL position c * CHUNK + e, which is dense.
    constexpr int CHUNK = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    constexpr int BPC = CHUNK / kBoxSplats;                    // boxes per chunk: 8, 16 or 32
    static_assert(CHUNK % kBoxSplats == 0 && kBoxSplats % 64 == 0 && THREADS >= 256, "a wave row lies in one box");
    __shared__ uint32_t s_hist[kWsMaxBins];
    __shared__ uint32_t s_min[WAVES];
    __shared__ uint32_t s_lpre[CULL ? 257 : 1], s_box[CULL ? BPC : 1], s_tmpw[WAVES];
    const bool compact = CULL && lb.list != nullptr;
    // the group table of the pass before this one (its consumer finished one launch ago) is cleared for the next frame
    if (gsum_zero != nullptr)
        for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < gsum_zero_words; i += gridDim.x * THREADS) gsum_zero[i] = 0u;
    if (CULL && blockIdx.x == 0 && threadIdx.x == 0) *minkey_next = 0xFFFFFFFFu;      // the other frame parity's word
    uint32_t n = d_n ? *d_n : n_static;
    if (n > n_cap) n = n_cap;
    if (compact) {
        live_prefix<WAVES>(lb, s_lpre, s_tmpw);
        n = s_lpre[256] * (uint32_t)kBoxSplats;                 // virtual positions (the cloud's last box may be partial: see `in`)
    }
    int shift, bits;
    ws_digit_range(pass, CULL ? 0u : *minkey_cur, shift, bits);
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    const uint32_t nchunks = (n + CHUNK - 1) / CHUNK;
    uint32_t mk = 0xFFFFFFFFu;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        for (uint32_t d = threadIdx.x; d < nbins; d += THREADS) s_hist[d] = 0u;
        if (compact && threadIdx.x < (uint32_t)BPC) s_box[threadIdx.x] = live_box_at(lb, s_lpre, chunk * BPC + threadIdx.x);
        __syncthreads();
        // unconditional (clamped) loads first, so that all of them are in flight together: under `if (i < n)` the
        // compiler waits for each load before it issues the next (r3, seen in the ISA).  The cull pass over 8192-key chunks
        // takes its chunk in two halves: 16 positions in flight cost 126 VGPRs = 2 workgroups per CU = 512 slots for the 733
        // chunks of 6 M splats (a second, half-empty round); 8 in flight fit 3 per CU.
        constexpr int SUB = (CULL && ITEMS == 16) ? 2 : 1;
        constexpr int IPS = ITEMS / SUB;
#pragma unroll 1
        for (int sub = 0; sub < SUB; ++sub) {
        const uint32_t base = chunk * CHUNK + (uint32_t)sub * (IPS * THREADS);
        float4 pp[CULL ? IPS : 1];
        uint32_t kk[CULL ? 1 : IPS];
        // storage index of virtual position i (== i without a list); `in`: the position holds a splat
        auto locate = [&](uint32_t i, bool& in) -> uint32_t {
            if (!compact) { in = i < n; return i; }
            const uint32_t e = i - chunk * CHUNK, bx = s_box[e / kBoxSplats], st = bx * kBoxSplats + (e % kBoxSplats);
            in = bx != 0xFFFFFFFFu && st < lb.n_storage;
            return st;
        };
#pragma unroll
        for (int r = 0; r < IPS; ++r) {
            const uint32_t i = base + r * THREADS + threadIdx.x;
            if (CULL) {
                bool in;
                const uint32_t st = locate(i, in);
                pp[r] = pos[in ? st : 0u];                                              // (the cloud has >= 1 splat inside this loop)
            } else {
                kk[r] = keys_in[min(i, n - 1u)];                                        // n >= 1 inside this loop
            }
        }
#pragma unroll
        for (int r = 0; r < IPS; ++r) {
            const uint32_t i = base + r * THREADS + threadIdx.x;
            uint32_t key = 0u;
            bool ok = false;
            if (CULL) {
                bool in;
                (void)locate(i, in);
                if (in) ok = cull_key(pp[r], fp, key);
            } else if (i < n) {
                key = kk[r];
                ok = true;
            }
            if (CULL) {
                const unsigned long long m = __ballot(ok);
                if (i < n) {
                    raw_keys[i] = key;
                    if ((threadIdx.x & 63) == 0) vmask[i >> 6] = m;        // i is a multiple of 64 here
                    if (ok) mk = min(mk, key);
                }
            }
            if (ok) atomicAdd(&s_hist[(key >> shift) & dmask], 1u);
        }
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < nbins; d += THREADS) {
            const uint32_t c = s_hist[d];
            hist[(size_t)chunk * nbins + d] = c;
            if (c != 0u)
                (void)__hip_atomic_fetch_add(&gsum_acc[(size_t)(chunk >> gshift) * nbins + d], c, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (CULL) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mk = min(mk, (uint32_t)__shfl_xor((int)mk, d, 64));
        if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = mk;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = s_min[0];
#pragma unroll
            for (int k = 1; k < WAVES; ++k) m = min(m, s_min[k]);
            if (m != 0xFFFFFFFFu) (void)atomicMin(minkey_cur, m);
        }
    }
}

// dynamic LDS of ws_downsweep<., ITEMS, THREADS>: keys + values of the chunk, packed per-wave counters, per-digit deltas,
// scan scratch
constexpr size_t ws_downsweep_lds(int items, int threads = kWsThreads)
{
    return (size_t)threads * items * 8 + (size_t)(threads / 64) * (kWsMaxBins / 2) * 4 + (size_t)kWsMaxBins * 4 + 64;
}

template <bool CULL, int ITEMS, int THREADS = kWsThreads>
__global__ __launch_bounds__(THREADS, ITEMS == 8 ? 4 : 2) void ws_downsweep(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, const unsigned long long* __restrict__ vmask,
    const uint32_t* __restrict__ d_n, uint32_t n_static, uint32_t n_cap, int pass, const uint32_t* __restrict__ minkey_cur,
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ gsum, int gshift, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, uint32_t* __restrict__ d_count_out, int xcd_map, LiveBoxes lb = LiveBoxes{nullptr, nullptr, 0u, 0u})
{
    // lb.list != nullptr (CULL): pass 0 over the listed boxes only, see ws_upsweep -- keys_in / vmask are indexed by virtual
    // position, the value written is the splat's STORAGE index.
    // xcd_map: workgroup b runs on XCD b % 8; chunk = xcd_contiguous(b) gives every XCD a contiguous range of chunks, so
    // the digit runs that neighbouring chunks write next to each other meet in ONE L2 instead of being written to HBM
    // as partial lines by several (the per-XCD L2s are not coherent; every one writes back its own bytes of a shared line)
    constexpr int CHUNK = THREADS * ITEMS;
    constexpr int WAVES = THREADS / 64;
    constexpr int QPT = ws_qpt(THREADS);                        // quads (4 digits) per thread in the per-digit steps: 1 or 2
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint32_t* s_keys = s_dyn;                                   // CHUNK
    uint32_t* s_vals = s_keys + CHUNK;                          // CHUNK
    uint32_t* s_cnt = s_vals + CHUNK;                           // WAVES x (nbins / 2) packed 16-bit counters, then bases
    uint32_t* s_gd = s_cnt + WAVES * (kWsMaxBins / 2);          // nbins: global position minus chunk-local position
    uint32_t* s_tmp = s_gd + kWsMaxBins;                        // WAVES words
    uint4* s_part = reinterpret_cast<uint4*>(s_keys);           // THREADS uint4 of row-sum scratch (s_keys not live yet)
    constexpr int BPC = CHUNK / kBoxSplats;
    __shared__ uint32_t s_lpre[CULL ? 257 : 1], s_box[CULL ? BPC : 1];
    const bool compact = CULL && lb.list != nullptr;

    uint32_t n = d_n ? *d_n : n_static;
    if (n > n_cap) n = n_cap;
    if (compact) {
        live_prefix<WAVES>(lb, s_lpre, s_tmp);
        n = s_lpre[256] * (uint32_t)kBoxSplats;
    }
    int shift, bits;
    ws_digit_range(pass, CULL ? 0u : *minkey_cur, shift, bits);
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u, Q = nbins >> 2, half = nbins >> 1;
    const uint32_t nchunks = (n + CHUNK - 1) / CHUNK;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    // digit totals = sum of all group rows; their exclusive scan = where each digit's run starts (kept in registers).
    // Thread t owns the quads t + k THREADS (k < QPT) that exist; the scan runs over the quads in (k, t) order, i.e. the
    // digits in ascending order: first all k = 0 quads, then -- offset by their total -- the k = 1 quads.
    uint32_t gbase[QPT][4];
    {
        const uint32_t ng = (nchunks + (1u << gshift) - 1u) >> gshift;
        uint4 tot[QPT];
        ws_row_sum<THREADS>(gsum, ng, gsum, 0u, nbins, bits, s_part, tot);
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const bool own = (uint32_t)t + (uint32_t)k * THREADS < Q;
            const uint32_t tsum = own ? tot[k].x + tot[k].y + tot[k].z + tot[k].w : 0u;
            uint32_t total;
            const uint32_t e = run + ws_block_incl_scan<WAVES>(tsum, s_tmp, total) - tsum;
            gbase[k][0] = e; gbase[k][1] = e + tot[k].x; gbase[k][2] = gbase[k][1] + tot[k].y; gbase[k][3] = gbase[k][2] + tot[k].z;
            run += total;
        }
        if (d_count_out != nullptr && blockIdx.x == 0 && t == 0) *d_count_out = run;
    }

    for (uint32_t cidx = blockIdx.x; cidx < nchunks; cidx += gridDim.x) {
        const uint32_t chunk = (xcd_map && (gridDim.x >= nchunks || (gridDim.x & 7u) == 0u)) ? xcd_contiguous(cidx, nchunks) : cidx;
        if (compact && t < BPC) s_box[t] = live_box_at(lb, s_lpre, chunk * BPC + (uint32_t)t);      // (barriers follow before its use)
        // this chunk's exclusive prefix per digit: the group rows before its group + the chunk rows before it in the group
        const uint32_t g = chunk >> gshift;
        uint4 pre[QPT];
        ws_row_sum<THREADS>(gsum, g, hist + (size_t)(g << gshift) * nbins, chunk - (g << gshift), nbins, bits, s_part, pre);
        __syncthreads();            // (the wide form of ws_row_sum has no barrier: s_cnt below is not the scratch, but keep the phases apart)
        for (uint32_t i = t; i < (uint32_t)WAVES * (nbins >> 3); i += THREADS) reinterpret_cast<uint4*>(s_cnt)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();

        uint32_t key[ITEMS], val[ITEMS], lrank[ITEMS];
        bool valid[ITEMS];
        // wave w owns the contiguous sub-chunk [w * 64 * ITEMS, (w + 1) * 64 * ITEMS): keeps the sort stable
        const uint32_t base = chunk * CHUNK + (uint32_t)w * (64 * ITEMS);
        // unconditional (clamped) loads, all in flight together (see ws_upsweep); n >= 1 inside this loop
        unsigned long long vm[CULL ? ITEMS : 1];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t ic = min(base + r * 64 + lane, n - 1u);
            key[r] = keys_in[ic];
            if (CULL) vm[r] = vmask[min(base + r * 64, n - 1u) >> 6];          // wave-uniform address
            else val[r] = vals_in[ic];
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = base + r * 64 + lane;
            valid[r] = i < n;
            if (CULL) {
                valid[r] = valid[r] && ((vm[r] >> lane) & 1ull);
                val[r] = i;
                if (compact) {                        // virtual position -> storage index (a row of 64 lies in one box)
                    const uint32_t e = i - chunk * CHUNK;
                    val[r] = s_box[e / kBoxSplats] * kBoxSplats + (e % kBoxSplats);
                }
            }
        }
        uint32_t* wcnt = s_cnt + (uint32_t)w * half;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            // ds_add_rtn_u32 serves the lanes of one wave instruction in ascending lane order and a wave's DS instructions
            // in program order (lds_atomic_order_probe), so the returned half-word IS the stable rank inside the wave
            const uint32_t d = (key[r] >> shift) & dmask, sh = (d & 1u) << 4;
            lrank[r] = 0u;
            if (valid[r]) lrank[r] = (atomicAdd(&wcnt[d >> 1], 1u << sh) >> sh) & 0xFFFFu;
        }
        __syncthreads();
        // per digit: counts of the waves -> chunk-local exclusive positions -> per-wave bases (16 bit: < CHUNK <= 8192)
        uint32_t chunk_count = 0;
#pragma unroll
        for (int kq = 0; kq < QPT; ++kq) {
            const uint32_t qd = (uint32_t)t + (uint32_t)kq * THREADS;            // this thread's quad (digits 4 qd .. 4 qd + 3)
            const bool own = qd < Q;
            // (the waves' counts are read twice -- once for the totals, once for the bases -- instead of being kept: 16 waves
            //  x 4 digits would be 64 registers)
            uint32_t tot[4] = {0u, 0u, 0u, 0u};
            if (own) {
#pragma unroll
                for (int k = 0; k < WAVES; ++k) {
                    const uint2 x = *reinterpret_cast<const uint2*>(s_cnt + (uint32_t)k * half + 2u * qd);
                    tot[0] += x.x & 0xFFFFu; tot[1] += x.x >> 16; tot[2] += x.y & 0xFFFFu; tot[3] += x.y >> 16;
                }
            }
            const uint32_t tsum = tot[0] + tot[1] + tot[2] + tot[3];
            uint32_t part_total;
            const uint32_t e = chunk_count + ws_block_incl_scan<WAVES>(tsum, s_tmp, part_total) - tsum;
            chunk_count += part_total;
            if (own) {
                uint32_t run[4] = {e, e + tot[0], e + tot[0] + tot[1], e + tot[0] + tot[1] + tot[2]};
                const uint32_t pr[4] = {pre[kq].x, pre[kq].y, pre[kq].z, pre[kq].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) s_gd[4u * qd + j] = gbase[kq][j] + pr[j] - run[j];
#pragma unroll
                for (int k = 0; k < WAVES; ++k) {
                    uint2* slot = reinterpret_cast<uint2*>(s_cnt + (uint32_t)k * half + 2u * qd);
                    const uint2 c = *slot;
                    uint2 x;
                    x.x = run[0] | (run[1] << 16);
                    x.y = run[2] | (run[3] << 16);
                    *slot = x;
                    run[0] += c.x & 0xFFFFu; run[1] += c.x >> 16; run[2] += c.y & 0xFFFFu; run[3] += c.y >> 16;
                }
            }
        }
        __syncthreads();
        // local sort through LDS, then a coalesced write-out: position p of the chunk's digit-sorted order goes to
        // p + s_gd[digit], so neighbouring threads write neighbouring words of a digit run
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (valid[r]) {
                const uint32_t d = (key[r] >> shift) & dmask, sh = (d & 1u) << 4;
                const uint32_t p = ((wcnt[d >> 1] >> sh) & 0xFFFFu) + lrank[r];
                s_keys[p] = key[r];
                s_vals[p] = val[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t p = k * THREADS + t;
            if (p < chunk_count) {
                const uint32_t kk = s_keys[p];
                const uint32_t dst = p + s_gd[(kk >> shift) & dmask];
                keys_out[dst] = kk;
                vals_out[dst] = s_vals[p];
            }
        }
        __syncthreads();
    }
}

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

template <bool FULL_SH>
__global__ __launch_bounds__(kProjThreads) void project_kernel(const uint32_t* __restrict__ sorted_idx,
                                                               const uint32_t* __restrict__ d_V,
                                                               const float4* __restrict__ recs,
                                                               FrameParams fp,
                                                               float4* __restrict__ out_rec,
                                                               uint32_t* __restrict__ out_rect,
                                                               uint32_t* __restrict__ out_zq, uint32_t* __restrict__ d_Veff = nullptr)
{
    // Records are 256 B (full SH) or 128 B (base) and line aligned.  The gather by sorted index is
    // done cooperatively: F4 consecutive lanes fetch one whole record (coalesced 256/128 B), the wave
    // stages 64 records in LDS, then every lane reads its own record back (stride 68/36 dwords keeps
    // the ds_read_b128 accesses conflict free).
    constexpr int F4 = FULL_SH ? 16 : 8;
    constexpr int RPI = 64 / F4;              // records fetched per wave-wide load instruction
    constexpr int STRIDE = F4 * 4 + 4;        // dwords
    __shared__ __attribute__((aligned(16))) float s_stage[64 * STRIDE];
    const uint32_t V = *d_V;
    const int lane = threadIdx.x;
    // two views in one chain (FrameParams.views == 2): ranks [0, V) are view 0, [V1, V1 + V) view 1; the gap gets empty rectangles
    const uint32_t V1 = (V + 63u) & ~63u;
    const uint32_t total = fp.views == 2 ? V1 + V : V;
    if (d_Veff != nullptr && blockIdx.x == 0 && lane == 0) *d_Veff = total;       // what the binning passes walk
    if (blockIdx.x * kProjThreads >= total) return;
    const uint32_t r = blockIdx.x * kProjThreads + lane;
    const bool second = fp.views == 2 && blockIdx.x * kProjThreads >= V1;         // wave-uniform
    const uint32_t rl = second ? r - V1 : r;                                       // rank inside the view
    const bool valid = rl < V;
    if (!valid && r < total) out_rect[r] = kRectEmpty;                             // (the gap between the views, and nothing else)
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
    const float* vm = second ? fp.view1 : fp.view;
    const float* pm = second ? fp.proj1 : fp.proj;
    const float* eye = second ? fp.eye1 : fp.eye;

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
            if (second) { ty0 += fp.rows_view; ty1 += fp.rows_view; }         // the second view's bins follow the first's
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

// ------------------------------------------------------------------------------------------
// tile binning.  The splats are already in global depth order (rank).  Two STABLE partitions of the
// (splat, tile) pairs -- first by tile column, then by tile row -- leave every tile's list in
// draw order without ever sorting on depth again:
//   pass 1 (bin1_*):  pairs are enumerated on the fly from the rank-ordered rectangles and
//                     partitioned by column tx;       word = (row << 24) | rank
//   pass 2 (radix_*<MODE_PAIR>): partition by the row byte; the downsweep rewrites the word to
//                     (tx << 24) | rank (tx recovered from the input position), so that inside a
//                     row the words are ascending and tile_start_kernel can binary-search them.
// ------------------------------------------------------------------------------------------

template <int BIN_CHUNK>
__global__ __launch_bounds__(kThreads) void bin1_upsweep(const uint32_t* __restrict__ rect,
                                                         const uint32_t* __restrict__ d_V,
                                                         uint32_t* __restrict__ hist, uint32_t hist_stride,
                                                         uint32_t* __restrict__ d_overflow,
                                                         uint32_t* __restrict__ gsum_acc,
                                                         uint32_t* __restrict__ gsum_zero, uint32_t gsum_zero_rows,
                                                         uint32_t* __restrict__ heavy, uint32_t* __restrict__ heavy_next,
                                                         uint8_t* __restrict__ heavy_flag, uint32_t heavy_slots, uint32_t gsup)
{
    // Heavy chunks (r3).  The ranks are in depth order, so the huge far-away splats of a real scene (sky, background) are the
    // FIRST ranks: a few chunks hold half of all the pairs (scene-like 6 M cloud: 25 of 2344 chunks, 500 k pairs each against
    // 11 k), and the column pass lasted as long as the slowest of them.  A chunk with more than kHeavyPairs pairs is put on a
    // list (heavy[0] = count, heavy[1..] = chunk numbers, order irrelevant) and bin1_downsweep gives it kHeavyParts workgroups,
    // one per block of columns: columns are independent in that pass (a cursor per column), so the parts need no hand-off.
    // heavy_next is the other frame parity's counter: cleared here for the next frame.  heavy_slots <= kHeavyCap = the split
    // chunks the downsweep's grid has helper workgroups for (the host sizes it from an earlier frame's count; a chunk that
    // gets no slot is processed unsplit -- slower, never wrong).
    // per-frame reset of the sticky overflow flag (set later in the frame by bin1_downsweep): saves a memset launch
    if (blockIdx.x == 0 && threadIdx.x == 0) { *d_overflow = 0u; heavy_next[0] = 0u; }
    if (gsum_zero != nullptr)      // scan-free path, see radix_upsweep
        for (uint32_t row = blockIdx.x; row < gsum_zero_rows; row += gridDim.x) gsum_zero[(size_t)row * 256 + threadIdx.x] = 0u;
    __shared__ uint32_t s_diff[kThreads + 1];
    __shared__ uint32_t s_tmp[4];
    const uint32_t V = *d_V;
    const uint32_t nchunks = (V + BIN_CHUNK - 1) / BIN_CHUNK;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        s_diff[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_diff[kThreads] = 0;
        __syncthreads();
        uint32_t rcs[BIN_CHUNK / kThreads];          // clamped loads, all in flight together (V >= 1 here)
#pragma unroll
        for (int k = 0; k < BIN_CHUNK / kThreads; ++k) rcs[k] = rect[min(chunk * BIN_CHUNK + k * kThreads + threadIdx.x, V - 1u)];
#pragma unroll
        for (int k = 0; k < BIN_CHUNK / kThreads; ++k) {
            const uint32_t r = chunk * BIN_CHUNK + k * kThreads + threadIdx.x;
            if (r < V) {
                const uint32_t rc = rcs[k];
                const uint32_t tx0 = rc & 255u, ty0 = (rc >> 8) & 255u, tx1 = (rc >> 16) & 255u, ty1 = rc >> 24;
                if (tx0 <= tx1) {
                    // pairs per column = sum of row counts of the rectangles covering it: difference array
                    const uint32_t rows = ty1 - ty0 + 1u;
                    atomicAdd(&s_diff[tx0], rows);
                    atomicAdd(&s_diff[tx1 + 1u], 0u - rows);
                }
            }
        }
        __syncthreads();
        uint32_t total;
        const uint32_t incl = block_incl_scan(s_diff[threadIdx.x], s_tmp, total);   // wraps mod 2^32: exact
        hist[(size_t)chunk * 256 + threadIdx.x] = incl;
        if (gsum_acc != nullptr && incl != 0u) {
            (void)__hip_atomic_fetch_add(&gsum_acc[(size_t)(gsup + (chunk >> kGroupShift)) * 256 + threadIdx.x], incl,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&gsum_acc[(size_t)(chunk >> kSuperShift) * 256 + threadIdx.x], incl, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t psum = incl;                                  // pairs of this chunk = sum of its column counts
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) psum += __shfl_xor(psum, d, 64);
        if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = psum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint8_t flag = 0;
            if (s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3] > kHeavyPairs) {
                const uint32_t slot = atomicAdd(&heavy[0], 1u);
                if (slot < heavy_slots) { heavy[1u + slot] = chunk; flag = 1; }
            }
            if (heavy_slots != 0u) heavy_flag[chunk] = flag;
        }
        __syncthreads();
    }
}

// Splat-parallel stable partition by column.  Items = (rank, column) "column pairs" in (rank, tx)
// order, weight = number of tile rows; wave w takes a contiguous quarter of the chunk's items, so
// (wave, round, lane) order == item order.  Ranking inside a wave: ballot-match on the column byte,
// weighted prefix from 9 ballots over the bits of the weight (rows <= 256).
// (5 waves per SIMD = 5 workgroups per CU, as in r2: the r3 additions had pushed the kernel to 106 VGPRs = 4, which cost
//  the frames-in-flight mode throughput)
template <bool ATOMIC_RANK, int BIN_CHUNK>
__global__ __launch_bounds__(kThreads, (ATOMIC_RANK && BIN_CHUNK == kBinChunk) ? 5 : 2) void bin1_downsweep(const uint32_t* __restrict__ rect,
                                                           const uint32_t* __restrict__ d_V,
                                                           const uint32_t* __restrict__ hist, uint32_t hist_stride,
                                                           const uint32_t* __restrict__ totals,
                                                           uint32_t* __restrict__ pairs_out, uint32_t cap,
                                                           uint32_t* __restrict__ d_D,
                                                           uint32_t* __restrict__ d_overflow,
                                                           uint32_t* __restrict__ host_words, int report_overflow,
                                                           const uint32_t* __restrict__ gsum,
                                                           uint32_t* __restrict__ totals_out, int xcd_map,
                                                           const uint32_t* __restrict__ heavy,
                                                           const uint8_t* __restrict__ heavy_flag, uint32_t nhelp, int tiles_x,
                                                           uint32_t gsup, const uint32_t* __restrict__ d_V_report = nullptr)
{
    // d_V_report: the Sort's own V for the host-mapped hint (with two views in one chain d_V counts the ranks of both)
    // The first nhelp workgroups are helpers for the heavy chunks (bin1_upsweep; first, so that they start with the launch):
    // helper h takes column block 1 + h % (kHeavyParts - 1) of chunk heavy[1 + h / (kHeavyParts - 1)] and exits at once when
    // there is no such chunk; the other nmain workgroups walk the chunks (grid-stride), a heavy chunk's main workgroup keeps
    // block 0.  A part sees every rectangle of the chunk clipped to its columns.
    // gsum != nullptr: scan-free path (hist = raw per-chunk column counts, see radix_upsweep); workgroup 0 then also
    // publishes the column totals in totals_out for the row pass.  host_words (host-mapped): [0] pairs needed by an
    // overflowed device-output frame, [1] V and [2] D of the latest frame (read by the host without synchronising,
    // only to choose between the scan-free and the 3-kernel path for the NEXT frame's row pass)
    constexpr int PER = BIN_CHUNK / kThreads;          // rectangles per thread (blocked)
    __shared__ uint32_t s_off[BIN_CHUNK + 1];          // exclusive scan of the rectangle widths
    __shared__ uint32_t s_rect[BIN_CHUNK];
    __shared__ uint32_t s_cnt[4][256];                 // per-wave column weights, then per-wave cursors
    __shared__ uint32_t s_base[kThreads];
    __shared__ uint32_t s_tmp[4];
    // item -> owner rectangle table (chunks with at most kOwnerCap items; larger ones binary-search s_off):
    // one LDS read per item instead of a 10-step dependent search, twice per item
    constexpr uint32_t kOwnerCap = 8u * BIN_CHUNK;       // 8192 / 16384 items: 16 / 32 KB
    __shared__ __attribute__((aligned(16))) uint16_t s_owner[kOwnerCap];
    uint4* s_part = reinterpret_cast<uint4*>(s_owner);      // 16 KB, not live while the row sums run
    const uint32_t V = *d_V;
    const uint32_t nchunks = (V + BIN_CHUNK - 1) / BIN_CHUNK;
    const bool helper = blockIdx.x < nhelp;
    const uint32_t nmain = gridDim.x - nhelp, mb = blockIdx.x - nhelp;       // main workgroups / this one's index among them
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    {
        // (s_part: 4 KB scratch for the cooperative row sums of the scan-free path)
        const uint32_t t = (gsum != nullptr) ? group_total(gsum, nchunks, s_part) : totals[threadIdx.x];
        uint32_t tot;
        const uint32_t incl = block_incl_scan(t, s_tmp, tot);
        s_base[threadIdx.x] = incl - t;
        if (totals_out != nullptr && !helper && mb == 0u) totals_out[threadIdx.x] = t;
        if (!helper && mb == 0u && threadIdx.x == 255) {
            *d_D = incl;
            if (host_words != nullptr) {
                __hip_atomic_store(host_words + 1, d_V_report ? *d_V_report : V, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(host_words + 2, incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(host_words + 3, heavy[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // heavy chunks wanted
            }
            if (incl > cap) {
                *d_overflow = incl;
                // device-output renders never synchronise: leave the pair count this frame needed in host-mapped
                // memory, where the next msplat_sort / msplat_render / msplat_synchronize on the context finds it
                if (host_words != nullptr && report_overflow)
                    __hip_atomic_store(host_words, incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();

    uint32_t hchunk = 0, hpart = 0;
    if (helper) {
        const uint32_t h = blockIdx.x, slot = h / (kHeavyParts - 1u);
        if (slot >= min(heavy[0], nhelp / (kHeavyParts - 1u))) return;      // workgroup-uniform: no such heavy chunk this frame
        hchunk = heavy[1u + slot];
        hpart = 1u + h % (kHeavyParts - 1u);
    }
    const uint32_t cpp = ((uint32_t)tiles_x + kHeavyParts - 1u) / kHeavyParts;      // columns per part
    for (uint32_t cidx = helper ? hchunk : mb; cidx < nchunks; cidx += nmain) {
        // (xcd_map: see ws_downsweep -- the (chunk, column) runs of neighbouring chunks are adjacent in memory)
        const uint32_t chunk = (!helper && xcd_map > 1 && (nmain >= nchunks || nmain % (8u * (uint32_t)xcd_map) == 0u))
                                   ? xcd_grouped(cidx, nchunks, (uint32_t)xcd_map)
                                   : ((!helper && xcd_map == 1 && (nmain >= nchunks || (nmain & 7u) == 0u)) ? xcd_contiguous(cidx, nchunks) : cidx);
        // this workgroup's columns of the chunk: all of them, or one block of a heavy chunk
        uint32_t c_lo = 0u, c_hi = 255u;
        if (helper || (nhelp != 0u && heavy_flag[chunk])) {       // (no helpers launched: no chunk is split, no flag to read)
            c_lo = hpart * cpp;
            c_hi = c_lo + cpp - 1u;
        }
        const uint32_t chunk_pre = (gsum != nullptr) ? group_prefix(hist, gsum, chunk, s_part, gsup) : hist[(size_t)chunk * 256 + threadIdx.x];
        const uint32_t rbase = chunk * BIN_CHUNK;
        uint32_t rc[PER], woff[PER], wsum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t r = rbase + threadIdx.x * PER + k;
            uint32_t x = (r < V) ? rect[r] : kRectEmpty;
            {   // clip to [c_lo, c_hi] (a no-op for 0 .. 255)
                const uint32_t a = max(x & 255u, c_lo), b = min((x >> 16) & 255u, c_hi);
                x = (a <= b && (x & 255u) <= ((x >> 16) & 255u)) ? ((x & 0xFF00FF00u) | a | (b << 16)) : kRectEmpty;
            }
            rc[k] = x;
            woff[k] = wsum;
            wsum += rect_width(rc[k]);
        }
        uint32_t M;
        const uint32_t incl = block_incl_scan(wsum, s_tmp, M);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            s_off[threadIdx.x * PER + k] = incl - wsum + woff[k];
            s_rect[threadIdx.x * PER + k] = rc[k];
        }
        if (threadIdx.x == 0) s_off[BIN_CHUNK] = M;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_cnt[k][threadIdx.x] = 0;
        const bool owner_table = M <= kOwnerCap;              // block-uniform
        if (owner_table) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t first = incl - wsum + woff[k], wd = rect_width(rc[k]);
                for (uint32_t q = 0; q < wd; ++q) s_owner[first + q] = (uint16_t)(threadIdx.x * PER + k);
            }
        }
        __syncthreads();

        const uint32_t per_wave = (((M + 3u) >> 2) + 63u) & ~63u;    // multiple of 64
        const uint32_t wbeg = (uint32_t)w * per_wave;
        const uint32_t wend = min(M, wbeg + per_wave);

        // item k -> (owner rectangle j, column tx, rows, first row)
        auto locate = [&](uint32_t k, uint32_t& tx, uint32_t& rows, uint32_t& ty0, uint32_t& rank) {
            uint32_t lo = 0, hi = BIN_CHUNK - 1;      // last j with s_off[j] <= k (BIN_CHUNK candidates: 10 / 11 steps)
            if (owner_table) {
                lo = s_owner[k];
            } else {
#pragma unroll
                for (int s = 0; (1 << s) < BIN_CHUNK; ++s) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (s_off[mid] <= k) lo = mid; else hi = mid - 1u;
                }
            }
            const uint32_t r = s_rect[lo];
            tx = (r & 255u) + (k - s_off[lo]);
            ty0 = (r >> 8) & 255u;
            rows = (r >> 24) - ty0 + 1u;
            rank = rbase + lo;
        };

        // pass A: column weights per wave
        for (uint32_t k = wbeg + lane; k < wend; k += 64) {
            uint32_t tx, rows, ty0, rank;
            locate(k, tx, rows, ty0, rank);
            atomicAdd(&s_cnt[w][tx], rows);
        }
        __syncthreads();
        {
            const int d = threadIdx.x;
            const uint32_t g = s_base[d] + chunk_pre;
            const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d];
            s_cnt[0][d] = g;
            s_cnt[1][d] = g + c0;
            s_cnt[2][d] = g + c0 + c1;
            s_cnt[3][d] = g + c0 + c1 + c2;
        }
        __syncthreads();

        // pass B: rank inside the wave, advance the wave's column cursors, emit the words
        for (uint32_t kb = wbeg; kb < wend; kb += 64) {          // wave-uniform trip count
            const uint32_t k = kb + lane;
            const bool valid = k < wend;
            uint32_t tx = 0, rows = 0, ty0 = 0, rank = 0;
            if (valid) locate(k, tx, rows, ty0, rank);
            uint32_t pos = 0;
            if (ATOMIC_RANK) {
                // weighted stable rank straight from the LDS atomic (lane-ordered, see radix_downsweep)
                if (valid) pos = atomicAdd(&s_cnt[w][tx], rows);
            } else {
                uint64_t m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool bit = (tx >> b) & 1u;
                    const uint64_t bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                uint32_t pre = 0, tot = 0;
#pragma unroll
                for (int b = 0; b < 9; ++b) {
                    const uint64_t bal = __ballot(valid && ((rows >> b) & 1u)) & m;
                    pre += (uint32_t)__popcll(bal & lt_mask) << b;
                    tot += (uint32_t)__popcll(bal) << b;
                }
                uint32_t prev = 0;
                if (valid) prev = s_cnt[w][tx];
                __builtin_amdgcn_wave_barrier();
                if (valid && (m & lt_mask) == 0) s_cnt[w][tx] = prev + tot;
                __builtin_amdgcn_wave_barrier();
                pos = prev + pre;
            }
            // emit: item j writes `rows` consecutive words.  (r3: one wave-wide store loop per tall item, and a cooperative
            // expansion of the batch's words by binary search, were both measured on the scene-like cloud and dropped -- 674 /
            // 392 us; what fixed that workload is splitting the heavy CHUNKS over workgroups, above.)
            // The pass is bound by the NUMBER of store requests (every lane of a store instruction hits its own line): an
            // item's words are consecutive, so they go out as 8- and 16-byte stores once pos is aligned.
            if (valid) {
                const uint32_t w0 = (ty0 << 24) | rank;
                if (pos + rows <= cap) {
                    uint32_t q = 0;
                    if ((pos & 1u) && rows != 0u) { pairs_out[pos] = w0; q = 1u; }
                    if (((pos + q) & 2u) && q + 2u <= rows) {
                        *reinterpret_cast<uint2*>(pairs_out + pos + q) = make_uint2(w0 + (q << 24), w0 + ((q + 1u) << 24));
                        q += 2u;
                    }
                    for (; q + 4u <= rows; q += 4u)
                        *reinterpret_cast<uint4*>(pairs_out + pos + q) =
                            make_uint4(w0 + (q << 24), w0 + ((q + 1u) << 24), w0 + ((q + 2u) << 24), w0 + ((q + 3u) << 24));
                    if (q + 2u <= rows) {
                        *reinterpret_cast<uint2*>(pairs_out + pos + q) = make_uint2(w0 + (q << 24), w0 + ((q + 1u) << 24));
                        q += 2u;
                    }
                    if (q < rows) pairs_out[pos + q] = w0 + (q << 24);
                } else {
                    for (uint32_t q = 0; q < rows; ++q)
                        if (pos + q < cap) pairs_out[pos + q] = w0 + (q << 24);
                }
            }
        }
        __syncthreads();
        if (helper) break;           // a helper serves one (chunk, column block)
    }
}

// per bin: first position of its list in the final pair array.  The array is sorted by (row, word)
// with word = (tx << 24) | rank, so inside row vty the words are ascending: lower_bound(tx << 24).
// One WAVE per bin and a 64-ary search: every step probes 64 evenly spaced words of the remaining range with
// one gather, so a row segment of 100 k words needs 3 dependent loads instead of 17 (this kernel was a
// 2 k-thread latency chain: 7.8 us at 1920x1080).
// (the compositors' sharded work queue, see queue_next below)
constexpr int kTileStartBins = kThreads / 64;      // bins per workgroup
__global__ __launch_bounds__(kThreads) void tile_start_kernel(const uint32_t* __restrict__ pairs,
                                                              const uint32_t* __restrict__ row_totals,
                                                              const uint32_t* __restrict__ d_D, uint32_t cap,
                                                              int tiles_x, int ntiles,
                                                              uint32_t* __restrict__ tile_start,
                                                              uint32_t* __restrict__ gsum_zero, uint32_t gsum_zero_rows,
                                                              uint32_t* __restrict__ queue_reset)
{
    // (the compositors' work queue starts empty every frame; tile_order_kernel does it when it runs)
    if (queue_reset != nullptr && blockIdx.x == 0 && threadIdx.x < kQueueShards) queue_reset[threadIdx.x * kQueueStride] = 0u;
    __shared__ uint32_t s_row[kThreads + 1];
    __shared__ uint32_t s_tmp[4];
    if (gsum_zero != nullptr)      // scan-free path: the row pass's group table for the NEXT frame, see radix_upsweep
        for (uint32_t row = blockIdx.x; row < gsum_zero_rows; row += gridDim.x) gsum_zero[(size_t)row * 256 + threadIdx.x] = 0u;
    {
        const uint32_t t = row_totals[threadIdx.x];
        uint32_t tot;
        const uint32_t incl = block_incl_scan(t, s_tmp, tot);
        s_row[threadIdx.x] = incl - t;
        if (threadIdx.x == 255) s_row[256] = incl;
    }
    __syncthreads();
    const uint32_t D = min(*d_D, cap);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kTileStartBins + (threadIdx.x >> 6);
    if (tile == 0 && lane == 0) tile_start[ntiles] = D;
    if (tile >= ntiles) return;
    const int vty = tile / tiles_x;
    const uint32_t tx = (uint32_t)(tile - vty * tiles_x);
    uint32_t lo = min(s_row[vty], D), hi = min(s_row[vty + 1], D);     // the answer lies in [lo, hi]
    const uint32_t key = tx << 24;
    while (lo < hi) {                                                  // wave-uniform
        const uint32_t len = hi - lo;
        const uint32_t step = (len + 64u) / 65u;                       // >= 1
        const uint32_t p = lo + ((uint32_t)lane + 1u) * step - 1u;     // probe j = lane: ascending positions
        const bool below = (p < hi) && (pairs[p] < key);
        const uint32_t c = (uint32_t)__popcll(__ballot(below));        // probes 0..c-1 are below the key (sorted input)
        const uint32_t pc = lo + (c + 1u) * step - 1u;                 // probe c: first probe not below, if it exists
        const uint32_t nlo = c ? lo + c * step : lo;                   // = p[c-1] + 1
        const uint32_t nhi = (c < 64u && pc < hi) ? pc : hi;
        lo = nlo;
        hi = nhi;
    }
    if (lane == 0) tile_start[tile] = lo;
}

// Self-check of the two ordering contracts everything downstream relies on (the stable ranking rests on
// ds_add_rtn handing out values in lane order, which is probed once per context but not documented hardware
// behaviour): (1) the sorted keys ascend and equal keys keep ascending splat indices, (2) every bin list ascends in
// draw-order rank.  bad[0] / bad[1] count the violations.  On demand only (msplat_debug_verify_order).
__global__ __launch_bounds__(kThreads) void verify_order_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ idx,
                                                                const uint32_t* __restrict__ d_V,
                                                                const uint32_t* __restrict__ tile_start,
                                                                const uint32_t* __restrict__ pairs, uint32_t cap,
                                                                int nbins, uint32_t* __restrict__ bad)
{
    const uint32_t V = *d_V;
    uint32_t b0 = 0, b1 = 0;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i + 1u < V; i += gridDim.x * kThreads) {
        const uint32_t k0 = keys[i], k1 = keys[i + 1u];
        if (k0 > k1 || (k0 == k1 && idx[i] >= idx[i + 1u])) ++b0;
    }
    if (tile_start != nullptr) {
        const int lane = threadIdx.x & 63;
        for (int bin = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); bin < nbins; bin += gridDim.x * (kThreads / 64)) {
            const uint32_t s = min(tile_start[bin], cap), e = min(tile_start[bin + 1], cap);
            for (uint32_t i = s + lane; i + 1u < e; i += 64u)
                if ((pairs[i] & kRankMask) >= (pairs[i + 1u] & kRankMask)) ++b1;
        }
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
}

// The compositors' work queue.  One queue head serves only ~90 returning atomics per microsecond (measured r2: a
// half-tile launch pulling 8 k items from one head spent ~90 us queueing), so the head is sharded: item i lives
// in shard i % 32, a workgroup pulls from the shard of its index and, when that one is drained, from up to two
// neighbours (checked with a plain load first, so drained shards are not hammered by the exiting waves).
// Where the items are numbered heaviest-first (every item on its own wave) every shard hands out its
// share heaviest-first too; persistent waves otherwise walk the bins in storage order (`tile_order` + 65536).  The first item
// of every workgroup is static (its own index): queue[s] counts only the items of shard s taken dynamically.
__device__ __forceinline__ uint32_t queue_next(uint32_t* __restrict__ queue, uint32_t nitems)
{
    const uint32_t home = blockIdx.x % kQueueShards;
    for (uint32_t t = 0; t < 3u; ++t) {
        const uint32_t s = (home + t) % kQueueShards;
        const uint32_t stat = (gridDim.x + kQueueShards - 1u - s) / kQueueShards;      // items of shard s taken statically
        uint32_t* head = queue + s * kQueueStride;
        if (t != 0u) {
            const uint32_t cur = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint64_t)(cur + stat) * kQueueShards + s >= nitems) continue;
        }
        const uint32_t k = atomicAdd(head, 1u) + stat;
        const uint64_t item = (uint64_t)k * kQueueShards + s;
        if (item < nitems) return (uint32_t)item;
    }
    return 0xFFFFFFFFu;
}

// 0, 1, 2, ...: the bin order persistent compositor waves use (filled once, at msplat_create)
__global__ __launch_bounds__(kThreads) void iota_kernel(uint32_t* __restrict__ dst)
{
    dst[blockIdx.x * kThreads + threadIdx.x] = blockIdx.x * kThreads + threadIdx.x;
}

// tiles ordered by descending list length (counting sort on len/16): the compositor's waves pull tiles
// from this list through an atomic queue, heaviest first (longest-processing-time-first scheduling)
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ tile_start, int ntiles,
                                                          uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ queue)
{
    if (threadIdx.x < kQueueShards) queue[threadIdx.x * kQueueStride] = 0u;      // the compositors' work queue starts empty every frame

    __shared__ uint32_t s_cnt[256];
    __shared__ uint32_t s_off[256];
    if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += 1024) {
        const uint32_t len = tile_start[i + 1] - tile_start[i];
        atomicAdd(&s_cnt[255u - min(len >> 4, 255u)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {          // one wave scans the 256 buckets (4 per lane)
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = s_cnt[threadIdx.x * 4 + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if ((int)threadIdx.x >= d) incl += t;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_off[threadIdx.x * 4 + k] = run; run += c[k]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += 1024) {
        const uint32_t len = tile_start[i + 1] - tile_start[i];
        const uint32_t pos = atomicAdd(&s_off[255u - min(len >> 4, 255u)], 1u);
        order[pos] = (uint32_t)i;      // order inside a bucket is irrelevant (tiles are independent)
    }
}

// ------------------------------------------------------------------------------------------
// composite: one 16x16 workgroup per tile, front-to-back over the tile's depth-ordered list
// (reverse of the reference's back-to-front ROP blend; algebraically identical -- SURVEY 8a-12):
//   C = sum_i T_i w_i c_i,  T_i = prod_{j nearer}(1 - w_j),  A = 1
// splat_frag.glsl:18-42 defines w and the discard (w <= 1/256); app.cpp:153-160 the blend/clear.
// ------------------------------------------------------------------------------------------

constexpr int kCompThreads = 64;   // one wave per 16x16 tile, 4 pixels (one per 16x4 strip) per lane
constexpr int kCompOcc = 5;        // waves per SIMD the register allocation leaves room for (76 VGPRs; 6+ measured slower)

// Work item = (bin, quadrant): one wave composites one 16x16 tile of a 32x32 bin.  The kernel is VALU bound: 21.5 VALU
// instructions per record in the inner loop (12 packed, 4 v_exp_f32, 3 scalar FMAs: ~147 pipe cycles) and ~70 per staged
// batch.  Formulations that were built, measured and removed (numbers in DESIGN.md 4): four waves per tile with 8x8 sub-block
// queues (LDS-pipe bound), one wave per 16x8 half tile, per-strip-pair masks, 6-8 waves per SIMD.
// Discard by underflow: the fragment shader's discard (w <= 1/256, splat_frag.glsl:37-40) would cost a compare and a select
// per pixel.  Here it is free: the exponent is biased by -118, so that w' = exp2(e - 118) is a NORMAL float exactly when
// e >= -8 and underflows otherwise, and the wave runs with fp32 denormals flushed (MODE.FP_DENORM, set below): the
// underflowing weights come out of v_exp_f32 as exact zeros.  The transmittance is carried scaled by 2^118 (Ts = 2^118 T), so
// tw = Ts w' = T w exactly as before (powers of two), colours accumulate unchanged and Ts -= 2^118 tw.  The bias costs 4 bits
// of the exponent's absolute precision (|e - 118| ~ 2^7 instead of <= 2^3): a relative error of 3e-6 in w.  Only e == -8
// exactly (w == 1/256, which the reference discards) is kept: a measure-zero threshold flip.
template <bool F16>
__global__ __launch_bounds__(kCompThreads, kCompOcc) void composite_kernel(const uint32_t* __restrict__ tile_start,
                                                                 const uint32_t* __restrict__ pairs,
                                                                 const float4* __restrict__ rec,
                                                                 void* __restrict__ out, size_t pitch_bytes,
                                                                 FrameParams fp, uint32_t cap,
                                                                 const uint32_t* __restrict__ order,
                                                                 uint32_t* __restrict__ queue, uint32_t ntiles,
                                                                 uint32_t* __restrict__ probe, int prio_levels,
                                                                 void* __restrict__ out1 = nullptr)
{
    // out1: the second view's target (FrameParams.views == 2: bin rows >= rows_view belong to it)
    // Lane (lx, ly) owns pixels (x0+lx, y0 + 4k + ly), k = 0..3: strip k is the 16x4 pixel block of
    // rows 4k..4k+3.  Per splat the exponent is split into a part shared by the four strips and a
    // 2-FMA part per strip; strips the splat's y-range cannot reach, or whose 64 pixels are all
    // saturated, are skipped with scalar branches.
    __shared__ float4 s_rec[(kCompThreads + 1) * 3];
    // (r4, measured and removed: the exponents of four staged records at a time from the matrix pipe -- e is a K = 5 contraction
    //  of per-record coefficients with per-pixel monomials on top of c0; v_mfma_f32_4x4x1_16b_f32, 3.5 per record instead of 3
    //  scalar + 4 packed FMAs; bit-compatible images.  20 % fewer non-MFMA VALU instructions, and the launch 14 % LONGER (81 -> 92 us):
    //  the MFMAs take the same issue port, SQ_ACTIVE_INST_VALU fell by 4 % only.  DESIGN.md 4, profiles/r04_pmc_sq_compositor_mfma.txt)

    // Persistent waves + dynamic queue: per-tile work varies by >10x (list length, early saturation),
    // so tiles are pulled heaviest-first from `order` instead of being bound to a workgroup index.
    // The first tile of every wave is static (its workgroup index): same-address atomics are served
    // at only ~8 ns each, so thousands of waves pulling at launch would queue up for tens of us.
    // Work item = (bin, quadrant): the four 16x16 tiles of a 32x32 bin share the bin's list.
    constexpr float kBias = 118.0f;
    constexpr float kScale = 0x1p118f;
    __builtin_amdgcn_s_setreg(1 | (4 << 6) | ((2 - 1) << 11), 0);      // MODE[5:4] = 0: flush fp32 denormals
    constexpr int NP = 2;                            // strip pairs per work item (the whole 16x16 tile)
    constexpr int NS = 2 * NP;                       // 16x4 strips per work item
    constexpr int ROWS = 4 * NS;                     // pixel rows per work item
    for (uint32_t qpos = blockIdx.x; qpos < ntiles;) {
    const int tile = (int)qpos;                       // probe slot
    const uint32_t tpos = qpos;                       // (bin, quadrant) index
    // The four tiles of a bin walk the SAME list, and workgroup b runs on XCD b % 8 (each XCD has its own L2): inside every
    // group of 32 items the quadrants of one bin are the items r, r + 8, r + 16, r + 24, i.e. on one XCD, as the first
    // (static) item of a wave and -- shard = item % 32, home shard = workgroup % 32 -- as a pulled one.  Three of the four
    // waves then find the list words and records in their XCD's L2 instead of fetching them from HBM again.
    uint32_t slot = tpos >> 2, quadrant = tpos & 3u;
    if (tpos < (ntiles & ~31u)) {
        slot = (tpos >> 5) * 8u + (tpos & 7u);
        quadrant = (tpos >> 3) & 3u;
    }
    const int bin = (int)order[slot];
    const int quad = (int)quadrant;
    const int bvy = bin / fp.tiles_x;
    const int tx = (bin - bvy * fp.tiles_x) * 2 + (quad & 1);
    const bool second = fp.views == 2 && bvy >= fp.rows_view;
    const int ty = (second ? bvy - fp.rows_view : band_real_row(fp, bvy)) * 2 + (quad >> 1);
    if (tx * kTile >= fp.width || ty * kTile >= fp.height) {      // work item entirely outside the image
        if (gridDim.x >= ntiles) break;
        uint32_t nq = 0;
        if (threadIdx.x == 0) nq = queue_next(queue, ntiles);
        qpos = __builtin_amdgcn_readfirstlane(nq);
        continue;
    }
    // The launch lasts as long as its heaviest work item (the probe: max / mean item clocks = 2.0, and the heaviest item
    // spans the whole launch although it starts first), because a wave that shares its SIMD with four others gets a
    // fifth of the issue slots.  Items are numbered heaviest-first, so the wave's issue priority follows the item
    // number: the heaviest thousand items run at the single-wave issue rate from the start and the light ones fill
    // the slots they leave (SIMD arbitration is priority first, then age -- MI355X_MICROARCH.md).
    if (prio_levels == 1) {
        const uint32_t band = max(ntiles >> 3, 1u);                    // an eighth of the items per priority step
        const uint32_t lvl = qpos / band;
        if (lvl == 0u) __builtin_amdgcn_s_setprio(3);
        else if (lvl == 1u) __builtin_amdgcn_s_setprio(2);
        else if (lvl <= 3u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    const int lane = threadIdx.x;
    const int lx = lane & 15, ly = lane >> 4;
    const int x = tx * kTile + lx, ybase = ty * kTile + ly;
    const float fx = (float)x + 0.5f;
    const float fy0 = (float)ybase + 0.5f;
    const float tile_y0 = (float)(ty * kTile);

    uint32_t start = tile_start[bin], end = tile_start[bin + 1];
    if (start > cap) start = cap;
    if (end > cap) end = cap;

    // Accumulators are kept as strip PAIRS (0,1) and (2,3): gfx950 executes a plain wave64 fp32 VALU
    // op in ~4 cycles but a packed v_pk_{fma,mul,add}_f32 does two per lane in the same slot (measured:
    // SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.4 cycles), and this kernel is VALU bound.
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f T[NP], cr[NP], cg[NP], cb[NP];
    bool inside[NS];
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        T[h] = (v2f){kScale, kScale};                // the transmittance scaled by 2^118
        cr[h] = (v2f){0.0f, 0.0f}; cg[h] = (v2f){0.0f, 0.0f}; cb[h] = (v2f){0.0f, 0.0f};
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) inside[k] = (x < fp.width) && (ybase + 4 * k < fp.height);
    // The exponent is evaluated as a polynomial in TILE-CENTRED pixel coordinates (|u|, |v| <= 7.5: no cancellation
    // trouble): e(u, v) = c0 + c1 u + c2 v + c3 u^2 + c4 u v + c5 v^2, coefficients per staged record.  Per record and lane
    // that is 3 scalar FMAs for the u part plus 2 packed FMAs per strip pair -- the centre-relative form (dx, dy, base,
    // lin) needed 4 + 3: 162 instead of 186 VALU pipe cycles per record.
    const float xc = (float)(tx * kTile) + 0.5f * (float)kTile, yc = tile_y0 + 0.5f * (float)ROWS;
    const float u = fx - xc;
    v2f vp[NP];
#pragma unroll
    for (int h = 0; h < NP; ++h) vp[h] = (v2f){fy0 + 8.0f * h - yc, fy0 + 8.0f * h + 4.0f - yc};
    uint32_t alive = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) alive |= (__ballot(inside[k]) != 0ull) ? (1u << k) : 0u;

    // Three-stage software pipeline over batches of 64 list entries (nearest first):
    //   ranks of batch b+2 and records of batch b+1 are in flight while batch b is composited,
    // so a tile whose entries are mostly culled pays one memory latency per batch instead of two
    // dependent ones (that latency chain, not ALU work, is the critical path of the long tiles).
    uint32_t hiA = end;                                        // entries [start, hiA) not yet rank-loaded
    uint32_t cntA = min((uint32_t)kCompThreads, hiA - start);  // batch whose ranks are in rankA
    // rankA holds the RAW pair word; the rank mask is applied where the word is used.  Masking right after the
    // load made the compiler wait (s_waitcnt vmcnt(0)) for it -- and with it for the record loads issued just
    // before -- in front of the inner loop: the whole prefetch pipeline was serialised (27 % of the kernel).
    uint32_t rankA = 0;
    if (lane < (int)cntA) rankA = pairs[hiA - 1u - lane];     // j = 0 is the nearest splat
    hiA -= cntA;
    uint32_t cnt = cntA;                                       // batch whose records are in p0..p2
    float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
    if (lane < (int)cnt) {
        uint32_t rk = rankA & kRankMask;
        asm volatile("" : "+v"(rk));              // keep the mask out of the address arithmetic (see composite_depth_kernel)
        const float4* src = rec + (size_t)rk * 3;
        p0 = src[0]; p1 = src[1]; p2 = src[2];
    }
    cntA = min((uint32_t)kCompThreads, hiA - start);
    if (lane < (int)cntA) rankA = pairs[hiA - 1u - lane];
    hiA -= cntA;
    const uint64_t probe_t0 = probe ? clock64() : 0ull;
    uint32_t probe_n = 0, probe_batches = 0;
    uint64_t probe_inner = 0;
    // pair words / records whose loads have been issued so far (the prefetch pipeline runs two / one batches ahead)
    uint32_t probe_words = min(end - start, 2u * (uint32_t)kCompThreads), probe_recs = cnt;
    while (cnt != 0u && alive != 0u) {
        // stage: every lane turns its list entry into the coefficients of e(u, v) in tile-centred coordinates and tests
        // it against the tile; the survivors are compacted into LDS in list order (near to far).  Straight-line code on
        // purpose: the CU has ONE scalar unit for its four SIMDs and this is the dependent chain between two batches --
        // the branchy form (per-strip y tests, the exact test under an EXEC mask) was ~115 VALU + ~80 scalar instructions
        // per batch, this one is ~60 + ~15.
        uint32_t n;
        {
            constexpr float U = 0.5f * (float)(kTile - 1);          // box of pixel centres: |u| <= U, |v| <= Vh
            constexpr float Vh = 0.5f * (float)(ROWS - 1);
            const float a = p0.x - xc, b = p0.y - yc;               // splat centre, tile-centred
            const float qa = p0.z, qb = p0.w, qc = p1.x;            // c3, c4, c5
            const float Aa = qa * a, Bb = qb * b, Cb = qc * b, Ba = qb * a;
            const float c1 = __builtin_fmaf(-2.0f, Aa, -Bb);
            const float c2 = __builtin_fmaf(-2.0f, Cb, -Ba);
            const float c0 = __builtin_fmaf(Aa + Bb, a, __builtin_fmaf(Cb, b, p1.y - kBias));   // the exponent bias rides on log2(alpha)
            // y reach of the footprint against the strips that are still live (strip k: v in [4k - Vh, 4k + 3 - Vh])
            const float vlo = b - p2.w, vhi = b + p2.w;
            bool rel = lane < (int)cnt && vhi >= -Vh && vlo <= Vh;
            if (alive != (1u << NS) - 1u) {                          // wave-uniform; only once strips have saturated
                bool any = false;
#pragma unroll
                for (int k = 0; k < NS; ++k)
                    any = any || ((alive & (1u << k)) && vhi >= 4.0f * k - Vh && vlo <= 4.0f * k + 3.0f - Vh);
                rel = rel && any;
            }
            // exact footprint-vs-tile test (the list was built from bounding rectangles): e is a concave quadratic, so
            // unless the centre lies inside the box its maximum over the box is on one of the four edges (1-D maximiser,
            // clamped).  v_rcp_f32 instead of IEEE divisions: the maximiser only has to be good to the 0.05 slack below.
            const bool inside_box = fabsf(a) <= U && fabsf(b) <= Vh;
            const float i2c = -0.5f * __builtin_amdgcn_rcpf(qc), i2a = -0.5f * __builtin_amdgcn_rcpf(qa);
            const float ku = __builtin_fmaf(qa, U * U, c0), kv = __builtin_fmaf(qc, Vh * Vh, c0);
            float emax;
            {
                const float lp = __builtin_fmaf(qb, U, c2), lm = __builtin_fmaf(qb, -U, c2);       // edges u = +-U
                const float kp = __builtin_fmaf(c1, U, ku), km = __builtin_fmaf(c1, -U, ku);
                const float vp = fminf(fmaxf(lp * i2c, -Vh), Vh), vm = fminf(fmaxf(lm * i2c, -Vh), Vh);
                const float ep = __builtin_fmaf(__builtin_fmaf(qc, vp, lp), vp, kp);
                const float em = __builtin_fmaf(__builtin_fmaf(qc, vm, lm), vm, km);
                const float mp = __builtin_fmaf(qb, Vh, c1), mm = __builtin_fmaf(qb, -Vh, c1);    // edges v = +-Vh
                const float hp = __builtin_fmaf(c2, Vh, kv), hm = __builtin_fmaf(c2, -Vh, kv);
                const float up = fminf(fmaxf(mp * i2a, -U), U), um = fminf(fmaxf(mm * i2a, -U), U);
                const float fp_ = __builtin_fmaf(__builtin_fmaf(qa, up, mp), up, hp);
                const float fm_ = __builtin_fmaf(__builtin_fmaf(qa, um, mm), um, hm);
                emax = fmaxf(fmaxf(ep, em), fmaxf(fp_, fm_));
            }
            rel = rel && (inside_box || emax > -8.05f - kBias);
            const uint64_t relmask = __ballot(rel);
            n = (uint32_t)__popcll(relmask);
            if (rel) {
                const int slot = __popcll(relmask & ((1ull << lane) - 1ull));
                // the four values the packed instructions broadcast (c5, r, g, b) sit at even dwords of the 16-byte reads: they
                // land in even VGPRs, which a packed operand can name directly (an odd one costs a v_mov)
                s_rec[slot * 3 + 0] = make_float4(qc, c0, p1.z, c1);
                s_rec[slot * 3 + 1] = make_float4(p1.w, c2, p2.x, qa);
                s_rec[slot * 3 + 2] = make_float4(qb, 0.0f, 0.0f, 0.0f);
            }
        }
        __syncthreads();
        cnt = cntA;
        if (lane < (int)cnt) {
            uint32_t rk = rankA & kRankMask;
            asm volatile("" : "+v"(rk));
            const float4* src = rec + (size_t)rk * 3;
            p0 = src[0]; p1 = src[1]; p2 = src[2];
        }
        cntA = min((uint32_t)kCompThreads, hiA - start);
        if (lane < (int)cntA) rankA = pairs[hiA - 1u - lane];
        hiA -= cntA;
        probe_n += n;
        ++probe_batches;
        probe_words += cntA;
        probe_recs += cnt;
        const uint64_t probe_t1 = probe ? clock64() : 0ull;
        if (n != 0u) {
            float4 a = s_rec[0];          // c5, c0, r, c1
            float4 b = s_rec[1];          // g, c2, b, c3
            float c4 = s_rec[2].x;
#pragma unroll 2
            for (uint32_t j = 0; j < n; ++j) {
                // next record (slot n is a harmless over-read inside the 65-slot array)
                const float4 na = s_rec[(j + 1) * 3 + 0];
                const float4 nb = s_rec[(j + 1) * 3 + 1];
                const float nc4 = s_rec[(j + 1) * 3 + 2].x;
                const float base = __builtin_fmaf(__builtin_fmaf(b.w, u, a.w), u, a.y);      // c0 + c1 u + c3 u^2
                const float lin = __builtin_fmaf(c4, u, b.y);                                // c2 + c4 u
                const v2f vbase = (v2f){base, base}, vlin = (v2f){lin, lin}, vC = (v2f){a.x, a.x};
                const v2f vr = (v2f){a.z, a.z}, vg = (v2f){b.x, b.x}, vb = (v2f){b.z, b.z};
                // Branch-free on purpose: the strips are independent dependency chains inside one basic
                // block, so the in-order wave can overlap them.  w = 0 where the fragment shader discards.
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    const v2f e = __builtin_elementwise_fma(vp[h], __builtin_elementwise_fma(vC, vp[h], vlin), vbase);
                    // splat_frag.glsl:37-40 discard: w = exp2(e) > 1/256  <=>  e > -8
                    v2f w;           // discard by underflow (see the kernel's header)
                    w.x = __builtin_amdgcn_exp2f(e.x);
                    w.y = __builtin_amdgcn_exp2f(e.y);
                    const v2f tw = T[h] * w;
                    cr[h] = __builtin_elementwise_fma(tw, vr, cr[h]);
                    cg[h] = __builtin_elementwise_fma(tw, vg, cg[h]);
                    cb[h] = __builtin_elementwise_fma(tw, vb, cb[h]);
                    T[h] = __builtin_elementwise_fma(tw, (v2f){-kScale, -kScale}, T[h]);
                }
                a = na; b = nb; c4 = nc4;
            }
        }
        if (probe) probe_inner += clock64() - probe_t1;
        // strips whose 64 pixels are all saturated (or outside the image) are finished
        uint32_t na = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            na |= (__ballot(inside[k] && T[k >> 1][k & 1] >= fp.t_eps * kScale) != 0ull) ? (1u << k) : 0u;
        alive = na;
        __syncthreads();
    }

    if (probe != nullptr && lane == 0) {
        probe[tile * 8 + 0] = (uint32_t)(clock64() - probe_t0);        // shader clocks, whole tile
        probe[tile * 8 + 1] = probe_n;          // splats composited (after culling / saturation)
        probe[tile * 8 + 2] = probe_batches;    // batches of 64 list entries staged
        probe[tile * 8 + 3] = (uint32_t)probe_inner;   // shader clocks spent in the inner loops
        probe[tile * 8 + 4] = probe_words;      // 4-byte pair words loaded
        probe[tile * 8 + 5] = probe_recs;       // 48-byte projected records loaded
        probe[tile * 8 + 6] = end - start;      // length of the bin list
        probe[tile * 8 + 7] = 1u;               // work item ran
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (inside[k]) {
            char* row = (char*)(second ? out1 : out) + (size_t)(ybase + 4 * k) * pitch_bytes;
            if (F16) {
                union { _Float16 h[4]; uint2 u; } pk;
                pk.h[0] = (_Float16)cr[k >> 1][k & 1]; pk.h[1] = (_Float16)cg[k >> 1][k & 1]; pk.h[2] = (_Float16)cb[k >> 1][k & 1]; pk.h[3] = (_Float16)1.0f;
                ((uint2*)row)[x] = pk.u;
            } else {
                ((float4*)row)[x] = make_float4(cr[k >> 1][k & 1], cg[k >> 1][k & 1], cb[k >> 1][k & 1], 1.0f);
            }
        }
    }
    __syncthreads();      // s_rec is reused by the next tile
    if (gridDim.x >= ntiles) break;       // every work item has its own wave: nothing to pull, no exit atomic
    uint32_t nq = 0;
    if (threadIdx.x == 0) nq = queue_next(queue, ntiles);
    qpos = __builtin_amdgcn_readfirstlane(nq);
    }   // persistent tile loop
}

// ------------------------------------------------------------------------------------------
// composite with an emulated depth buffer.  The reference enables GL_DEPTH_TEST
// (app.cpp:163, GL_LESS, depth writes on); it is live whenever the target has a depth attachment
// (default back buffer, XR swapchains) and inert for the colour-only --fp16/--fp32 FBO that the
// main compositor models.  With a depth buffer a fragment that survives the discard also has to
// pass z < zbuf and then writes its z: splats whose quantised depths tie, or that are drawn out
// of depth order (second XR eye re-using the first eye's sort), lose their later fragments.
// Whether a fragment passes depends on everything drawn BEFORE it, so this variant walks the list
// in draw order (far to near) with the literal "over" blend and cannot terminate early.
//
// The same draw-order walk also emulates what the render target does to the running colour (fp.rop, SURVEY 8a-12,
// src/app.cpp:1012-1020): the default RGBA8 back buffer clamps source, destination and result to [0,1] and stores 8-bit
// unorm after EVERY blend (GL 4.6 17.3.6), the --fp16 target rounds to fp16 after every blend; the main compositor
// accumulates in fp32 and rounds once.  fp.depth_bits = 0 then means "no depth test".
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float rop_store(float x, int rop)
{
    if (rop == 1) return floorf(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f + 0.5f) / 255.0f;
    if (rop == 2) return (float)(_Float16)x;            // round to nearest even, like the fp16 target
    return x;
}

template <bool HALF>
__global__ __launch_bounds__(kCompThreads) void composite_depth_kernel(const uint32_t* __restrict__ tile_start,
                                                                       const uint32_t* __restrict__ pairs,
                                                                       const float4* __restrict__ rec,
                                                                       const uint32_t* __restrict__ zq,
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
            for (int k = 0; k < 4; ++k) { cr[k] = 0.0f; cg[k] = 0.0f; cb[k] = 0.0f; zbuf[k] = 0xFFFFFFFFu; }   // cleared to 1.0
            const float X0 = (float)(tx * kTile) + 0.5f, X1 = X0 + (float)(kTile - 1);
            const float Y0 = (float)(ty * kTile) + 0.5f, Y1 = Y0 + (float)(kTile - 1);
            for (uint32_t base = start; base < end; base += kCompThreads) {      // ascending = draw order
                const uint32_t cnt = min((uint32_t)kCompThreads, end - base);
                float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
                uint32_t z = 0;
                bool rel = false;
                if (lane < (int)cnt) {
                    uint32_t rank = pairs[base + lane] & kRankMask;
                    // hipcc 7.2 (gfx950) folds "(x & 0xFFFFFF) * 48 + base" in ONE basic block into v_mad_u64_u32
                    // on the UNMASKED word (mul24 known-bits combine; seen in the ISA, faulted on the GPU): keep
                    // the masked value opaque.  composite_kernel masks in a different block and is not affected.
                    asm volatile("" : "+v"(rank));
                    const float4* src = rec + (size_t)rank * 3;
                    p0 = src[0]; p1 = src[1]; p2 = src[2];
                    if (fp.depth_bits != 0) z = zq[rank];
                    // same exact footprint-vs-tile test as composite_kernel
                    const float qa = p0.z, qb = p0.w, qc = p1.x, la = p1.y;
                    const float dxl = X0 - p0.x, dxh = X1 - p0.x, dyl = Y0 - p0.y, dyh = Y1 - p0.y;
                    rel = true;
                    if (!(dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f)) {
                        float emax = -1e30f;
                        const float i2c = __builtin_amdgcn_rcpf(2.0f * qc), i2a = __builtin_amdgcn_rcpf(2.0f * qa);
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const float dx = s ? dxh : dxl;
                            const float dy = fminf(fmaxf(-qb * dx * i2c, dyl), dyh);
                            emax = fmaxf(emax, (qc * dy + qb * dx) * dy + qa * dx * dx + la);
                            const float ey = s ? dyh : dyl;
                            const float ex = fminf(fmaxf(-qb * ey * i2a, dxl), dxh);
                            emax = fmaxf(emax, (qa * ex + qb * ey) * ex + qc * ey * ey + la);
                        }
                        rel = emax > -8.05f;
                    }
                }
                const uint64_t relmask = __ballot(rel);
                const uint32_t n = (uint32_t)__popcll(relmask);
                if (rel) {
                    const int slot = __popcll(relmask & ((1ull << lane) - 1ull));     // keeps draw order
                    s_rec[slot * 3 + 0] = p0;
                    s_rec[slot * 3 + 1] = p1;
                    s_rec[slot * 3 + 2] = p2;
                    s_z[slot] = z;
                }
                __syncthreads();
                for (uint32_t j = 0; j < n; ++j) {
                    const float4 a = s_rec[j * 3 + 0];      // px, py, A, B
                    const float4 b = s_rec[j * 3 + 1];      // C, log2(alpha), r, g
                    const float blue = s_rec[j * 3 + 2].x;
                    const uint32_t zj = s_z[j];
                    const float dx = fx - a.x;
                    const float base_e = __builtin_fmaf(a.z * dx, dx, b.y);
                    const float lin = a.w * dx;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dy = ((float)(ybase + 4 * k) + 0.5f) - a.y;
                        const float e = __builtin_fmaf(dy, __builtin_fmaf(b.x, dy, lin), base_e);
                        // splat_frag.glsl:37-40 discard, then GL_LESS against the emulated depth buffer
                        if (e > -8.0f && (fp.depth_bits == 0 || zj < zbuf[k])) {
                            const float w = __builtin_amdgcn_exp2f(e);
                            // splat_frag.glsl:27-28: out = (w rgb, w); GL_ONE, GL_ONE_MINUS_SRC_ALPHA
                            float sr = w * b.z, sg = w * b.w, sb = w * blue;
                            if (fp.rop == 1) {      // fixed-point target: the source colour is clamped before the blend
                                sr = fminf(fmaxf(sr, 0.0f), 1.0f); sg = fminf(fmaxf(sg, 0.0f), 1.0f); sb = fminf(fmaxf(sb, 0.0f), 1.0f);
                            }
                            const float oma = 1.0f - w;
                            cr[k] = rop_store(sr + oma * cr[k], fp.rop);
                            cg[k] = rop_store(sg + oma * cg[k], fp.rop);
                            cb[k] = rop_store(sb + oma * cb[k], fp.rop);
                            zbuf[k] = zj;
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
