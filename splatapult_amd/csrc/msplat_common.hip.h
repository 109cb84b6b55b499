// msplat_common.hip.h -- constants, per-frame parameters, block scans, XCD-aware index maps, the presort cull and the
// box-level cull shared by the kernels of the splat hot path (hand-written CDNA4: gfx950, wave64).  Part of msplat_kernels.hip.h.
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
// Wave issue priorities (s_setprio; SIMD arbitration is priority first, then age).  With frames in flight the waves of different
// frames' kernels share SIMDs: the short, latency-bound sort / binning kernels run at 3, the HBM-bound projection at 2 and the
// VALU-bound compositor at 1 (its heaviest items) or 0, so a chain kernel's few instructions between two memory waits are issued
// ahead of the compositor's arithmetic instead of queueing behind it.  Measured r5 (tools/gpu_r5_d.sh, eight combinations,
// driver protocol, same box): config 2 5741 -> 6028 frames/s (6208 -> 6352 in 500-frame blocks), config 5 2890 -> 3036,
// 6 M / 1080p 2313 -> 2349, 6 M / 4096^2 1312 -> 1338, scene-like 5217 -> 5328; one frame at a time: unchanged.
constexpr int kChainPrio = 3, kProjPrio = 2, kCompPrioMax = 1;
#define MSPLAT_CHAIN_ENTER() __builtin_amdgcn_s_setprio(msplat::kChainPrio)
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

// Workgroup residency stamps (r6; DIAGNOSTIC builds only: -DMSPLAT_STAMPS, tools/stamp_timeline.py).  rocprofv3 serialises dispatches
// while it collects counters and this box has no PC sampling, so what frames IN FLIGHT do to each other was unknown.  With stamps
// every workgroup of the frame's kernels leaves {kernel id, block, HW_ID, XCC_ID, start, end} (s_memrealtime: 100 MHz) in a hashed
// table -- plain stores, no atomics, nothing returns to the wave -- while a device flag is on.  The product build compiles the
// macro to nothing.
#ifdef MSPLAT_STAMPS
struct StampRec { uint32_t kid, blk, hwid, xcc; unsigned long long t0, t1; };
__device__ StampRec* g_stamp_buf = nullptr;
__device__ uint32_t g_stamp_mask = 0u;           // table size - 1 while stamping is on, else 0
struct StampScope {
    unsigned long long t0;
    uint32_t kid, mask;
    __device__ __forceinline__ explicit StampScope(uint32_t k) : t0(0ull), kid(k), mask(0u)
    {
        if (threadIdx.x == 0) {
            mask = __builtin_nontemporal_load(&g_stamp_mask);
            if (mask) t0 = wall_clock64();
        }
    }
    __device__ __forceinline__ ~StampScope()
    {
        if (threadIdx.x == 0 && mask) {
            StampRec r;
            r.kid = kid; r.blk = blockIdx.x;
            r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
            r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);        // HW_REG_XCC_ID
            r.t0 = t0; r.t1 = wall_clock64();
            uint32_t h = (uint32_t)t0 * 2654435761u ^ (blockIdx.x * 40503u) ^ (kid << 26) ^ (r.hwid * 97u) ^ (r.xcc << 13);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            g_stamp_buf[h & mask] = r;
        }
    }
};
#define MSPLAT_STAMP(KID) msplat::StampScope stamp_scope_((uint32_t)(KID))
#else
#define MSPLAT_STAMP(KID)
#endif
// kernel ids of the stamps (tools/stamp_timeline.py names them)
enum { KID_WS_UP_CULL = 1, KID_WS_DOWN_CULL = 2, KID_WS_UP = 3, KID_WS_DOWN = 4, KID_PROJECT = 5, KID_BIN1_UP = 6, KID_BIN1_DOWN = 7,
       KID_ROW_UP = 8, KID_ROW_DOWN = 9, KID_TILE_START = 10, KID_COMPOSITE = 11, KID_BOX_CULL = 12, KID_RADIX_UP = 13, KID_RADIX_DOWN = 14 };

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

// What the compositor needs of a frame (r5): eleven words instead of FrameParams' five matrices in its kernarg segment / SGPRs.
struct CompParams {
    int width, height, tiles_x;
    int banded, band_first, band_block, band_stride;     // as in FrameParams
    int rows_view;                                       // two views in one chain: bin rows of the first view
    float t_eps;
};
__host__ inline CompParams comp_params(const FrameParams& fp)
{
    return CompParams{fp.width, fp.height, fp.tiles_x, fp.banded, fp.band_first, fp.band_block, fp.band_stride, fp.rows_view, fp.t_eps};
}

// What project_kernel needs of a frame (r5): one view's matrices and the raster geometry -- no mvp (the sort's), no second view.
struct ProjParams {
    float view[16], proj[16], eye[3];
    float W, H, X0, Y0;
    int width, height, tiles_y;
    int banded, band_first, band_block, band_stride;
    float band_inv_stride;
    int srgb, depth_bits, rows_view;
};
struct ProjView1 { float view[16], proj[16], eye[3]; };   // the second view of a two-view chain (project_kernel<*, PROJ_TWO_VIEWS> only)
struct ProjNoView1 { int unused; };
__host__ inline ProjParams proj_params(const FrameParams& fp)
{
    ProjParams p;
    for (int i = 0; i < 16; ++i) { p.view[i] = fp.view[i]; p.proj[i] = fp.proj[i]; }
    for (int i = 0; i < 3; ++i) p.eye[i] = fp.eye[i];
    p.W = fp.W; p.H = fp.H; p.X0 = fp.X0; p.Y0 = fp.Y0;
    p.width = fp.width; p.height = fp.height; p.tiles_y = fp.tiles_y;
    p.banded = fp.banded; p.band_first = fp.band_first; p.band_block = fp.band_block; p.band_stride = fp.band_stride;
    p.band_inv_stride = fp.band_inv_stride;
    p.srgb = fp.srgb; p.depth_bits = fp.depth_bits; p.rows_view = fp.rows_view;
    return p;
}
__host__ inline ProjView1 proj_view1(const FrameParams& fp)
{
    ProjView1 v;
    for (int i = 0; i < 16; ++i) { v.view[i] = fp.view1[i]; v.proj[i] = fp.proj1[i]; }
    for (int i = 0; i < 3; ++i) v.eye[i] = fp.eye1[i];
    return v;
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

// band geometry (see FrameParams): real bin row of virtual row vy
template <class P>
__host__ __device__ __forceinline__ int band_real_row(const P& fp, int vy)
{
    if (!fp.banded) return vy;
    const int k = vy / fp.band_block;
    return fp.band_first + k * fp.band_stride + (vy - k * fp.band_block);
}
// d / band_stride for 0 <= d < 65536 without an integer division (~30 instructions on this hardware, twice per splat in the
// band-culled sort and in project_kernel): (d + 0.5) / s lies at least 0.5 / s away from every integer, far more than the
// float error of the product
template <class P>
__host__ __device__ __forceinline__ int band_quot(const P& fp, int d)
{
    return (int)(((float)d + 0.5f) * fp.band_inv_stride);
}
// virtual index of the first owned row >= t (>= tiles_y: there is none)
template <class P>
__host__ __device__ __forceinline__ int band_first_owned_from(const P& fp, int t)
{
    if (!fp.banded) return t < 0 ? 0 : t;
    if (t <= fp.band_first) return 0;
    const int d = t - fp.band_first, k = band_quot(fp, d), j = d - k * fp.band_stride;
    return j < fp.band_block ? k * fp.band_block + j : (k + 1) * fp.band_block;
}
// virtual index of the last owned row <= t (-1: there is none; may be >= tiles_y: clamp)
template <class P>
__host__ __device__ __forceinline__ int band_last_owned_upto(const P& fp, int t)
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
    // never 0 for a splat that can be drawn: a zero (or underflowing) covariance still has the 0.3-pixel low-pass footprint, and
    // "bound == 0" means exactly "alpha <= 1/256" to the band cull and the two-pass gate
    const float bnd = (float)((double)rho2 * lmax * (1.0 + 1e-5));
    return bnd > 1.17549435e-38f ? bnd : 1.17549435e-38f;
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
// BAND = false (r6): an instantiation for frames WITHOUT the band-restricted cull (every single-GPU frame) -- the branch below
// keeps ~25 uniform values live (view rows, proj[5], the layout), which is what pushed pass 0's upsweep over the SGPR file.
template <bool BAND = true>
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
        if (BAND && fp.band_cull) {
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

// two-pass frames (msplat_occlusion.hip.h): first rank of pass 1 for V visible splats and a share of them in pass 1
__device__ __forceinline__ uint32_t occ_cut(uint32_t V, float share)
{
    uint32_t r1 = (uint32_t)((float)V * share);
    r1 = (r1 + 63u) & ~63u;
    if (r1 < 64u) r1 = 64u;
    const uint32_t cut = r1 >= V ? 0u : V - r1;
    return cut & ~1023u;                      // whole chunks of the column pass (and waves of the projection) lie on one side of it
}

}  // namespace msplat
