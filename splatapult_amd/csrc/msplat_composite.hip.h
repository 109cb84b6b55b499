// msplat_composite.hip.h -- the compositors: work queue, front-to-back composite_kernel (GL rasteriser + shader/splat_frag.glsl + ROP blend) and the
// draw-order variant with an emulated depth buffer / render-target rounding
// (one of the parts of msplat_kernels.hip.h)
#pragma once

#include "msplat_common.hip.h"
#include "msplat_binning.hip.h"

#pragma clang fp contract(off)

namespace msplat {

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

// ------------------------------------------------------------------------------------------
// composite: one 16x16 workgroup per tile, front-to-back over the tile's depth-ordered list
// (reverse of the reference's back-to-front ROP blend; algebraically identical -- SURVEY 8a-12):
//   C = sum_i T_i w_i c_i,  T_i = prod_{j nearer}(1 - w_j),  A = 1
// splat_frag.glsl:18-42 defines w and the discard (w <= 1/256); app.cpp:153-160 the blend/clear.
// ------------------------------------------------------------------------------------------

constexpr int kCompThreads = 64;   // one wave per 16x16 tile, 4 pixels (one per 16x4 strip) per lane
constexpr int kCompOcc = 5;        // occupancy bound handed to the compiler (<= 102 VGPRs); the kernel needs 76, so 6 waves fit a SIMD (tools/kres.sh).
                                   // A bound of 6+ was measured slower (r3: the allocator then squeezes the inner loop), DESIGN.md 4

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
// Template parameters instead of run-time arguments (r5): F16 = RGBA16F target; OCC = 0 one pass, 1 / 2 the passes of a two-pass
// frame; TWO_VIEWS = both eyes in one chain; PROBE = per-item counters.  The plain frame's instantiation <F16, 0, false, false>
// carries none of the others' registers (r4 passed them at run time on top of the ~500-byte FrameParams: 81 VGPRs / 14 SGPR spills
// against r3's 79 / 5) and takes the eleven words of CompParams instead of five 4x4 matrices.
struct CompExtra {
    void* out1;                  // TWO_VIEWS: the second view's target (bin rows >= rows_view belong to it)
    uint32_t* fin;               // OCC != 0: per (bin, quadrant) 0xFFFFFFFF = final, else entries composited by pass 1
    float4* state;               // OCC != 0: (r, g, b, 2^118 T) per pixel of the tiles pass 1 leaves unfinished
    const uint32_t* d_nbins;     // OCC == 2: `order` lists *d_nbins bins -- the unfinished ones -- and the items are theirs alone
    uint32_t* probe;             // PROBE: 8 words per work item
};

template <bool F16, int OCC, bool TWO_VIEWS, bool PROBE>
__global__ __launch_bounds__(kCompThreads, kCompOcc) void composite_kernel(const uint32_t* __restrict__ tile_start,
                                                                 const uint32_t* __restrict__ pairs,
                                                                 const float4* __restrict__ rec,
                                                                 void* __restrict__ out, size_t pitch_bytes,
                                                                 CompParams fp, uint32_t cap,
                                                                 const uint32_t* __restrict__ order,
                                                                 uint32_t* __restrict__ queue, uint32_t ntiles,
                                                                 int prio_levels, CompExtra ex)
{
    constexpr int occ_pass = OCC;
    MSPLAT_STAMP(KID_COMPOSITE);
    uint32_t* __restrict__ const fin = ex.fin;
    float4* __restrict__ const state = ex.state;
    uint32_t* __restrict__ const probe = PROBE ? ex.probe : nullptr;
    if (OCC == 2) ntiles = *ex.d_nbins * 4u;
    // Two-pass frame (msplat_occlusion.hip.h), occ_pass 1 / 2; fin[bin * 4 + quadrant] and state[y * width + x] = (r, g, b, 2^118 T)
    // carry a tile from one to the other.  Pass 1 walks WHOLE batches only: a tile whose strips are all saturated after one of
    // them is final (fin = 0xFFFFFFFF, pixels written); otherwise it stops in front of the first incomplete batch, leaves its
    // accumulators in `state` and the number of entries it composited in fin.  Pass 2 skips the final tiles and RESUMES the others
    // at that entry of their complete list.  Batches count from the list's end and pass 1's list is a suffix of the complete one,
    // so the walk is cut at a batch boundary of the single pass: same batches, same strip tests between them, same pixels.
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
    const bool second = TWO_VIEWS && bvy >= fp.rows_view;
    const int ty = (second ? bvy - fp.rows_view : band_real_row(fp, bvy)) * 2 + (quad >> 1);
    if (tx * kTile >= fp.width || ty * kTile >= fp.height || (occ_pass == 2 && fin[bin * 4 + quad] == 0xFFFFFFFFu)) {
        // work item entirely outside the image, or (pass 2 of a two-pass frame) the tile was finished by pass 1
        if (occ_pass == 1 && threadIdx.x == 0) fin[bin * 4 + quad] = 0xFFFFFFFFu;
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
        // (r5: one step only, kCompPrioMax = 1 -- the levels above it belong to the other frames' short kernels, msplat_common.hip.h)
        if (lvl <= 3u) __builtin_amdgcn_s_setprio(kCompPrioMax);
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
    if (occ_pass == 2) end -= min(fin[bin * 4 + quad], end - start);       // resume: the entries pass 1 composited are the list's last

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
    if (occ_pass == 2) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (inside[k]) {
                const float4 st = state[(size_t)(ybase + 4 * k) * fp.width + x];
                cr[k >> 1][k & 1] = st.x; cg[k >> 1][k & 1] = st.y; cb[k >> 1][k & 1] = st.z; T[k >> 1][k & 1] = st.w;
            }
        }
    }
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
    for (int k = 0; k < NS; ++k)
        alive |= (__ballot(inside[k] && (occ_pass != 2 || T[k >> 1][k & 1] >= fp.t_eps * kScale)) != 0ull) ? (1u << k) : 0u;

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
    const uint64_t probe_t0 = PROBE ? clock64() : 0ull;
    uint32_t probe_n = 0, probe_batches = 0, probe_useful = 0;      // probe_useful: (pixel, record) evaluations with w > 0
    uint64_t probe_inner = 0;
    // pair words / records whose loads have been issued so far (the prefetch pipeline runs two / one batches ahead)
    uint32_t probe_words = min(end - start, 2u * (uint32_t)kCompThreads), probe_recs = cnt;
    uint32_t consumed = 0;                            // (pass 1) entries composited: whole batches
    while (cnt != 0u && alive != 0u && !(occ_pass == 1 && cnt < (uint32_t)kCompThreads)) {
        consumed += cnt;
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
        const uint64_t probe_t1 = PROBE ? clock64() : 0ull;
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
                    if (PROBE) probe_useful += (w.x > 0.0f ? 1u : 0u) + (w.y > 0.0f ? 1u : 0u);      // per lane; summed over the wave at the end
                    const v2f tw = T[h] * w;
                    cr[h] = __builtin_elementwise_fma(tw, vr, cr[h]);
                    cg[h] = __builtin_elementwise_fma(tw, vg, cg[h]);
                    cb[h] = __builtin_elementwise_fma(tw, vb, cb[h]);
                    T[h] = __builtin_elementwise_fma(tw, (v2f){-kScale, -kScale}, T[h]);
                }
                a = na; b = nb; c4 = nc4;
            }
        }
        if (PROBE) probe_inner += clock64() - probe_t1;
        // strips whose 64 pixels are all saturated (or outside the image) are finished
        uint32_t na = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            na |= (__ballot(inside[k] && T[k >> 1][k & 1] >= fp.t_eps * kScale) != 0ull) ? (1u << k) : 0u;
        alive = na;
        __syncthreads();
    }

    // (pass 1 of a two-pass frame) final, or to be resumed by pass 2?
    const bool carry = occ_pass == 1 && alive != 0u;
    if (occ_pass == 1 && lane == 0) fin[bin * 4 + quad] = carry ? consumed : 0xFFFFFFFFu;
    if (PROBE) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) probe_useful += (uint32_t)__shfl_xor((int)probe_useful, d, 64);
    }
    if (PROBE && lane == 0) {
        probe[tile * 8 + 0] = (uint32_t)(clock64() - probe_t0);        // shader clocks, whole tile
        probe[tile * 8 + 1] = probe_n;          // splats composited (after culling / saturation)
        probe[tile * 8 + 2] = probe_batches;    // batches of 64 list entries staged
        probe[tile * 8 + 3] = (uint32_t)probe_inner;   // shader clocks spent in the inner loops
        probe[tile * 8 + 4] = probe_words;      // 4-byte pair words loaded
        probe[tile * 8 + 5] = probe_recs;       // 48-byte projected records loaded
        probe[tile * 8 + 6] = end - start;      // length of the bin list
        probe[tile * 8 + 7] = 1u + probe_useful;      // != 0: the work item ran; - 1 = evaluations whose weight survived the discard (w > 0)
    }
    if (carry) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (inside[k])
                state[(size_t)(ybase + 4 * k) * fp.width + x] = make_float4(cr[k >> 1][k & 1], cg[k >> 1][k & 1], cb[k >> 1][k & 1], T[k >> 1][k & 1]);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (inside[k] && !carry) {
            char* row = (char*)(second ? ex.out1 : out) + (size_t)(ybase + 4 * k) * pitch_bytes;
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

}  // namespace msplat
