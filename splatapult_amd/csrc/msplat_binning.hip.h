// msplat_binning.hip.h -- tile binning: the column partition of the (splat, bin) pairs (bin1_*), the bins' list offsets and work order
// (one of the parts of msplat_kernels.hip.h)
#pragma once

#include "msplat_common.hip.h"
#include "msplat_sort.hip.h"
#include "msplat_project.hip.h"

#pragma clang fp contract(off)

namespace msplat {

// ------------------------------------------------------------------------------------------
// tile binning.  The splats are already in global depth order (rank).  Two STABLE partitions of the
// (splat, tile) pairs -- first by tile column, then by tile row -- leave every tile's list in
// draw order without ever sorting on depth again:
//   pass 1 (bin1_*):  pairs are enumerated on the fly from the rank-ordered rectangles and
//                     partitioned by column tx;       word = (row << 24) | rank
//   pass 2 (radix_*<MODE_PAIR>): partition by the row byte; the downsweep rewrites the word to
//                     (tx << 24) | rank (tx recovered from the input position), so that inside a
//                     row the words are ascending (a bin's list is one run of its row).
// ------------------------------------------------------------------------------------------

template <int BIN_CHUNK>
__global__ __launch_bounds__(kThreads) void bin1_upsweep(const uint32_t* __restrict__ rect,
                                                         const uint32_t* __restrict__ d_V,
                                                         uint32_t* __restrict__ hist, uint32_t hist_stride,
                                                         uint32_t* __restrict__ d_overflow,
                                                         uint32_t* __restrict__ gsum_acc,
                                                         uint32_t* __restrict__ gsum_zero, uint32_t gsum_zero_rows,
                                                         uint32_t* __restrict__ heavy, uint32_t* __restrict__ heavy_next,
                                                         uint8_t* __restrict__ heavy_flag, uint32_t heavy_slots, uint32_t gsup,
                                                         int keep_overflow = 0, const uint32_t* __restrict__ d_first = nullptr)
{
    MSPLAT_CHAIN_ENTER();
    MSPLAT_STAMP(KID_BIN1_UP);
    // d_first (pass 1 of a two-pass frame): the chunks start at rank *d_first (a multiple of BIN_CHUNK): the ranks below it have
    // empty rectangles and are not walked
    // keep_overflow: second binning chain of a two-pass frame -- the first chain's overflow verdict stays
    // Heavy chunks (r3).  The ranks are in depth order, so the huge far-away splats of a real scene (sky, background) are the
    // FIRST ranks: a few chunks hold half of all the pairs (scene-like 6 M cloud: 25 of 2344 chunks, 500 k pairs each against
    // 11 k), and the column pass lasted as long as the slowest of them.  A chunk with more than kHeavyPairs pairs is put on a
    // list (heavy[0] = count, heavy[1..] = chunk numbers, order irrelevant) and bin1_downsweep gives it kHeavyParts workgroups,
    // one per block of columns: columns are independent in that pass (a cursor per column), so the parts need no hand-off.
    // heavy_next is the other frame parity's counter: cleared here for the next frame.  heavy_slots <= kHeavyCap = the split
    // chunks the downsweep's grid has helper workgroups for (the host sizes it from an earlier frame's count; a chunk that
    // gets no slot is processed unsplit -- slower, never wrong).
    // per-frame reset of the sticky overflow flag (set later in the frame by bin1_downsweep): saves a memset launch
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (!keep_overflow) *d_overflow = 0u; heavy_next[0] = 0u; }
    if (gsum_zero != nullptr)      // scan-free path, see radix_upsweep
        for (uint32_t row = blockIdx.x; row < gsum_zero_rows; row += gridDim.x) gsum_zero[(size_t)row * 256 + threadIdx.x] = 0u;
    __shared__ uint32_t s_diff[kThreads + 1];
    __shared__ uint32_t s_tmp[4];
    const uint32_t V = *d_V;
    const uint32_t first = d_first != nullptr ? min(*d_first, V) : 0u;
    const uint32_t nchunks = (V - first + BIN_CHUNK - 1) / BIN_CHUNK;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        s_diff[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_diff[kThreads] = 0;
        __syncthreads();
        uint32_t rcs[BIN_CHUNK / kThreads];          // clamped loads, all in flight together (V >= 1 here)
#pragma unroll
        for (int k = 0; k < BIN_CHUNK / kThreads; ++k) rcs[k] = rect[min(first + chunk * BIN_CHUNK + k * kThreads + threadIdx.x, V - 1u)];
#pragma unroll
        for (int k = 0; k < BIN_CHUNK / kThreads; ++k) {
            const uint32_t r = first + chunk * BIN_CHUNK + k * kThreads + threadIdx.x;
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
__global__ __launch_bounds__(kThreads, (ATOMIC_RANK && BIN_CHUNK == kBinChunk) ? 4 : 2) void bin1_downsweep(const uint32_t* __restrict__ rect,
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
                                                           uint32_t gsup, const uint32_t* __restrict__ d_V_report = nullptr,
                                                           uint32_t* __restrict__ host_D2 = nullptr, uint32_t seq = 0u,
                                                           const uint32_t* __restrict__ d_first = nullptr)
{
    MSPLAT_CHAIN_ENTER();
    MSPLAT_STAMP(KID_BIN1_DOWN);
    // d_first: see bin1_upsweep
    // host_D2 (host-mapped): second binning chain of a two-pass frame -- its pair count, for the host's choice of the next share
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
    const uint32_t first = d_first != nullptr ? min(*d_first, V) : 0u;
    const uint32_t nchunks = (V - first + BIN_CHUNK - 1) / BIN_CHUNK;
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
            if (host_D2 != nullptr) {
                __hip_atomic_store(host_D2, incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(host_D2 + 2, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // ... of which frame
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
        const uint32_t rbase = first + chunk * BIN_CHUNK;
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

}  // namespace msplat
