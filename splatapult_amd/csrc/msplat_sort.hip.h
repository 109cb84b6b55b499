// msplat_sort.hip.h -- stable partitions: the 8-bit scan-free radix passes (sort fallback, the Morton sort at upload, binning's row
// partition radix_*<MODE_PAIR>) and the wide-digit three-pass depth sort (ws_*) with the cull fused into pass 0
//   shader/presort_compute.glsl + shader/multi_radixsort*.glsl (reference) -> one launch chain
// (one of the parts of msplat_kernels.hip.h)
#pragma once

#include "msplat_common.hip.h"

#pragma clang fp contract(off)

namespace msplat {

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

// (BAND = false: MODE_CULL for frames without the band-restricted cull, see cull_key)
template <int MODE, int SORT_ITEMS = kSortItems, bool BAND = true>
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
    MSPLAT_CHAIN_ENTER();
    MSPLAT_STAMP(MODE == MODE_PAIR ? KID_ROW_UP : KID_RADIX_UP);
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
                    if (in && cull_key<BAND>(pp[r], fp, key)) atomicAdd(&s_hist[digit_of<MODE>(key, shift)], 1u);
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

template <int MODE, bool HAS_VALUES, bool ATOMIC_RANK, int SORT_ITEMS = kSortItems, bool BAND = true>
__global__ __launch_bounds__(kThreads, (ATOMIC_RANK && MODE == MODE_KEYS && SORT_ITEMS == kSortItems) ? 5
                                       : (ATOMIC_RANK && MODE == MODE_CULL && SORT_ITEMS == kSortItems) ? 4 : 2) void radix_downsweep(const uint32_t* __restrict__ keys_in,
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
    MSPLAT_CHAIN_ENTER();
    MSPLAT_STAMP(MODE == MODE_PAIR ? KID_ROW_DOWN : KID_RADIX_DOWN);
    // lb.list != nullptr (MODE_CULL): pass 0 over the listed live boxes only (virtual positions), see ws_upsweep
    // gsum != nullptr: scan-free path -- hist holds raw per-chunk counts, prefixes come from the group tables;
    // otherwise hist holds exclusive prefixes and totals the digit totals (radix_scan*).
    // totals_out != nullptr: workgroup 0 publishes the digit totals (the row totals the list offsets are built from).
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
                valid[r] = in && cull_key<BAND>(pp[r % kPosBatch], fp, key[r]);
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
                // (r6: validity then lives in lrank: ITEMS lane masks kept in SGPR pairs across the phases were what spilled)
                lrank[r] = 0xFFFFFFFFu;
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
            if (ATOMIC_RANK ? lrank[r] != 0xFFFFFFFFu : valid[r]) {
                const uint32_t d = digit_of<MODE>(key[r], shift);
                const uint32_t p = s_cnt[w][d] + lrank[r];
                uint32_t kout = key[r];
                if (MODE == MODE_PAIR) {
                    // input is ordered by (column, rank): recover the column from the input position and
                    // store (tx << 24) | rank, so each row of the result is ascending.
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
// (CULL: 0 = keys of an earlier pass, 1 = pass 0 with the presort cull, 2 = the same with the band-restricted cull of a multi-GPU rank)
template <int CULL, int ITEMS, int THREADS = kWsThreads>
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
    MSPLAT_CHAIN_ENTER();
    MSPLAT_STAMP(CULL ? KID_WS_UP_CULL : KID_WS_UP);
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
                if (in) ok = cull_key<CULL == 2>(pp[r], fp, key);
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
    uint32_t* __restrict__ vals_out, uint32_t* __restrict__ d_count_out, LiveBoxes lb = LiveBoxes{nullptr, nullptr, 0u, 0u})
{
    MSPLAT_CHAIN_ENTER();
    MSPLAT_STAMP(CULL ? KID_WS_DOWN_CULL : KID_WS_DOWN);
    // lb.list != nullptr (CULL): pass 0 over the listed boxes only, see ws_upsweep -- keys_in / vmask are indexed by virtual
    // position, the value written is the splat's STORAGE index.
    // workgroup b runs on XCD b % 8; chunk = xcd_contiguous(b) gives every XCD a contiguous range of chunks, so
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

    uint32_t n = CULL ? n_static : *d_n;         // (pass 0 walks the cloud, the later passes the visible count left on the device)
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
        const uint32_t chunk = (gridDim.x >= nchunks || (gridDim.x & 7u) == 0u) ? xcd_contiguous(cidx, nchunks) : cidx;
        if (compact && t < BPC) s_box[t] = live_box_at(lb, s_lpre, chunk * BPC + (uint32_t)t);      // (barriers follow before its use)
        // this chunk's exclusive prefix per digit: the group rows before its group + the chunk rows before it in the group
        const uint32_t g = chunk >> gshift;
        uint4 pre[QPT];
        ws_row_sum<THREADS>(gsum, g, hist + (size_t)(g << gshift) * nbins, chunk - (g << gshift), nbins, bits, s_part, pre);
        __syncthreads();            // (the wide form of ws_row_sum has no barrier: s_cnt below is not the scratch, but keep the phases apart)
        for (uint32_t i = t; i < (uint32_t)WAVES * (nbins >> 3); i += THREADS) reinterpret_cast<uint4*>(s_cnt)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();

        uint32_t key[ITEMS], val[ITEMS], lrank[ITEMS];
        constexpr uint32_t kNoRank = 0xFFFFFFFFu;               // lrank of a position that holds no (visible) key
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
        uint32_t* wcnt = s_cnt + (uint32_t)w * half;
        // (r6: validity lives in lrank -- ITEMS lane masks in SGPR pairs across three phases were what spilled 13-30 SGPRs)
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = base + r * 64 + lane;
            bool ok = i < n;
            if (CULL) {
                ok = ok && ((vm[r] >> lane) & 1ull);
                val[r] = i;
                if (compact) {                        // virtual position -> storage index (a row of 64 lies in one box)
                    const uint32_t e = i - chunk * CHUNK;
                    val[r] = s_box[e / kBoxSplats] * kBoxSplats + (e % kBoxSplats);
                }
            }
            // ds_add_rtn_u32 serves the lanes of one wave instruction in ascending lane order and a wave's DS instructions
            // in program order (lds_atomic_order_probe), so the returned half-word IS the stable rank inside the wave
            const uint32_t d = (key[r] >> shift) & dmask, sh = (d & 1u) << 4;
            lrank[r] = kNoRank;
            if (ok) lrank[r] = (atomicAdd(&wcnt[d >> 1], 1u << sh) >> sh) & 0xFFFFu;
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
                const uint32_t* pc = s_cnt + 2u * qd;         // (a running pointer: WAVES scalar products k * half cost SGPRs)
#pragma unroll
                for (int k = 0; k < WAVES; ++k, pc += half) {
                    const uint2 x = *reinterpret_cast<const uint2*>(pc);
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
                uint32_t* ps = s_cnt + 2u * qd;
#pragma unroll
                for (int k = 0; k < WAVES; ++k, ps += half) {
                    uint2* slot = reinterpret_cast<uint2*>(ps);
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
            if (lrank[r] != kNoRank) {
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

}  // namespace msplat
