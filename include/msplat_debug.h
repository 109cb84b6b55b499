/*
 * msplat_debug.h -- parity-test taps, probes and experiment handles of libmsplat.so.  Nothing here is needed to bind
 * SplatRenderer::Sort / Render (that is msplat.h); these entry points let tests/ compare intermediate results with the oracle,
 * let bench.py count what the compositor fetched, and let a test pin what the library otherwise steers by itself.
 * All of them synchronise the context.
 */
#ifndef MSPLAT_DEBUG_H
#define MSPLAT_DEBUG_H

#include "msplat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- intermediate results of the latest Render ---- */
/* per draw-order rank r < V: 12 floats {px, py, A, B, C, log2(alpha), r, g, b, alpha, 0, 0} with
 * w(dx, dy) = exp2(A dx^2 + B dx dy + C dy^2 + log2 alpha); rect = bin rectangle tx0 | ty0<<8 | tx1<<16 | ty1<<24 (tx0 > tx1: empty) */
int msplat_debug_get_projected(msplat_ctx* ctx, float* rec12, uint32_t* rect, uint32_t cap);
/* tile_start has tiles_x * tiles_y + 1 entries; pairs[k] & 0xFFFFFF = draw-order rank.  After a two-pass Render both (and
 * msplat_get_stats().pairs / drawn) describe the SECOND pass */
int msplat_debug_get_tile_lists(msplat_ctx* ctx, uint32_t* tile_start, uint32_t tile_cap, uint32_t* pairs, uint64_t pair_cap);
/* on-device self-check of the ordering contracts (sorted keys ascending, ties by ascending storage slot; every bin list
 * ascending in draw-order rank): violation counts, both 0 on a healthy context.  Guards the lane-ordered LDS-atomic ranking,
 * which msplat_create probes but the hardware does not document (msplat_config.rank_mode = MSPLAT_RANK_BALLOT selects the ballot path) */
int msplat_debug_verify_order(msplat_ctx* ctx, uint32_t* key_violations, uint32_t* list_violations);
/* chunk-level cull for the latest Sort's camera: live / all bounding boxes (of 256 stored splats; 0 / 0 for a cloud in upload
 * order); *listed (may be NULL) = 1 when that Sort's first pass walked only the listed live boxes */
int msplat_debug_get_cull_boxes(msplat_ctx* ctx, uint32_t* live, uint32_t* total, int* listed);

/* ---- two-pass frames (msplat_config.two_pass) ---- */
/* share > 0 pins the share of the visible splats in the first pass (any value gives the same pixels), 0 hands it back to the
 * feedback loop.  two_pass_frames: Renders of the context that ran in two passes; share_now: what the next one would use */
int msplat_debug_two_pass(msplat_ctx* ctx, float share, uint64_t* two_pass_frames, float* share_now);
/* the context's latest two-pass Render: out[0] = two-pass Renders so far (0: the rest is meaningless), [1] = splats projected by
 * pass 1, [2] = splats behind the cut that passed the gate (pass 2), [3] / [4] = (splat, bin) pairs binned by pass 1 / 2,
 * [5] = bins pass 1 left unfinished, [6] = bins, [7] = visible splats */
int msplat_get_two_pass_info(msplat_ctx* ctx, uint64_t out[8]);

/* ---- compositor probe (bench statistics): per (bin, quadrant) work item 8 words {shader clocks, records composited, batches
 * staged, inner-loop clocks, pair words fetched, records fetched, bin-list length, 1 + evaluations with w > 0}.  Off by default (a few clock reads per
 * batch; a probed context renders in one pass) ---- */
typedef struct msplat_composite_work {     /* sums over the work items of the latest render */
    uint64_t work_items;          /* 16x16 tiles composited */
    uint64_t list_entries;        /* sum of the bin-list lengths: what it would fetch without early termination */
    uint64_t pair_words_fetched;  /* 4-byte list entries whose loads were issued */
    uint64_t records_fetched;     /* 48-byte projected records whose loads were issued */
    uint64_t records_composited;  /* records that passed the exact footprint test of their 16x16 tile */
    uint64_t pixel_evals;         /* (pixel, splat) evaluations = 256 per composited record */
    uint64_t batches;             /* 64-entry batches staged */
    uint64_t clocks_sum, clocks_max, inner_clocks_sum;   /* shader clocks per work item (probe overhead included) */
    uint64_t useful_evals;        /* r6: evaluations whose weight survived the discard (w > 1/256): the rest of pixel_evals is the price
                                     of evaluating a whole 16x16 tile per record (splat_frag.glsl:37-40 runs per covered fragment) */
} msplat_composite_work;
int msplat_set_tile_probe(msplat_ctx* ctx, int enable);
/* (the record size is in the name: a caller built for the 4-word records of the first release fails to link) */
int msplat_debug_get_tile_probe8(msplat_ctx* ctx, uint32_t* dst8, uint32_t tile_cap);
int msplat_get_composite_work(msplat_ctx* ctx, msplat_composite_work* out);

/* ---- msplat_band_exchange on ONE rank (tests on a one-GPU box): the runs of bin rows that rank `rank` of `world` owns travel
 * from src to dst through ncclSend / ncclRecv to the calling rank itself (`comm` = a 1-rank communicator) ---- */
int msplat_debug_band_exchange_loopback(msplat_ctx* ctx, void* comm, int32_t kind, int32_t block_rows, int32_t world, int32_t rank,
                                        const void* src, void* dst, uint64_t pitch_bytes, int32_t width, int32_t height, int32_t flags);

/* ---- msplat_config.cu_partition: the MSPLAT_CU_* the context's own stream really got (MSPLAT_CU_ALL on a caller's stream or when
 * the runtime refused the mask); mask8 != NULL: the stream's CU mask as hipExtStreamGetCUMask reports it (8 words) ---- */
int msplat_debug_cu_partition(msplat_ctx* ctx, uint32_t* mask8);

#ifdef __cplusplus
}
#endif
#endif /* MSPLAT_DEBUG_H */
