/*
 * msplat.h -- C ABI of libmsplat.so, the MI355X-native (gfx950 / CDNA4, HIP) replacement for
 * the reference's SplatRenderer::Sort()/Render() hot path.
 *
 * The reference has no plugin/FFI layer: the seam is the C++ class SplatRenderer
 * (/root/reference/src/splatrenderer.h:23-67) fed by GaussianCloud (src/gaussiancloud.h:17-91).
 * Every entry point below names the reference interface it replaces.  The C++ shim
 * splatapult_amd/host/msplat_host.hpp re-creates the reference's class surface on top of this
 * ABI; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions (identical to the reference): matrices are float[16], column-major like glm;
 * cameraMat = camera-to-world; viewport = (x, y, W, H); nearFar = (near, far).
 * All functions return 0 (MSPLAT_OK) on success, a negative MSPLAT_ERR_* otherwise, and never
 * throw.  A context is not thread-safe: calls on one context are serialised by the caller
 * (same as the reference's single GL thread).  No torch / framework types cross this boundary.
 */
#ifndef MSPLAT_H
#define MSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSPLAT_VERSION 1

enum {
    MSPLAT_OK = 0,
    MSPLAT_ERR_INVALID_ARG = -1,
    MSPLAT_ERR_NO_DEVICE = -2,     /* no HIP device / HIP runtime failure at create         */
    MSPLAT_ERR_HIP = -3,           /* a HIP call failed; see msplat_last_error               */
    MSPLAT_ERR_NO_CLOUD = -4,      /* sort/render before upload                              */
    MSPLAT_ERR_NO_SORT = -5,       /* render before sort                                     */
    MSPLAT_ERR_UNSUPPORTED = -6,   /* e.g. viewport larger than 8192x8192                    */
    MSPLAT_ERR_PAIR_OVERFLOW = -7, /* (splat,bin) pair buffer too small: host-output renders grow it and retry,   */
                                   /* device-output renders report it on the NEXT call (see msplat_render)     */
    MSPLAT_ERR_IO = -8,            /* PLY open/parse failure                                 */
    MSPLAT_ERR_PAIR_OVERFLOW_EARLIER = -9  /* msplat_sort / msplat_render: THIS call did its work (the sort ran, the image
                                      was written), but an EARLIER device-output render on the context had overflowed the
                                      pair buffer (that frame lacks splats; the buffer has been grown unless its capacity is
                                      fixed).  A warning about a past frame, not a failure of the call.  msplat_synchronize
                                      reports the same event as MSPLAT_ERR_PAIR_OVERFLOW. */
};

enum { MSPLAT_FB_RGBA32F = 0, MSPLAT_FB_RGBA16F = 1 };
/* render-target emulation (msplat_set_target_emulation) */
enum { MSPLAT_ROP_NONE = 0, MSPLAT_ROP_RGBA8 = 1, MSPLAT_ROP_RGBA16F = 2 };

typedef struct msplat_ctx msplat_ctx;
typedef struct msplat_cloud msplat_cloud;

/* Construction parameters.  Replaces the implicit GL state the reference's renderer lives in:
 * device = the GL context's GPU (sdl_main.cpp:98-100); fb_format = App's --fp16/--fp32 FBO
 * choice (app.cpp:1000-1035); srgb = SplatRenderer::Init's isFramebufferSRGBEnabled
 * (splatrenderer.cpp:60-72). */
typedef struct msplat_config {
    uint32_t struct_size;      /* sizeof(msplat_config), for ABI evolution                   */
    int32_t device;            /* HIP device ordinal                                         */
    int32_t fb_format;         /* MSPLAT_FB_*                                                */
    int32_t srgb;              /* FRAMEBUFFER_SRGB path of splat_vert.glsl:129-151,209-218   */
    float t_epsilon;           /* front-to-back early-out: stop a pixel when its             */
                               /* transmittance < t_epsilon. 0 = never (exact).              */
                               /* negative = library default (2^-14)                         */
    uint64_t pair_capacity;    /* max (splat,tile) pairs per render; 0 = auto (grows)        */
    void* stream;              /* hipStream_t to launch on; NULL = library-owned stream      */
    int32_t enable_timing;     /* n > 0: record per-stage hipEvents on every n-th sort/render */
                               /* (msplat_get_timings averages them); 0 = never             */
    int32_t compositor_waves;  /* persistent compositor waves per render; 0 = default (8192, the  */
                               /* measured best for one frame at a time; the SplatRenderer shims   */
                               /* use 1280 with frames in flight so that frames share the CUs)     */
    int32_t rank_mode;         /* MSPLAT_RANK_*: how the stable radix / binning passes rank the    */
                               /* keys of one wave.  Added after the first release of the struct:  */
                               /* a struct_size that ends before this field selects MSPLAT_RANK_AUTO */
    int32_t frame_mode;        /* MSPLAT_FRAMES_*: is this context the only one working on the GPU, or one of */
                               /* several frames in flight?  Occupies what was padding after rank_mode: any   */
                               /* value other than the two named ones means AUTO                               */
    int32_t spatial_order;     /* MSPLAT_SPATIAL_* (r4): may the library store the cloud in its own (Morton)  */
                               /* order so that the cull can skip whole chunks?  A shorter struct_size = AUTO  */
    int32_t async_submit;      /* r4: != 0: msplat_sort and device-output msplat_render return at once; their   */
                               /* launches are issued by a worker thread of the context (frames in flight)     */
    int32_t two_pass;          /* MSPLAT_TWO_PASS_* (r4): may a Render run as two passes with occlusion feedback */
                               /* (same pixels, less work)?  A shorter struct_size = AUTO                      */
    int32_t reserved0;         /* 0 */
} msplat_config;

/* msplat_config.two_pass -- a frame in two passes with occlusion feedback (splatapult_amd/csrc/msplat_occlusion.hip.h).
 * The compositor stops a tile when its pixels are saturated, but projection and binning process every visible splat: most of
 * the (splat, bin) pairs they produce are never read (76 % at BASELINE config 2, 87 % at config 4, 98 % with a camera inside a
 * scene).  A two-pass Render first projects, bins and composites the NEAREST splats only; tiles saturated by them are final;
 * a 16-byte test per remaining splat (centre + footprint bound against the map of unfinished bins) then decides whether its
 * 256-byte record is fetched at all, and only the unfinished bins are binned and composited again -- from complete lists, from
 * scratch.  The image is bit for bit the image of the single pass (every finished tile saw exactly the list entries the
 * single pass would have consumed, in the same batches), whatever share of the splats goes into the first pass; that share is
 * steered from counts an earlier frame left in host-mapped memory, never waited for.
 * AUTO: clouds of >= 262 144 splats on the splat compositor (no emulated depth test / render-target rounding / points / two
 * views in one chain), after the context's first 8 frames, not while the tile probe is on.  A two-pass frame costs nine more
 * launches, so AUTO probes (four frames, then one pass until their counts are in) and keeps two passes only where they pay:
 * where more than 30 % of the bins never saturate (a 1 M-splat cloud seen from outside) it goes back to one pass and tries
 * again after 1024 frames, then 2048, ...
 * After a two-pass Render, msplat_get_stats().pairs / drawn and the debug list getters describe the SECOND pass. */
enum {
    MSPLAT_TWO_PASS_AUTO = 0,
    MSPLAT_TWO_PASS_ON = 1,
    MSPLAT_TWO_PASS_OFF = 2
};

/* msplat_config.async_submit.  Issuing a frame (~14 kernel launches) costs the host ~55 us.  A caller that keeps several frames
 * in flight on several contexts from ONE thread therefore starts the k-th context k x 55 us after the first -- a stagger that a
 * short block of frames pays at both ends.  With async_submit every such context owns a worker thread: msplat_sort and
 * msplat_render with out_is_device = 1 copy their arguments (the caller's arrays may be reused at once), queue the call and return
 * MSPLAT_OK; the worker issues the launches in call order.  Every other entry point of the context first waits until the worker
 * has ISSUED what is queued (that is host work, microseconds; never a wait for the GPU), so msplat_stream_wait / msplat_wait_event /
 * the getters behave as without it.  A queued call that fails is reported by the next msplat_synchronize (MSPLAT_ERR_* of the
 * first failure).  Host-output renders stay synchronous.  The SplatRenderer shims switch it on in SetFramesInFlight. */

/* msplat_config.spatial_order -- the STORAGE order of the uploaded cloud and the tie rule of the sort.
 * The reference culls per splat over the whole cloud every frame (shader/presort_compute.glsl:31-57 dispatched over N,
 * src/splatrenderer.cpp:188-189) and so did pass 0 of this sort: 24 us of 249 when everything is visible, but most of the sort
 * when little is -- a rank of a row-sharded frame keeps 17 % of the splats, a camera inside a scene 40 %.  With spatial order the
 * upload stores the cloud sorted by the Morton code of the positions (ties in upload order), keeps one bounding box per 256
 * stored splats, and a Sort whose predecessor saw less than 70 % of the cloud first lists the boxes that can hold a visible
 * splat and runs its cull + first radix pass over those alone.  The box test is conservative -- the visible set, the keys and
 * every pixel are what the per-splat test alone gives.
 * What it changes is the TIE RULE.  Draw order = ascending 32-bit depth key (splatrenderer.cpp:165-169), and splats with EQUAL
 * keys are drawn in ascending storage slot.  The reference's own tie order is undefined (the slots come from an atomic counter,
 * presort_compute.glsl:50), so any rule is one of its outcomes; this one is deterministic and reproducible:
 * msplat_get_storage_order returns the permutation (slot -> upload index), and a CPU oracle fed the cloud in that order gives the
 * same frame bit for bit (tests/).  Without reordering the storage slot IS the upload index.  Everything the API reports about
 * splats (msplat_get_sorted_indices, msplat_download_cloud) is in upload numbering either way.
 * AUTO: on from 262 144 splats (below that a pass-0 chunk is a large part of the cloud and there is nothing to skip); point
 * clouds are never reordered.  MSPLAT_SPATIAL_ORDER=0|1 in the environment overrides the field. */
enum {
    MSPLAT_SPATIAL_AUTO = 0,
    MSPLAT_SPATIAL_ON = 1,
    MSPLAT_SPATIAL_OFF = 2
};

/* msplat_config.frame_mode.  Pixels, keys and lists are identical in both modes; what changes is which kernels run.
 * MSPLAT_FRAMES_SERIAL (= AUTO): one frame at a time -- the shortest single frame: the three-pass wide-digit sort
 *   (58 us instead of 73 at 1 M splats, 162 instead of 204 at 6 M) and the bins' list offsets from the row pass's pair
 *   counts (two launches fewer); 12 launches per frame.
 * MSPLAT_FRAMES_IN_FLIGHT: the context shares the GPU with other contexts' frames (SetFramesInFlight, bench.py's default
 *   mode).  Launch latency is hidden by the other frames there and what counts is total work and how easily a workgroup
 *   finds a free CU: the three-pass sort runs in its 4-wave form (256 threads, 40 KB of LDS, against 8 waves and 72 KB)
 *   up to 2 M splats and the four 8-bit passes take over beyond (4 waves, 24 KB; at 6 M they measure 8 % more frames/s
 *   than three 4-wave passes), and the list offsets come from the search kernels (the persistent compositor needs no
 *   bin order).  Measured r3 at config 2, 4 frames in flight, same box: 6045 frames/s against 5800 with the SERIAL
 *   kernels.  The SplatRenderer shims select it for their in-flight contexts.
 * In the environment MSPLAT_SORT=lsd8|wide and MSPLAT_TILE_TABLE=search|count override the choices one by one. */
enum {
    MSPLAT_FRAMES_AUTO = 0,
    MSPLAT_FRAMES_SERIAL = 1,
    MSPLAT_FRAMES_IN_FLIGHT = 2
};

/* msplat_config.rank_mode */
enum {
    MSPLAT_RANK_AUTO = 0,      /* lane-ordered LDS atomics if the probe run by msplat_create confirms */
                               /* that ds_add_rtn hands out return values in lane order, else ballots */
    MSPLAT_RANK_BALLOT = 1     /* always the ballot / popcount ranking (no reliance on that ordering) */
};

/* Byte offsets of the attributes inside one AoS record, i.e. the BinaryAttribute offsets that
 * SplatRenderer::BuildVertexArrayObject binds (splatrenderer.cpp:345-391;
 * gaussiancloud.cpp:633-657).  r_sh1..b_sh3 are ignored unless full_sh. */
typedef struct msplat_attr_offsets {
    uint32_t pos_with_alpha;
    uint32_t r_sh0, g_sh0, b_sh0;
    uint32_t cov3_col0, cov3_col1, cov3_col2;
    uint32_t r_sh1, r_sh2, r_sh3;
    uint32_t g_sh1, g_sh2, g_sh3;
    uint32_t b_sh1, b_sh2, b_sh3;
} msplat_attr_offsets;

typedef struct msplat_stats {
    uint64_t num_splats;       /* N                                                          */
    uint32_t sort_count;       /* V: splats that survived the presort cull (sortCount)       */
    uint32_t drawn;            /* splats that passed the geometry-stage guard band           */
    uint64_t pairs;            /* (splat, bin) pairs actually binned (bins of msplat_tile_size()) */
    uint32_t tiles_x, tiles_y;
    uint32_t width, height;
    uint64_t pair_capacity;
    uint64_t device_bytes;     /* device memory held by the context                          */
    uint64_t pairs_tile16;     /* (splat, 16x16 tile) pairs covered by the footprints: the D */
                               /* of the algorithmic byte count               */
} msplat_stats;

/* what the compositor (the dominant kernel) did in the last render, summed over its (bin, quadrant) work items;
 * filled by msplat_get_composite_work when the tile probe is on (msplat_set_tile_probe) */
typedef struct msplat_composite_work {
    uint64_t work_items;          /* 16x16 tiles composited                                               */
    uint64_t list_entries;        /* sum of the bin-list lengths: what it would fetch without early-out   */
    uint64_t pair_words_fetched;  /* 4-byte list entries whose loads were issued                          */
    uint64_t records_fetched;     /* 48-byte projected records whose loads were issued                    */
    uint64_t records_composited;  /* records that passed the exact footprint test of their work item: a 16x16 */
                                  /* tile (default), a 16x8 half tile or an 8x8 sub-block (MSPLAT_COMPOSITOR)  */
    uint64_t pixel_evals;         /* (pixel, splat) evaluations = 128 / 256 / 64 per composited record    */
    uint64_t batches;             /* 64-entry batches staged                                              */
    uint64_t clocks_sum, clocks_max, inner_clocks_sum;   /* shader clocks per work item (probe overhead included) */
} msplat_composite_work;

typedef struct msplat_timings {
    /* milliseconds, last frame, valid when enable_timing; names follow the reference's Tracy
     * zones (splatrenderer.cpp:156,172,208,297,318) */
    float sort_total;          /* "SplatRenderer::Sort"  (pre-sort + sort)                   */
    float render_total;        /* "SplatRenderer::Render"                                    */
    float project;             /* vertex+geometry stage equivalent                           */
    float binning;             /* tile lists (count, scan, two stable partition passes)      */
    float composite;           /* fragment+blend equivalent (the dominant kernel)            */
    float reserved[3];         /* [0] = frames averaged, [1] = compositor KERNEL time (exact    */
                               /* dispatch begin/end events), [2] = launches in that average   */
} msplat_timings;

/* ---- context ---------------------------------------------------------------------------- */
/* replaces SplatRenderer::SplatRenderer / ~SplatRenderer (splatrenderer.cpp:41-48) */
int msplat_create(msplat_ctx** out, const msplat_config* cfg);
void msplat_destroy(msplat_ctx* ctx);
const char* msplat_last_error(const msplat_ctx* ctx);   /* ctx may be NULL: global last error */
const char* msplat_version_string(void);
/* edge of the square screen bins (pixels) that tile lists and msplat_set_band rows refer to */
int msplat_tile_size(void);

/* replaces SplatRenderer::Init + BuildVertexArrayObject (splatrenderer.cpp:50-151,345-391):
 * copies the interleaved cloud to the device (the caller may free it afterwards, as the
 * reference does not retain the shared_ptr). `aos` is host memory, n records of stride_bytes. */
int msplat_upload_cloud(msplat_ctx* ctx, const void* aos, uint64_t n, uint32_t stride_bytes,
                        const msplat_attr_offsets* off, int full_sh);

/* GPU ingest: GaussianCloud::ImportPly's per-vertex math (gaussiancloud.cpp:254-361:
 * sigmoid(opacity), exp(scale), quaternion -> R S S^T R^T, SH repack) as a HIP kernel over the raw PLY
 * vertex block, building the renderer's device cloud directly.  Byte offsets of the float properties inside
 * one vertex; -1 = property absent (reads as 0, like BinaryAttribute::Read).  f_rest is optional: if any
 * is absent, or full_sh == 0, the cloud is SH degree 0 (gaussiancloud.cpp:188-205). */
typedef struct msplat_ply_layout {
    uint32_t vertex_size;
    int32_t x, y, z;
    int32_t f_dc[3];
    int32_t f_rest[45];
    int32_t opacity;
    int32_t scale[3];
    int32_t rot[4];
} msplat_ply_layout;
int msplat_upload_ply_vertices(msplat_ctx* ctx, const void* vertices, uint64_t n,
                               const msplat_ply_layout* layout, int full_sh);
/* Ply::Parse (ply.cpp:72-87) on the host + msplat_upload_ply_vertices: replaces
 * GaussianCloud::ImportPly + SplatRenderer::Init for callers that do not need the host-side cloud */
int msplat_upload_ply(msplat_ctx* ctx, const char* path, int import_full_sh);
/* the device cloud in the reference's interleaved layout (100 B / 244 B records); parity tests */
int msplat_download_cloud(msplat_ctx* ctx, void* aos_out, uint64_t cap_bytes);

/* Multi-GPU tile-row sharding (no reference counterpart).  Restricts this
 * context to tile rows t with t % row_mod == row_rem (rows of msplat_tile_size() pixels, row 0 = GL
 * bottom).
 * row_mod = 1 (default) = whole image.  The framebuffer handed to msplat_render is always the
 * full W x H image; only rows owned by the band are written. */
int msplat_set_band(msplat_ctx* ctx, int32_t row_mod, int32_t row_rem);
/* The general form (r3): the context owns blocks of `block` consecutive bin rows starting at first_row,
 * first_row + stride, first_row + 2 stride, ... (stride >= block), at most row_count rows in total (0 = as many as the
 * image has).  Contiguous band g of G over R rows (the north star's "tiles row-sharded"): (g R / G, (g+1) R / G - g R / G,
 * that count, anything >= it); interleaved rows: (g, 0, 1, G); blocks of k rows dealt round-robin: (g k, 0, k, G k).
 * msplat_band_plan computes these.  Pixels are bit-identical to the unbanded frame for every layout. */
int msplat_set_band_layout(msplat_ctx* ctx, int32_t first_row, int32_t row_count, int32_t block, int32_t stride);
enum { MSPLAT_BANDS_CONTIGUOUS = 0, MSPLAT_BANDS_INTERLEAVED = 1, MSPLAT_BANDS_BLOCK_INTERLEAVED = 2 };
/* layout parameters of rank `rank` of `world` for `rows_full` = ceil(H / msplat_tile_size()) bin rows; block_rows is
 * only read for MSPLAT_BANDS_BLOCK_INTERLEAVED.  Host arithmetic only (works without a GPU). */
int msplat_band_plan(int32_t kind, int32_t rows_full, int32_t world, int32_t rank, int32_t block_rows, int32_t* first_row,
                     int32_t* row_count, int32_t* block, int32_t* stride);
/* Band-restricted cull: with a band set, msplat_sort additionally drops splats whose
 * footprint (conservative bound) cannot reach a row owned by this context, so sort / projection / binning
 * shrink with the number of ranks.  Pixels are unchanged, but msplat_sort_count and the sorted list then
 * describe the band only, and every msplat_render must use the camera of the preceding msplat_sort
 * (mono rendering).  Default off; ignored without a band. */
int msplat_set_band_cull(msplat_ctx* ctx, int enable);

/* Depth-buffer emulation.  The reference enables GL_DEPTH_TEST (app.cpp:163, GL_LESS,
 * depth writes on).  It is inert on the colour-only --fp16/--fp32 FBO (app.cpp:1027) -- bits = 0, the
 * default and the configuration every other entry point models -- and live on the default back buffer
 * (24-bit, sdl_main.cpp:79) and the XR swapchains: there a fragment that survives the discard must also
 * pass z < depth buffer, so later-drawn splats lose fragments where quantised depths tie or where the
 * draw order is not the depth order (second eye rendered with the first eye's sort).  bits = 24: 24-bit
 * unorm depth; 32: float depth.  Renders then walk every tile list in draw order without early
 * termination (several times slower); meant for diffing against the GL app's output. */
int msplat_set_depth_test(msplat_ctx* ctx, int depth_bits);

/* The blend as the GL app's render target performs it (src/app.cpp:1012-1020).  The reference's
 * default RGBA8 back buffer clamps source, destination and result to [0,1] and stores 8-bit unorm after EVERY blend
 * (GL 4.6 17.3.6); its --fp16 target rounds to fp16 after every blend; its --fp32 target (what every other entry
 * point models) does neither.  rop = MSPLAT_ROP_RGBA8 / MSPLAT_ROP_RGBA16F reproduce the first two for callers who
 * diff against the GL app's pixels: renders then walk every bin list in draw order with the literal blend and no
 * early termination (several times slower).  The values are written in the context's fb_format (RGBA8 results are
 * multiples of 1/255).  The arithmetic inside a ROP is implementation-defined: this restates the specification. */
int msplat_set_target_emulation(msplat_ctx* ctx, int rop);

/* replaces SplatRenderer::Sort (splatrenderer.cpp:153-312): cull + depth key
 * (presort_compute.glsl:31-57), stable ascending 32-bit radix sort, sorted index list kept as
 * context state for subsequent renders.  Asynchronous: no host readback (the reference's
 * 4-byte glMapBufferRange stall, splatrenderer.cpp:195-204, is not reproduced). */
int msplat_sort(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                const float viewport[4], const float nearFar[2]);

/* replaces SplatRenderer::Render (splatrenderer.cpp:315-343) *plus* the GL pipeline behind
 * its glDrawElements: vertex (splat_vert.glsl), geometry (splat_geom.glsl), fragment
 * (splat_frag.glsl) and the blend/clear state of app.cpp:144-164.  Writes W x H RGBA
 * (float or half per cfg.fb_format), row 0 = GL bottom row, alpha = 1, into `rgba`.
 * out_is_device != 0: `rgba` is a device pointer, the call is asynchronous on the stream.
 * out_is_device == 0: `rgba` is host memory, the call returns after the copy completed.
 * pitch_bytes = bytes between rows (0 = tightly packed).
 * Pair-buffer overflow (more (splat, bin) pairs than msplat_stats.pair_capacity; default 32 per splat): a
 * host-output render grows the buffer and retries before it returns.  A device-output render cannot know:
 * the binning kernel leaves the needed pair count in host-mapped memory, and the NEXT msplat_sort /
 * msplat_render / msplat_synchronize on the context (which still does its own work) grows the buffer -- unless
 * msplat_config.pair_capacity fixed it -- and reports it once (msplat_synchronize: MSPLAT_ERR_PAIR_OVERFLOW; msplat_sort /
 * msplat_render, whose own work is still done: MSPLAT_ERR_PAIR_OVERFLOW_EARLIER): the frame that overflowed
 * lacks splats in its last bin columns and should be re-rendered.
 * Limits: at most 2^24 splats per cloud (24-bit rank field in the pair words) and viewports up to 8192 x 8192
 * (256 x 256 bins of 32 px); beyond them msplat_upload_* / msplat_sort return MSPLAT_ERR_UNSUPPORTED. */
int msplat_render(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16],
                  const float viewport[4], const float nearFar[2],
                  void* rgba, uint64_t pitch_bytes, int out_is_device);

/* Two views of the latest Sort in ONE chain of launches -- the reference's VR frame: Sort with the first eye's matrices, then
 * Render per eye (src/app.cpp:603-607).  Same arguments and results as
 *     msplat_render(ctx, cameraMat0, projMat0, viewport, nearFar, rgba0, pitch_bytes, out_is_device);
 *     msplat_render(ctx, cameraMat1, projMat1, viewport, nearFar, rgba1, pitch_bytes, out_is_device);
 * bit for bit, but projection, binning and compositing each run once over both views' work (6 launches instead of 12:
 * BASELINE configs[4], 2 x 2016 x 2240, 0.50 -> 0.44 ms per stereo frame).  Device targets of a plain splat context take that
 * form; host targets, banded contexts, point clouds, depth-test / render-target emulation, clouds beyond 2^23 splats and
 * viewports taller than 4096 px are rendered view after view. */
int msplat_render_stereo(msplat_ctx* ctx, const float cameraMat0[16], const float projMat0[16], const float cameraMat1[16],
                         const float projMat1[16], const float viewport[4], const float nearFar[2], void* rgba0, void* rgba1,
                         uint64_t pitch_bytes, int out_is_device);

/* blocks until everything queued on the context's stream has finished */
int msplat_synchronize(msplat_ctx* ctx);

/* ---- frames in flight -------------------------------------------------------------------
 * The reference queues Sort and Render of successive frames on one GL command stream and lets the
 * driver overlap them.  Here a frame's ~20 launches are a dependent chain on one HIP stream, so the
 * overlap is made explicit: create one context per frame in flight (each has its own stream and
 * per-frame buffers), upload the cloud into the first and attach it to the others (no copy), then
 * issue frame k's Sort + Render(s) on context k % depth.  Results are bit-identical to a single
 * context.  The C++ / Python SplatRenderer shims do this rotation (SetFramesInFlight).
 * The ROCm runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (environment variable, default 4,
 * read when the runtime initialises); streams sharing a queue serialise, so a process that wants 4 frames
 * in flight next to its own streams should start with GPU_MAX_HW_QUEUES=8 (bench.py does). */
/* `ctx` renders `owner`'s cloud (same device).  A later upload into either context detaches it. */
int msplat_attach_cloud(msplat_ctx* ctx, msplat_ctx* owner);
/* makes `stream` (hipStream_t; NULL = default stream) wait, on the device, for everything queued so
 * far on the context's stream -- e.g. before a collective or a readback of the frame just rendered */
int msplat_stream_wait(msplat_ctx* ctx, void* stream);
/* the reverse join: the context's stream waits for `event` (a hipEvent_t the caller recorded, e.g. after
 * the consumer of a framebuffer that the next frame issued on this context will overwrite) */
int msplat_wait_event(msplat_ctx* ctx, void* event);

/* the hipStream_t the context launches on (the one given in msplat_config.stream, or the library's own) */
void* msplat_get_stream(msplat_ctx* ctx);

/* ---- several GPUs, one process (no reference counterpart: the reference is one single-threaded C++ process
 * calling Sort / Render, src/app.cpp:1067-1068, and this keeps that shape) ------------------------------------------------
 * One group = one context per listed device.  The cloud is replicated, the screen's bin rows are partitioned over the
 * devices (msplat_group_set_layout; default MSPLAT_BANDS_CONTIGUOUS, the north star's "tiles row-sharded"), every device
 * runs the whole pipeline on its rows, and the ONLY exchange is the row gather: with out_is_device != 0 `rgba` is memory of
 * devices[0] and every other device's compositor stores its rows straight into it through the peer mapping (xGMI, one
 * direct link per device, no staging, no RCCL); where no peer mapping exists the rows are staged locally and copied.
 * Pixels are bit-identical to one context rendering the whole frame.  msplat_group_render returns once every device's work
 * has been ISSUED; the stream of context 0 (msplat_get_stream(msplat_group_context(g, 0))) waits for the others, so
 * synchronising it -- or msplat_group_synchronize -- means the frame is complete.  Calls on one group are serialised by
 * the caller (like a context); devices beyond the first are driven by the library's own worker threads.
 * devices may repeat an ordinal (several contexts on one GPU: tests).  cfg: as msplat_create (device is ignored; stream
 * must be NULL when n > 1). */
typedef struct msplat_group msplat_group;
int msplat_group_create(msplat_group** out, const int32_t* devices, uint32_t n, const msplat_config* cfg);
void msplat_group_destroy(msplat_group* g);
const char* msplat_group_last_error(const msplat_group* g);     /* g may be NULL */
uint32_t msplat_group_size(const msplat_group* g);
msplat_ctx* msplat_group_context(msplat_group* g, uint32_t i);  /* borrowed: stats, timings, parity taps of rank i */
int msplat_group_peer_store(const msplat_group* g, uint32_t i); /* 1: rank i writes device 0's framebuffer directly */
/* replicated uploads (msplat_upload_cloud / msplat_upload_gaussian_cloud / msplat_upload_ply on every device) */
int msplat_group_upload_cloud(msplat_group* g, const void* aos, uint64_t n, uint32_t stride_bytes,
                              const msplat_attr_offsets* off, int full_sh);
int msplat_group_upload_gaussian_cloud(msplat_group* g, const msplat_cloud* c);
int msplat_group_upload_ply(msplat_group* g, const char* path, int import_full_sh);
/* MSPLAT_BANDS_*; block_rows only for MSPLAT_BANDS_BLOCK_INTERLEAVED */
int msplat_group_set_layout(msplat_group* g, int32_t kind, int32_t block_rows);
/* msplat_set_band_cull on every context: mono rendering only (every Render uses its Sort's camera) */
int msplat_group_set_band_cull(msplat_group* g, int enable);
/* SplatRenderer::Sort / Render over the group: same arguments as msplat_sort / msplat_render */
int msplat_group_sort(msplat_group* g, const float cameraMat[16], const float projMat[16], const float viewport[4],
                      const float nearFar[2]);
int msplat_group_render(msplat_group* g, const float cameraMat[16], const float projMat[16], const float viewport[4],
                        const float nearFar[2], void* rgba, uint64_t pitch_bytes, int out_is_device);
int msplat_group_synchronize(msplat_group* g);

/* sortCount of the last Sort (splatrenderer.cpp:198-199); synchronises */
int msplat_sort_count(msplat_ctx* ctx, uint32_t* v);
/* the element buffer the reference fills at splatrenderer.cpp:296-311: upload indices of the visible splats in draw order
 * (ascending key = far to near; equal keys in ascending storage slot, see msplat_config.spatial_order); synchronises */
int msplat_get_sorted_indices(msplat_ctx* ctx, uint32_t* dst, uint32_t cap);
int msplat_get_sorted_keys(msplat_ctx* ctx, uint32_t* dst, uint32_t cap);
/* storage order of the uploaded cloud: dst[slot] = upload index (identity unless the cloud was reordered; *reordered, if not
 * NULL, says which).  dst may be NULL to ask only whether.  cap = entries dst can take (>= N). */
int msplat_get_storage_order(msplat_ctx* ctx, uint32_t* dst, uint64_t cap, int* reordered);
/* chunk-level cull, evaluated for the latest Sort's camera: bounding boxes (of 256 stored splats) that can hold a visible splat /
 * boxes in the cloud (0 / 0 for a cloud in upload order); *listed (may be NULL) = 1 when that Sort's first pass walked only the
 * listed live boxes (it does once an earlier frame saw less than 70 % of the cloud).  Synchronises. */
int msplat_debug_get_cull_boxes(msplat_ctx* ctx, uint32_t* live, uint32_t* total, int* listed);

int msplat_get_stats(msplat_ctx* ctx, msplat_stats* out);      /* synchronises */
int msplat_get_timings(msplat_ctx* ctx, msplat_timings* out);  /* synchronises */

/* ---- parity-test taps (intermediate results of the last render; synchronise) ------------ */
/* per drawn-order splat r < V: 12 floats {px, py, A, B, C, log2(alpha), r, g, b, alpha, 0, 0}
 * where w(dx,dy) = exp2(A dx^2 + B dx dy + C dy^2 + log2 alpha); rect = packed tile rectangle
 * tx0 | ty0<<8 | tx1<<16 | ty1<<24 (0x0000FFFFu-style empty when tx0 > tx1) */
int msplat_debug_get_projected(msplat_ctx* ctx, float* rec12, uint32_t* rect, uint32_t cap);
/* tile_start has tiles_x*tiles_y+1 entries; pairs[k] & 0xFFFFFF = draw-order rank */
int msplat_debug_get_tile_lists(msplat_ctx* ctx, uint32_t* tile_start, uint32_t tile_cap,
                                uint32_t* pairs, uint64_t pair_cap);

/* on-device self-check of the ordering contracts (sorted keys ascending, ties by ascending storage slot; every bin list
 * ascending in draw-order rank): counts of violations, both 0 on a healthy context.  Guards the lane-ordered LDS-atomic
 * ranking, which msplat_create probes but the hardware does not document (MSPLAT_BALLOT_RANK=1 selects the ballot path) */
int msplat_debug_verify_order(msplat_ctx* ctx, uint32_t* key_violations, uint32_t* list_violations);
/* two-pass frames (msplat_config.two_pass): share > 0 pins the share of the visible splats that goes into the first pass (tests:
 * any value gives the same pixels), 0 hands it back to the feedback loop.  two_pass_frames: Renders of the context that ran in
 * two passes so far; share_now: the share the next one would use. */
int msplat_debug_two_pass(msplat_ctx* ctx, float share, uint64_t* two_pass_frames, float* share_now);
/* what the context's latest two-pass Render did (synchronises): out[0] = two-pass Renders so far (0: the rest is meaningless),
 * [1] = visible splats projected by pass 1, [2] = splats behind the cut that passed the gate and were projected by pass 2,
 * [3] = (splat, bin) pairs binned by pass 1, [4] = by pass 2, [5] = bins pass 1 left unfinished, [6] = bins, [7] = visible splats */
int msplat_get_two_pass_info(msplat_ctx* ctx, uint64_t out[8]);

/* compositor probe (performance analysis, bench statistics): per (bin, quadrant) work item 8 words
 * {shader clocks, records composited, batches staged, inner-loop clocks, pair words fetched, records fetched,
 *  bin-list length, ran}.  Off by default (a few clock reads per batch); MSPLAT_TILE_PROBE=1 in the environment
 * turns it on at msplat_create. */
int msplat_set_tile_probe(msplat_ctx* ctx, int enable);
/* (the getter carries the record size in its name: an out-of-tree caller built for the 4-word records of the first
 * release fails to link instead of overrunning its buffer) */
int msplat_debug_get_tile_probe8(msplat_ctx* ctx, uint32_t* dst8, uint32_t tile_cap);
int msplat_get_composite_work(msplat_ctx* ctx, msplat_composite_work* out);

/* ---- scene data: GaussianCloud / Ply surface (gaussiancloud.h:17-91, ply.h:19-46) -------- */
/* replaces GaussianCloud::GaussianCloud(Options{importFullSH}) */
msplat_cloud* msplat_cloud_create(int import_full_sh);
void msplat_cloud_destroy(msplat_cloud* c);
/* replaces GaussianCloud::ImportPly (gaussiancloud.cpp:138-365) */
int msplat_cloud_import_ply(msplat_cloud* c, const char* path);
/* same per-vertex math as ImportPly's lambda (gaussiancloud.cpp:254-361) applied to raw
 * attribute arrays instead of a file (synthetic scenes); f_rest may be NULL */
int msplat_cloud_from_attributes(msplat_cloud* c, uint64_t n, const float* xyz, const float* f_dc,
                                 const float* f_rest, const float* opacity, const float* log_scale,
                                 const float* rot);
/* GaussianCloud::ExportPly / InitDebugCloud / PruneSplats (gaussiancloud.cpp:367-626) */
int msplat_cloud_export_ply(msplat_cloud* c, const char* path);
int msplat_cloud_init_debug(msplat_cloud* c);
int msplat_cloud_prune(msplat_cloud* c, const float origin[3], uint32_t keep);
uint64_t msplat_cloud_num_gaussians(const msplat_cloud* c);   /* GetNumGaussians */
uint64_t msplat_cloud_stride(const msplat_cloud* c);          /* GetStride       */
uint64_t msplat_cloud_total_size(const msplat_cloud* c);      /* GetTotalSize    */
const void* msplat_cloud_raw_data(const msplat_cloud* c);     /* GetRawDataPtr   */
int msplat_cloud_has_full_sh(const msplat_cloud* c);          /* HasFullSH       */
int msplat_cloud_attr_offsets(const msplat_cloud* c, msplat_attr_offsets* out); /* Get*Attrib */
/* msplat_upload_cloud(ctx, raw, n, stride, offsets, has_full_sh) in one call */
int msplat_upload_gaussian_cloud(msplat_ctx* ctx, const msplat_cloud* c);

/* ---- scene config files + image output (host only) -------------------------- */
/* CamerasConfig::ImportJson (camerasconfig.cpp:20-67): cameras.json -> camera-to-world matrices (float[16]
 * each, column-major, -z forward / +y up) and the two fov angles; count_out = cameras in the file */
int msplat_cameras_import_json(const char* path, float* mats16_out, float* fovs2_out, uint32_t cap,
                               uint32_t* count_out);
/* CamerasConfig::EstimateFloorPlane (camerasconfig.cpp:69-95) */
int msplat_cameras_floor_plane(const char* path, float normal_out[3], float pos_out[3]);
/* VrConfig::ImportJson / ExportJson (vrconfig.cpp:20-65): <scene>_vr.json floor matrix */
int msplat_vrconfig_import_json(const char* path, float floor_mat_out[16]);
int msplat_vrconfig_export_json(const char* path, const float floor_mat[16]);
/* FindConfigFile (app.cpp:89-119): looks in the PLY's directory, its parent and grandparent */
int msplat_find_config_file(const char* ply_path, const char* config_name, char* out, uint32_t cap);
/* W x H float RGBA framebuffer (row 0 = bottom) -> 8-bit image file, top row first: clamp + round as an RGBA8
 * target does (the reference's back buffer), optional LinearToSRGB (util.cpp:357-367); ".ppm" or PNG */
int msplat_write_image(const char* path, const float* rgba, int width, int height, int encode_srgb);
/* 8-bit gray / gray+alpha / RGB / RGBA non-interlaced PNG (what Image::Load accepts, core/image.cpp:72-101) ->
 * RGBA8, top row first.  rgba8_out may be NULL to query the size; cap = bytes available */
int msplat_read_image(const char* path, uint8_t* rgba8_out, uint64_t cap, uint32_t* width_out, uint32_t* height_out);

/* ---- point-cloud renderer ------------------------------------------------
 * PointCloud (pointcloud.h:15-48) + PointRenderer (pointrenderer.h:23-57, pointrenderer.cpp:48-196): the SfM
 * points of <scene>/input.ply drawn as depth-sorted textured sprites.  A context holds EITHER a splat cloud or a
 * point cloud; with points, msplat_sort runs the same presort + radix sort (pointrenderer.cpp:113-166) and
 * msplat_render the sprite pipeline (point_vert/geom/frag.glsl + the blend state of app.cpp:153-156):
 * PointRenderer::Render == msplat_sort + msplat_render with the same matrices. */
typedef struct msplat_points msplat_points;
msplat_points* msplat_points_create(int use_linear_colors);            /* PointCloud::PointCloud */
void msplat_points_destroy(msplat_points* p);
int msplat_points_import_ply(msplat_points* p, const char* path);      /* PointCloud::ImportPly  */
int msplat_points_export_ply(const msplat_points* p, const char* path);/* PointCloud::ExportPly  */
void msplat_points_init_debug(msplat_points* p);                       /* PointCloud::InitDebugCloud */
uint64_t msplat_points_num(const msplat_points* p);
uint32_t msplat_points_stride(const msplat_points* p);                 /* 32: position.xyzw, color.rgba */
const void* msplat_points_data(const msplat_points* p);
/* replaces PointRenderer::Init's buffer setup (pointrenderer.cpp:95-110,198-225): n records of stride bytes,
 * float4 position and float4 colour at the given offsets */
int msplat_upload_points(msplat_ctx* ctx, const void* aos, uint64_t n, uint32_t stride_bytes,
                         uint32_t position_offset, uint32_t color_offset);
int msplat_upload_point_cloud(msplat_ctx* ctx, const msplat_points* p);
/* the sprite texture (texture/sphere.png in the reference, pointrenderer.cpp:54-64): RGBA8, top row first
 * as decoded from the file.  Applies Image::Load's row flip and 8-bit alpha pre-multiplication
 * (core/image.cpp:108-114,128-158), builds the mip chain (2x2 box filter) and samples it LinearMipmapLinear /
 * Linear / ClampToEdge; cfg.srgb decodes texels sRGB -> linear (GL_SRGB8_ALPHA8, core/texture.cpp:63-70).
 * rgba8 == NULL: a built-in procedural sphere sprite (the reference's asset is not shipped with this library). */
int msplat_set_point_sprite(msplat_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* ---- host matrix helpers used by the shims (glm closed forms; app.cpp:1042, util.cpp:420) - */
void msplat_mat4_inverse(const float m[16], float out[16]);
void msplat_mat4_mul(const float a[16], const float b[16], float out[16]);
void msplat_perspective(float fovy, float aspect, float zn, float zf, float out[16]);
void msplat_create_projection(float tanL, float tanR, float tanU, float tanD, float zn, float zf,
                              float out[16]);

#ifdef __cplusplus
}
#endif
#endif /* MSPLAT_H */
