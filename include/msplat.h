/*
 * msplat.h -- C ABI of libmsplat.so, the MI355X-native (gfx950 / CDNA4, HIP) replacement for the reference's
 * SplatRenderer::Sort()/Render() hot path.  This header is what an integrator binds; parity-test taps, probes and experiment
 * switches are in msplat_debug.h; the reasoning behind every default is in INTEGRATION.md (Appendix A) and DESIGN.md.
 *
 * The reference has no plugin/FFI layer: the seam is the C++ class SplatRenderer (/root/reference/src/splatrenderer.h:23-67)
 * fed by GaussianCloud (src/gaussiancloud.h:17-91).  Every entry point names the reference interface it replaces; the C++ shim
 * splatapult_amd/host/msplat_host.hpp re-creates the reference's class surface on top of this ABI.
 *
 * Conventions (the reference's): matrices are float[16], column-major like glm; cameraMat = camera-to-world; viewport =
 * (x, y, W, H); nearFar = (near, far).  Functions return 0 (MSPLAT_OK) or a negative MSPLAT_ERR_* and never throw.  A context
 * is not thread-safe (the reference's single GL thread).  No torch / framework / HIP types cross this boundary.
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
    MSPLAT_ERR_NO_DEVICE = -2,     /* no HIP device / runtime failure at create: there is no CPU fallback */
    MSPLAT_ERR_HIP = -3,           /* a HIP call failed; see msplat_last_error */
    MSPLAT_ERR_NO_CLOUD = -4,      /* sort / render before upload */
    MSPLAT_ERR_NO_SORT = -5,       /* render before sort */
    MSPLAT_ERR_UNSUPPORTED = -6,   /* more than 2^24 splats, viewport beyond 8192 x 8192 */
    MSPLAT_ERR_PAIR_OVERFLOW = -7, /* (splat, bin) pair buffer too small, see msplat_render */
    MSPLAT_ERR_IO = -8,            /* PLY / JSON / image open or parse failure */
    MSPLAT_ERR_PAIR_OVERFLOW_EARLIER = -9  /* a WARNING: this call did its work, but an earlier device-output render of the
                                      context had overflowed the pair buffer (that frame lacks splats; the buffer has grown) */
};

enum { MSPLAT_FB_RGBA32F = 0, MSPLAT_FB_RGBA16F = 1 };
enum { MSPLAT_ROP_NONE = 0, MSPLAT_ROP_RGBA8 = 1, MSPLAT_ROP_RGBA16F = 2 };          /* msplat_set_target_emulation */
enum { MSPLAT_TWO_PASS_AUTO = 0, MSPLAT_TWO_PASS_ON = 1, MSPLAT_TWO_PASS_OFF = 2 };  /* msplat_config.two_pass */
enum { MSPLAT_SPATIAL_AUTO = 0, MSPLAT_SPATIAL_ON = 1, MSPLAT_SPATIAL_OFF = 2 };     /* msplat_config.spatial_order */
enum { MSPLAT_FRAMES_AUTO = 0, MSPLAT_FRAMES_SERIAL = 1, MSPLAT_FRAMES_IN_FLIGHT = 2 }; /* msplat_config.frame_mode */
enum { MSPLAT_RANK_AUTO = 0, MSPLAT_RANK_BALLOT = 1 };                               /* msplat_config.rank_mode */
/* msplat_config.cu_partition: every CU, or the even / odd CU positions of every XCD (4 frames in flight alternate: +3-5 %) */
enum { MSPLAT_CU_ALL = 0, MSPLAT_CU_EVEN = 1, MSPLAT_CU_ODD = 2 };
enum { MSPLAT_BANDS_CONTIGUOUS = 0, MSPLAT_BANDS_INTERLEAVED = 1, MSPLAT_BANDS_BLOCK_INTERLEAVED = 2, MSPLAT_BANDS_ROOT_WEIGHTED = 3 };

typedef struct msplat_ctx msplat_ctx;
typedef struct msplat_cloud msplat_cloud;
typedef struct msplat_group msplat_group;
typedef struct msplat_points msplat_points;

/* Construction parameters: the implicit GL state the reference's renderer lives in -- device = the GL context's GPU
 * (sdl_main.cpp:98-100), fb_format = App's --fp16 / --fp32 FBO (app.cpp:1000-1035), srgb = SplatRenderer::Init's
 * isFramebufferSRGBEnabled (splatrenderer.cpp:60-72).  Fields were appended over time: a struct_size that ends before a field
 * selects its AUTO value.  Pixels, keys and lists do not depend on any field below `stream`. */
typedef struct msplat_config {
    uint32_t struct_size;      /* sizeof(msplat_config) */
    int32_t device;            /* HIP device ordinal */
    int32_t fb_format;         /* MSPLAT_FB_* */
    int32_t srgb;              /* FRAMEBUFFER_SRGB path of splat_vert.glsl:129-151,209-218 */
    float t_epsilon;           /* front-to-back early-out below this transmittance; 0 = never, negative = default 2^-14 */
    uint64_t pair_capacity;    /* max (splat, bin) pairs per render; 0 = automatic (grows) */
    void* stream;              /* hipStream_t to launch on; NULL = a stream of the library's own */
    int32_t enable_timing;     /* n > 0: per-stage events on every n-th sort / render (msplat_get_timings) */
    int32_t compositor_waves;  /* persistent compositor waves per render; 0 = default */
    int32_t rank_mode;         /* MSPLAT_RANK_*: lane-ordered LDS atomics (probed at create) or ballots for the stable ranking */
    int32_t frame_mode;        /* MSPLAT_FRAMES_*: the only context working on the GPU, or one of several frames in flight */
    int32_t spatial_order;     /* MSPLAT_SPATIAL_*: may the cloud be STORED in Morton order (chunk-level cull; ties of the sort
                                  are then in storage order, msplat_get_storage_order) */
    int32_t async_submit;      /* != 0: msplat_sort / device-output msplat_render return at once, a worker thread of the context
                                  issues their launches; a queued call's failure or overflow warning is returned by the next
                                  msplat_synchronize / msplat_stream_wait */
    int32_t two_pass;          /* MSPLAT_TWO_PASS_*: may a Render run as two passes with occlusion feedback (same pixels) */
    int32_t cu_partition;      /* MSPLAT_CU_*: the CUs a stream the library creates itself (stream == NULL) may use; frames in flight */
} msplat_config;

/* Byte offsets of the attributes inside one AoS record: the BinaryAttribute offsets SplatRenderer::BuildVertexArrayObject binds
 * (splatrenderer.cpp:345-391; gaussiancloud.cpp:633-657).  r_sh1 .. b_sh3 are ignored unless full_sh. */
typedef struct msplat_attr_offsets {
    uint32_t pos_with_alpha;
    uint32_t r_sh0, g_sh0, b_sh0;
    uint32_t cov3_col0, cov3_col1, cov3_col2;
    uint32_t r_sh1, r_sh2, r_sh3;
    uint32_t g_sh1, g_sh2, g_sh3;
    uint32_t b_sh1, b_sh2, b_sh3;
} msplat_attr_offsets;

typedef struct msplat_stats {
    uint64_t num_splats;       /* N */
    uint32_t sort_count;       /* V: splats that survived the presort cull (sortCount) */
    uint32_t drawn;            /* splats that passed the geometry-stage guard band */
    uint64_t pairs;            /* (splat, bin) pairs binned (bins of msplat_tile_size() pixels) */
    uint32_t tiles_x, tiles_y;
    uint32_t width, height;
    uint64_t pair_capacity;
    uint64_t device_bytes;     /* device memory held by the context */
    uint64_t pairs_tile16;     /* (splat, 16x16 tile) pairs covered by the footprints: SURVEY.md 8d's D */
} msplat_stats;

typedef struct msplat_timings {   /* milliseconds, averaged; names follow the reference's Tracy zones (splatrenderer.cpp:156-318) */
    float sort_total;          /* "SplatRenderer::Sort" (cull + key + sort) */
    float render_total;        /* "SplatRenderer::Render" */
    float project;             /* vertex + geometry stage */
    float binning;             /* bin lists */
    float composite;           /* fragment + blend */
    float reserved[3];         /* [0] frames averaged, [1] compositor KERNEL time (dispatch begin / end events), [2] its launches */
} msplat_timings;

/* ---- context: SplatRenderer::SplatRenderer / ~SplatRenderer (splatrenderer.cpp:41-48) ---- */
int msplat_create(msplat_ctx** out, const msplat_config* cfg);
void msplat_destroy(msplat_ctx* ctx);
const char* msplat_last_error(const msplat_ctx* ctx);   /* ctx may be NULL: global last error */
const char* msplat_version_string(void);
int msplat_tile_size(void);                             /* edge of the square screen bins (pixels) band rows refer to */

/* ---- upload: SplatRenderer::Init + BuildVertexArrayObject (splatrenderer.cpp:50-151,345-391).  Copies the interleaved
 * cloud (host memory, n records of stride_bytes) to the device; the caller may free it afterwards. */
int msplat_upload_cloud(msplat_ctx* ctx, const void* aos, uint64_t n, uint32_t stride_bytes,
                        const msplat_attr_offsets* off, int full_sh);
/* GPU ingest: GaussianCloud::ImportPly's per-vertex math (gaussiancloud.cpp:254-361) as a HIP kernel over the
 * raw PLY vertex block.  Byte offsets of the float properties inside one vertex; -1 = absent (reads as 0, like
 * BinaryAttribute::Read); without all of f_rest, or with full_sh == 0, the cloud is SH degree 0 (gaussiancloud.cpp:188-205). */
typedef struct msplat_ply_layout {
    uint32_t vertex_size;
    int32_t x, y, z;
    int32_t f_dc[3];
    int32_t f_rest[45];
    int32_t opacity;
    int32_t scale[3];
    int32_t rot[4];
} msplat_ply_layout;
int msplat_upload_ply_vertices(msplat_ctx* ctx, const void* vertices, uint64_t n, const msplat_ply_layout* layout, int full_sh);
/* Ply::Parse (ply.cpp:72-87) on the host + the ingest kernel: GaussianCloud::ImportPly + SplatRenderer::Init in one call */
int msplat_upload_ply(msplat_ctx* ctx, const char* path, int import_full_sh);
/* the device cloud in the reference's interleaved layout (100 B / 244 B records), upload numbering */
int msplat_download_cloud(msplat_ctx* ctx, void* aos_out, uint64_t cap_bytes);

/* ---- SplatRenderer::Sort (splatrenderer.cpp:153-312): cull + depth key (presort_compute.glsl:31-57), stable ascending 32-bit
 * radix sort; the sorted index list stays context state for the following renders.  Asynchronous: the reference's 4-byte
 * readback stall (splatrenderer.cpp:195-204) is not reproduced. */
int msplat_sort(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16], const float viewport[4], const float nearFar[2]);

/* ---- SplatRenderer::Render (splatrenderer.cpp:315-343) plus the GL pipeline behind its glDrawElements: splat_vert / _geom /
 * _frag.glsl and the blend / clear state of app.cpp:144-164.  Writes W x H RGBA (float or half), row 0 = GL bottom row,
 * alpha = 1.  out_is_device != 0: `rgba` is device memory, the call is asynchronous on the stream; else host memory, the call
 * returns after the copy.  pitch_bytes = bytes between rows (0 = tight).
 * Pair-buffer overflow: a host-output render grows the buffer and retries.  A device-output render cannot know; the NEXT
 * msplat_sort / msplat_render / msplat_synchronize of the context grows the buffer (unless pair_capacity fixed it) and reports
 * it once -- msplat_synchronize: MSPLAT_ERR_PAIR_OVERFLOW; sort / render, whose own work is done: MSPLAT_ERR_PAIR_OVERFLOW_EARLIER.
 * The frame that overflowed lacks splats in its last bin columns and should be rendered again. */
int msplat_render(msplat_ctx* ctx, const float cameraMat[16], const float projMat[16], const float viewport[4],
                  const float nearFar[2], void* rgba, uint64_t pitch_bytes, int out_is_device);
/* the reference's VR frame -- Sort with the first eye, Render per eye (app.cpp:603-607) -- as ONE chain of launches; the same
 * pixels as two msplat_render calls, bit for bit.  Host targets, banded contexts, points and the emulations go view by view. */
int msplat_render_stereo(msplat_ctx* ctx, const float cameraMat0[16], const float projMat0[16], const float cameraMat1[16],
                         const float projMat1[16], const float viewport[4], const float nearFar[2], void* rgba0, void* rgba1,
                         uint64_t pitch_bytes, int out_is_device);
/* blocks until everything queued on the context (and issued by its worker thread) has finished */
int msplat_synchronize(msplat_ctx* ctx);

/* ---- frames in flight.  The reference queues successive frames on one GL command stream and the driver overlaps them; here
 * the overlap is explicit: one context per frame in flight (own stream and per-frame buffers), the cloud uploaded into the
 * first and attached to the others, frame k issued on context k % depth.  Bit-identical to a single context.  The shims do
 * the rotation (SetFramesInFlight).  Start the process with GPU_MAX_HW_QUEUES=8. */
int msplat_attach_cloud(msplat_ctx* ctx, msplat_ctx* owner);        /* `ctx` renders `owner`'s cloud (same device), no copy */
int msplat_stream_wait(msplat_ctx* ctx, void* stream);              /* `stream` (hipStream_t) waits for the context's work so far */
int msplat_wait_event(msplat_ctx* ctx, void* event);                /* the context's stream waits for `event` (hipEvent_t) */
void* msplat_get_stream(msplat_ctx* ctx);                           /* the hipStream_t the context launches on */
int msplat_get_fb_format(const msplat_ctx* ctx);                    /* MSPLAT_FB_* the context was created with (-1: NULL) */

/* ---- rows of the screen on several GPUs (no reference counterpart).  The context owns blocks of `block`
 * consecutive bin rows (msplat_tile_size() pixels; row 0 = GL bottom) starting at first_row, first_row + stride, ..., at most
 * row_count rows (0 = all).  The framebuffer handed to msplat_render is always the full image; only owned rows are written;
 * pixels are bit-identical to the unbanded frame.  msplat_set_band(mod, rem) = interleaved single rows. */
int msplat_set_band(msplat_ctx* ctx, int32_t row_mod, int32_t row_rem);
int msplat_set_band_layout(msplat_ctx* ctx, int32_t first_row, int32_t row_count, int32_t block, int32_t stride);
/* the standard layouts (MSPLAT_BANDS_*) for rank `rank` of `world` over rows_full bin rows; host arithmetic only */
int msplat_band_plan(int32_t kind, int32_t rows_full, int32_t world, int32_t rank, int32_t block_rows, int32_t* first_row,
                     int32_t* row_count, int32_t* block, int32_t* stride);
/* MSPLAT_BANDS_ROOT_WEIGHTED: contiguous bands, rank 0 (the gather's root: sends nothing) weighted `block_rows` PERCENT of another rank;
 * msplat_band_root_weight picks the percentage from a linear cost model; _plan_weighted: rows ~ weights[], bounds[world + 1] */
int msplat_band_plan_weighted(int32_t rows_full, int32_t world, const float* weights, int32_t* bounds_out);
int msplat_band_root_weight(int32_t rows_full, int32_t world, double fixed_ms, double ms_per_row, double row_bytes, double link_gbps, int overlap);
/* msplat_sort also drops splats whose footprint cannot reach an owned row (mono rendering; msplat_sort_count describes the band only) */
int msplat_set_band_cull(msplat_ctx* ctx, int enable);
/* The exchange, one process per GPU (the north star's "RCCL over xGMI only for the final row gather"): rank `root` posts one
 * receive per run of foreign rows straight into its framebuffer, the owners send their runs from where the compositor left
 * them, all in ONE ncclGroupStart/End, on the context's stream.  `comm` = the caller's ncclComm_t (librccl is loaded at the
 * first call: no link-time dependency), `kind` / `block_rows` = the layout every rank set with msplat_band_plan.  `rgba` =
 * device memory of `height` rows of pitch_bytes, pixels of the context's fb_format (tight rows: one message per run; else row by
 * row in the same group, so a window of a wider surface keeps its neighbours).  world == 1: nothing to do.
 * flags: MSPLAT_EXCHANGE_WIRE_FP16 (RGBA32F targets only): rows cross the link as RGBA16F -- half the bytes; the gathered rows
 * then differ from the owners' by one fp16 rounding, |d| <= 2^-11 |value| (values beyond 65504 become inf), root's own rows not. */
enum { MSPLAT_EXCHANGE_WIRE_FP16 = 1 };
int msplat_band_exchange(msplat_ctx* ctx, void* comm, int32_t rank, int32_t world, int32_t root, int32_t kind, int32_t block_rows,
                         void* rgba, uint64_t pitch_bytes, int32_t width, int32_t height, int32_t flags);

/* ---- several GPUs, ONE process: the reference's shape, a single-threaded host calling Sort / Render (app.cpp:1067-1068).
 * One context per listed device, replicated cloud, bin rows partitioned (default MSPLAT_BANDS_CONTIGUOUS).  The only exchange
 * is the row gather into `rgba` on devices[0]: peer stores from the other devices' compositors (default; no staging), RCCL
 * send / recv (msplat_group_set_exchange), or a 2-D copy where no peer mapping exists.  Context 0's stream waits for the
 * others: synchronising it, or msplat_group_synchronize, means the frame is complete.  cfg as msplat_create (device ignored). */
enum { MSPLAT_EXCHANGE_PEER_STORE = 0, MSPLAT_EXCHANGE_RCCL = 1, MSPLAT_EXCHANGE_COPY = 2 };
int msplat_group_create(msplat_group** out, const int32_t* devices, uint32_t n, const msplat_config* cfg);
void msplat_group_destroy(msplat_group* g);
const char* msplat_group_last_error(const msplat_group* g);     /* g may be NULL */
uint32_t msplat_group_size(const msplat_group* g);
msplat_ctx* msplat_group_context(msplat_group* g, uint32_t i);  /* borrowed: stats, timings of rank i */
int msplat_group_peer_store(const msplat_group* g, uint32_t i); /* 1: rank i writes device 0's framebuffer directly */
int msplat_group_set_exchange(msplat_group* g, int32_t exchange);  /* MSPLAT_EXCHANGE_*; also MSPLAT_GROUP_EXCHANGE=rccl|copy|peer */
int msplat_group_get_exchange(const msplat_group* g);              /* the exchange the latest msplat_group_render used */
int msplat_group_upload_cloud(msplat_group* g, const void* aos, uint64_t n, uint32_t stride_bytes,
                              const msplat_attr_offsets* off, int full_sh);
int msplat_group_upload_gaussian_cloud(msplat_group* g, const msplat_cloud* c);
int msplat_group_upload_ply(msplat_group* g, const char* path, int import_full_sh);
int msplat_group_set_layout(msplat_group* g, int32_t kind, int32_t block_rows);
int msplat_group_set_band_cull(msplat_group* g, int enable);
int msplat_group_sort(msplat_group* g, const float cameraMat[16], const float projMat[16], const float viewport[4],
                      const float nearFar[2]);
int msplat_group_render(msplat_group* g, const float cameraMat[16], const float projMat[16], const float viewport[4],
                        const float nearFar[2], void* rgba, uint64_t pitch_bytes, int out_is_device);
int msplat_group_synchronize(msplat_group* g);

/* ---- results of the latest Sort / Render (these synchronise) ---- */
int msplat_sort_count(msplat_ctx* ctx, uint32_t* v);            /* sortCount (splatrenderer.cpp:198-199) */
/* the element buffer of splatrenderer.cpp:296-311: upload indices of the visible splats in draw order (ascending key = far to
 * near; equal keys in ascending storage slot) */
int msplat_get_sorted_indices(msplat_ctx* ctx, uint32_t* dst, uint32_t cap);
int msplat_get_sorted_keys(msplat_ctx* ctx, uint32_t* dst, uint32_t cap);
/* dst[slot] = upload index (identity unless the cloud was reordered; *reordered says which); dst may be NULL */
int msplat_get_storage_order(msplat_ctx* ctx, uint32_t* dst, uint64_t cap, int* reordered);
int msplat_get_stats(msplat_ctx* ctx, msplat_stats* out);
int msplat_get_timings(msplat_ctx* ctx, msplat_timings* out);

/* ---- what the GL app's render target does (for callers who diff against its pixels; draw-order walk, several times slower).
 * Depth test: GL_DEPTH_TEST is on (app.cpp:163) and live wherever the target has a depth attachment (default
 * back buffer, 24 bits, sdl_main.cpp:79; XR swapchains); bits = 0 (default) models the colour-only --fp16 / --fp32 FBO.
 * Target rounding (SURVEY.md 8a-12, app.cpp:1012-1020): the RGBA8 back buffer clamps and stores 8-bit unorm after EVERY blend,
 * the --fp16 target rounds to fp16 after every blend; MSPLAT_ROP_NONE (default) accumulates in fp32 and rounds once. */
int msplat_set_depth_test(msplat_ctx* ctx, int depth_bits);
int msplat_set_target_emulation(msplat_ctx* ctx, int rop);

/* ---- scene data: GaussianCloud / Ply (gaussiancloud.h:17-91, ply.h:19-46) ---- */
msplat_cloud* msplat_cloud_create(int import_full_sh);          /* GaussianCloud::GaussianCloud(Options{importFullSH}) */
void msplat_cloud_destroy(msplat_cloud* c);
int msplat_cloud_import_ply(msplat_cloud* c, const char* path); /* GaussianCloud::ImportPly (gaussiancloud.cpp:138-365) */
/* ImportPly's per-vertex math (gaussiancloud.cpp:254-361) on raw attribute arrays (synthetic scenes); f_rest may be NULL */
int msplat_cloud_from_attributes(msplat_cloud* c, uint64_t n, const float* xyz, const float* f_dc, const float* f_rest,
                                 const float* opacity, const float* log_scale, const float* rot);
int msplat_cloud_export_ply(msplat_cloud* c, const char* path); /* ExportPly / InitDebugCloud / PruneSplats */
int msplat_cloud_init_debug(msplat_cloud* c);                   /*   (gaussiancloud.cpp:367-626) */
int msplat_cloud_prune(msplat_cloud* c, const float origin[3], uint32_t keep);
uint64_t msplat_cloud_num_gaussians(const msplat_cloud* c);     /* GetNumGaussians */
uint64_t msplat_cloud_stride(const msplat_cloud* c);            /* GetStride */
uint64_t msplat_cloud_total_size(const msplat_cloud* c);        /* GetTotalSize */
const void* msplat_cloud_raw_data(const msplat_cloud* c);       /* GetRawDataPtr */
int msplat_cloud_has_full_sh(const msplat_cloud* c);            /* HasFullSH */
int msplat_cloud_attr_offsets(const msplat_cloud* c, msplat_attr_offsets* out);   /* Get*Attrib */
int msplat_upload_gaussian_cloud(msplat_ctx* ctx, const msplat_cloud* c);

/* ---- scene config files + image output (host only) ---- */
/* CamerasConfig::ImportJson (camerasconfig.cpp:20-67): camera-to-world matrices (float[16] each) and the two fov angles */
int msplat_cameras_import_json(const char* path, float* mats16_out, float* fovs2_out, uint32_t cap, uint32_t* count_out);
int msplat_cameras_floor_plane(const char* path, float normal_out[3], float pos_out[3]);   /* EstimateFloorPlane (:69-95) */
int msplat_vrconfig_import_json(const char* path, float floor_mat_out[16]);                /* VrConfig (vrconfig.cpp:20-65) */
int msplat_vrconfig_export_json(const char* path, const float floor_mat[16]);
int msplat_find_config_file(const char* ply_path, const char* config_name, char* out, uint32_t cap);   /* app.cpp:89-119 */
/* W x H float RGBA (row 0 = bottom) -> 8-bit ".ppm" / PNG, top row first: clamp + round like an RGBA8 target, optional
 * LinearToSRGB (util.cpp:357-367) */
int msplat_write_image(const char* path, const float* rgba, int width, int height, int encode_srgb);
/* 8-bit non-interlaced PNG (what Image::Load accepts, core/image.cpp:72-101) -> RGBA8, top row first; NULL queries the size */
int msplat_read_image(const char* path, uint8_t* rgba8_out, uint64_t cap, uint32_t* width_out, uint32_t* height_out);

/* ---- point-cloud renderer: PointCloud (pointcloud.h:15-48) + PointRenderer (pointrenderer.h:23-57,
 * pointrenderer.cpp:48-196).  A context holds EITHER a splat cloud or a point cloud; with points, msplat_sort is the same
 * presort + radix sort and msplat_render the sprite pipeline (point_*.glsl + the blend state of app.cpp:153-156):
 * PointRenderer::Render == msplat_sort + msplat_render with the same matrices. */
msplat_points* msplat_points_create(int use_linear_colors);
void msplat_points_destroy(msplat_points* p);
int msplat_points_import_ply(msplat_points* p, const char* path);
int msplat_points_export_ply(const msplat_points* p, const char* path);
void msplat_points_init_debug(msplat_points* p);
uint64_t msplat_points_num(const msplat_points* p);
uint32_t msplat_points_stride(const msplat_points* p);          /* 32: position.xyzw, color.rgba */
const void* msplat_points_data(const msplat_points* p);
int msplat_upload_points(msplat_ctx* ctx, const void* aos, uint64_t n, uint32_t stride_bytes, uint32_t position_offset,
                         uint32_t color_offset);                /* PointRenderer::Init's buffers (pointrenderer.cpp:95-110) */
int msplat_upload_point_cloud(msplat_ctx* ctx, const msplat_points* p);
/* the sprite (texture/sphere.png, pointrenderer.cpp:54-64): RGBA8, top row first (Image::Load's flip + premultiplication, mip chain); NULL = built-in sphere */
int msplat_set_point_sprite(msplat_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* ---- host matrix helpers used by the shims (glm closed forms; app.cpp:1042, util.cpp:420) ---- */
void msplat_mat4_inverse(const float m[16], float out[16]);
void msplat_mat4_mul(const float a[16], const float b[16], float out[16]);
void msplat_perspective(float fovy, float aspect, float zn, float zf, float out[16]);
void msplat_create_projection(float tanL, float tanR, float tanU, float tanD, float zn, float zf, float out[16]);

#ifdef __cplusplus
}
#endif
#endif /* MSPLAT_H */
