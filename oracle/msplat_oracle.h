/*
 * msplat_oracle.h -- CPU ORACLE for the splatapult Sort()/Render() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libmsplat.so) never links,
 * loads or calls anything in oracle/.
 *
 * PARITY PINNED (round 4): the reference's arithmetic for this path lives in GLSL 4.60 shaders behind an OpenGL driver and the
 * reference ships no tests / golden vectors for them.  This file is a literal C restatement of those
 * shaders.  It is pinned by running the shaders themselves: oracle/glref (a window-system-free loader for Mesa's llvmpipe)
 * compiles /root/reference/shader/presort_compute.glsl and splat_{vert,geom,frag}.glsl where they lie and executes them with
 * the reference's GL state; tests/test_reference_shaders.py compares -- visible set and keys exact, framebuffers within SURVEY
 * 8c's tolerance -- and tests/golden/glref_*.npz carry those outputs to machines without the reference.  Also pinned: the PLY
 * parser (oracle/_ref/libref_ply.so, built from /root/reference/src/ply.cpp by oracle/Makefile).  Cross-checks that predate
 * the shader runs stay: an independent numpy restatement (oracle/np_oracle.py), SURVEY 8c's hand-computed values.
 *
 * Conventions: all matrices are float[16], column-major like glm (m[col*4+row]);
 * cameraMat = camera-to-world; viewport = (x, y, W, H); nearFar = (near, far).
 * Everything is fp32 and compiled with -ffp-contract=off so that the operation order
 * written here is the operation order executed.
 */
#ifndef MSPLAT_ORACLE_H
#define MSPLAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* float offsets inside one AoS record (reference: src/gaussiancloud.cpp:32-56) */
enum {
    ORC_OFF_POS = 0,    /* x y z alpha */
    ORC_OFF_R_SH0 = 4,
    ORC_OFF_G_SH0 = 8,
    ORC_OFF_B_SH0 = 12,
    ORC_OFF_COV0 = 16,  /* 3 floats per column */
    ORC_OFF_COV1 = 19,
    ORC_OFF_COV2 = 22,
    ORC_BASE_FLOATS = 25,   /* 100 B */
    ORC_OFF_R_SH1 = 25, ORC_OFF_R_SH2 = 29, ORC_OFF_R_SH3 = 33,
    ORC_OFF_G_SH1 = 37, ORC_OFF_G_SH2 = 41, ORC_OFF_G_SH3 = 45,
    ORC_OFF_B_SH1 = 49, ORC_OFF_B_SH2 = 53, ORC_OFF_B_SH3 = 57,
    ORC_FULL_FLOATS = 61    /* 244 B */
};

/* One projected splat, as the vertex+geometry stages hand it to the fragment stage. */
typedef struct orc_splat2d {
    float px, py;        /* screen-space centre, GL window coords (origin bottom-left)   */
    float cov[4];        /* cov2D as the vec4 (m00, m01, m10, m11), +0.3 on the diagonal  */
    float inv[4];        /* cov2Dinv as the vec4 (i00, i01, i10, i11)                     */
    float rgb[3];        /* 0.5 + SH, unclamped (sRGB->linear applied if requested)       */
    float alpha;
    float ndc[3];        /* clip.xyz / clip.w                                             */
    float depth;         /* clip.w                                                        */
    float hx, hy;        /* half extents of the AABB of the 3.5-sigma quad (pixels)       */
    int32_t reject;      /* 1 if the geometry stage drops it / GL clips it away           */
    uint32_t index;      /* original splat index                                          */
} orc_splat2d;

/* ---- matrix helpers (restating glm's published closed forms; glm is unpinned) -------- */
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);      /* out = a*b   */
void orc_mat4_inverse(const float m[16], float out[16]);                      /* cofactors   */
void orc_perspective(float fovy, float aspect, float zn, float zf, float out[16]);
void orc_create_projection(float tanL, float tanR, float tanU, float tanD,
                           float zn, float zf, float out[16]);

/* ---- load time: PLY vertex attributes -> AoS record (gaussiancloud.cpp:254-361) ------ */
void orc_build_cloud(size_t n, const float* xyz, const float* f_dc, const float* f_rest,
                     const float* opacity, const float* log_scale, const float* rot,
                     int full_sh, float* aos_out);

/* ---- per frame ------------------------------------------------------------------------ */
/* presort_compute.glsl:31-57. Compacts in ascending index order. Returns V. */
uint32_t orc_presort(size_t n, const float* aos, size_t stride_floats, const float mvp[16],
                     float zfar, uint32_t* keys_out, uint32_t* idx_out);
/* single-key variant used by unit tests; returns 0 if culled */
int orc_cull_key(const float xyz[3], const float mvp[16], float zfar, uint32_t* key_out);

/* splatrenderer.cpp:223-264: ascending key, stable (ties keep input order). */
void orc_sort(uint32_t v, uint32_t* keys, uint32_t* idx);

/* splat_vert.glsl:153-222 + splat_geom.glsl:34-87 for idx[0..v). */
void orc_project(uint32_t v, const uint32_t* idx, const float* aos, size_t stride_floats,
                 int full_sh, int srgb, const float viewMat[16], const float projMat[16],
                 const float viewport[4], const float nearFar[2], const float eye[3],
                 orc_splat2d* out);

/* splat_frag.glsl:18-42 + blend state app.cpp:153-160. Back-to-front in array order.
 * rgba is W*H*4 floats, row 0 = GL bottom row. nthreads>1 splits pixel rows across OpenMP
 * threads (each pixel still sees the identical blend sequence). Rows [row0,row1) only. */
void orc_composite(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba,
                   int row0, int row1, int nthreads);
/* the same, plus per pixel the most that discard-threshold flips (|w - 1/256| <= flip_rel/256) could move it */
void orc_composite_flip(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba,
                        int row0, int row1, int nthreads, float* flip_budget, float flip_rel);
/* orc_composite plus the GL_LESS depth test the reference leaves enabled (app.cpp:160,163), against an
 * emulated depth buffer of depth_bits = 24 (unorm, the default back buffer) or 32 (float).  SURVEY.md 8f-4. */
uint32_t orc_quantise_depth(float ndcz, int depth_bits);
void orc_composite_depth(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba, int depth_bits,
                         int nthreads);
/* the blend as the render target performs it: rop 0 = float accumulation, 1 = RGBA8 (clamp + 8-bit unorm after every
 * blend, GL 4.6 17.3.6; the default back buffer), 2 = RGBA16F (fp16 rounding after every blend; --fp16);
 * depth_bits 0 = no depth test.  src/app.cpp:1012-1020, SURVEY.md 8a-12. */
void orc_composite_rop(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba, int depth_bits, int rop,
                       int nthreads);
/* same, double-precision accumulation; used only to calibrate tolerances */
void orc_composite_f64(uint32_t v, const orc_splat2d* s, int W, int H, double* rgba,
                       int nthreads);

/* ---- point-cloud renderer: pointrenderer.cpp:113-196 + shader/point_*.glsl ---------- */
typedef struct orc_point2d {
    float cx, cy;        /* quad centre, pixels (origin bottom-left)                        */
    float hx, hy;        /* quad half size, pixels                                           */
    float rgba[4];       /* vertex colour                                                    */
    float lambda;        /* texture level of detail                                          */
    float ndcz;
    int32_t reject;      /* clipped away by the near / far plane, or degenerate              */
    uint32_t index;
} orc_point2d;
int orc_build_sprite(const uint8_t* rgba8_top_first, int w, int h, int srgb, float* chain, uint32_t* off);
void orc_points_project(uint32_t v, const uint32_t* idx, const float* points, const float viewMat[16],
                        const float projMat[16], const float viewport[4], int tex_w, int tex_h, orc_point2d* out);
void orc_points_composite(uint32_t v, const orc_point2d* pts, const float* chain, const uint32_t* off, int tex_w,
                          int tex_h, int levels, int W, int H, float* rgba, int depth_bits);

/* Whole frame: Sort(cameraMat, projMat, ...) then Render(cameraMat2, projMat2, ...).
 * Pass the same matrices twice for the desktop path. Optional outputs may be NULL.
 * Returns V. */
uint32_t orc_render_frame(size_t n, const float* aos, size_t stride_floats, int full_sh,
                          int srgb,
                          const float sortCameraMat[16], const float sortProjMat[16],
                          const float renderCameraMat[16], const float renderProjMat[16],
                          const float viewport[4], const float nearFar[2],
                          float* rgba, uint32_t* sorted_idx_out, uint32_t* sorted_keys_out,
                          orc_splat2d* splats_out, int nthreads);

/* The TIMED CPU baseline (msplat_cpu_tiled.c): the same frame with parallel cull / stable radix
 * sort / projection, 16x16-tile binning and a front-to-back compositor that stops a pixel at T < t_eps, on
 * `nthreads` host threads.  stage_ms (may be NULL) receives {cull, sort, project, bin, composite, total} in ms.
 * Rows [row0, row1) are produced.  Returns V (0xFFFFFFFF: out of memory). */
uint32_t orc_render_frame_tiled(size_t n, const float* aos, size_t stride_floats, int full_sh, int srgb,
                                const float sortCameraMat[16], const float sortProjMat[16],
                                const float renderCameraMat[16], const float renderProjMat[16],
                                const float viewport[4], const float nearFar[2], float* rgba,
                                uint32_t* sorted_idx_out, uint32_t* sorted_keys_out, float t_eps, int nthreads,
                                int row0, int row1, double* stage_ms);

/* pixel-splat evaluations performed by the last orc_composite call (for baselines) */
uint64_t orc_last_fragment_count(void);

#ifdef __cplusplus
}
#endif
#endif
