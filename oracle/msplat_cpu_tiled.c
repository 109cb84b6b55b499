/*
 * msplat_cpu_tiled.c -- the TIMED CPU baseline of SURVEY.md 8d(ii): the same Sort()/Render() arithmetic as the
 * literal oracle (msplat_oracle.c), organised the way a CPU renderer would be: parallel cull + ordered compaction,
 * parallel stable LSD radix sort, parallel projection, 16x16-tile binning, and a front-to-back compositor with early
 * termination, all on the host's threads (OpenMP worker pool).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as msplat_oracle.h): only tests/, bench.py's cpu_baseline
 * leg and __graft_entry__ may load it.  The reference has no CPU implementation of this path (src/sdl_main.cpp:28 is a
 * commented-out define); this file is the "reference's CPU-side sort+raster path timed on the host cores" that
 * BASELINE.json's north_star asks to report next to the GPU number.
 *
 * Arithmetic: cull + key = orc_cull_key (presort_compute.glsl:38-55); order = ascending key, stable
 * (splatrenderer.cpp:223-264); projection = orc_project (splat_vert.glsl:153-222, splat_geom.glsl:22-54); fragments =
 * splat_frag.glsl:20-41 with the same per-fragment expression as orc_composite.  The blend is the algebraically
 * identical front-to-back form  C = sum_i T_i w_i c_i,  T <- T (1 - w)  (SURVEY.md 8a-12), a pixel stops when
 * T < t_eps.  tests/test_oracle.py checks it against the literal oracle on the committed golden scenes.
 */
#include "msplat_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define TILE 16

typedef struct {
    size_t n_cap, pair_cap, cnt_cap, tile_cap;
    uint32_t *keyA, *keyB, *idxA, *idxB;     /* n_cap each */
    orc_splat2d* splats;                      /* n_cap */
    uint32_t* rect;                           /* 4 * n_cap: tile rectangle tx0, ty0, tx1, ty1 (tx0 > tx1: none) */
    uint32_t* pairs;                          /* pair_cap: draw-order ranks, grouped by tile */
    uint32_t* cnt;                            /* blocks x ntiles */
    uint32_t* tile_start;                     /* ntiles + 1 */
} tiled_ws;

static tiled_ws g_ws;

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

static int grow(void** p, size_t* cap, size_t need, size_t elem)
{
    if (*cap >= need && *p) return 0;
    free(*p);
    *p = malloc((need ? need : 1) * elem);
    *cap = *p ? need : 0;
    return *p ? 0 : -1;
}

static inline void pixel_range(float centre, float half, int lo_clip, int hi_clip, int* lo, int* hi)
{
    /* identical to msplat_oracle.c: pixels whose centre lies within the quad's AABB, padded by one pixel */
    float a = floorf(centre - half - 0.5f) - 1.0f;
    float b = ceilf(centre + half - 0.5f) + 1.0f;
    if (a < (float)lo_clip) a = (float)lo_clip;
    if (b > (float)(hi_clip - 1)) b = (float)(hi_clip - 1);
    if (!(a <= b)) { *lo = 0; *hi = -1; return; }
    *lo = (int)a;
    *hi = (int)b;
}

/* exp(x) for x <= 0, branch-free so that the compositor's row loop vectorises: Cody-Waite range reduction and the degree-5
 * polynomial of the classic single-precision expf (relative error ~1e-7: the baseline need not be bit-identical to the
 * literal oracle, tests/test_oracle.py holds it to the framebuffer tolerance) */
static inline float exp_neg(float x)
{
    x = x < -87.0f ? -87.0f : x;
    const float t = x * 1.44269504088896341f;
    const int n = (int)(t - 0.5f);                           /* nearest integer for t <= 0 */
    const float fn = (float)n;
    float r = x - fn * 0.693359375f;
    r = r - fn * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * r * r + r + 1.0f;
    union { int32_t i; float f; } s;
    s.i = (n + 127) << 23;
    return p * s.f;
}

/* one pixel row segment of one splat inside a tile: n <= 16 pixels starting at tile-local index p0, first pixel centre
 * offset (dx0, dy) from the splat centre.  splat_frag.glsl:20-41 + the front-to-back blend; discarded fragments and
 * saturated pixels get weight 0 (no branches). */
__attribute__((target_clones("avx2", "default")))
static void blend_row(float* __restrict__ T, float* __restrict__ cr, float* __restrict__ cg, float* __restrict__ cb, int n,
                      float dx0, float dy, float i0, float i1, float i2, float i3, float alpha, float r, float g, float b, float t_eps)
{
#pragma omp simd
    for (int k = 0; k < n; ++k) {
        const float dx = dx0 + (float)k;
        const float mx = i0 * dx + i2 * dy;
        const float my = i1 * dx + i3 * dy;
        const float q = dx * mx + dy * my;
        const float sa = alpha * exp_neg(-0.5f * q);
        const float t = T[k];
        const int keep = (sa > (1.0f / 256.0f)) & (t >= t_eps);
        const float w = keep ? sa : 0.0f;
        const float tw = t * w;
        cr[k] += tw * r;
        cg[k] += tw * g;
        cb[k] += tw * b;
        T[k] = t - tw;
    }
}

/* stable LSD radix sort of (key, idx), 4 x 8 bit, `nb` blocks of contiguous input; a pass whose digit is the same for
 * every key is skipped.  Result in keyA / idxA. */
static void par_sort(uint32_t v, uint32_t** pk, uint32_t** pi, uint32_t** pk2, uint32_t** pi2, int nb)
{
    uint32_t *ks = *pk, *is = *pi, *kd = *pk2, *id = *pi2;
    if (nb < 1) nb = 1;
    size_t* hist = (size_t*)malloc((size_t)nb * 256 * sizeof(size_t));
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = pass * 8;
#pragma omp parallel for schedule(static, 1) num_threads(nb)
        for (int b = 0; b < nb; ++b) {
            size_t* h = hist + (size_t)b * 256;
            memset(h, 0, 256 * sizeof(size_t));
            const uint32_t i0 = (uint32_t)(((uint64_t)v * b) / nb), i1 = (uint32_t)(((uint64_t)v * (b + 1)) / nb);
            for (uint32_t i = i0; i < i1; ++i) h[(ks[i] >> sh) & 255u]++;
        }
        size_t run = 0;
        int constant = 0;
        for (int d = 0; d < 256; ++d) {
            size_t tot = 0;
            for (int b = 0; b < nb; ++b) {
                const size_t c = hist[(size_t)b * 256 + d];
                hist[(size_t)b * 256 + d] = run + tot;
                tot += c;
            }
            if (tot == v) constant = 1;
            run += tot;
        }
        if (constant) continue;                 /* the order does not change: nothing to move */
#pragma omp parallel for schedule(static, 1) num_threads(nb)
        for (int b = 0; b < nb; ++b) {
            size_t* h = hist + (size_t)b * 256;
            const uint32_t i0 = (uint32_t)(((uint64_t)v * b) / nb), i1 = (uint32_t)(((uint64_t)v * (b + 1)) / nb);
            for (uint32_t i = i0; i < i1; ++i) {
                const size_t d = h[(ks[i] >> sh) & 255u]++;
                kd[d] = ks[i];
                id[d] = is[i];
            }
        }
        uint32_t* t;
        t = ks; ks = kd; kd = t;
        t = is; is = id; id = t;
    }
    free(hist);
    *pk = ks; *pi = is; *pk2 = kd; *pi2 = id;
}

/* Whole frame.  Returns V (0xFFFFFFFF on allocation failure).  stage_ms (may be NULL): cull, sort, project, bin,
 * composite, total.  sorted_idx_out / sorted_keys_out may be NULL.  rows [row0, row1) of the image are produced (tiles
 * that intersect them), the rest of `rgba` is left untouched. */
uint32_t orc_render_frame_tiled(size_t n, const float* aos, size_t stride, int full_sh, int srgb,
                                const float sortCameraMat[16], const float sortProjMat[16],
                                const float renderCameraMat[16], const float renderProjMat[16],
                                const float viewport[4], const float nearFar[2], float* rgba,
                                uint32_t* sorted_idx_out, uint32_t* sorted_keys_out, float t_eps, int nthreads,
                                int row0, int row1, double* stage_ms)
{
    tiled_ws* ws = &g_ws;
    if (nthreads < 1) nthreads = 1;
    omp_set_num_threads(nthreads);
    const int W = (int)viewport[2], H = (int)viewport[3];
    if (row0 < 0) row0 = 0;
    if (row1 > H) row1 = H;
    const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
    const size_t ntiles = (size_t)tiles_x * tiles_y;
    const int nb = nthreads;
    double t0 = now_ms(), t1;
    double st[6] = {0, 0, 0, 0, 0, 0};

    if (ws->n_cap < n) {
        free(ws->keyA); free(ws->keyB); free(ws->idxA); free(ws->idxB); free(ws->splats); free(ws->rect);
        const size_t c = n ? n : 1;
        ws->keyA = (uint32_t*)malloc(c * 4); ws->keyB = (uint32_t*)malloc(c * 4);
        ws->idxA = (uint32_t*)malloc(c * 4); ws->idxB = (uint32_t*)malloc(c * 4);
        ws->splats = (orc_splat2d*)malloc(c * sizeof(orc_splat2d));
        ws->rect = (uint32_t*)malloc(c * 16);
        ws->n_cap = c;
        if (!ws->keyA || !ws->keyB || !ws->idxA || !ws->idxB || !ws->splats || !ws->rect) { ws->n_cap = 0; return 0xFFFFFFFFu; }
    }
    if (grow((void**)&ws->cnt, &ws->cnt_cap, (size_t)nb * ntiles, 4)) return 0xFFFFFFFFu;
    if (grow((void**)&ws->tile_start, &ws->tile_cap, ntiles + 1, 4)) return 0xFFFFFFFFu;

    /* ---- cull + key, compacted in ascending index order (presort_compute.glsl:38-55) ---- */
    float viewS[16], mvp[16];
    orc_mat4_inverse(sortCameraMat, viewS);
    orc_mat4_mul(sortProjMat, viewS, mvp);
    uint32_t* bcount = (uint32_t*)calloc((size_t)nb + 1, sizeof(uint32_t));
#pragma omp parallel for schedule(static, 1) num_threads(nb)
    for (int b = 0; b < nb; ++b) {
        const size_t i0 = (n * (size_t)b) / nb, i1 = (n * (size_t)(b + 1)) / nb;
        uint32_t c = 0;
        for (size_t i = i0; i < i1; ++i) {
            uint32_t key;
            if (orc_cull_key(aos + i * stride, mvp, nearFar[1], &key)) {
                ws->keyB[i0 + c] = key;          /* compact inside the block first */
                ws->idxB[i0 + c] = (uint32_t)i;
                ++c;
            }
        }
        bcount[b + 1] = c;
    }
    for (int b = 0; b < nb; ++b) bcount[b + 1] += bcount[b];
    const uint32_t V = bcount[nb];
#pragma omp parallel for schedule(static, 1) num_threads(nb)
    for (int b = 0; b < nb; ++b) {
        const size_t i0 = (n * (size_t)b) / nb;
        const uint32_t c = bcount[b + 1] - bcount[b];
        memcpy(ws->keyA + bcount[b], ws->keyB + i0, (size_t)c * 4);
        memcpy(ws->idxA + bcount[b], ws->idxB + i0, (size_t)c * 4);
    }
    free(bcount);
    t1 = now_ms(); st[0] = t1 - t0; t0 = t1;

    /* ---- stable ascending sort (splatrenderer.cpp:223-264) ---- */
    par_sort(V, &ws->keyA, &ws->idxA, &ws->keyB, &ws->idxB, nb);
    if (sorted_idx_out) memcpy(sorted_idx_out, ws->idxA, (size_t)V * 4);
    if (sorted_keys_out) memcpy(sorted_keys_out, ws->keyA, (size_t)V * 4);
    t1 = now_ms(); st[1] = t1 - t0; t0 = t1;

    /* ---- vertex + geometry stage in draw order (orc_project is OpenMP-parallel) ---- */
    float viewR[16];
    orc_mat4_inverse(renderCameraMat, viewR);
    const float eye[3] = {renderCameraMat[12], renderCameraMat[13], renderCameraMat[14]};
    orc_project(V, ws->idxA, aos, stride, full_sh, srgb, viewR, renderProjMat, viewport, nearFar, eye, ws->splats);
    t1 = now_ms(); st[2] = t1 - t0; t0 = t1;

    /* ---- binning: per block of ranks, pairs per tile; lists stay in draw order ---- */
    const int ty_lo = row0 / TILE, ty_hi = (row1 > row0) ? (row1 - 1) / TILE : -1;
    memset(ws->cnt, 0, (size_t)nb * ntiles * 4);
#pragma omp parallel for schedule(static, 1) num_threads(nb)
    for (int b = 0; b < nb; ++b) {
        uint32_t* cnt = ws->cnt + (size_t)b * ntiles;
        const uint32_t r0 = (uint32_t)(((uint64_t)V * b) / nb), r1 = (uint32_t)(((uint64_t)V * (b + 1)) / nb);
        for (uint32_t r = r0; r < r1; ++r) {
            const orc_splat2d* g = &ws->splats[r];
            uint32_t* rc = ws->rect + (size_t)r * 4;
            rc[0] = 1; rc[2] = 0;
            /* degenerate covariance: nothing sensible to rasterise (the HIP path rejects the same splats) */
            const float det = g->cov[0] * g->cov[3] - g->cov[1] * g->cov[2];
            if (g->reject || !(det > 0.0f) || !(g->cov[0] > 0.0f) || !(g->cov[3] > 0.0f) || !(g->alpha > 1.0f / 256.0f)) continue;
            /* fragments survive the discard only inside w > 1/256 <=> d^T inv d < rho2 = 2 ln(256 alpha), whose axis-aligned
             * extents sqrt(rho2 cov_xx), sqrt(rho2 cov_yy) lie inside the 3.5-sigma quad's AABB (rho <= 3.33): a tighter
             * rectangle that drops no contributing pixel */
            const float rho2 = 2.0f * logf(256.0f * g->alpha);
            const float ex = fminf(g->hx, sqrtf(rho2 * g->cov[0]) * 1.0001f + 0.01f);
            const float ey = fminf(g->hy, sqrtf(rho2 * g->cov[3]) * 1.0001f + 0.01f);
            ws->splats[r].hx = ex;
            ws->splats[r].hy = ey;
            int xa, xb, ya, yb;
            pixel_range(g->px, ex, 0, W, &xa, &xb);
            pixel_range(g->py, ey, 0, H, &ya, &yb);
            if (xa > xb || ya > yb) continue;
            int tx0 = xa / TILE, tx1 = xb / TILE, ty0 = ya / TILE, ty1 = yb / TILE;
            if (ty0 < ty_lo) ty0 = ty_lo;
            if (ty1 > ty_hi) ty1 = ty_hi;
            if (ty0 > ty1) continue;
            rc[0] = (uint32_t)tx0; rc[1] = (uint32_t)ty0; rc[2] = (uint32_t)tx1; rc[3] = (uint32_t)ty1;
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) cnt[(size_t)ty * tiles_x + tx]++;
        }
    }
    /* offsets: tile-major, inside a tile by block (= by rank) */
#pragma omp parallel for schedule(static) num_threads(nb)
    for (int64_t t = 0; t < (int64_t)ntiles; ++t) {
        uint32_t tot = 0;
        for (int b = 0; b < nb; ++b) tot += ws->cnt[(size_t)b * ntiles + t];
        ws->tile_start[t + 1] = tot;
    }
    ws->tile_start[0] = 0;
    for (size_t t = 0; t < ntiles; ++t) ws->tile_start[t + 1] += ws->tile_start[t];
    const size_t D = ws->tile_start[ntiles];
    if (grow((void**)&ws->pairs, &ws->pair_cap, D + 1, 4)) return 0xFFFFFFFFu;
#pragma omp parallel for schedule(static) num_threads(nb)
    for (int64_t t = 0; t < (int64_t)ntiles; ++t) {
        uint32_t run = ws->tile_start[t];
        for (int b = 0; b < nb; ++b) {
            const uint32_t c = ws->cnt[(size_t)b * ntiles + t];
            ws->cnt[(size_t)b * ntiles + t] = run;
            run += c;
        }
    }
#pragma omp parallel for schedule(static, 1) num_threads(nb)
    for (int b = 0; b < nb; ++b) {
        uint32_t* cur = ws->cnt + (size_t)b * ntiles;
        const uint32_t r0 = (uint32_t)(((uint64_t)V * b) / nb), r1 = (uint32_t)(((uint64_t)V * (b + 1)) / nb);
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t* rc = ws->rect + (size_t)r * 4;
            if (rc[0] > rc[2]) continue;
            for (uint32_t ty = rc[1]; ty <= rc[3]; ++ty)
                for (uint32_t tx = rc[0]; tx <= rc[2]; ++tx) ws->pairs[cur[(size_t)ty * tiles_x + tx]++] = r;
        }
    }
    t1 = now_ms(); st[3] = t1 - t0; t0 = t1;

    /* ---- composite: one tile at a time, nearest splat first, stop when the whole tile is saturated ---- */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nb)
    for (int64_t t = 0; t < (int64_t)ntiles; ++t) {
        const int ty = (int)(t / tiles_x), tx = (int)(t - (int64_t)ty * tiles_x);
        if (ty < ty_lo || ty > ty_hi) continue;
        const int x0 = tx * TILE, y0 = ty * TILE;
        const int x1 = x0 + TILE < W ? x0 + TILE : W, y1 = y0 + TILE < H ? y0 + TILE : H;
        float T[TILE * TILE], cr[TILE * TILE], cg[TILE * TILE], cb[TILE * TILE];
        for (int k = 0; k < TILE * TILE; ++k) { T[k] = 1.0f; cr[k] = 0.0f; cg[k] = 0.0f; cb[k] = 0.0f; }
        const uint32_t s = ws->tile_start[t], e = ws->tile_start[t + 1];
        uint32_t since_check = 0;
        for (uint32_t k = e; k > s; --k) {
            const orc_splat2d* g = &ws->splats[ws->pairs[k - 1]];
            int xa, xb, ya, yb;
            pixel_range(g->px, g->hx, x0, x1, &xa, &xb);
            pixel_range(g->py, g->hy, y0, y1, &ya, &yb);
            if (xa > xb) continue;
            const float dx0 = ((float)xa + 0.5f) - g->px;
            for (int y = ya; y <= yb; ++y) {
                const int p0 = (y - y0) * TILE + (xa - x0);
                /* splat_frag.glsl:20-41 (same expression as orc_composite); a saturated pixel (T < t_eps) takes no more */
                blend_row(T + p0, cr + p0, cg + p0, cb + p0, xb - xa + 1, dx0, ((float)y + 0.5f) - g->py, g->inv[0], g->inv[1],
                          g->inv[2], g->inv[3], g->alpha, g->rgb[0], g->rgb[1], g->rgb[2], t_eps);
            }
            if (++since_check == 16 && t_eps > 0.0f) {         /* every 16 splats: is the whole tile saturated? */
                since_check = 0;
                int live = 0;
                for (int yy = 0; yy < y1 - y0; ++yy)
                    for (int xx = 0; xx < x1 - x0; ++xx) live += T[yy * TILE + xx] >= t_eps;
                if (!live) break;
            }
        }
        for (int y = y0; y < y1; ++y) {
            if (y < row0 || y >= row1) continue;
            for (int x = x0; x < x1; ++x) {
                const int p = (y - y0) * TILE + (x - x0);
                float* d = rgba + ((size_t)y * W + x) * 4;
                d[0] = cr[p]; d[1] = cg[p]; d[2] = cb[p]; d[3] = 1.0f;     /* dst.a stays 1 (app.cpp:158-160) */
            }
        }
    }
    t1 = now_ms(); st[4] = t1 - t0;
    st[5] = st[0] + st[1] + st[2] + st[3] + st[4];
    if (stage_ms) memcpy(stage_ms, st, sizeof(st));
    return V;
}
