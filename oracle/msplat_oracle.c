/*
 * msplat_oracle.c -- CPU ORACLE (test infrastructure; see msplat_oracle.h for the rules).
 * PARITY PINNED (round 4) for the shader arithmetic: the reference's own shader FILES, executed on Mesa llvmpipe by oracle/glref,
 * give the same visible sets and 32-bit keys bit for bit and the same framebuffers within SURVEY 8c's tolerance
 * (tests/test_reference_shaders.py: 15 cases; committed outputs tests/golden/glref_*.npz).  Not pinned: glm's host-side closed
 * forms (below), the ROP's internal precision on 8-bit / fp16 targets (implementation-defined in GL), the point-cloud shaders.
 *
 * Literal restatement, function by function, of the reference's hot path:
 *   shader/presort_compute.glsl:31-57            -> orc_cull_key / orc_presort
 *   src/splatrenderer.cpp:223-264 (sort contract) -> orc_sort
 *   shader/splat_vert.glsl:51-127,153-222        -> sh_radiance / project_one
 *   shader/splat_geom.glsl:22-87                 -> project_one (inverse, reject, extents)
 *   shader/splat_frag.glsl:18-42 + src/app.cpp:153-160 -> orc_composite
 *   src/gaussiancloud.cpp:86-94,119-122,254-361  -> orc_build_cloud
 *   src/splatrenderer.cpp:161,175,327-335        -> orc_render_frame (host matrices)
 *   src/core/util.cpp:420-480                    -> orc_create_projection
 * glm (vcpkg "latest", unpinned, not under /root/reference) closed forms restated:
 *   inverse(mat4) cofactor expansion, mat4*mat4, perspective (RH, -1..1), quat->mat3.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include "msplat_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define M(m, c, r) ((m)[(c) * 4 + (r)])

/* ------------------------------------------------------------------------------------- */
/* matrices                                                                              */
/* ------------------------------------------------------------------------------------- */

void orc_mat4_mul(const float a[16], const float b[16], float out[16])
{
    /* glm: Result[c] = A[0]*B[c][0] + A[1]*B[c][1] + A[2]*B[c][2] + A[3]*B[c][3] */
    float tmp[16];
    for (int c = 0; c < 4; ++c) {
        for (int r = 0; r < 4; ++r) {
            float s = M(a, 0, r) * M(b, c, 0);
            s = s + M(a, 1, r) * M(b, c, 1);
            s = s + M(a, 2, r) * M(b, c, 2);
            s = s + M(a, 3, r) * M(b, c, 3);
            tmp[c * 4 + r] = s;
        }
    }
    memcpy(out, tmp, sizeof(tmp));
}

void orc_mat4_inverse(const float m[16], float out[16])
{
    /* glm::inverse(mat4): 2x2 sub-determinant ("Coef") form */
    float c00 = M(m,2,2) * M(m,3,3) - M(m,3,2) * M(m,2,3);
    float c02 = M(m,1,2) * M(m,3,3) - M(m,3,2) * M(m,1,3);
    float c03 = M(m,1,2) * M(m,2,3) - M(m,2,2) * M(m,1,3);
    float c04 = M(m,2,1) * M(m,3,3) - M(m,3,1) * M(m,2,3);
    float c06 = M(m,1,1) * M(m,3,3) - M(m,3,1) * M(m,1,3);
    float c07 = M(m,1,1) * M(m,2,3) - M(m,2,1) * M(m,1,3);
    float c08 = M(m,2,1) * M(m,3,2) - M(m,3,1) * M(m,2,2);
    float c10 = M(m,1,1) * M(m,3,2) - M(m,3,1) * M(m,1,2);
    float c11 = M(m,1,1) * M(m,2,2) - M(m,2,1) * M(m,1,2);
    float c12 = M(m,2,0) * M(m,3,3) - M(m,3,0) * M(m,2,3);
    float c14 = M(m,1,0) * M(m,3,3) - M(m,3,0) * M(m,1,3);
    float c15 = M(m,1,0) * M(m,2,3) - M(m,2,0) * M(m,1,3);
    float c16 = M(m,2,0) * M(m,3,2) - M(m,3,0) * M(m,2,2);
    float c18 = M(m,1,0) * M(m,3,2) - M(m,3,0) * M(m,1,2);
    float c19 = M(m,1,0) * M(m,2,2) - M(m,2,0) * M(m,1,2);
    float c20 = M(m,2,0) * M(m,3,1) - M(m,3,0) * M(m,2,1);
    float c22 = M(m,1,0) * M(m,3,1) - M(m,3,0) * M(m,1,1);
    float c23 = M(m,1,0) * M(m,2,1) - M(m,2,0) * M(m,1,1);

    float f0[4] = {c00, c00, c02, c03};
    float f1[4] = {c04, c04, c06, c07};
    float f2[4] = {c08, c08, c10, c11};
    float f3[4] = {c12, c12, c14, c15};
    float f4[4] = {c16, c16, c18, c19};
    float f5[4] = {c20, c20, c22, c23};
    float v0[4] = {M(m,1,0), M(m,0,0), M(m,0,0), M(m,0,0)};
    float v1[4] = {M(m,1,1), M(m,0,1), M(m,0,1), M(m,0,1)};
    float v2[4] = {M(m,1,2), M(m,0,2), M(m,0,2), M(m,0,2)};
    float v3[4] = {M(m,1,3), M(m,0,3), M(m,0,3), M(m,0,3)};
    static const float sa[4] = {+1.0f, -1.0f, +1.0f, -1.0f};
    static const float sb[4] = {-1.0f, +1.0f, -1.0f, +1.0f};
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        float i0 = (v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i];
        float i1 = (v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i];
        float i2 = (v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i];
        float i3 = (v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i];
        inv[0 * 4 + i] = i0 * sa[i];
        inv[1 * 4 + i] = i1 * sb[i];
        inv[2 * 4 + i] = i2 * sa[i];
        inv[3 * 4 + i] = i3 * sb[i];
    }
    float d0 = M(m,0,0) * inv[0 * 4 + 0];
    float d1 = M(m,0,1) * inv[1 * 4 + 0];
    float d2 = M(m,0,2) * inv[2 * 4 + 0];
    float d3 = M(m,0,3) * inv[3 * 4 + 0];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * ood;
}

void orc_perspective(float fovy, float aspect, float zn, float zf, float out[16])
{
    /* glm::perspective, right-handed, clip z in [-1,1]  (app.cpp:1042) */
    float t = tanf(fovy / 2.0f);
    memset(out, 0, 16 * sizeof(float));
    M(out, 0, 0) = 1.0f / (aspect * t);
    M(out, 1, 1) = 1.0f / t;
    M(out, 2, 2) = -(zf + zn) / (zf - zn);
    M(out, 2, 3) = -1.0f;
    M(out, 3, 2) = -(2.0f * zf * zn) / (zf - zn);
}

void orc_create_projection(float tanL, float tanR, float tanU, float tanD,
                           float zn, float zf, float m[16])
{
    /* util.cpp:420-480, GRAPHICS_OPENGL branch (offsetZ = nearZ, height = up - down) */
    const float w = tanR - tanL;
    const float h = tanU - tanD;
    const float offsetZ = zn;
    memset(m, 0, 16 * sizeof(float));
    m[0] = 2 / w;
    m[8] = (tanR + tanL) / w;
    m[5] = 2 / h;
    m[9] = (tanU + tanD) / h;
    m[11] = -1;
    if (zf <= zn) {
        m[10] = -1;
        m[14] = -(zn + offsetZ);
    } else {
        m[10] = -(zf + offsetZ) / (zf - zn);
        m[14] = -(zf * (zn + offsetZ)) / (zf - zn);
    }
}

/* ------------------------------------------------------------------------------------- */
/* load time                                                                             */
/* ------------------------------------------------------------------------------------- */

static void mat3_mul(const float a[9], const float b[9], float out[9])
{
    /* column-major 3x3, glm operator*: out[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2] */
    float t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            float s = a[0 * 3 + r] * b[c * 3 + 0];
            s = s + a[1 * 3 + r] * b[c * 3 + 1];
            s = s + a[2 * 3 + r] * b[c * 3 + 2];
            t[c * 3 + r] = s;
        }
    memcpy(out, t, sizeof(t));
}

static void mat3_transpose(const float a[9], float out[9])
{
    float t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) t[c * 3 + r] = a[r * 3 + c];
    memcpy(out, t, sizeof(t));
}

static void cov_from_rot_scale(const float rot[4], const float scale[3], float V[9])
{
    /* gaussiancloud.cpp:86-94: q = (w=rot0, x=rot1, y=rot2, z=rot3) normalised; R S S^T R^T */
    float w = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float t0 = w * w, t1 = x * x, t2 = y * y, t3 = z * z;
    float len = sqrtf((t0 + t1) + (t2 + t3));
    if (len <= 0.0f) { w = 1.0f; x = y = z = 0.0f; }
    else { float ool = 1.0f / len; w *= ool; x *= ool; y *= ool; z *= ool; }
    float qxx = x * x, qyy = y * y, qzz = z * z;
    float qxz = x * z, qxy = x * y, qyz = y * z;
    float qwx = w * x, qwy = w * y, qwz = w * z;
    float R[9];
    R[0] = 1.0f - 2.0f * (qyy + qzz);
    R[1] = 2.0f * (qxy + qwz);
    R[2] = 2.0f * (qxz - qwy);
    R[3] = 2.0f * (qxy - qwz);
    R[4] = 1.0f - 2.0f * (qxx + qzz);
    R[5] = 2.0f * (qyz + qwx);
    R[6] = 2.0f * (qxz + qwy);
    R[7] = 2.0f * (qyz - qwx);
    R[8] = 1.0f - 2.0f * (qxx + qyy);
    float S[9] = {scale[0], 0, 0, 0, scale[1], 0, 0, 0, scale[2]};
    float St[9], Rt[9], A[9], B[9];
    mat3_transpose(S, St);
    mat3_transpose(R, Rt);
    mat3_mul(R, S, A);
    mat3_mul(A, St, B);
    mat3_mul(B, Rt, V);
}

void orc_build_cloud(size_t n, const float* xyz, const float* f_dc, const float* f_rest,
                     const float* opacity, const float* log_scale, const float* rot,
                     int full_sh, float* aos_out)
{
    const size_t stride = full_sh ? ORC_FULL_FLOATS : ORC_BASE_FLOATS;
    for (size_t i = 0; i < n; ++i) {
        float* o = aos_out + i * stride;
        o[0] = xyz[i * 3 + 0];
        o[1] = xyz[i * 3 + 1];
        o[2] = xyz[i * 3 + 2];
        o[3] = 1.0f / (1.0f + expf(-opacity[i]));            /* gaussiancloud.cpp:119-122 */
        if (full_sh) {
            /* gaussiancloud.cpp:262-314: channel c, coeff k: k=0 -> f_dc[c]; k>0 -> f_rest[c*15 + k-1] */
            static const int off0[3] = {ORC_OFF_R_SH0, ORC_OFF_G_SH0, ORC_OFF_B_SH0};
            static const int off1[3] = {ORC_OFF_R_SH1, ORC_OFF_G_SH1, ORC_OFF_B_SH1};
            for (int c = 0; c < 3; ++c) {
                o[off0[c] + 0] = f_dc[i * 3 + c];
                for (int k = 1; k < 4; ++k) o[off0[c] + k] = f_rest[i * 45 + c * 15 + (k - 1)];
                for (int k = 4; k < 16; ++k) o[off1[c] + (k - 4)] = f_rest[i * 45 + c * 15 + (k - 1)];
            }
        } else {
            /* gaussiancloud.cpp:316-332 */
            o[ORC_OFF_R_SH0] = f_dc[i * 3 + 0]; o[ORC_OFF_R_SH0 + 1] = o[ORC_OFF_R_SH0 + 2] = o[ORC_OFF_R_SH0 + 3] = 0.0f;
            o[ORC_OFF_G_SH0] = f_dc[i * 3 + 1]; o[ORC_OFF_G_SH0 + 1] = o[ORC_OFF_G_SH0 + 2] = o[ORC_OFF_G_SH0 + 3] = 0.0f;
            o[ORC_OFF_B_SH0] = f_dc[i * 3 + 2]; o[ORC_OFF_B_SH0 + 1] = o[ORC_OFF_B_SH0 + 2] = o[ORC_OFF_B_SH0 + 3] = 0.0f;
        }
        float sc[3] = {expf(log_scale[i * 3 + 0]), expf(log_scale[i * 3 + 1]), expf(log_scale[i * 3 + 2])};
        float V[9];
        cov_from_rot_scale(rot + i * 4, sc, V);
        for (int k = 0; k < 9; ++k) o[ORC_OFF_COV0 + k] = V[k];   /* col0, col1, col2 */
    }
}

/* ------------------------------------------------------------------------------------- */
/* cull + key                                                                            */
/* ------------------------------------------------------------------------------------- */

static inline void mat4_mul_point(const float m[16], float x, float y, float z, float p[4])
{
    /* m * vec4(x,y,z,1): ((m0*x + m1*y) + m2*z) + m3*1, one rounding per op */
    for (int r = 0; r < 4; ++r) {
        float s = M(m, 0, r) * x;
        s = s + M(m, 1, r) * y;
        s = s + M(m, 2, r) * z;
        s = s + M(m, 3, r);
        p[r] = s;
    }
}

int orc_cull_key(const float xyz[3], const float mvp[16], float zfar, uint32_t* key_out)
{
    float p[4];
    mat4_mul_point(mvp, xyz[0], xyz[1], xyz[2], p);
    float depth = p[3];
    float xx = p[0] / depth;
    float yy = p[1] / depth;
    const float CLIP = 1.5f;
    if (depth > 0.0f && xx < CLIP && xx > -CLIP && yy < CLIP && yy > -CLIP) {
        /* keyMax - uint((depth / far) * keyMax); float(0xFFFFFFFFu) == 2^32 */
        float f = (depth / zfar) * 4294967296.0f;
        uint32_t q;
        if (f >= 4294967296.0f) q = 0xFFFFFFFFu;   /* GLSL: undefined; we saturate (far-clipped anyway) */
        else q = (uint32_t)f;                      /* truncation; f > 0 here */
        *key_out = 0xFFFFFFFFu - q;
        return 1;
    }
    return 0;
}

uint32_t orc_presort(size_t n, const float* aos, size_t stride, const float mvp[16],
                     float zfar, uint32_t* keys_out, uint32_t* idx_out)
{
    uint32_t v = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t key;
        if (orc_cull_key(aos + i * stride, mvp, zfar, &key)) {
            keys_out[v] = key;
            idx_out[v] = (uint32_t)i;
            ++v;
        }
    }
    return v;
}

/* ------------------------------------------------------------------------------------- */
/* sort: ascending key, stable                                                           */
/* ------------------------------------------------------------------------------------- */

void orc_sort(uint32_t v, uint32_t* keys, uint32_t* idx)
{
    /* LSD radix, 4 x 8 bit, exactly the reference's contract (stable, ascending) */
    if (v == 0) return;
    uint32_t* k2 = (uint32_t*)malloc((size_t)v * sizeof(uint32_t));
    uint32_t* i2 = (uint32_t*)malloc((size_t)v * sizeof(uint32_t));
    uint32_t *ks = keys, *is = idx, *kd = k2, *id = i2;
    for (int pass = 0; pass < 4; ++pass) {
        size_t hist[257];
        memset(hist, 0, sizeof(hist));
        const int sh = pass * 8;
        for (uint32_t i = 0; i < v; ++i) hist[((ks[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (uint32_t i = 0; i < v; ++i) {
            size_t d = hist[(ks[i] >> sh) & 255u]++;
            kd[d] = ks[i];
            id[d] = is[i];
        }
        uint32_t* t;
        t = ks; ks = kd; kd = t;
        t = is; is = id; id = t;
    }
    /* 4 passes: result is back in keys/idx */
    free(k2);
    free(i2);
}

/* ------------------------------------------------------------------------------------- */
/* vertex + geometry stage                                                               */
/* ------------------------------------------------------------------------------------- */

static void sh_radiance(const float* rec, int full_sh, const float v[3], float rgb[3])
{
    /* splat_vert.glsl:51-127 */
    float b[16];
    float vx = v[0], vy = v[1], vz = v[2];
    float vx2 = vx * vx, vy2 = vy * vy, vz2 = vz * vz;
    b[0] = 0.28209479177387814f;
    float k1 = 0.4886025119029199f;
    b[1] = -k1 * vy;
    b[2] = k1 * vz;
    b[3] = -k1 * vx;
    const float* sh0[3] = {rec + ORC_OFF_R_SH0, rec + ORC_OFF_G_SH0, rec + ORC_OFF_B_SH0};
    if (full_sh) {
        float k2 = 1.0925484305920792f, k3 = 0.31539156525252005f, k4 = 0.5462742152960396f;
        b[4] = k2 * vy * vx;
        b[5] = -k2 * vy * vz;
        b[6] = k3 * (3.0f * vz2 - 1.0f);
        b[7] = -k2 * vx * vz;
        b[8] = k4 * (vx2 - vy2);
        float k5 = 0.5900435899266435f, k6 = 2.8906114426405543f, k7 = 0.4570457994644658f;
        float k8 = 0.37317633259011546f, k9 = 1.4453057213202771f;
        b[9] = -k5 * vy * (3.0f * vx2 - vy2);
        b[10] = k6 * vy * vx * vz;
        b[11] = -k7 * vy * (5.0f * vz2 - 1.0f);
        b[12] = k8 * vz * (5.0f * vz2 - 3.0f);
        b[13] = -k7 * vx * (5.0f * vz2 - 1.0f);
        b[14] = k9 * vz * (vx2 - vy2);
        b[15] = -k5 * vx * (vx2 - 3.0f * vy2);
        const float* sh1[3] = {rec + ORC_OFF_R_SH1, rec + ORC_OFF_G_SH1, rec + ORC_OFF_B_SH1};
        for (int c = 0; c < 3; ++c) {
            float s = b[0] * sh0[c][0];
            for (int k = 1; k < 4; ++k) s = s + b[k] * sh0[c][k];
            for (int k = 4; k < 16; ++k) s = s + b[k] * sh1[c][k - 4];
            rgb[c] = 0.5f + s;
        }
    } else {
        for (int c = 0; c < 3; ++c) {
            float s = b[0] * sh0[c][0];
            for (int k = 1; k < 4; ++k) s = s + b[k] * sh0[c][k];
            rgb[c] = 0.5f + s;
        }
    }
}

static float srgb_to_linear(float s)
{
    /* splat_vert.glsl:129-141 */
    if (s <= 0.04045f) return s / 12.92f;
    return powf((s + 0.055f) / 1.055f, 2.4f);
}

static void project_one(const float* rec, int full_sh, int srgb, const float viewMat[16],
                        const float projMat[16], const float viewport[4], const float nearFar[2],
                        const float eye[3], orc_splat2d* o)
{
    float alpha = rec[3];
    float t[4];
    mat4_mul_point(viewMat, rec[0], rec[1], rec[2], t);

    float X0 = viewport[0] * (0.00001f * nearFar[0]);   /* splat_vert.glsl:160 */
    float Y0 = viewport[1];
    float WIDTH = viewport[2];
    float HEIGHT = viewport[3];
    float Z_NEAR = nearFar[0];
    float Z_FAR = nearFar[1];

    float SX = M(projMat, 0, 0);
    float SY = M(projMat, 1, 1);
    float WZ = M(projMat, 3, 2);
    float tzSq = t[2] * t[2];
    float jsx = -(SX * WIDTH) / (2.0f * t[2]);
    float jsy = -(SY * HEIGHT) / (2.0f * t[2]);
    float jtx = (SX * t[0] * WIDTH) / (2.0f * tzSq);
    float jty = (SY * t[1] * HEIGHT) / (2.0f * tzSq);
    float jtz = ((Z_FAR - Z_NEAR) * WZ) / (2.0f * tzSq);
    /* GLSL mat3(vec3,vec3,vec3) takes COLUMNS */
    float J[9] = {jsx, 0.0f, 0.0f, 0.0f, jsy, 0.0f, jtx, jty, jtz};
    float W[9] = {M(viewMat,0,0), M(viewMat,0,1), M(viewMat,0,2),
                  M(viewMat,1,0), M(viewMat,1,1), M(viewMat,1,2),
                  M(viewMat,2,0), M(viewMat,2,1), M(viewMat,2,2)};
    float V[9];
    for (int k = 0; k < 9; ++k) V[k] = rec[ORC_OFF_COV0 + k];
    float JW[9], JWt[9], A[9], Vp[9];
    mat3_mul(J, W, JW);
    mat3_transpose(JW, JWt);
    mat3_mul(JW, V, A);
    mat3_mul(A, JWt, Vp);

    /* mat2(V_prime): columns (Vp[0][0],Vp[0][1]), (Vp[1][0],Vp[1][1]) */
    float m00 = Vp[0 * 3 + 0], m01 = Vp[0 * 3 + 1], m10 = Vp[1 * 3 + 0], m11 = Vp[1 * 3 + 1];
    m00 += 0.3f;
    m11 += 0.3f;
    o->cov[0] = m00; o->cov[1] = m01; o->cov[2] = m10; o->cov[3] = m11;

    float p4[4];
    {
        /* projMat * t (t is a full vec4 with w = 1 from the affine viewMat) */
        for (int r = 0; r < 4; ++r) {
            float s = M(projMat, 0, r) * t[0];
            s = s + M(projMat, 1, r) * t[1];
            s = s + M(projMat, 2, r) * t[2];
            s = s + M(projMat, 3, r) * t[3];
            p4[r] = s;
        }
    }
    float gx = p4[0] / p4[3];
    float gy = p4[1] / p4[3];
    o->px = 0.5f * (WIDTH + (gx * WIDTH) + (2.0f * X0));
    o->py = 0.5f * (HEIGHT + (gy * HEIGHT) + (2.0f * Y0));

    float d[3] = {rec[0] - eye[0], rec[1] - eye[1], rec[2] - eye[2]};
    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float v[3] = {d[0] / len, d[1] / len, d[2] / len};
    sh_radiance(rec, full_sh, v, o->rgb);
    o->alpha = alpha;
    if (srgb) {
        for (int c = 0; c < 3; ++c) o->rgb[c] = srgb_to_linear(o->rgb[c]);
    }

    /* ---- geometry stage: splat_geom.glsl:34-87 ---- */
    float det = m00 * m11 - m01 * m10;
    o->inv[0] = m11 / det;
    o->inv[1] = -m01 / det;
    o->inv[2] = -m10 / det;
    o->inv[3] = m00 / det;

    o->ndc[0] = p4[0] / p4[3];
    o->ndc[1] = p4[1] / p4[3];
    o->ndc[2] = p4[2] / p4[3];
    o->depth = p4[3];
    int reject = 0;
    if (o->ndc[2] < 0.25f || o->ndc[0] > 2.0f || o->ndc[0] < -2.0f ||
        o->ndc[1] > 2.0f || o->ndc[1] < -2.0f) reject = 1;
    /* fixed-function clip of the whole quad against the far plane (all 4 vertices share z,w);
       NaN centres never rasterise */
    if (!(o->ndc[2] <= 1.0f)) reject = 1;
    if (!(p4[3] > 0.0f)) reject = 1;

    /* oriented 3.5-sigma quad -> AABB half extents (superset of the rasterised pixels) */
    float k = 3.5f;
    float a = m00, b = m01, c = m11;
    float apco2 = (a + c) / 2.0f;
    float amco2 = (a - c) / 2.0f;
    float term = sqrtf(amco2 * amco2 + b * b);
    float maj = apco2 + term;
    float mn = apco2 - term;
    float theta;
    if (b == 0.0f) theta = (a >= c) ? 0.0f : 1.57079632679489661923f;
    else theta = atan2f(maj - a, b);
    float r1 = k * sqrtf(maj);
    float r2 = k * sqrtf(mn);
    float majx = r1 * cosf(theta), majy = r1 * sinf(theta);
    float minx = r2 * cosf(theta + 1.57079632679489661923f), miny = r2 * sinf(theta + 1.57079632679489661923f);
    o->hx = fabsf(majx) + fabsf(minx);
    o->hy = fabsf(majy) + fabsf(miny);
    if (!(o->hx == o->hx) || !(o->hy == o->hy)) reject = 1;   /* NaN quad: nothing rasterised */
    o->reject = reject;
}

void orc_project(uint32_t v, const uint32_t* idx, const float* aos, size_t stride,
                 int full_sh, int srgb, const float viewMat[16], const float projMat[16],
                 const float viewport[4], const float nearFar[2], const float eye[3],
                 orc_splat2d* out)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)v; ++r) {
        uint32_t i = idx ? idx[r] : (uint32_t)r;
        project_one(aos + (size_t)i * stride, full_sh, srgb, viewMat, projMat, viewport, nearFar,
                    eye, &out[r]);
        out[r].index = i;
    }
}

/* ------------------------------------------------------------------------------------- */
/* fragment stage + blend                                                                */
/* ------------------------------------------------------------------------------------- */

static uint64_t g_fragments = 0;
uint64_t orc_last_fragment_count(void) { return g_fragments; }

static inline void pixel_range(float centre, float half, int limit, int lo_clip, int hi_clip,
                               int* lo, int* hi)
{
    /* pixels i whose centre i+0.5 lies in [centre-half, centre+half], padded by one pixel */
    float a = floorf(centre - half - 0.5f) - 1.0f;
    float b = ceilf(centre + half - 0.5f) + 1.0f;
    if (a < (float)lo_clip) a = (float)lo_clip;
    if (b > (float)(hi_clip - 1)) b = (float)(hi_clip - 1);
    (void)limit;
    if (!(a <= b)) { *lo = 0; *hi = -1; return; }
    *lo = (int)a;
    *hi = (int)b;
}

static void composite_impl(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba,
                           int row0, int row1, int nthreads, float* flip_budget, float flip_rel)
{
    if (row0 < 0) row0 = 0;
    if (row1 > H) row1 = H;
    if (nthreads < 1) nthreads = 1;
    uint64_t frags = 0;
    const int rows = row1 - row0;
    if (rows <= 0) return;
    if (nthreads > rows) nthreads = rows;
#pragma omp parallel for schedule(static, 1) num_threads(nthreads) reduction(+ : frags)
    for (int band = 0; band < nthreads; ++band) {
        const int y0 = row0 + (int)(((int64_t)rows * band) / nthreads);
        const int y1 = row0 + (int)(((int64_t)rows * (band + 1)) / nthreads);
        /* clear: (0,0,0,1)  app.cpp:158-160 */
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W; ++x) {
                float* d = rgba + ((size_t)y * W + x) * 4;
                d[0] = 0.0f; d[1] = 0.0f; d[2] = 0.0f; d[3] = 1.0f;
            }
        for (uint32_t k = 0; k < v; ++k) {        /* draw order = array order = far -> near */
            const orc_splat2d* g = &s[k];
            if (g->reject) continue;
            int xa, xb, ya, yb;
            pixel_range(g->px, g->hx, W, 0, W, &xa, &xb);
            pixel_range(g->py, g->hy, H, y0, y1, &ya, &yb);
            for (int y = ya; y <= yb; ++y) {
                for (int x = xa; x <= xb; ++x) {
                    /* splat_frag.glsl:20-41 */
                    float dx = ((float)x + 0.5f) - g->px;
                    float dy = ((float)y + 0.5f) - g->py;
                    /* cov2Dinv * d : columns (inv0,inv1),(inv2,inv3) */
                    float mx = g->inv[0] * dx + g->inv[2] * dy;
                    float my = g->inv[1] * dx + g->inv[3] * dy;
                    float q = dx * mx + dy * my;
                    float e = expf(-0.5f * q);
                    float sa = g->alpha * e;
                    ++frags;
                    float* d = rgba + ((size_t)y * W + x) * 4;
                    if (flip_budget && fabsf(sa - (1.0f / 256.0f)) <= flip_rel * (1.0f / 256.0f)) {
                        /* a fragment this close to the discard threshold may fall on the other side in an
                         * implementation whose w differs by a few ulp; blending it or not moves the pixel
                         * by at most w * (|c| + |dst|) (later blends only attenuate the difference) */
                        float cm = fmaxf(fabsf(g->rgb[0]), fmaxf(fabsf(g->rgb[1]), fabsf(g->rgb[2])));
                        float dm = fmaxf(fabsf(d[0]), fmaxf(fabsf(d[1]), fabsf(d[2])));
                        flip_budget[(size_t)y * W + x] += sa * (cm + dm);
                    }
                    if (sa <= (1.0f / 256.0f)) continue;        /* discard */
                    float oma = 1.0f - sa;
                    /* GL_ONE, GL_ONE_MINUS_SRC_ALPHA  (app.cpp:153-156) */
                    d[0] = (sa * g->rgb[0]) + oma * d[0];
                    d[1] = (sa * g->rgb[1]) + oma * d[1];
                    d[2] = (sa * g->rgb[2]) + oma * d[2];
                    d[3] = sa + oma * d[3];
                }
            }
        }
    }
    g_fragments = frags;
}

void orc_composite(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba,
                   int row0, int row1, int nthreads)
{
    composite_impl(v, s, W, H, rgba, row0, row1, nthreads, NULL, 0.0f);
}

/* orc_composite that also reports, per pixel, how far discard-threshold flips could move it:
 * flip_budget[y*W+x] (caller-zeroed, W*H floats) += w (|c| + |dst|) for every fragment with
 * |w - 1/256| <= flip_rel/256.  Lets the parity tests explain every pixel outside the tight tolerance. */
void orc_composite_flip(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba,
                        int row0, int row1, int nthreads, float* flip_budget, float flip_rel)
{
    composite_impl(v, s, W, H, rgba, row0, row1, nthreads, flip_budget, flip_rel);
}

/* Window-space depth of a splat's fragments as an order-preserving uint32.  The quad's four vertices
 * share the centre's clip z and w (splat_geom.glsl:93-101 offsets x and y only), so every fragment of a
 * splat has z_ndc = ndc[2]; window z = 0.5 z_ndc + 0.5 (default glDepthRange); a 24-bit buffer stores
 * round(z_w (2^24 - 1)) (sdl_main.cpp:79 asks for 24 bits), a float buffer the value itself. */
uint32_t orc_quantise_depth(float ndcz, int depth_bits)
{
    float zw = 0.5f * ndcz + 0.5f;
    if (!(zw >= 0.0f)) return 0u;
    if (depth_bits == 24) {
        double q = floor((double)zw * 16777215.0 + 0.5);
        return q >= 16777215.0 ? 16777215u : (uint32_t)q;
    }
    uint32_t u;
    memcpy(&u, &zw, 4);
    return u;
}

/* orc_composite with the depth test the reference leaves enabled (app.cpp:163: glEnable(GL_DEPTH_TEST),
 * default func GL_LESS, depth writes on, buffer cleared to 1.0 by app.cpp:160): a fragment that survives
 * the discard is blended only if its depth is LESS than the stored one, and then stores its depth.
 * Discarded fragments write nothing (splat_frag.glsl:37-40 `discard`). */
/* fp32 -> fp16 (round to nearest even, no traps) -> fp32: what an RGBA16F render target stores */
static float round_to_half(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return f;                          /* inf / nan */
    if (a >= 0x477FF000u) {                                   /* >= 65520: rounds to inf */
        const uint32_t inf = sign | 0x7F800000u;
        float r; memcpy(&r, &inf, 4); return r;
    }
    if (a < 0x38800000u) {                                    /* below 2^-14: half subnormal, quantum 2^-24 */
        float m; memcpy(&m, &a, 4);
        const float q = 5.9604644775390625e-08f;              /* 2^-24 */
        float r = nearbyintf(m / q) * q;                      /* default rounding mode: nearest even */
        uint32_t ru; memcpy(&ru, &r, 4); ru |= sign; memcpy(&r, &ru, 4);
        return r;
    }
    /* normal half: keep 10 mantissa bits */
    const uint32_t rem = a & 0x1FFFu, keep = a & ~0x1FFFu;
    uint32_t res = keep;
    if (rem > 0x1000u || (rem == 0x1000u && (keep & 0x2000u))) res += 0x2000u;
    res |= sign;
    float r; memcpy(&r, &res, 4); return r;
}

static float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }      /* NaN -> 1 is irrelevant here */
static float to_unorm8(float x) { return floorf(clamp01(x) * 255.0f + 0.5f) / 255.0f; }

/* The blend as the render-target hardware (ROP) performs it, in draw order, for the targets the GL app actually uses
 * (src/app.cpp:1012-1020):
 *   rop = 0  float accumulation, nothing rounded (the colour-only fp32 FBO: orc_composite's semantics)
 *   rop = 1  RGBA8 (the default back buffer): GL 4.6 17.3.6 -- source colour, destination colour and the result are
 *            clamped to [0,1]; the result is stored as 8-bit unorm (round to nearest) after EVERY blend
 *   rop = 2  RGBA16F (--fp16): no clamping, the result is rounded to fp16 after every blend
 * depth_bits = 0: no depth test; 24 / 32: the GL_LESS test the reference leaves enabled (app.cpp:160,163).
 * The arithmetic precision inside a ROP is implementation-defined: this restates the specification, not a driver. */
void orc_composite_rop(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba, int depth_bits, int rop,
                       int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > H) nthreads = H;
    if (H <= 0 || W <= 0) return;
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
    for (int band = 0; band < nthreads; ++band) {
        const int y0 = (int)(((int64_t)H * band) / nthreads);
        const int y1 = (int)(((int64_t)H * (band + 1)) / nthreads);
        uint32_t* zbuf = (uint32_t*)malloc((size_t)(y1 - y0 > 0 ? y1 - y0 : 1) * W * sizeof(uint32_t));
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W; ++x) {
                float* d = rgba + ((size_t)y * W + x) * 4;
                d[0] = 0.0f; d[1] = 0.0f; d[2] = 0.0f; d[3] = 1.0f;
                zbuf[(size_t)(y - y0) * W + x] = 0xFFFFFFFFu;           /* glClear: depth 1.0 */
            }
        for (uint32_t k = 0; k < v; ++k) {
            const orc_splat2d* g = &s[k];
            if (g->reject) continue;
            const uint32_t zq = depth_bits ? orc_quantise_depth(g->ndc[2], depth_bits) : 0u;
            int xa, xb, ya, yb;
            pixel_range(g->px, g->hx, W, 0, W, &xa, &xb);
            pixel_range(g->py, g->hy, H, y0, y1, &ya, &yb);
            for (int y = ya; y <= yb; ++y) {
                for (int x = xa; x <= xb; ++x) {
                    float dx = ((float)x + 0.5f) - g->px;
                    float dy = ((float)y + 0.5f) - g->py;
                    float mx = g->inv[0] * dx + g->inv[2] * dy;
                    float my = g->inv[1] * dx + g->inv[3] * dy;
                    float q = dx * mx + dy * my;
                    float e = expf(-0.5f * q);
                    float sa = g->alpha * e;
                    if (sa <= (1.0f / 256.0f)) continue;        /* discard: no colour, no depth write */
                    if (depth_bits) {
                        uint32_t* zb = &zbuf[(size_t)(y - y0) * W + x];
                        if (!(zq < *zb)) continue;                  /* GL_LESS */
                        *zb = zq;
                    }
                    float* d = rgba + ((size_t)y * W + x) * 4;
                    /* splat_frag.glsl:27-28: out = (a g rgb, a g); GL_ONE, GL_ONE_MINUS_SRC_ALPHA (app.cpp:153-156) */
                    float sr = sa * g->rgb[0], sg = sa * g->rgb[1], sb = sa * g->rgb[2], aa = sa;
                    if (rop == 1) { sr = clamp01(sr); sg = clamp01(sg); sb = clamp01(sb); aa = clamp01(aa); }
                    float oma = 1.0f - aa;
                    float r0 = sr + oma * d[0], r1 = sg + oma * d[1], r2 = sb + oma * d[2], r3 = aa + oma * d[3];
                    if (rop == 1) { r0 = to_unorm8(r0); r1 = to_unorm8(r1); r2 = to_unorm8(r2); r3 = to_unorm8(r3); }
                    else if (rop == 2) { r0 = round_to_half(r0); r1 = round_to_half(r1); r2 = round_to_half(r2); r3 = round_to_half(r3); }
                    d[0] = r0; d[1] = r1; d[2] = r2; d[3] = r3;
                }
            }
        }
        free(zbuf);
    }
}

void orc_composite_depth(uint32_t v, const orc_splat2d* s, int W, int H, float* rgba, int depth_bits,
                         int nthreads)
{
    orc_composite_rop(v, s, W, H, rgba, depth_bits, 0, nthreads);
}

void orc_composite_f64(uint32_t v, const orc_splat2d* s, int W, int H, double* rgba, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > H) nthreads = H;
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
    for (int band = 0; band < nthreads; ++band) {
        const int y0 = (int)(((int64_t)H * band) / nthreads);
        const int y1 = (int)(((int64_t)H * (band + 1)) / nthreads);
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W; ++x) {
                double* d = rgba + ((size_t)y * W + x) * 4;
                d[0] = 0.0; d[1] = 0.0; d[2] = 0.0; d[3] = 1.0;
            }
        for (uint32_t k = 0; k < v; ++k) {
            const orc_splat2d* g = &s[k];
            if (g->reject) continue;
            int xa, xb, ya, yb;
            pixel_range(g->px, g->hx, W, 0, W, &xa, &xb);
            pixel_range(g->py, g->hy, H, y0, y1, &ya, &yb);
            for (int y = ya; y <= yb; ++y)
                for (int x = xa; x <= xb; ++x) {
                    double dx = ((double)x + 0.5) - (double)g->px;
                    double dy = ((double)y + 0.5) - (double)g->py;
                    double mx = (double)g->inv[0] * dx + (double)g->inv[2] * dy;
                    double my = (double)g->inv[1] * dx + (double)g->inv[3] * dy;
                    double q = dx * mx + dy * my;
                    double sa = (double)g->alpha * exp(-0.5 * q);
                    if (sa <= 1.0 / 256.0) continue;
                    double* d = rgba + ((size_t)y * W + x) * 4;
                    double oma = 1.0 - sa;
                    d[0] = sa * (double)g->rgb[0] + oma * d[0];
                    d[1] = sa * (double)g->rgb[1] + oma * d[1];
                    d[2] = sa * (double)g->rgb[2] + oma * d[2];
                    d[3] = sa + oma * d[3];
                }
        }
    }
}

/* ------------------------------------------------------------------------------------- */
/* point-cloud renderer: src/pointrenderer.cpp:113-196, shader/point_*.glsl */
/* ------------------------------------------------------------------------------------- */

/* Texture preparation as the reference does it before glTexImage2D: Image::Load flips the rows (t = 0 is the
 * image's bottom row, core/image.cpp:108-111) and pre-multiplies colour by alpha in 8 bits with truncation
 * (image.cpp:144-157); a texture flagged sRGB (pointrenderer.cpp:60) decodes texels to linear on fetch; mip levels
 * from glGenerateMipmap (core/texture.cpp:76) restated as a 2x2 box filter.  Returns the number of levels;
 * off[l] = texel offset of level l in chain (float4 texels).  chain needs 4 * (4/3 w h + 16) floats. */
int orc_build_sprite(const uint8_t* rgba8_top_first, int w, int h, int srgb, float* chain, uint32_t* off)
{
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            const uint8_t* s = rgba8_top_first + ((size_t)(h - 1 - j) * w + i) * 4;
            float* o = chain + ((size_t)j * w + i) * 4;
            float alpha = (float)s[3] / 255.0f;
            for (int c = 0; c < 3; ++c) {
                uint8_t pm = (uint8_t)((((float)s[c] / 255.0f) * alpha) * 255.0f);
                float v = (float)pm / 255.0f;
                o[c] = srgb ? srgb_to_linear(v) : v;
            }
            o[3] = alpha;
        }
    int lw = w, lh = h, level = 0;
    size_t cur = 0;
    off[0] = 0;
    while ((lw > 1 || lh > 1) && level + 1 < 14) {
        int nw = lw / 2 > 1 ? lw / 2 : 1, nh = lh / 2 > 1 ? lh / 2 : 1;
        size_t next = cur + (size_t)lw * lh;
        for (int j = 0; j < nh; ++j)
            for (int i = 0; i < nw; ++i) {
                int i0 = 2 * i < lw - 1 ? 2 * i : lw - 1, i1 = 2 * i + 1 < lw - 1 ? 2 * i + 1 : lw - 1;
                int j0 = 2 * j < lh - 1 ? 2 * j : lh - 1, j1 = 2 * j + 1 < lh - 1 ? 2 * j + 1 : lh - 1;
                for (int c = 0; c < 4; ++c) {
                    float a = chain[(cur + (size_t)j0 * lw + i0) * 4 + c], b = chain[(cur + (size_t)j0 * lw + i1) * 4 + c];
                    float cc = chain[(cur + (size_t)j1 * lw + i0) * 4 + c], d = chain[(cur + (size_t)j1 * lw + i1) * 4 + c];
                    float m = (((a + b) + cc) + d) * 0.25f;
                    /* a derived level has the base level's format (GL 4.6 8.14.4): 8-bit unorm, sRGB-encoded for colour when the
                     * texture is GL_SRGB8_ALPHA8 -- the filter runs on decoded values, the result is stored at 8 bits (round to
                     * nearest; llvmpipe's own levels are within one 8-bit step of this: tests/test_reference_shaders.py) */
                    if (srgb && c < 3) {
                        float e = m <= 0.0031308f ? m * 12.92f : 1.055f * powf(m, 1.0f / 2.4f) - 0.055f;
                        m = srgb_to_linear(floorf(e * 255.0f + 0.5f) / 255.0f);
                    } else {
                        m = floorf(m * 255.0f + 0.5f) / 255.0f;
                    }
                    chain[(next + (size_t)j * nw + i) * 4 + c] = m;
                }
            }
        cur = next;
        lw = nw;
        lh = nh;
        off[++level] = (uint32_t)cur;
    }
    return level + 1;
}

/* point_vert.glsl:22-25 + point_geom.glsl:22-46 for idx[0..v): clip position, the quad's half size (an offset of
 * (pointSize * invAspectRatio, pointSize) added in CLIP space, pointrenderer.cpp:170-177), level of detail.
 * points: 8 floats per point (position.xyzw, color.rgba), pointcloud.cpp:19-23. */
void orc_points_project(uint32_t v, const uint32_t* idx, const float* points, const float viewMat[16],
                        const float projMat[16], const float viewport[4], int tex_w, int tex_h, orc_point2d* out)
{
    const float WIDTH = viewport[2], HEIGHT = viewport[3];
    for (uint32_t r = 0; r < v; ++r) {
        uint32_t i = idx ? idx[r] : r;
        const float* rec = points + (size_t)i * 8;
        orc_point2d* o = &out[r];
        float t[4], p4[4];
        mat4_mul_point(viewMat, rec[0], rec[1], rec[2], t);
        for (int k = 0; k < 4; ++k) {
            float s = M(projMat, 0, k) * t[0];
            s = s + M(projMat, 1, k) * t[1];
            s = s + M(projMat, 2, k) * t[2];
            s = s + M(projMat, 3, k) * t[3];
            p4[k] = s;
        }
        float w = p4[3];
        /* the four vertices share z and w: near/far clipping keeps or drops the whole quad */
        int reject = !(w > 0.0f) || !(p4[2] >= -w) || !(p4[2] <= w);
        float ndcx = p4[0] / w, ndcy = p4[1] / w;
        o->ndcz = p4[2] / w;
        /* viewport transform, viewport origin = image origin (app.cpp:148 glViewport(0, 0, w, h)) */
        o->cx = (ndcx + 1.0f) * (0.5f * WIDTH);
        o->cy = (ndcy + 1.0f) * (0.5f * HEIGHT);
        float invAspect = 1.0f / (WIDTH / HEIGHT);
        o->hx = ((0.02f * invAspect) / w) * (0.5f * WIDTH);
        o->hy = (0.02f / w) * (0.5f * HEIGHT);
        if (!(o->hx > 0.0f) || !(o->hy > 0.0f) || !(o->cx == o->cx) || !(o->cy == o->cy)) reject = 1;
        float rx = (float)tex_w / (2.0f * o->hx), ry = (float)tex_h / (2.0f * o->hy);
        o->lambda = log2f(rx > ry ? rx : ry);
        for (int c = 0; c < 4; ++c) o->rgba[c] = rec[4 + c];
        o->reject = reject;
        o->index = i;
    }
}

static void sprite_tap(const float* chain, uint32_t off, int sw, int sh, float u, float v, float out[4])
{
    float x = u * (float)sw - 0.5f, y = v * (float)sh - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float ax = x - xf, ay = y - yf;
    int i0 = (int)xf, i1 = (int)xf + 1, j0 = (int)yf, j1 = (int)yf + 1;
    /* ClampToEdge */
    i0 = i0 < 0 ? 0 : (i0 > sw - 1 ? sw - 1 : i0);
    i1 = i1 < 0 ? 0 : (i1 > sw - 1 ? sw - 1 : i1);
    j0 = j0 < 0 ? 0 : (j0 > sh - 1 ? sh - 1 : j0);
    j1 = j1 < 0 ? 0 : (j1 > sh - 1 ? sh - 1 : j1);
    const float* t00 = chain + ((size_t)off + (size_t)j0 * sw + i0) * 4;
    const float* t10 = chain + ((size_t)off + (size_t)j0 * sw + i1) * 4;
    const float* t01 = chain + ((size_t)off + (size_t)j1 * sw + i0) * 4;
    const float* t11 = chain + ((size_t)off + (size_t)j1 * sw + i1) * 4;
    float bx = 1.0f - ax, by = 1.0f - ay;
    for (int c = 0; c < 4; ++c) out[c] = (t00[c] * bx + t10[c] * ax) * by + (t01[c] * bx + t11[c] * ax) * ay;
}

static void sprite_sample(const float* chain, const uint32_t* off, int w, int h, int levels, float u, float v,
                          float lambda, float out[4])
{
    if (!(lambda > 0.0f)) { sprite_tap(chain, off[0], w, h, u, v, out); return; }     /* magnification: Linear */
    float lf = floorf(lambda);
    int l0 = (int)lf < levels - 1 ? (int)lf : levels - 1;
    int l1 = l0 + 1 < levels - 1 ? l0 + 1 : levels - 1;
    float a[4], b[4];
    sprite_tap(chain, off[l0], (w >> l0) > 1 ? (w >> l0) : 1, (h >> l0) > 1 ? (h >> l0) : 1, u, v, a);
    if (l1 == l0) { for (int c = 0; c < 4; ++c) out[c] = a[c]; return; }
    sprite_tap(chain, off[l1], (w >> l1) > 1 ? (w >> l1) : 1, (h >> l1) > 1 ? (h >> l1) : 1, u, v, b);
    float f = lambda - lf, g = 1.0f - f;
    for (int c = 0; c < 4; ++c) out[c] = a[c] * g + b[c] * f;
}

/* point_frag.glsl:17-25 + blend state app.cpp:153-156 in array (draw) order; coverage = pixel centres inside the
 * quad [c - h, c + h); depth_bits != 0 adds the GL_LESS depth test (no discard in this shader: every covered
 * fragment that passes writes depth, even where the sprite is transparent). */
void orc_points_composite(uint32_t v, const orc_point2d* pts, const float* chain, const uint32_t* off, int tex_w,
                          int tex_h, int levels, int W, int H, float* rgba, int depth_bits)
{
    uint32_t* zbuf = (uint32_t*)malloc((size_t)(W > 0 ? W : 1) * (H > 0 ? H : 1) * sizeof(uint32_t));
    for (int p = 0; p < W * H; ++p) {
        rgba[p * 4 + 0] = 0.0f; rgba[p * 4 + 1] = 0.0f; rgba[p * 4 + 2] = 0.0f; rgba[p * 4 + 3] = 1.0f;
        zbuf[p] = 0xFFFFFFFFu;
    }
    for (uint32_t k = 0; k < v; ++k) {
        const orc_point2d* g = &pts[k];
        if (g->reject) continue;
        uint32_t zq = depth_bits ? orc_quantise_depth(g->ndcz, depth_bits) : 0u;
        float xlo = g->cx - g->hx, xhi = g->cx + g->hx, ylo = g->cy - g->hy, yhi = g->cy + g->hy;
        int xa = (int)fmaxf(floorf(xlo - 1.0f), 0.0f), xb = (int)fminf(ceilf(xhi + 1.0f), (float)(W - 1));
        int ya = (int)fmaxf(floorf(ylo - 1.0f), 0.0f), yb = (int)fminf(ceilf(yhi + 1.0f), (float)(H - 1));
        if (!(xlo < (float)W) || !(xhi > 0.0f) || !(ylo < (float)H) || !(yhi > 0.0f)) continue;
        for (int y = ya; y <= yb; ++y)
            for (int x = xa; x <= xb; ++x) {
                float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
                if (!(fx >= xlo && fx < xhi && fy >= ylo && fy < yhi)) continue;
                size_t p = (size_t)y * W + x;
                if (depth_bits && !(zq < zbuf[p])) continue;
                zbuf[p] = zq;
                float u = (fx - xlo) / (2.0f * g->hx), vv = (fy - ylo) / (2.0f * g->hy);
                float tex[4];
                sprite_sample(chain, off, tex_w, tex_h, levels, u, vv, g->lambda, tex);
                float sa = g->rgba[3] * tex[3];
                float oma = 1.0f - sa;
                float* d = rgba + p * 4;
                d[0] = ((g->rgba[3] * g->rgba[0]) * tex[0]) + oma * d[0];
                d[1] = ((g->rgba[3] * g->rgba[1]) * tex[1]) + oma * d[1];
                d[2] = ((g->rgba[3] * g->rgba[2]) * tex[2]) + oma * d[2];
                d[3] = sa + oma * d[3];
            }
    }
    free(zbuf);
}

/* ------------------------------------------------------------------------------------- */
/* whole frame                                                                           */
/* ------------------------------------------------------------------------------------- */

uint32_t orc_render_frame(size_t n, const float* aos, size_t stride, int full_sh, int srgb,
                          const float sortCameraMat[16], const float sortProjMat[16],
                          const float renderCameraMat[16], const float renderProjMat[16],
                          const float viewport[4], const float nearFar[2],
                          float* rgba, uint32_t* sorted_idx_out, uint32_t* sorted_keys_out,
                          orc_splat2d* splats_out, int nthreads)
{
    /* Sort(): splatrenderer.cpp:161,175 */
    float modelView[16], mvp[16];
    orc_mat4_inverse(sortCameraMat, modelView);
    orc_mat4_mul(sortProjMat, modelView, mvp);
    uint32_t* keys = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* idx = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t v = orc_presort(n, aos, stride, mvp, nearFar[1], keys, idx);
    orc_sort(v, keys, idx);
    if (sorted_idx_out) memcpy(sorted_idx_out, idx, (size_t)v * sizeof(uint32_t));
    if (sorted_keys_out) memcpy(sorted_keys_out, keys, (size_t)v * sizeof(uint32_t));

    /* Render(): splatrenderer.cpp:327-335 */
    if (rgba || splats_out) {
        float viewMat[16];
        orc_mat4_inverse(renderCameraMat, viewMat);
        float eye[3] = {M(renderCameraMat, 3, 0), M(renderCameraMat, 3, 1), M(renderCameraMat, 3, 2)};
        orc_splat2d* sp = splats_out ? splats_out
                                     : (orc_splat2d*)malloc((v ? v : 1) * sizeof(orc_splat2d));
#ifdef _OPENMP
        int saved = omp_get_max_threads();
        omp_set_num_threads(nthreads > 0 ? nthreads : 1);
#endif
        orc_project(v, idx, aos, stride, full_sh, srgb, viewMat, renderProjMat, viewport, nearFar,
                    eye, sp);
#ifdef _OPENMP
        omp_set_num_threads(saved);
#endif
        if (rgba) {
            int W = (int)viewport[2], H = (int)viewport[3];
            orc_composite(v, sp, W, H, rgba, 0, H, nthreads);
        }
        if (!splats_out) free(sp);
    }
    free(keys);
    free(idx);
    return v;
}
