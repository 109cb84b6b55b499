"""Independent numpy restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Written separately from msplat_oracle.c (vectorised, per-stage) so that the two restatements
check each other (r4: the C restatement is also pinned against the reference's shaders executed on Mesa llvmpipe, oracle/glref).

Follows: shader/presort_compute.glsl:31-57, shader/splat_vert.glsl:51-127,153-222,
shader/splat_geom.glsl:22-54, shader/splat_frag.glsl:18-42, src/app.cpp:153-160,
src/gaussiancloud.cpp:86-94,119-122,254-361.
"""
import numpy as np

F = np.float32


def col(m, c):
    return np.asarray(m, F).reshape(4, 4)[c]          # column c of a column-major float[16]


def as_rows(m):
    """column-major float[16] -> ordinary (row, col) 4x4 ndarray"""
    return np.asarray(m, F).reshape(4, 4).T.copy()


def to_colmajor(a):
    return np.asarray(a, F).T.reshape(16).copy()


def perspective(fovy, aspect, zn, zf):
    t = np.tan(F(fovy) / F(2.0), dtype=F)
    m = np.zeros((4, 4), F)          # m[col][row]
    m[0][0] = F(1.0) / (F(aspect) * t)
    m[1][1] = F(1.0) / t
    m[2][2] = -(F(zf) + F(zn)) / (F(zf) - F(zn))
    m[2][3] = F(-1.0)
    m[3][2] = -(F(2.0) * F(zf) * F(zn)) / (F(zf) - F(zn))
    return m.reshape(16)


def transform_points(m, xyz):
    """m * vec4(xyz, 1) with the canonical left-to-right, unfused fp32 order."""
    m = np.asarray(m, F).reshape(4, 4)   # m[c][r]
    x, y, z = (xyz[:, 0].astype(F), xyz[:, 1].astype(F), xyz[:, 2].astype(F))
    out = np.empty((xyz.shape[0], 4), F)
    for r in range(4):
        s = m[0][r] * x
        s = s + m[1][r] * y
        s = s + m[2][r] * z
        s = s + m[3][r]
        out[:, r] = s
    return out


def cull_keys(xyz, mvp, zfar):
    """returns (visible mask, key uint32 array)   presort_compute.glsl:38-55"""
    p = transform_points(mvp, xyz)
    depth = p[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        xx = p[:, 0] / depth
        yy = p[:, 1] / depth
        vis = (depth > 0) & (xx < F(1.5)) & (xx > F(-1.5)) & (yy < F(1.5)) & (yy > F(-1.5))
        f = (depth / F(zfar)) * F(4294967296.0)
    fq = np.where(vis, f, F(0)).astype(np.float64)
    q = np.where(fq >= 4294967296.0, 4294967295, np.floor(fq)).astype(np.uint64)
    key = (np.uint64(0xFFFFFFFF) - q).astype(np.uint32)
    return vis, key


def build_cloud(xyz, f_dc, f_rest, opacity, log_scale, rot, full_sh):
    n = xyz.shape[0]
    out = np.zeros((n, 61 if full_sh else 25), F)
    out[:, 0:3] = xyz
    out[:, 3] = F(1) / (F(1) + np.exp(-opacity.astype(F), dtype=F))
    for c, base in enumerate((4, 8, 12)):
        out[:, base] = f_dc[:, c]
        if full_sh:
            out[:, base + 1:base + 4] = f_rest[:, c * 15:c * 15 + 3]
            hi = 25 + c * 12
            out[:, hi:hi + 12] = f_rest[:, c * 15 + 3:c * 15 + 15]
    s = np.exp(log_scale.astype(F), dtype=F)
    q = rot.astype(F)
    ln = np.sqrt((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + (q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3]), dtype=F)
    q = q * (F(1) / ln)[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((n, 3, 3), F)          # R[:, row, col]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    RS = (R * s[:, None, :]) * s[:, None, :]          # (R*S)*S^T  column scaling twice
    V = np.einsum("nik,njk->nij", RS, R).astype(F)    # * R^T   (summation order not pinned)
    # stored as 3 columns
    out[:, 16:19] = V[:, :, 0]
    out[:, 19:22] = V[:, :, 1]
    out[:, 22:25] = V[:, :, 2]
    return out


_K = dict(k0=0.28209479177387814, k1=0.4886025119029199, k2=1.0925484305920792,
          k3=0.31539156525252005, k4=0.5462742152960396, k5=0.5900435899266435,
          k6=2.8906114426405543, k7=0.4570457994644658, k8=0.37317633259011546,
          k9=1.4453057213202771)


def sh_basis(v, full):
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    k = {n: F(val) for n, val in _K.items()}
    b = [np.full_like(x, k["k0"]), -k["k1"] * y, k["k1"] * z, -k["k1"] * x]
    if full:
        x2, y2, z2 = x * x, y * y, z * z
        b += [k["k2"] * y * x, -k["k2"] * y * z, k["k3"] * (F(3) * z2 - F(1)), -k["k2"] * x * z,
              k["k4"] * (x2 - y2),
              -k["k5"] * y * (F(3) * x2 - y2), k["k6"] * y * x * z, -k["k7"] * y * (F(5) * z2 - F(1)),
              k["k8"] * z * (F(5) * z2 - F(3)), -k["k7"] * x * (F(5) * z2 - F(1)), k["k9"] * z * (x2 - y2),
              -k["k5"] * x * (x2 - F(3) * y2)]
    return np.stack(b, axis=1).astype(F)


def project(aos, idx, full_sh, srgb, viewMat, projMat, viewport, nearFar, eye):
    """returns dict of per-splat arrays in idx order"""
    rec = aos[idx]
    xyz = rec[:, 0:3]
    t = transform_points(viewMat, xyz)
    P = np.asarray(projMat, F).reshape(4, 4)
    Wd, Hd = F(viewport[2]), F(viewport[3])
    SX, SY = P[0][0], P[1][1]
    tz = t[:, 2]
    tz2 = tz * tz
    jsx = -(SX * Wd) / (F(2) * tz)
    jsy = -(SY * Hd) / (F(2) * tz)
    jtx = (SX * t[:, 0] * Wd) / (F(2) * tz2)
    jty = (SY * t[:, 1] * Hd) / (F(2) * tz2)
    n = rec.shape[0]
    Jr = np.zeros((n, 2, 3), F)                     # rows 0,1 of J
    Jr[:, 0, 0] = jsx; Jr[:, 0, 2] = jtx
    Jr[:, 1, 1] = jsy; Jr[:, 1, 2] = jty
    W3 = as_rows(viewMat)[:3, :3]
    Mx = np.einsum("nij,jk->nik", Jr, W3).astype(F)          # 2x3
    Sig = np.empty((n, 3, 3), F)
    Sig[:, :, 0] = rec[:, 16:19]; Sig[:, :, 1] = rec[:, 19:22]; Sig[:, :, 2] = rec[:, 22:25]
    cov = np.einsum("nij,njk,nlk->nil", Mx, Sig, Mx).astype(F)
    a = cov[:, 0, 0] + F(0.3)
    c = cov[:, 1, 1] + F(0.3)
    b01 = cov[:, 1, 0]       # mat2 element [0][1] = column 0, row 1
    b10 = cov[:, 0, 1]
    det = a * c - b01 * b10
    inv = np.stack([c / det, -b01 / det, -b10 / det, a / det], axis=1).astype(F)
    t4 = t.copy()
    p4 = np.empty((n, 4), F)
    for r in range(4):
        s = P[0][r] * t4[:, 0]
        s = s + P[1][r] * t4[:, 1]
        s = s + P[2][r] * t4[:, 2]
        s = s + P[3][r] * t4[:, 3]
        p4[:, r] = s
    with np.errstate(divide="ignore", invalid="ignore"):
        ndc = p4[:, :3] / p4[:, 3:4]
    X0 = F(viewport[0]) * (F(0.00001) * F(nearFar[0]))
    Y0 = F(viewport[1])
    px = F(0.5) * (Wd + ndc[:, 0] * Wd + F(2) * X0)
    py = F(0.5) * (Hd + ndc[:, 1] * Hd + F(2) * Y0)
    d = xyz - np.asarray(eye, F)[None, :]
    v = d / np.sqrt((d * d).sum(axis=1, dtype=F), dtype=F)[:, None]
    B = sh_basis(v, full_sh)
    rgb = np.empty((n, 3), F)
    for ch, base in enumerate((4, 8, 12)):
        coef = rec[:, base:base + 4]
        if full_sh:
            hi = 25 + ch * 12
            coef = np.concatenate([coef, rec[:, hi:hi + 12]], axis=1)
        rgb[:, ch] = F(0.5) + (B * coef).sum(axis=1, dtype=F)
    if srgb:
        lo = rgb / F(12.92)
        with np.errstate(invalid="ignore"):
            hi_ = np.power((rgb + F(0.055)) / F(1.055), F(2.4), dtype=F)
        rgb = np.where(rgb <= F(0.04045), lo, hi_).astype(F)
    with np.errstate(invalid="ignore"):
        reject = (ndc[:, 2] < 0.25) | (np.abs(ndc[:, 0]) > 2) | (np.abs(ndc[:, 1]) > 2) | ~(ndc[:, 2] <= 1) \
            | ~(p4[:, 3] > 0)
    return dict(px=px, py=py, cov=np.stack([a, b01, b10, c], axis=1), inv=inv, rgb=rgb, alpha=rec[:, 3],
                ndc=ndc, depth=p4[:, 3], reject=reject)


def composite(pr, W, H):
    """Literal per-pixel back-to-front blend over ALL splats (O(V*W*H)); small cases only."""
    img = np.zeros((H, W, 4), F)
    img[..., 3] = 1
    ys, xs = np.meshgrid(np.arange(H, dtype=F) + F(0.5), np.arange(W, dtype=F) + F(0.5), indexing="ij")
    n = pr["px"].shape[0]
    for k in range(n):
        if pr["reject"][k]:
            continue
        dx = xs - pr["px"][k]
        dy = ys - pr["py"][k]
        i0, i1, i2, i3 = pr["inv"][k]
        q = dx * (i0 * dx + i2 * dy) + dy * (i1 * dx + i3 * dy)
        with np.errstate(over="ignore", under="ignore"):
            g = np.exp(F(-0.5) * q, dtype=F)
        sa = (pr["alpha"][k] * g).astype(F)
        m = sa > F(1.0 / 256.0)
        if not m.any():
            continue
        oma = F(1) - sa
        for ch in range(3):
            img[..., ch] = np.where(m, sa * pr["rgb"][k, ch] + oma * img[..., ch], img[..., ch])
        img[..., 3] = np.where(m, sa + oma * img[..., 3], img[..., 3])
    return img
