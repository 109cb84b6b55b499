"""Counter-based synthetic Gaussian-cloud generator.

Language independent: for splat i and channel c (< 64)
    h  = splitmix64(seed XOR (i*64 + c))
    u1 = ((h >> 40) + 0.5) / 2^24,  u2 = (((h >> 16) & 0xFFFFFF) + 0.5) / 2^24
    n  = sqrt(-2 ln u1) * cos(2 pi u2)                       (Box-Muller, one normal per hash)
channels: 0-2 position, 3-5 log-scale, 6-9 quaternion, 10 opacity logit, 11-13 f_dc, 14-58 f_rest.
The output is PLY-style raw attributes (the inputs of GaussianCloud::ImportPly's per-vertex math).
"""
import numpy as np

SEED_1M = 0x5EED1234
SEED_6M = 0x5EED6000


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _normals(seed, i0, n, channels):
    """standard normals, shape (n, len(channels)), float64"""
    idx = (np.arange(i0, i0 + n, dtype=np.uint64)[:, None] * np.uint64(64) +
           np.asarray(channels, np.uint64)[None, :])
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) ^ idx)
    u1 = ((h >> np.uint64(40)).astype(np.float64) + 0.5) / 16777216.0
    u2 = (((h >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64) + 0.5) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def generate(n, seed=SEED_1M, pos_sigma=1.5, pos_clip=6.0, log_scale_mean=-4.0, log_scale_sigma=0.6,
             opacity_mean=0.5, opacity_sigma=2.0, dc_sigma=0.8, rest_sigma=0.15, full_sh=True, chunk=1 << 16,
             workers=None):
    """returns dict(xyz, f_dc, f_rest|None, opacity, log_scale, rot) of float32 arrays.
    Chunks are independent (counter-based generator), so they are filled by a thread pool: numpy releases the
    GIL inside its loops, and the values do not depend on the chunking or on the number of workers."""
    xyz = np.empty((n, 3), np.float32)
    f_dc = np.empty((n, 3), np.float32)
    f_rest = np.empty((n, 45), np.float32) if full_sh else None
    opacity = np.empty(n, np.float32)
    log_scale = np.empty((n, 3), np.float32)
    rot = np.empty((n, 4), np.float32)

    def fill(i0):
        m = min(chunk, n - i0)
        sl = slice(i0, i0 + m)
        xyz[sl] = np.clip(_normals(seed, i0, m, [0, 1, 2]) * pos_sigma, -pos_clip, pos_clip)
        log_scale[sl] = log_scale_mean + log_scale_sigma * _normals(seed, i0, m, [3, 4, 5])
        q = _normals(seed, i0, m, [6, 7, 8, 9])
        rot[sl] = q / np.linalg.norm(q, axis=1, keepdims=True)
        opacity[sl] = opacity_mean + opacity_sigma * _normals(seed, i0, m, [10])[:, 0]
        f_dc[sl] = dc_sigma * _normals(seed, i0, m, [11, 12, 13])
        if full_sh:
            f_rest[sl] = rest_sigma * _normals(seed, i0, m, list(range(14, 59)))

    starts = list(range(0, n, chunk))
    if workers is None:
        import os
        workers = min(32, os.cpu_count() or 1)
    if workers <= 1 or len(starts) <= 1:
        for i0 in starts:
            fill(i0)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(fill, starts))
    return dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=opacity, log_scale=log_scale, rot=rot)


def make_cloud(n, seed=SEED_1M, full_sh=True, **kw):
    """GaussianCloud filled through the same per-vertex math as ImportPly (C++ side)."""
    from .scene import GaussianCloud
    a = generate(n, seed=seed, full_sh=full_sh, **kw)
    gc = GaussianCloud(GaussianCloud.Options(importFullSH=full_sh, exportFullSH=full_sh))
    ok = gc.FromAttributes(a["xyz"], a["f_dc"], a["f_rest"], a["opacity"], a["log_scale"], a["rot"])
    if not ok:
        raise RuntimeError("FromAttributes failed")
    return gc


def write_ply(path, attrs):
    """write the raw attributes as an Inria-style binary PLY (62 float properties)"""
    n = attrs["xyz"].shape[0]
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    has_rest = attrs.get("f_rest") is not None
    if has_rest:
        names += ["f_rest_%d" % i for i in range(45)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    cols = [attrs["xyz"], np.zeros((n, 3), np.float32), attrs["f_dc"]]
    if has_rest:
        cols.append(attrs["f_rest"])
    cols += [attrs["opacity"][:, None], attrs["log_scale"], attrs["rot"]]
    data = np.concatenate([np.asarray(c, np.float32) for c in cols], axis=1)
    assert data.shape[1] == len(names)
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n).encode())
        for nm in names:
            f.write(("property float %s\n" % nm).encode())
        f.write(b"end_header\n")
        f.write(np.ascontiguousarray(data, "<f4").tobytes())


# ------------------------------------------------------------------------------------------------
# A scene-LIKE cloud (round 3): what a trained 3DGS scene looks like to the renderer, which the isotropic blobs
# above do not -- positions concentrated on surfaces, heavy-tailed anisotropic scales, a small population of huge
# background splats, bimodal opacity, and cameras INSIDE the cloud (stand-in for BASELINE configs[2], the Inria
# "bicycle" scene, which is not on the box).  Same counter-based generator: channels 59..63 pick the population.
# ------------------------------------------------------------------------------------------------
def _uniforms(seed, i0, n, channels):
    idx = (np.arange(i0, i0 + n, dtype=np.uint64)[:, None] * np.uint64(64) + np.asarray(channels, np.uint64)[None, :])
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) ^ idx)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def generate_scene(n, seed=0x5CE11E, objects=48, full_sh=True, chunk=1 << 16, workers=None):
    """returns the same dict as generate().  Populations (by a per-splat uniform): 30 % ground plane y = -1, 55 % shells
    of `objects` ellipsoids inside radius 5, 14 % a far dome (radius 12-20), 1 % background splats (radius 25-40, metres
    wide: each spans a quarter of a 45-degree view and more).  log-scales ~ N(-4.3, 1.2^2) clipped to [-7.5, -1], one axis 1.5
    smaller (flat, surface-like); opacity logits 60 % N(4, 1) / 40 % N(-2, 1.5)."""
    xyz = np.empty((n, 3), np.float32)
    f_dc = np.empty((n, 3), np.float32)
    f_rest = np.empty((n, 45), np.float32) if full_sh else None
    opacity = np.empty(n, np.float32)
    log_scale = np.empty((n, 3), np.float32)
    rot = np.empty((n, 4), np.float32)
    # the objects: centres, radii (deterministic from the seed)
    on = _normals(seed ^ 0xABCDEF, 0, objects, [0, 1, 2, 3, 4, 5])
    centres = np.stack([np.clip(on[:, 0] * 2.2, -5, 5), np.clip(0.3 + 0.6 * on[:, 1], -0.7, 1.8), np.clip(on[:, 2] * 2.2, -5, 5)], 1)
    radii = 0.25 + 0.45 * np.abs(on[:, 3:6])

    def fill(i0):
        m = min(chunk, n - i0)
        sl = slice(i0, i0 + m)
        u = _uniforms(seed, i0, m, [59, 60, 61])
        g = _normals(seed, i0, m, [0, 1, 2])
        pop = u[:, 0]
        p = np.empty((m, 3))
        d = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-9)          # unit directions
        ground, obj, dome, back = pop < 0.30, (pop >= 0.30) & (pop < 0.85), (pop >= 0.85) & (pop < 0.99), pop >= 0.99
        p[ground] = np.stack([np.clip(6.0 * g[ground, 0], -15, 15), -1.0 + 0.02 * g[ground, 1], np.clip(6.0 * g[ground, 2], -15, 15)], 1)
        j = np.minimum((u[obj, 1] * objects).astype(np.int64), objects - 1)
        p[obj] = centres[j] + radii[j] * d[obj] * (1.0 + 0.01 * (u[obj, 2:3] - 0.5))
        dd = d[dome].copy(); dd[:, 1] = np.abs(dd[:, 1])
        p[dome] = dd * (12.0 + 8.0 * u[dome, 1:2])
        p[back] = d[back] * (25.0 + 15.0 * u[back, 1:2])
        xyz[sl] = p
        s = np.clip(-4.3 + 1.2 * _normals(seed, i0, m, [3]), -7.5, -1.0)
        ls = s + 0.35 * _normals(seed, i0, m, [3, 4, 5]) * np.array([0.0, 1.0, 1.0])
        ls[:, 2] -= 1.5                                                              # flat: surface-like
        ls[dome] += 1.6                                                              # far geometry is coarser
        ls[back] = 0.5 + 0.5 * _normals(seed, i0, m, [3, 4, 5])[back]
        log_scale[sl] = ls
        q = _normals(seed, i0, m, [6, 7, 8, 9])
        rot[sl] = q / np.linalg.norm(q, axis=1, keepdims=True)
        on_ = _normals(seed, i0, m, [10])[:, 0]
        opaque = u[:, 2] < 0.6
        opacity[sl] = np.where(opaque, 4.0 + on_, -2.0 + 1.5 * on_)
        f_dc[sl] = 0.8 * _normals(seed, i0, m, [11, 12, 13])
        if full_sh:
            f_rest[sl] = 0.15 * _normals(seed, i0, m, list(range(14, 59)))

    starts = list(range(0, n, chunk))
    if workers is None:
        import os
        workers = min(32, os.cpu_count() or 1)
    if workers <= 1 or len(starts) <= 1:
        for i0 in starts:
            fill(i0)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(fill, starts))
    return dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=opacity, log_scale=log_scale, rot=rot)


def scene_cameras(count=64, radius=2.0, height=0.2):
    """camera-to-world matrices (float32[16], column-major, -z forward / +y up) on a circle INSIDE the scene of
    generate_scene, each looking across the middle of the scene at the objects on the other side"""
    cams = []
    for k in range(count):
        a = 2.0 * np.pi * k / count
        pos = np.array([radius * np.cos(a), height, radius * np.sin(a)])
        target = np.array([-1.2 * np.cos(a + 0.4), 0.1, -1.2 * np.sin(a + 0.4)])
        z = pos - target
        z /= np.linalg.norm(z)
        x = np.cross([0.0, 1.0, 0.0], z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, pos
        cams.append(m.T.astype(np.float32).reshape(16).copy())
    return cams


def write_cameras_json(path, cams, width, height, fovy):
    """Inria-style cameras.json for camera-to-world matrices: CamerasConfig::ImportJson (camerasconfig.cpp:20-67) builds
    mat = [R col 0, -R col 1, -R col 2, position] from the file's rotation R, so R = [cam x, -cam y, -cam z]"""
    import json
    fy = 0.5 * height / np.tan(0.5 * fovy)
    out = []
    for k, c in enumerate(cams):
        m = np.asarray(c, np.float64).reshape(4, 4).T           # m[:, c] = column c
        R = np.stack([m[:3, 0], -m[:3, 1], -m[:3, 2]], axis=1)
        out.append({"id": k, "img_name": "synthetic_%04d" % k, "width": int(width), "height": int(height),
                    "position": [float(v) for v in m[:3, 3]], "rotation": [[float(v) for v in row] for row in R],
                    "fx": float(fy), "fy": float(fy)})
    with open(path, "w") as f:
        json.dump(out, f)
