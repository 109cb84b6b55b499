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
