"""Scenes of the reference-shader fixtures (tests/golden/glref_*.npz): shared by the generator (tests/golden/make_glref_golden.py,
which runs the reference's shaders on Mesa llvmpipe in the build container) and by the GPU test that compares the HIP path with
those outputs (tests/test_gpu_reference_shaders.py).  Clouds come from the seeded, language-independent generator; the fixture
stores a digest of the records it was made from."""
import hashlib
import os

import numpy as np

from splatapult_amd import camera
from tests import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def digest(aos):
    return hashlib.sha256(np.ascontiguousarray(aos, np.float32).tobytes()).hexdigest()


def _cfg1():
    g = np.load(os.path.join(GOLDEN, "test_ply_cfg1.npz"))
    return g["aos_nosh"], g["cam"], g["proj"]


def cases():
    """name -> dict(aos, full_sh, srgb, cam, proj, W, H, nf, render_cam, render_proj)"""
    out = {}
    aos, cam, proj = _cfg1()
    out["cfg1_test_ply_nosh"] = dict(aos=aos, full_sh=False, srgb=False, cam=cam, proj=proj, W=640, H=480)
    c = scenes.synth_cloud(3000, 31, log_scale_mean=-3.0)
    cam, proj, _, _ = scenes.default_view(256, 144, yaw=0.0)
    out["synth_sh3"] = dict(aos=c.as_array(), full_sh=True, srgb=False, cam=cam, proj=proj, W=256, H=144)
    c = scenes.synth_cloud(8000, 33, full_sh=False, log_scale_mean=-3.0)
    cam, proj, _, _ = scenes.default_view(256, 144, yaw=2.2, z=1.0)
    out["synth_sh0_inside"] = dict(aos=c.as_array(), full_sh=False, srgb=False, cam=cam, proj=proj, W=256, H=144)
    c = scenes.cloud_from_attrs(scenes.hard_attrs(3000, 11))
    cam, proj, _, _ = scenes.default_view(256, 192, yaw=0.7)
    out["hard_cases"] = dict(aos=c.as_array(), full_sh=True, srgb=False, cam=cam, proj=proj, W=256, H=192)
    c = scenes.synth_cloud(4000, 35, log_scale_mean=-3.2)
    cam, proj, _, _ = scenes.default_view(256, 144, yaw=0.3)
    out["srgb_define"] = dict(aos=c.as_array(), full_sh=True, srgb=True, cam=cam, proj=proj, W=256, H=144)
    c = scenes.synth_cloud(6000, 36, log_scale_mean=-3.1)
    cam0 = camera.pose((0.0, 0.0, 6.0))
    eyes = [camera.translate_local(cam0, dx=-0.032), camera.translate_local(cam0, dx=+0.032)]
    projs = [camera.create_projection(-1.0, 0.8, 0.95, -0.95), camera.create_projection(-0.8, 1.0, 0.95, -0.95)]
    out["second_eye"] = dict(aos=c.as_array(), full_sh=True, srgb=False, cam=eyes[0], proj=projs[0], W=216, H=240,
                             render_cam=eyes[1], render_proj=projs[1])
    for v in out.values():
        v.setdefault("render_cam", None)
        v.setdefault("render_proj", None)
        v["nf"] = list(scenes.NF)
    return out


# BASELINE configs[1] -- the configuration the metric is quoted on: 1 M synthetic splats, SH3, 1920x1080, camera at (0, 0, 7).  The
# reference's shaders render the whole frame (3 s on llvmpipe); a full float image is 25 MB, so the fixture keeps a 512 x 256 window
# around the centre, an 8x8 box-mean of the whole frame, and digests of the shader's keys / indices.
CFG2_WINDOW = (412, 668, 704, 1216)          # rows y0:y1, columns x0:x1 (GL convention: row 0 = bottom)


def config2():
    from splatapult_amd import synthetic
    cloud = synthetic.make_cloud(1_000_000, seed=0x5EED1234, full_sh=True, pos_sigma=1.5)
    cam, proj, _, _ = scenes.default_view(1920, 1080, z=7.0)
    return dict(aos=cloud.as_array(), full_sh=True, srgb=False, cam=cam, proj=proj, W=1920, H=1080, nf=list(scenes.NF),
                render_cam=None, render_proj=None)


def box_mean8(rgb):
    H, W = rgb.shape[0] // 8 * 8, rgb.shape[1] // 8 * 8
    return rgb[:H, :W].astype(np.float64).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3)).astype(np.float32)


def u32_digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.uint32).tobytes()).hexdigest()


# ---- PointRenderer: shader/point_{vert,geom,frag}.glsl + the sprite texture, on llvmpipe ----
def point_sprite(size=64):
    from tests.test_points import smooth_sprite
    return smooth_sprite(size, size, seed=9)          # 2^k x 2^k: every mip reduction is an exact 2x2 box in any GL


def point_cases():
    """name -> dict(points (N, 8), sprite (h, w, 4) uint8 top row first, cam, proj, W, H, nf, srgb, depth_bits, kind)
    kind "magnified": every sprite is larger on screen than the texture (GL_LINEAR only: no level of detail involved);
    kind "minified": sprites of 1-6 pixels (LinearMipmapLinear between deep levels)"""
    out = {}
    rng = np.random.default_rng(77)
    W, H = 256, 192
    proj = camera.perspective(camera.FOVY, W / H)
    # an 8 x 8 sprite drawn 10-19 pixels wide (w = 0.2 ... 0.4: half size 0.01 H / w = 4.8 ... 9.6 pixels)
    near = np.zeros((60, 8), np.float32)
    near[:, :3] = rng.uniform(-1.0, 1.0, size=(60, 3)) * np.array([0.16, 0.12, 0.1])
    near[:, 3] = 1.0
    near[:, 4:7] = rng.integers(64, 256, size=(60, 3)).astype(np.float32) / np.float32(255.0)
    near[:, 7] = rng.integers(128, 256, size=60).astype(np.float32) / np.float32(255.0)
    out["points_magnified"] = dict(points=near, cam=camera.pose((0.0, 0.0, 0.3)), proj=proj, W=W, H=H, srgb=False, depth_bits=0,
                                   kind="magnified", sprite=point_sprite(8))
    far = np.zeros((600, 8), np.float32)
    far[:, :3] = rng.normal(0, 1.0, size=(600, 3))
    far[:, 3] = 1.0
    far[:, 4:7] = rng.integers(0, 256, size=(600, 3)).astype(np.float32) / np.float32(255.0)
    far[:, 7] = 1.0
    out["points_minified"] = dict(points=far, cam=camera.pose((0.2, 0.1, 2.6), yaw=0.2), proj=proj, W=W, H=H, srgb=False,
                                  depth_bits=0, kind="minified")
    out["points_minified_srgb_depth24"] = dict(points=far, cam=camera.pose((0.2, 0.1, 2.6), yaw=0.2), proj=proj, W=W, H=H, srgb=True,
                                               depth_bits=24, kind="minified")
    for v in out.values():
        v.setdefault("sprite", point_sprite())
        v["nf"] = list(scenes.NF)
    return out


def point_edge_mask(pts, W, H, eps=1.0 / 400.0):
    """pixels whose centre lies within eps of an edge of some sprite's quad: GL snaps vertices to a sub-pixel grid (llvmpipe: 1 / 256
    pixel) before the top-left rule decides such a pixel; the oracle and the HIP path test the unsnapped interval [c - h, c + h)"""
    m = np.zeros((H, W), bool)
    for p in pts:
        if p["reject"]:
            continue
        x0, x1, y0, y1 = p["cx"] - p["hx"], p["cx"] + p["hx"], p["cy"] - p["hy"], p["cy"] + p["hy"]
        xs = [int(np.floor(e - 0.5 + k)) for e in (x0, x1) for k in (0, 1)]
        ys = [int(np.floor(e - 0.5 + k)) for e in (y0, y1) for k in (0, 1)]
        ya, yb = max(0, int(np.floor(y0)) - 1), min(H, int(np.ceil(y1)) + 1)
        xa, xb = max(0, int(np.floor(x0)) - 1), min(W, int(np.ceil(x1)) + 1)
        for x in xs:
            if 0 <= x < W and min(abs(x + 0.5 - x0), abs(x + 0.5 - x1)) <= eps:
                m[ya:yb, x] = True
        for y in ys:
            if 0 <= y < H and min(abs(y + 0.5 - y0), abs(y + 0.5 - y1)) <= eps:
                m[y, xa:xb] = True
    return m


# llvmpipe's level of detail is 0.045 below log2(rho) of the GL specification's formula (measured: with that offset its trilinear
# results agree with the oracle's to one 8-bit step of the mip levels; tests/test_reference_shaders.py).  GL leaves rho's
# approximation and the mip levels' rounding to the implementation, so for minified sprites the comparison with ANY driver has
# this form: exact geometry and coverage, colours within the two levels' difference times the LOD error.
POINT_MINIFIED_TOL = 0.03
