"""CPU tests of the point-cloud surface: PointCloud (pointcloud.cpp) import/export/debug cloud, the
PNG reader that loads the sprite texture, and the oracle's restatement of PointRenderer's draw (point_*.glsl)."""
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import oracle as orc
from splatapult_amd import PointCloud, camera
from tests import scenes


def write_point_ply(path, xyz, rgb, doubles=False, with_color=True, extra=True):
    n = len(xyz)
    t = "double" if doubles else "float"
    props = ["property %s x" % t, "property %s y" % t, "property %s z" % t]
    if extra:
        props += ["property float nx", "property float ny", "property float nz"]
    if with_color:
        props += ["property uchar red", "property uchar green", "property uchar blue"]
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n%s\nend_header\n" % (n, "\n".join(props))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        for i in range(n):
            f.write(struct.pack("<3d" if doubles else "<3f", *xyz[i]))
            if extra:
                f.write(struct.pack("<3f", 0.0, 0.0, 0.0))
            if with_color:
                f.write(bytes(int(c) for c in rgb[i]))


def srgb_to_linear(s):
    s = np.asarray(s, np.float32)
    return np.where(s <= 0.04045, s / np.float32(12.92), np.power((s + np.float32(0.055)) / np.float32(1.055), np.float32(2.4))).astype(np.float32)


@pytest.mark.parametrize("doubles", [False, True])
def test_point_cloud_import_matches_reference_construction(tmp_path, doubles):
    rng = np.random.default_rng(1)
    xyz = rng.uniform(0.0, 3.0, size=(200, 3))
    rgb = rng.integers(0, 256, size=(200, 3))
    p = str(tmp_path / "input.ply")
    write_point_ply(p, xyz, rgb, doubles=doubles)
    pc = PointCloud(False)
    assert pc.ImportPly(p) and pc.GetNumPoints() == 200 and pc.GetStride() == 32 and pc.GetTotalSize() == 6400
    a = pc.as_array()
    np.testing.assert_array_equal(a[:, :3], xyz.astype(np.float32))                 # (float)double / float as is
    np.testing.assert_array_equal(a[:, 3], 1.0)
    np.testing.assert_array_equal(a[:, 4:7], rgb.astype(np.float32) / np.float32(255.0))   # pointcloud.cpp:97-99
    np.testing.assert_array_equal(a[:, 7], 1.0)
    # the reference's quirk: useLinearColors sends the POSITIONS through SRGBToLinear (pointcloud.cpp:84-95)
    pl = PointCloud(True)
    assert pl.ImportPly(p)
    np.testing.assert_allclose(pl.as_array()[:, :3], srgb_to_linear(xyz.astype(np.float32)), rtol=2e-6)
    np.testing.assert_array_equal(pl.as_array()[:, 4:7], a[:, 4:7])


def test_point_cloud_errors_missing_colour_export_roundtrip_and_debug_cloud(tmp_path):
    pc = PointCloud(False)
    assert not pc.ImportPly(str(tmp_path / "missing.ply"))
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    assert not pc.ImportPly(str(bad))
    nopos = str(tmp_path / "nopos.ply")
    with open(nopos, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\nend_header\n" + struct.pack("<2f", 1, 2))
    assert not pc.ImportPly(nopos)                                                  # missing position: fatal (:55-60)
    xyz = np.array([[0.25, -1.0, 2.0], [3.0, 4.0, -5.0]])
    nocol = str(tmp_path / "nocol.ply")
    write_point_ply(nocol, xyz, None, with_color=False)
    assert pc.ImportPly(nocol)                                                      # missing colour: logged only (:66-71)
    np.testing.assert_array_equal(pc.as_array()[:, 4:7], 0.0)
    rgb = np.array([[255, 0, 128], [1, 2, 3]])
    src = str(tmp_path / "src.ply")
    write_point_ply(src, xyz, rgb, extra=False)
    assert pc.ImportPly(src)
    out = str(tmp_path / "out.ply")
    assert pc.ExportPly(out)
    back = PointCloud(False)
    assert back.ImportPly(out)
    np.testing.assert_array_equal(back.as_array()[:, :4], pc.as_array()[:, :4])
    # colours are written as (uint8_t)(c * 255.0f): truncation (pointcloud.cpp:185-187)
    exp = np.floor(pc.as_array()[:, 4:7] * np.float32(255.0)).astype(np.float32) / np.float32(255.0)
    np.testing.assert_array_equal(back.as_array()[:, 4:7], exp)
    hdr = open(out, "rb").read(400).split(b"end_header")[0].decode()
    for name in ("x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"):
        assert ("property float %s\n" % name in hdr) or ("property uchar %s\n" % name in hdr)
    dbg = PointCloud(False)
    dbg.InitDebugCloud()                                                            # pointcloud.cpp:199-258
    a = dbg.as_array()
    assert a.shape == (15, 8)
    for axis in range(3):
        blk = a[axis * 5:(axis + 1) * 5]
        np.testing.assert_allclose(blk[:, axis], np.arange(5, dtype=np.float32) * np.float32(0.2), rtol=1e-6)
        assert (np.delete(blk[:, :3], axis, axis=1) == 0).all()
        assert (blk[:, 4 + axis] == 1).all() and (blk[:, 7] == 1).all() and (blk[:, 3] == 1).all()


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA"])
def test_png_reader_against_pillow(tmp_path, mode):
    from PIL import Image
    rng = np.random.default_rng(len(mode))
    h, w = 37, 53
    ch = {"RGBA": 4, "RGB": 3, "L": 1, "LA": 2}[mode]
    yy, xx = np.mgrid[0:h, 0:w]
    base = ((xx * 5 + yy * 3) % 256)[..., None] + rng.integers(0, 40, size=(h, w, ch))     # smooth + noise: mixed filters
    arr = (base % 256).astype(np.uint8)
    img = Image.fromarray(arr if ch > 1 else arr[..., 0], mode)
    exp = np.asarray(img.convert("RGBA"))
    for k, kw in enumerate([dict(compress_level=9, optimize=True), dict(compress_level=1), dict(compress_level=0)]):
        p = str(tmp_path / ("t%d.png" % k))
        img.save(p, **kw)
        np.testing.assert_array_equal(camera.read_image(p), exp)
    # fixed-Huffman blocks and a corrupted CRC
    raw = b"".join(b"\x00" + arr[y].tobytes() for y in range(h))
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    idat = co.compress(raw) + co.flush()
    ctype = {"RGBA": 6, "RGB": 2, "L": 0, "LA": 4}[mode]

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")
    p = str(tmp_path / "fixed.png")
    open(p, "wb").write(png)
    np.testing.assert_array_equal(camera.read_image(p), exp)
    broken = bytearray(png)
    broken[60] ^= 0xFF
    open(p, "wb").write(bytes(broken))
    with pytest.raises(IOError):
        camera.read_image(p)
    with pytest.raises(IOError):
        camera.read_image(str(tmp_path / "none.png"))


def test_png_reader_roundtrips_the_library_writer(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 1, size=(21, 34, 4)).astype(np.float32)
    p = str(tmp_path / "w.png")
    camera.write_image(p, img)
    exp = (np.clip(img[::-1], 0, 1) * 255.0 + 0.5).astype(np.uint8)
    np.testing.assert_array_equal(camera.read_image(p), exp)


def smooth_sprite(w, h, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    d = np.hypot((xx + 0.5) / w * 2 - 1, (yy + 0.5) / h * 2 - 1)
    t = np.zeros((h, w, 4), np.float64)
    t[..., 0] = 0.5 + 0.5 * np.sin(xx * 0.3)
    t[..., 1] = (yy + 0.5) / h                       # vertical gradient: catches a missing row flip
    t[..., 2] = 0.5 + 0.5 * np.cos(d * 5)
    t[..., 3] = np.clip((1 - d) * 6, 0, 1)
    t[..., :3] = np.clip(t[..., :3] + rng.normal(0, 0.02, size=(h, w, 3)), 0, 1)
    return (t * 255 + 0.5).astype(np.uint8)


def test_oracle_sprite_chain_and_single_point_geometry():
    tex = smooth_sprite(64, 48)
    L = orc.lib()
    import ctypes as C
    chain = np.zeros((64 * 48 * 4 // 3 + 64) * 4, np.float32)
    off = np.zeros(14, np.uint32)
    levels = L.orc_build_sprite(tex.ctypes.data, 64, 48, 0, chain.ctypes.data_as(C.POINTER(C.c_float)), off.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert levels == 7 and list(off[:7]) == [0, 3072, 3840, 4032, 4080, 4092, 4094]      # 64x48, 32x24, 16x12, 8x6, 4x3, 2x1, 1x1
    l0 = chain[:64 * 48 * 4].reshape(48, 64, 4)
    a = tex[::-1].astype(np.float32) / np.float32(255.0)                                   # rows flipped (image.cpp:108-111)
    np.testing.assert_array_equal(l0[..., 3], a[..., 3])
    pm = np.floor(a[..., :3] * a[..., 3:4] * np.float32(255.0)) / np.float32(255.0)        # 8-bit pre-multiply, truncating
    assert np.abs(l0[..., :3] - pm).max() <= 1.0 / 255.0 + 1e-6 and (np.abs(l0[..., :3] - pm) < 1e-6).mean() > 0.98
    l1 = chain[off[1] * 4:(off[1] + 32 * 24) * 4].reshape(24, 32, 4)
    # derived levels: 2x2 box of the level above, stored at 8 bits like the GL_RGBA8 texture's own levels (round to nearest)
    box = l0.astype(np.float64).reshape(24, 2, 32, 2, 4).mean(axis=(1, 3))
    assert np.abs(l1 - box).max() <= 0.5 / 255.0 + 1e-6
    np.testing.assert_allclose(l1 * 255.0, np.round(l1 * 255.0), atol=1e-4)
    assert (np.abs(l1 - np.floor(box * 255.0 + 0.5) / 255.0) < 1e-6).mean() > 0.999          # (float vs double box: a tie may flip)
    # one white point on the optical axis at distance d: quad half size = 0.01 * H / w pixels on both axes
    W, H, d = 640, 480, 2.0
    cam = camera.pose((0.0, 0.0, d))
    proj = camera.perspective(camera.FOVY, W / H)
    pts = np.array([[0, 0, 0, 1, 1, 1, 1, 1]], np.float32)
    fr = orc.points_frame(pts, tex, cam, proj, [0, 0, W, H], scenes.NF)
    q = fr["pts"][0]
    assert fr["V"] == 1 and not q["reject"]
    np.testing.assert_allclose([q["cx"], q["cy"]], [W / 2, H / 2], atol=1e-3)
    np.testing.assert_allclose([q["hx"], q["hy"]], [0.01 * H / d] * 2, rtol=1e-5)
    np.testing.assert_allclose(q["lambda"], np.log2(64 / (2 * 0.01 * H / d)), rtol=1e-5)
    img = fr["image"]
    cover = (img[..., :3].sum(axis=-1) > 0)
    ys, xs = np.nonzero(cover)
    assert xs.min() >= W / 2 - 3 and xs.max() <= W / 2 + 2 and ys.min() >= H / 2 - 3 and ys.max() <= H / 2 + 2
    np.testing.assert_allclose(img[..., 3], 1.0, atol=1e-6)
    # depth-tested: a farther point drawn first, a nearer one on top: the same image as without the test
    two = np.array([[0, 0, -1, 1, 1, 0, 0, 1], [0, 0, 0, 1, 0, 1, 0, 1]], np.float32)
    a = orc.points_frame(two, tex, cam, proj, [0, 0, W, H], scenes.NF)["image"]
    b = orc.points_frame(two, tex, cam, proj, [0, 0, W, H], scenes.NF, depth_bits=24)["image"]
    np.testing.assert_array_equal(a, b)


def test_oracle_reproduces_8f4_golden(golden_dir):
    """committed fixtures for the 8f-4 rows (tests/golden/make_golden.py::make_8f4): depth-tested second eye, points"""
    g = np.load(os.path.join(golden_dir, "fixtures_8f4.npz"))
    s = np.load(os.path.join(golden_dir, "synth_sh3.npz"))
    aos = orc.build_cloud(s["in_xyz"], s["in_f_dc"], s["in_f_rest"], s["in_opacity"], s["in_log_scale"], s["in_rot"], True)
    W, H = int(s["W"]), int(s["H"])
    res = orc.render_frame(aos, True, s["cam"], s["proj"], [0, 0, W, H], scenes.NF, render_cam=g["eye1"], render_proj=s["proj"],
                           want_splats=True, nthreads=4)
    np.testing.assert_allclose(res["image"], g["exp_eye1_plain"], atol=1e-6)
    for bits in (24, 32):
        np.testing.assert_allclose(orc.composite_depth(res["splats"], W, H, bits, nthreads=3), g["exp_eye1_depth%d" % bits], atol=1e-6)
    assert np.abs(g["exp_eye1_depth24"] - g["exp_eye1_plain"]).max() > 0.05          # the artifact is in the fixture
    PW, PH = int(g["PW"]), int(g["PH"])
    for srgb in (0, 1):
        for bits in (0, 24):
            img = orc.points_frame(g["points"], g["sprite"], g["pcam"], g["pproj"], [0, 0, PW, PH], scenes.NF, srgb=bool(srgb),
                                   depth_bits=bits)["image"]
            np.testing.assert_allclose(img, g["exp_points_srgb%d_depth%d" % (srgb, bits)], atol=1e-6)

