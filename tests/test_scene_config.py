"""CPU tests of the scene-config readers and the image writer: C++ CamerasConfig /
VrConfig / FindConfigFile / PresentRGBA8 / WritePNG against Python's own json / zlib as independent checks."""
import ctypes as C
import json
import os
import struct
import zlib

import numpy as np

from splatapult_amd import _capi, camera


def make_cameras_json(path, n=5, seed=0):
    rng = np.random.default_rng(seed)
    cams = []
    for i in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        cams.append({"id": i, "img_name": "img_%03d" % i, "width": 1959, "height": 1090,
                     "position": rng.normal(size=3).tolist(), "rotation": q.tolist(),
                     "fx": 1159.58 + i, "fy": 1164.66})
    json.dump(cams, open(path, "w"))
    return cams


def test_cameras_json_matches_reference_construction(tmp_path):
    p = str(tmp_path / "cameras.json")
    src = make_cameras_json(p)
    cams = camera.load_cameras_json(p)
    assert len(cams) == len(src)
    for (mat, fov), o in zip(cams, src):
        rot = np.array(o["rotation"], np.float32)               # row-major in the file
        m = mat.reshape(4, 4)                                   # m[c] = column c
        # camerasconfig.cpp:50-53: mat = [rot[0], -rot[1], -rot[2], pos] with rot[c] = file COLUMN c
        np.testing.assert_array_equal(m[0, :3], rot[:, 0])
        np.testing.assert_array_equal(m[1, :3], -rot[:, 1])
        np.testing.assert_array_equal(m[2, :3], -rot[:, 2])
        np.testing.assert_array_equal(m[3, :3], np.array(o["position"], np.float32))
        assert m[0, 3] == 0 and m[1, 3] == 0 and m[2, 3] == 0 and m[3, 3] == 1
        # both angles from fx (camerasconfig.cpp:47-48)
        fx = np.float32(o["fx"])
        np.testing.assert_allclose(fov, [2 * np.arctan(np.float32(1959) / (2 * fx)), 2 * np.arctan(np.float32(1090) / (2 * fx))],
                                   rtol=1e-6)


def test_cameras_json_errors_and_floor_plane(tmp_path):
    L = _capi.lib()
    n = C.c_uint32()
    assert L.msplat_cameras_import_json(str(tmp_path / "none.json").encode(), None, None, 0, C.byref(n)) == _capi.ERR_IO
    bad = tmp_path / "bad.json"
    bad.write_text('[{"id": 0, "position": [0, 0, 0]}]')            # missing keys -> false after logging
    assert L.msplat_cameras_import_json(str(bad).encode(), None, None, 0, C.byref(n)) == _capi.ERR_IO
    bad.write_text('[{"id": 0, "position": [0, 0')                  # truncated
    assert L.msplat_cameras_import_json(str(bad).encode(), None, None, 0, C.byref(n)) == _capi.ERR_IO
    p = str(tmp_path / "cameras.json")
    make_cameras_json(p, n=7, seed=3)
    cams = camera.load_cameras_json(p)
    ups = np.stack([m.reshape(4, 4)[1, :3] for m, _ in cams]).mean(axis=0)
    ups /= np.linalg.norm(ups)
    dist = np.mean([m.reshape(4, 4)[3, :3] @ ups for m, _ in cams])
    nrm = np.zeros(3, np.float32); pos = np.zeros(3, np.float32)
    f = C.POINTER(C.c_float)
    assert L.msplat_cameras_floor_plane(p.encode(), nrm.ctypes.data_as(f), pos.ctypes.data_as(f)) == 0
    np.testing.assert_allclose(nrm, ups, atol=1e-6)
    np.testing.assert_allclose(pos, ups * dist, atol=1e-5)


def test_vr_json_roundtrip_and_reference_fixture(golden_dir, tmp_path):
    fm = np.array(json.load(open(os.path.join(golden_dir, "test_vr.json")))["floorMat"], np.float32)   # rows
    got = camera.load_vr_json(os.path.join(golden_dir, "test_vr.json")).reshape(4, 4)                  # columns
    np.testing.assert_array_equal(got.T, fm)
    out = str(tmp_path / "x_vr.json")
    L = _capi.lib()
    assert L.msplat_vrconfig_export_json(out.encode(), got.reshape(16).ctypes.data_as(C.POINTER(C.c_float))) == 0
    back = np.array(json.load(open(out))["floorMat"], np.float32)      # valid JSON, same row-major convention
    np.testing.assert_allclose(back, fm, rtol=1e-5, atol=1e-9)


def test_find_config_file(tmp_path):
    d = tmp_path / "a" / "b" / "c"
    d.mkdir(parents=True)
    ply = d / "point_cloud.ply"
    ply.write_bytes(b"ply\n")
    assert camera.find_config_file(str(ply), "cameras.json") == ""
    (tmp_path / "a" / "cameras.json").write_text("[]")                 # grandparent: found
    assert camera.find_config_file(str(ply), "cameras.json") == str(tmp_path / "a" / "cameras.json")
    (d / "cameras.json").write_text("[]")                              # own directory wins
    assert camera.find_config_file(str(ply), "cameras.json") == str(d / "cameras.json")
    assert camera.find_config_file(str(d / "missing.ply"), "cameras.json") == ""


def read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        crc = struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        assert zlib.crc32(typ + data) & 0xFFFFFFFF == crc
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        if typ == b"IDAT":
            idat += data
        pos += 12 + n
    w, h = hdr[0], hdr[1]
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + w * 4)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 4)


def test_png_and_ppm_writer(tmp_path):
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, size=(37, 53, 4)).astype(np.float32)      # out-of-range values get clamped
    img[..., 3] = 1.0
    p = str(tmp_path / "o.png")
    camera.write_image(p, img)
    got = read_png(p)
    exp = (np.clip(img[::-1], 0, 1) * 255.0 + 0.5).astype(np.uint8)          # top row first
    np.testing.assert_array_equal(got, exp)
    camera.write_image(p, img, encode_srgb=True)
    lin = np.clip(img[::-1, :, :3], 0, None)
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
    exp_s = (np.clip(srgb, 0, 1) * 255.0 + 0.5).astype(np.uint8)
    got = read_png(p)
    assert np.abs(got[..., :3].astype(int) - exp_s.astype(int)).max() <= 1
    ppm = str(tmp_path / "o.ppm")
    camera.write_image(ppm, img)
    b = open(ppm, "rb").read()
    assert b.startswith(b"P6\n53 37\n255\n") and len(b) == len(b"P6\n53 37\n255\n") + 53 * 37 * 3
    np.testing.assert_array_equal(np.frombuffer(b[-53 * 37 * 3:], np.uint8).reshape(37, 53, 3), exp[..., :3])


def test_scene_like_generator_and_cameras_json_writer(tmp_path):
    """synthetic.generate_scene / scene_cameras / write_cameras_json (bench.py cfg3s): deterministic, chunk-independent,
    and the written cameras.json comes back through CamerasConfig::ImportJson as the matrices that were written"""
    import numpy as np
    from splatapult_amd import synthetic
    a = synthetic.generate_scene(5000, seed=11, chunk=1024, workers=1)
    b = synthetic.generate_scene(5000, seed=11, chunk=4096, workers=3)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert np.isfinite(a["xyz"]).all() and abs(np.linalg.norm(a["rot"], axis=1) - 1).max() < 1e-5
    r = np.linalg.norm(a["xyz"], axis=1)
    assert (r > 24).mean() > 0.003 and (np.abs(a["xyz"][:, 1] + 1) < 0.1).mean() > 0.2        # background shell, ground plane
    assert a["log_scale"].max() > 0.0 and a["log_scale"].min() < -7.0                           # heavy tail both ways
    cams = synthetic.scene_cameras(9)
    p = str(tmp_path / "cameras.json")
    synthetic.write_cameras_json(p, cams, 1920, 1080, camera.FOVY)
    back = camera.load_cameras_json(p)
    assert len(back) == 9
    for (m, fov), c in zip(back, cams):
        np.testing.assert_allclose(m, c, atol=1e-6)
        R = m.reshape(4, 4)[:3, :3]
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-5)                                # a rotation
