"""CPU tests of the host side of the boundary: Ply / GaussianCloud re-implementation against the
REAL reference parser (oracle/_ref, built from /root/reference/src/ply.cpp) and the oracle's
load-time math, plus the matrix helpers and the synthetic generator."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from splatapult_amd import GaussianCloud, _capi, camera, synthetic
from tests.golden.make_golden import PLY_NAMES

import ctypes as C

needs_ref = pytest.mark.skipif(orc.ref_ply_lib() is None, reason="oracle/_ref not built (no reference checkout)")


def write(path, header_lines, payload=b""):
    with open(path, "wb") as f:
        f.write(("\n".join(header_lines) + "\n").encode())
        f.write(payload)


def our_parse(path, full_sh=True):
    gc = GaussianCloud(GaussianCloud.Options(full_sh, full_sh))
    return gc.ImportPly(path), gc


# ---- Ply parser vs the reference's own parser -------------------------------------------------

@needs_ref
def test_test_ply_offsets_match_reference_parser(golden_dir):
    cnt, vs, props, raw = orc.ref_ply_read(os.path.join(golden_dir, "test.ply"), PLY_NAMES + ["nx", "ny", "nz"])
    assert cnt == 16 and vs == 248
    assert props["x"][2] == 0 and props["f_dc_0"][2] == 24 and props["f_rest_44"][2] == 212
    assert props["opacity"][2] == 216 and props["scale_0"][2] == 220 and props["rot_0"][2] == 232
    g = np.load(os.path.join(golden_dir, "test_ply_cfg1.npz"))
    assert int(g["vertex_count"]) == cnt and int(g["vertex_size"]) == vs


@needs_ref
def test_import_ply_equals_reference_parse_plus_oracle_math(golden_dir, tmp_path):
    """ImportPly == (reference Ply::Parse) o (oracle restatement of gaussiancloud.cpp:254-361), bit for bit"""
    a = synthetic.generate(300, seed=5)
    path = str(tmp_path / "s.ply")
    synthetic.write_ply(path, a)
    for p, full in ((os.path.join(golden_dir, "test.ply"), True), (path, True), (path, False)):
        cnt, vs, props, raw = orc.ref_ply_read(p, PLY_NAMES)
        v = raw.reshape(cnt, vs).view(np.float32)

        def c(n):
            return v[:, props[n][2] // 4]
        exp = orc.build_cloud(np.stack([c("x"), c("y"), c("z")], 1), np.stack([c("f_dc_%d" % i) for i in range(3)], 1),
                              np.stack([c("f_rest_%d" % i) for i in range(45)], 1), c("opacity"),
                              np.stack([c("scale_%d" % i) for i in range(3)], 1),
                              np.stack([c("rot_%d" % i) for i in range(4)], 1), full)
        ok, gc = our_parse(p, full)
        assert ok and gc.GetNumGaussians() == cnt and gc.HasFullSH() == full
        assert gc.GetStride() == (244 if full else 100) and gc.GetTotalSize() == cnt * gc.GetStride()
        np.testing.assert_array_equal(gc.as_array(), exp)


@needs_ref
def test_header_variants_accept_reject_like_reference(tmp_path):
    """comments, type aliases, odd property sets, malformed headers: same verdict as the reference parser"""
    pay = np.arange(12, dtype="<f4").tobytes()
    cases = {
        "ok_comments": (["ply", "comment hello", "format binary_little_endian 1.0", "comment x", "element vertex 3",
                         "property float x", "comment mid", "property float32 y", "property float z",
                         "property float opacity", "end_header"], pay),
        "ok_mixed_types": (["ply", "format binary_little_endian 1.0", "element vertex 1", "property uchar a",
                            "property int16 b", "property double c", "property uint d", "property float x",
                            "end_header"], bytes(19)),
        "bad_magic": (["plx", "format binary_little_endian 1.0", "element vertex 0", "end_header"], b""),
        "big_endian": (["ply", "format binary_big_endian 1.0", "element vertex 0", "end_header"], b""),
        "ascii": (["ply", "format ascii 1.0", "element vertex 0", "end_header"], b""),
        "no_element": (["ply", "format binary_little_endian 1.0", "property float x", "end_header"], b""),
        "element_face": (["ply", "format binary_little_endian 1.0", "element face 3", "end_header"], b""),
        "list_property": (["ply", "format binary_little_endian 1.0", "element vertex 1",
                           "property list uchar int vertex_indices", "end_header"], b""),
        "bad_type": (["ply", "format binary_little_endian 1.0", "element vertex 1", "property half x", "end_header"], b""),
        "not_property": (["ply", "format binary_little_endian 1.0", "element vertex 1", "element face 2",
                          "end_header"], b""),
        "truncated_header": (["ply", "format binary_little_endian 1.0", "element vertex 1", "property float x"], b""),
        "empty": ([], b""),
        "zero_vertices": (["ply", "format binary_little_endian 1.0", "element vertex 0", "property float x",
                           "end_header"], b""),
    }
    L = _capi.lib()
    for name, (hdr, payload) in cases.items():
        p = str(tmp_path / (name + ".ply"))
        write(p, hdr, payload)
        ref = orc.ref_ply_read(p, ["x", "y", "z", "a", "b", "c", "d", "opacity"])
        ok, gc = our_parse(p)
        assert ok == (ref is not None), name
        if ref is not None:
            assert gc.GetNumGaussians() == ref[0], name
    # offsets of the mixed-type header accumulate exactly like the reference's (ply.cpp:106-112)
    ref = orc.ref_ply_read(str(tmp_path / "ok_mixed_types.ply"), ["a", "b", "c", "d", "x"])
    assert {k: v[2] for k, v in ref[2].items()} == {"a": 0, "b": 1, "c": 3, "d": 11, "x": 15} and ref[1] == 19
    del L


def test_missing_file_and_missing_f_rest(tmp_path):
    ok, _ = our_parse(str(tmp_path / "nope.ply"))
    assert not ok                                              # gaussiancloud.cpp:143-147: false after logging
    a = synthetic.generate(10, seed=1, full_sh=False)
    p = str(tmp_path / "nosh.ply")
    synthetic.write_ply(p, a)
    ok, gc = our_parse(p, full_sh=True)                        # f_rest missing -> silently SH0 (:188-205)
    assert ok and not gc.HasFullSH() and gc.GetStride() == 100
    arr = gc.as_array()
    np.testing.assert_array_equal(arr[:, 5:8], 0)              # r_sh0 = (f_dc_0, 0, 0, 0)  (:316-332)


def test_attribute_offsets_are_the_reference_layout(golden_dir):
    ok, gc = our_parse(os.path.join(golden_dir, "test.ply"), True)
    off = gc.GetAttribOffsets()
    got = [getattr(off, n) for n, _ in _capi.AttrOffsets._fields_]
    # BaseGaussianData / FullGaussianData, gaussiancloud.cpp:32-56
    assert got == [0, 16, 32, 48, 64, 76, 88, 100, 116, 132, 148, 164, 180, 196, 212, 228]


def test_config1_cloud_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "test_ply_cfg1.npz"))
    ok, gc = our_parse(os.path.join(golden_dir, "test.ply"), False)
    np.testing.assert_array_equal(gc.as_array(), g["aos_nosh"])
    ok, gc = our_parse(os.path.join(golden_dir, "test.ply"), True)
    np.testing.assert_array_equal(gc.as_array(), g["aos_sh"])
    # sigma^2 = 0.0025, alpha = 1, f_dc = +-1.77245
    arr = g["aos_nosh"]
    np.testing.assert_allclose(arr[:, [16, 20, 24]], 0.0025, rtol=1e-5)
    assert (arr[:, 3] == 1.0).all()


def test_export_roundtrip(tmp_path):
    cloud = synthetic.make_cloud(200, seed=9)
    p = str(tmp_path / "out.ply")
    assert cloud.ExportPly(p)
    ok, back = our_parse(p, True)
    assert ok and back.GetNumGaussians() == 200 and back.HasFullSH()
    a, b = cloud.as_array(), back.as_array()
    np.testing.assert_allclose(b[:, :3], a[:, :3], rtol=0, atol=0)             # positions
    np.testing.assert_allclose(b[:, 3], a[:, 3], rtol=2e-5)                    # alpha through logit/sigmoid
    np.testing.assert_array_equal(b[:, 4:16], a[:, 4:16])                      # SH copied coefficient by coefficient
    np.testing.assert_array_equal(b[:, 25:61], a[:, 25:61])
    np.testing.assert_allclose(b[:, 16:25], a[:, 16:25], rtol=2e-3, atol=1e-7)  # covariance via eigen-decomposition


def test_debug_cloud_and_prune():
    gc = GaussianCloud()
    gc.InitDebugCloud()
    assert gc.GetNumGaussians() == 16 and gc.GetStride() == 244 and not gc.HasFullSH()   # gaussiancloud.cpp:509-511
    a = gc.as_array()
    np.testing.assert_allclose(a[:5, 0], [0.2, 0.4, 0.6, 0.8, 1.0], rtol=1e-6)
    assert (a[:, 3] == 1).all() and np.allclose(a[:, 16], 0.005)
    gc.PruneSplats([0.0, 0.0, 0.0], 4)
    assert gc.GetNumGaussians() == 4
    d = np.linalg.norm(gc.as_array()[:, :3], axis=1)
    assert (np.diff(d) >= 0).all() and d[0] == 0.0
    gc.PruneSplats([0, 0, 0], 100)                 # no-op when asking for more than there are
    assert gc.GetNumGaussians() == 4


def test_host_matrices_equal_oracle_bit_for_bit():
    L = _capi.lib()
    rng = np.random.default_rng(4)
    f = C.POINTER(C.c_float)
    for _ in range(50):
        m = camera.pose(rng.normal(size=3) * 5, rng.uniform(-3, 3), rng.uniform(-1, 1))
        out = np.zeros(16, np.float32)
        L.msplat_mat4_inverse(m.ctypes.data_as(f), out.ctypes.data_as(f))
        np.testing.assert_array_equal(out, orc.mat4_inverse(m))
        p = camera.perspective(camera.FOVY, rng.uniform(0.5, 2.5))
        out2 = np.zeros(16, np.float32)
        L.msplat_mat4_mul(p.ctypes.data_as(f), out.ctypes.data_as(f), out2.ctypes.data_as(f))
        np.testing.assert_array_equal(out2, orc.mat4_mul(p, out))
    np.testing.assert_array_equal(camera.perspective(camera.FOVY, 16 / 9),
                                  orc.perspective(np.float32(camera.FOVY), 16 / 9, 0.1, 1000.0))
    np.testing.assert_array_equal(camera.create_projection(-1.0, 0.8, 0.95, -0.95),
                                  orc.create_projection(-1.0, 0.8, 0.95, -0.95, 0.1, 1000.0))


def test_vr_json_camera(golden_dir):
    cam = camera.camera_from_vr_json(os.path.join(golden_dir, "test_vr.json"))
    np.testing.assert_allclose(cam[12:15], [-1.0419, -0.4425, -0.9787], atol=2e-4)
    assert cam[15] == 1.0


def test_generator_known_answers(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "generator_kat.json")))
    g = synthetic.generate(4, seed=synthetic.SEED_1M)
    for k, v in kat.items():
        np.testing.assert_allclose(np.asarray(g[k], np.float64), np.array(v), atol=2e-7)
    # counter-based: any sub-range reproduces the same splats
    big = synthetic.generate(1000, seed=77, chunk=128)
    again = synthetic.generate(1000, seed=77, chunk=1000)
    for k in big:
        np.testing.assert_array_equal(big[k], again[k])
    assert abs(float(np.linalg.norm(big["rot"], axis=1).mean()) - 1.0) < 1e-6
    assert np.abs(big["xyz"]).max() <= 6.0


def test_hostile_inputs_fail_without_throwing_across_the_c_abi(tmp_path):
    """a garbled vertex count, a decompression bomb and a deeply nested JSON must come back as error
    codes (the C ABI never throws; a C or ctypes caller would otherwise die in std::terminate)"""
    import struct
    import zlib
    import ctypes as C
    from splatapult_amd import GaussianCloud, PointCloud, _capi, camera
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %s\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
    for count in ("9223372036854775807", "4611686018427387904", "1152921504606846976"):       # overflow / bad_alloc
        p = tmp_path / ("huge_%s.ply" % count[:4])
        p.write_bytes((hdr % count).encode() + b"\0" * 36)
        assert not GaussianCloud().ImportPly(str(p))
        assert not PointCloud(False).ImportPly(str(p))
    # a short file is still accepted like the reference does (tail zero-filled, ply.cpp:80-84): 3 vertices announced, 1 present
    p = tmp_path / "short.ply"
    full = ("ply\nformat binary_little_endian 1.0\nelement vertex 3\n" + "".join(
        "property float %s\n" % n for n in ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1",
                                            "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]) + "end_header\n").encode()
    p.write_bytes(full + struct.pack("<14f", 1, 2, 3, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0))
    gc = GaussianCloud(GaussianCloud.Options(False, False))
    assert gc.ImportPly(str(p)) and gc.GetNumGaussians() == 3
    # PNG whose IDAT inflates to 64 MB behind a 4x4 IHDR: rejected as soon as the stream outgrows 4 * (1 + 4*4) bytes

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    bomb = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(b"\0" * (64 << 20), 9)) + chunk(b"IEND", b""))
    pb = tmp_path / "bomb.png"
    pb.write_bytes(bomb)
    w, h = C.c_uint32(), C.c_uint32()
    out = np.zeros((4, 4, 4), np.uint8)
    assert _capi.lib().msplat_read_image(str(pb).encode(), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)) != _capi.OK
    # JSON nested 10 000 deep: an error, not a stack overflow
    pj = tmp_path / "cameras.json"
    pj.write_text("[" * 10000 + "]" * 10000)
    with pytest.raises(Exception):
        camera.load_cameras_json(str(pj))


def test_host_library_is_clean_under_asan_ubsan(tmp_path):
    """the PLY / JSON / PNG parsers behind the never-throwing C ABI, built with -fsanitize=address,undefined and
    driven with the golden files + hostile inputs (tests/sanitize/host_sanitize_driver.cpp; `make sanitize`)"""
    import shutil
    import subprocess
    from tests.conftest import ROOT
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    if shutil.which("g++") is None or subprocess.run(["g++", "-fsanitize=address,undefined", str(probe), "-o", str(tmp_path / "probe")],
                                                     capture_output=True).returncode != 0:
        pytest.skip("g++ with libasan / libubsan is not available")
    p = subprocess.run(["make", "-C", ROOT, "sanitize", "SAN=" + str(tmp_path / "host_sanitize")], capture_output=True, text=True,
                       errors="replace")
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "0 failed expectation(s)" in p.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail


def _covariance_by_rodrigues(rot, log_scale):
    """Sigma = sum_i s_i^2 (R e_i)(R e_i)^T with R e_i from Rodrigues' rotation formula about the quaternion's axis, in float64 --
    no quaternion-to-matrix closed form, no matrix products: independent of the restatements of glm's mat3_cast it checks"""
    q = np.asarray(rot, np.float64)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)                 # glm::normalize (gaussiancloud.cpp:88-89); rot = (w, x, y, z)
    w, v = q[:, 0], q[:, 1:]
    sin_half = np.linalg.norm(v, axis=1)
    angle = 2.0 * np.arctan2(sin_half, w)
    axis = v / np.maximum(sin_half, 1e-300)[:, None]
    s2 = np.exp(np.asarray(log_scale, np.float64)) ** 2              # scale = exp(log_scale) (gaussiancloud.cpp:254-361)
    sigma = np.zeros((q.shape[0], 3, 3))
    for i in range(3):
        e = np.zeros((q.shape[0], 3)); e[:, i] = 1.0
        c, s = np.cos(angle)[:, None], np.sin(angle)[:, None]
        re = e * c + np.cross(axis, e) * s + axis * (axis * e).sum(1, keepdims=True) * (1.0 - c)
        sigma += s2[:, i, None, None] * re[:, :, None] * re[:, None, :]
    return sigma


def test_covariance_of_random_quaternions_by_rodrigues_rotation():
    """GaussianCloud::ImportPly's load math (a-2: R S S^T R^T from a quaternion and log-scales,
    gaussiancloud.cpp:86-94,349) checked by a PROPERTY that does not share the restatement's formulas: rotate the basis vectors
    about the quaternion's axis (Rodrigues) and sum s_i^2 (R e_i)(R e_i)^T.  Random non-identity, non-normalised quaternions,
    anisotropic scales over four decades; msplat_cloud_from_attributes must agree to fp32 rounding of the largest entry."""
    from tests import scenes
    rng = np.random.default_rng(77)
    n = 4000
    a = synthetic.generate(n, seed=5, full_sh=False)
    a["rot"] = (rng.normal(size=(n, 4)) * rng.uniform(0.2, 5.0, size=(n, 1))).astype(np.float32)      # not unit length
    a["rot"][:8] = np.array([[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [1, 1, 0, 0], [1, 0, 1, 0], [1, 0, 0, 1],
                             [1, 1, 1, 1], [-1, 2, -3, 4]], np.float32)                                # half turns, 90 and 120 degrees
    a["log_scale"] = rng.uniform(-9.0, 0.5, size=(n, 3)).astype(np.float32)
    cloud = scenes.cloud_from_attrs(a, full_sh=False)
    aos = cloud.as_array()
    got = aos[:, 16:25].reshape(n, 3, 3).transpose(0, 2, 1).astype(np.float64)      # stored column-major: [col][row]
    want = _covariance_by_rodrigues(a["rot"], a["log_scale"])
    scale = np.abs(want).max(axis=(1, 2), keepdims=True)
    err = (np.abs(got - want) / scale).max()
    print("covariance vs Rodrigues construction: worst error %.3g of the splat's largest entry" % err)
    assert err < 4e-6                                                # a few fp32 roundings of a product of three matrices
    np.testing.assert_allclose(got, got.transpose(0, 2, 1), rtol=0, atol=float(scale.max()) * 1e-6)   # symmetric
