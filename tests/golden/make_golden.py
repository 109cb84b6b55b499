#!/usr/bin/env python
"""Regenerates the golden fixtures under tests/golden/ (data only: inputs + expected outputs).

    python tests/golden/make_golden.py

The reference's hot path (GLSL shaders behind an OpenGL driver) cannot run in the build container
and the reference ships no tests or golden vectors, so the expected outputs here come
from this repo's CPU oracle (oracle/msplat_oracle.c).  (Since r4 the reference's shaders DO run here -- on Mesa llvmpipe through
oracle/glref -- and the fixtures made from THEIR outputs are tests/golden/glref_*.npz, make_glref_golden.py; the oracle is pinned
against them.)  What these older fixtures pin against the real reference:
  * test.ply / test_vr.json are the reference's own data files (data/test.ply, data/test_vr.json);
  * the PLY vertex block and property offsets in test_ply_cfg1.npz are read with the reference's own
    parser (oracle/_ref/libref_ply.so, built from /root/reference/src/ply.cpp).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from splatapult_amd import camera, synthetic  # noqa: E402
from tests import scenes  # noqa: E402

PLY_NAMES = (["x", "y", "z", "opacity"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(45)] +
             ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)])


def attrs_from_ref_ply(path):
    cnt, vs, props, raw = orc.ref_ply_read(path, PLY_NAMES)
    v = raw.reshape(cnt, vs).view(np.float32)

    def c(n):
        return v[:, props[n][2] // 4].copy()
    a = dict(xyz=np.stack([c("x"), c("y"), c("z")], 1), f_dc=np.stack([c("f_dc_%d" % i) for i in range(3)], 1),
             f_rest=np.stack([c("f_rest_%d" % i) for i in range(45)], 1), opacity=c("opacity"),
             log_scale=np.stack([c("scale_%d" % i) for i in range(3)], 1),
             rot=np.stack([c("rot_%d" % i) for i in range(4)], 1))
    return a, cnt, vs, {k: list(map(int, p)) for k, p in props.items()}


def frame(aos, full_sh, cam, proj, W, H, **kw):
    res = orc.render_frame(aos, full_sh, cam, proj, [0, 0, W, H], scenes.NF, want_splats=True, **kw)
    sp = res["splats"]
    return dict(V=np.int64(res["V"]), sorted_idx=res["sorted_idx"], sorted_keys=res["sorted_keys"],
                px=sp["px"], py=sp["py"], cov=sp["cov"], inv=sp["inv"], rgb=sp["rgb"], alpha=sp["alpha"],
                reject=sp["reject"], image=res["image"])


def make_8f4():
    """SURVEY 8f-4 fixtures: depth-buffer emulation (second-eye case) and the point-cloud renderer"""
    from tests.test_points import smooth_sprite
    g = np.load(os.path.join(HERE, "synth_sh3.npz"))
    aos = orc.build_cloud(g["in_xyz"], g["in_f_dc"], g["in_f_rest"], g["in_opacity"], g["in_log_scale"], g["in_rot"], True)
    W, H = int(g["W"]), int(g["H"])
    eye1 = camera.translate_local(camera.pose((1.0, 0.0, 7.0), yaw=0.55), dx=0.3)        # drawn in eye 0's order
    res = orc.render_frame(aos, True, g["cam"], g["proj"], [0, 0, W, H], scenes.NF, render_cam=eye1, render_proj=g["proj"],
                           want_splats=True)
    rng = np.random.default_rng(404)
    pts = np.zeros((400, 8), np.float32)
    pts[:, :3] = rng.normal(0, 1.0, size=(400, 3))
    pts[:, 3] = 1.0
    pts[:, 4:7] = rng.integers(0, 256, size=(400, 3)).astype(np.float32) / np.float32(255.0)
    pts[:, 7] = 1.0
    tex = smooth_sprite(32, 24, seed=5)
    PW, PH = 160, 120
    pcam, pproj, pvp, pnf = scenes.default_view(PW, PH, z=3.0, yaw=0.2)
    out = {}
    for srgb in (0, 1):
        for bits in (0, 24):
            out["exp_points_srgb%d_depth%d" % (srgb, bits)] = orc.points_frame(pts, tex, pcam, pproj, pvp, pnf, srgb=bool(srgb),
                                                                               depth_bits=bits)["image"]
    np.savez_compressed(os.path.join(HERE, "fixtures_8f4.npz"), eye1=eye1,
                        exp_eye1_plain=res["image"], exp_eye1_depth24=orc.composite_depth(res["splats"], W, H, 24),
                        exp_eye1_depth32=orc.composite_depth(res["splats"], W, H, 32),
                        points=pts, sprite=tex, pcam=pcam, pproj=pproj, PW=PW, PH=PH, **out)


def main():
    if "--only-8f4" in sys.argv:
        make_8f4()
        return
    assert orc.ref_ply_lib() is not None, "build oracle/_ref first (make -C oracle) -- needs /root/reference"
    # ---- config 1: the reference's only fixture --------------------------------------------
    a, cnt, vs, props = attrs_from_ref_ply(os.path.join(HERE, "test.ply"))
    W, H = 640, 480
    cam = camera.camera_from_vr_json(os.path.join(HERE, "test_vr.json"))
    proj = orc.perspective(np.float32(camera.FOVY), W / H, camera.Z_NEAR, camera.Z_FAR)
    aos = orc.build_cloud(a["xyz"], a["f_dc"], a["f_rest"], a["opacity"], a["log_scale"], a["rot"], False)
    out = frame(aos, False, cam, proj, W, H)
    np.savez_compressed(os.path.join(HERE, "test_ply_cfg1.npz"), vertex_count=cnt, vertex_size=vs,
                        prop_names=np.array(sorted(props)), prop_offsets=np.array([props[k][2] for k in sorted(props)]),
                        aos_nosh=aos, aos_sh=orc.build_cloud(a["xyz"], a["f_dc"], a["f_rest"], a["opacity"],
                                                             a["log_scale"], a["rot"], True),
                        cam=cam, proj=proj, W=W, H=H, **{"exp_" + k: v for k, v in out.items()})
    # ---- synthetic scenes (seeded, small) ----------------------------------------------------
    for name, attrs, (W, H), view in (
            ("synth_sh3", synthetic.generate(2000, seed=101, log_scale_mean=-3.2), (192, 144), dict(yaw=0.3, x=1.0)),
            ("synth_hard", scenes.hard_attrs(1500, seed=202), (200, 150), dict()),
    ):
        aos = orc.build_cloud(attrs["xyz"], attrs["f_dc"], attrs["f_rest"], attrs["opacity"], attrs["log_scale"],
                              attrs["rot"], True)
        cam, proj, vp, nf = scenes.default_view(W, H, **view)
        out = frame(aos, True, cam, proj, W, H)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cam=cam, proj=proj, W=W, H=H,
                            **{"in_" + k: v for k, v in attrs.items()}, **{"exp_" + k: v for k, v in out.items()})
    # ---- generator known answers (language-independent spec) ----------------------
    g = synthetic.generate(4, seed=synthetic.SEED_1M)
    json.dump({k: np.asarray(v, np.float64).round(7).tolist() for k, v in g.items() if v is not None},
              open(os.path.join(HERE, "generator_kat.json"), "w"), indent=0)
    make_8f4()
    for f in sorted(os.listdir(HERE)):
        print("%-24s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
