#!/usr/bin/env python
"""Writes tests/golden/glref_<case>.npz: the outputs of the REFERENCE'S OWN SHADERS for the scenes of tests/glref_cases.py.

Run in the build container only (needs /root/reference/shader and Mesa's software rasteriser):
    python tests/golden/make_glref_golden.py
oracle/glref compiles /root/reference/shader/presort_compute.glsl and splat_{vert,geom,frag}.glsl where they lie and executes
them on llvmpipe with the reference's GL state (oracle/glref/glref.c cites every line it mirrors).  A fixture holds DATA only:
  digest .............. sha-256 of the cloud's records (the scene is regenerated from its seed by tests/glref_cases.py)
  keys, idx ........... presort_compute.glsl: 32-bit depth keys and indices of the visible splats, by ascending index
  draw_order .......... the element buffer the image was drawn with (ascending key, ties by index)
  rgb ................. splat_vert + splat_geom + splat_frag through GL's rasteriser and blender, RGBA32F target: (H, W, 3) float32
                        (alpha is 1 everywhere -- asserted here -- and not stored)
  gl_version .......... the GL implementation that executed the shaders
glref_cfg2_1m_1080p.npz is BASELINE configs[1] itself (1 M splats, SH3, 1920x1080): sha-256 digests of the shader's keys / indices /
the draw order, a 512 x 256 window of its image and 8x8 box means of the whole frame (a full float image would be 25 MB).
glref_points.npz: the point-cloud renderer's shaders (point_vert / point_geom / point_frag + the sprite texture) for the scenes of
glref_cases.point_cases(): image and draw order per scene.
No reference source or shader text is stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import glref, oracle as orc          # noqa: E402
from tests import glref_cases                    # noqa: E402


def main():
    assert glref.available(), "needs /root/reference/shader and oracle/_ref/libglref.so"
    for name, c in glref_cases.cases().items():
        version = glref.init(c["full_sh"], c["srgb"])
        vp = [0, 0, c["W"], c["H"]]
        mvp = orc.mat4_mul(c["proj"], orc.mat4_inverse(c["cam"]))
        keys, idx = glref.presort(c["aos"], mvp, c["nf"])
        sk, si = orc.sort(keys, idx)             # a stable (key, index) sort of the SHADER's keys: the draw order
        rcam = c["cam"] if c["render_cam"] is None else c["render_cam"]
        rproj = c["proj"] if c["render_proj"] is None else c["render_proj"]
        eye = np.asarray(rcam, np.float32).reshape(16)[12:15].copy()
        img = glref.render(c["aos"], si, orc.mat4_inverse(rcam), rproj, vp, c["nf"], eye)
        assert (img[..., 3] == 1.0).all()
        path = os.path.join(glref_cases.GOLDEN, "glref_%s.npz" % name)
        np.savez_compressed(path, digest=glref_cases.digest(c["aos"]), keys=keys, idx=idx, draw_order=si, rgb=img[..., :3].copy(),
                            gl_version=version)
        print("%-24s V %6d  lit %6d  %7.0f KB  (%s)" % (name, keys.shape[0], int((img[..., :3].sum(-1) != 0).sum()),
                                                         os.path.getsize(path) / 1024.0, version))

    # BASELINE configs[1], whole frame through the reference's shaders; window + 8x8 box means + digests are kept
    c = glref_cases.config2()
    version = glref.init(True, False)
    vp = [0, 0, c["W"], c["H"]]
    mvp = orc.mat4_mul(c["proj"], orc.mat4_inverse(c["cam"]))
    keys, idx = glref.presort(c["aos"], mvp, c["nf"])
    sk, si = orc.sort(keys, idx)
    eye = np.asarray(c["cam"], np.float32).reshape(16)[12:15].copy()
    img = glref.render(c["aos"], si, orc.mat4_inverse(c["cam"]), c["proj"], vp, c["nf"], eye)
    assert (img[..., 3] == 1.0).all()
    y0, y1, x0, x1 = glref_cases.CFG2_WINDOW
    path = os.path.join(glref_cases.GOLDEN, "glref_cfg2_1m_1080p.npz")
    np.savez_compressed(path, digest=glref_cases.digest(c["aos"]), V=keys.shape[0], keys_digest=glref_cases.u32_digest(keys),
                        idx_digest=glref_cases.u32_digest(idx), order_digest=glref_cases.u32_digest(si),
                        window=img[y0:y1, x0:x1, :3].copy(), mean8=glref_cases.box_mean8(img[..., :3]), gl_version=version)
    print("%-24s V %6d  lit %6d  %7.0f KB" % ("cfg2_1m_1080p", keys.shape[0], int((img[..., :3].sum(-1) != 0).sum()), os.path.getsize(path) / 1024.0))

    # PointRenderer: shader/point_{vert,geom,frag}.glsl with the sprite texture (pointrenderer.cpp:54-63, 168-195)
    out = {}
    for name, c in glref_cases.point_cases().items():
        version = glref.init(True, False)
        vp = [0, 0, c["W"], c["H"]]
        mvp = orc.mat4_mul(c["proj"], orc.mat4_inverse(c["cam"]))
        keys, idx = glref.presort(np.ascontiguousarray(c["points"][:, :4]), mvp, c["nf"])
        sk, si = orc.sort(keys, idx)
        img = glref.points_render(c["points"], si, orc.mat4_inverse(c["cam"]), c["proj"], vp, c["sprite"], srgb=c["srgb"],
                                  depth_bits=c["depth_bits"])
        assert (img[..., 3] == 1.0).all()
        out[name + "_rgb"] = img[..., :3].copy()
        out[name + "_order"] = si
        out[name + "_digest"] = glref_cases.digest(c["points"]) + glref_cases.digest(c["sprite"].astype(np.float32))
    path = os.path.join(glref_cases.GOLDEN, "glref_points.npz")
    np.savez_compressed(path, gl_version=version, **out)
    print("%-24s %7.0f KB" % ("points", os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
