#!/usr/bin/env python
"""tools/probe_dump.py <workload> [orbit_step ...] -- dumps the compositor's per-work-item probe (msplat_set_tile_probe)
and the bin-list offsets of single frames to gpurun_out/probe_<workload>.npz for offline analysis."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from splatapult_amd import SplatRenderer, camera, synthetic  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = [int(a) for a in sys.argv[2:]] or [0, 16]
wl = bench.WORKLOADS[key]
W, H = wl["W"], wl["H"]
cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
r = SplatRenderer(device=0, fb_format=wl["fb"])
assert r.Init(cloud, False, False), r.last_error()
r.set_tile_probe(True)
proj = camera.perspective(camera.FOVY, W / H)
vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
out = {}
for s in steps:
    cam = camera.orbit(wl["cam_z"], 2.0 * math.pi * (s % 64) / 64.0)
    for rep in range(3):                # the probe of the last (warm) repetition is kept
        r.Sort(cam, proj, vp, nf)
        r.Render(cam, proj, vp, nf)
    probe = r.debug_tile_probe()
    ts, _ = r.debug_tile_lists()
    st = r.stats()
    w = r.composite_work()
    out["probe_%d" % s] = probe
    out["tile_start_%d" % s] = ts
    out["stats_%d" % s] = np.array([st["sort_count"], st["drawn"], st["pairs"], st["pairs_tile16"], st["tiles_x"], st["tiles_y"]], np.int64)
    ran = probe[:, 7] > 0
    print("step %d: V %d D16 %d Dbin %d | items %d  clocks max %d mean %.0f p99 %.0f | composited/item %.1f fetched/item %.1f list/item %.1f"
          % (s, st["sort_count"], st["pairs_tile16"], st["pairs"], ran.sum(), probe[ran, 0].max(), probe[ran, 0].mean(),
             np.percentile(probe[ran, 0], 99), probe[ran, 1].mean(), probe[ran, 5].mean(), probe[ran, 6].mean()))
    print("   work", w)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "probe_%s.npz" % key), **out)
