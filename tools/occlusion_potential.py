#!/usr/bin/env python
"""tools/occlusion_potential.py -- how much of a frame's projection and binning work is for pixels that are already opaque?

For one pose of a bench workload: the compositor's probe says how far every (bin, quadrant) item walked its bin's list before
its pixels were saturated.  With the bins' lists (rank-ordered) that gives each bin's STOP RANK -- the largest draw-order rank any of
its four tiles fetched -- and from it what a frame in two passes would have to generate: pass 1 = the ranks below R1 (every bin),
pass 2 = the complete lists of the bins that had not finished by R1.  Prints, per R1, the pairs and the
projected splats such a frame needs, and the floor of a frame with one slab per bin.  GPU box only (analysis, not a test)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
POOL = 1024          # msplat_config.compositor_waves: a wave pool smaller than the item count -- the bins are walked in storage order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3s")
    ap.add_argument("--pose", type=int, default=5)
    args = ap.parse_args()
    import bench
    from splatapult_amd import SplatRenderer, GaussianCloud, camera, synthetic, _capi
    wl = bench.WORKLOADS[args.workload]
    W, H = wl["W"], wl["H"]
    if wl.get("scene"):
        a = synthetic.generate_scene(wl["n"], seed=wl["seed"])
        gc = GaussianCloud()
        assert gc.FromAttributes(a["xyz"], a["f_dc"], a["f_rest"], a["opacity"], a["log_scale"], a["rot"])
        cloud = gc
        cam = synthetic.scene_cameras(64)[args.pose]
    else:
        cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
        cam = camera.orbit(wl["cam_z"], 2.0 * np.pi * args.pose / 64.0)
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    r = SplatRenderer(device=0, spatial_order=_capi.SPATIAL_OFF, compositor_waves=POOL)
    assert r.Init(cloud, False, False), r.last_error()
    r.set_tile_probe(True)
    r.Sort(cam, proj, vp, nf)
    r.Render(cam, proj, vp, nf)
    st = r.stats()
    V, D = int(st["sort_count"]), int(st["pairs"])
    tx, ty = int(st["tiles_x"]), int(st["tiles_y"])
    nb = tx * ty
    ts, pairs = r.debug_tile_lists()
    ts = ts.astype(np.int64)
    rank = (pairs & 0xFFFFFF).astype(np.int64)
    probe = np.zeros((nb * 8, 8), np.uint32)
    _capi.check(r._ctx, r._lib.msplat_debug_get_tile_probe8(r._ctx, probe.ctypes.data_as(C.POINTER(C.c_uint32)), probe.shape[0]))
    # probe slot = work item; item -> (bin, quadrant) as in composite_kernel (persistent waves walk the bins in storage order)
    ni = nb * 4
    t = np.arange(ni)
    slot = np.where(t < (ni & ~31), (t >> 5) * 8 + (t & 7), t >> 2)
    assert (probe[:ni, 7] == 1).all(), "every (bin, quadrant) item ran"
    length = np.diff(ts)
    assert (probe[:ni, 6].astype(np.int64) == length[slot]).all(), "item -> bin map"
    fetched = np.zeros(nb, np.int64)
    np.maximum.at(fetched, slot, probe[:ni, 4].astype(np.int64))          # the deepest walk of the bin's four tiles (incl. prefetch)
    finished = fetched < length                                           # all four tiles stopped before the list's end
    # the lists are stored in draw order (far to near, like the reference's element buffer) and walked from their END (nearest
    # first): the fetched entries are the last `fetched` ones, the deepest fetched entry is the one with the SMALLEST rank
    first = np.clip(ts[1:] - np.minimum(fetched, length), 0, max(D - 1, 0))
    stop = np.where(length > 0, rank[first], V)                           # smallest rank any tile of the bin fetched
    stop = np.where(finished, stop, -1)                                   # unfinished: needs every rank
    need = np.minimum(fetched, length).sum()
    print("%s pose %d: V %d, (splat, bin) pairs D %d, bins %d (%d with a list), finished before their list's end: %d (%.1f %% of the pairs lie in them)"
          % (args.workload, args.pose, V, D, nb, int((length > 0).sum()), int(finished.sum()), 100.0 * length[finished].sum() / max(D, 1)))
    print("one slab per bin (floor): %.1f %% of the pairs are ever fetched" % (100.0 * need / max(D, 1)))
    bin_of = np.repeat(np.arange(nb), length)
    for frac in (0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.6, 0.8):
        R1 = int(frac * V)
        cut = V - R1                                                      # pass 1 = the nearest R1 splats = ranks >= cut
        unfinished = stop < cut                                           # bins that pass 1 does not finish
        front = int((rank >= cut).sum())                                  # pass 1 bins every pair of its ranks
        back = int(length[unfinished].sum())                              # pass 2, ideal: the complete lists of the unfinished bins
        in_unf = unfinished[bin_of]
        touch = np.zeros(V + 1, bool)
        touch[rank[in_unf]] = True                                        # ranks that touch an unfinished bin
        ranks_back = int(touch[:cut].sum())                               # ... among the ranks pass 1 did not project: still projected
        back_rects = int(touch[rank].sum())                               # pass 2, rectangle form: every pair of every rank that touches one
        print("  R1 = %4.0f %% of V: bins finished in pass 1 %5.1f %%; pairs %5.1f %% (pass 1 %5.1f + pass 2 %5.1f; %5.1f with whole rectangles); projected splats %5.1f %% of V"
              % (100 * frac, 100.0 * (~unfinished & (length > 0)).sum() / max((length > 0).sum(), 1), 100.0 * (front + back) / D, 100.0 * front / D,
                 100.0 * back / D, 100.0 * back_rects / D, 100.0 * (R1 + ranks_back) / max(V, 1)))
    r.close()


if __name__ == "__main__":
    main()
