#!/usr/bin/env python
"""tools/band_rank_profile.py <workload> <world> <rank> <layout kind> <block rows> [frames] -- serial frames of ONE rank's share of a
row-sharded frame (band-culled sort), to be run under `rocprofv3 --kernel-trace --stats` for the rank's per-kernel durations."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    wlk, world, rank, kind, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    nfr = int(sys.argv[6]) if len(sys.argv) > 6 else 100
    import torch
    import bench
    from splatapult_amd import SplatRenderer, camera, synthetic, _capi
    wl = bench.WORKLOADS[wlk]
    W, H = wl["W"], wl["H"]
    cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
    dev = torch.device("cuda", 0)
    r = SplatRenderer(device=0, fb_format=wl["fb"], frames_in_flight=1)
    assert r.Init(cloud, False, False), r.last_error()
    T = _capi.lib().msplat_tile_size()
    R = (H + T - 1) // T
    r.set_band_plan(kind, R, world, rank, block_rows=k, band_cull=True)
    fb = torch.zeros((R * T, W, 4), dtype=torch.float32, device=dev)
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    for s in range(nfr):
        cam = camera.orbit(wl["cam_z"], 2.0 * math.pi * (s % 64) / 64.0)
        r.Sort(cam, proj, vp, nf)
        r.Render(cam, proj, vp, nf, out_ptr=fb.data_ptr(), pitch_bytes=W * 16)
    r.synchronize()
    print("V", r.sort_count(), "pairs", r.stats()["pairs"])


if __name__ == "__main__":
    main()
