#!/usr/bin/env python3
"""usage (GPU box): python tools/cu_partition_probe.py [--parts 1,2,1,2] [--frame-mode inflight|serial] [--pool N] [--blocks 20,200]
frames/s of config 2 with len(parts) frames in flight, context k on MSPLAT_CU_* parts[k] (0 all, 1 even, 2 odd CU positions); r6"""
import argparse
import math
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splatapult_amd import SplatRenderer, _capi, camera, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="1,2,1,2")
    ap.add_argument("--frame-mode", default="inflight")
    ap.add_argument("--pool", type=int, default=1280)
    ap.add_argument("--blocks", default="20,200")
    ap.add_argument("--label", default="")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4"])
    a = ap.parse_args()
    parts = [int(x) for x in a.parts.split(",")]
    P = len(parts)
    W, H = (4096, 4096) if a.workload == "cfg4" else (1920, 1080)
    if a.workload == "cfg2":
        cloud, cam_z = synthetic.make_cloud(1_000_000, seed=0x5EED1234, full_sh=True, pos_sigma=1.5), 7.0
    else:
        cloud, cam_z = synthetic.make_cloud(6_000_000, seed=0x5EED6000, full_sh=True, pos_sigma=3.0), 12.0
    fm = _capi.FRAMES_IN_FLIGHT if a.frame_mode == "inflight" else _capi.FRAMES_SERIAL
    r = SplatRenderer(device=0, fb_format="fp32", frames_in_flight=P, compositor_waves=a.pool, cu_partition=parts, frame_mode=fm)
    assert r.Init(cloud, False, False), r.last_error()
    dev = torch.device("cuda:0")
    fbs = [torch.zeros((((H + 31) // 32) * 32, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    poses = [camera.orbit(cam_z, 2.0 * math.pi * k / 64.0) for k in range(64)]
    step = [0]

    def frames(n):
        for _ in range(n):
            c = poses[step[0] % 64]
            step[0] += 1
            r.Sort(c, proj, vp, nf)
            r.Render(c, proj, vp, nf, out_ptr=fbs[r.frame_slot].data_ptr(), pitch_bytes=W * 16)
        r.synchronize()

    frames(400)
    out = []
    for blk in [int(x) for x in a.blocks.split(",")]:
        ts = []
        t_end = time.perf_counter() + 1.5
        while time.perf_counter() < t_end or len(ts) < 5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frames(blk)
            ts.append((time.perf_counter() - t0) / blk)
        out.append("%d-frame blocks %.0f frames/s" % (blk, 1.0 / float(np.median(ts))))
    print("%-44s %s parts %s got %s, %s, pool %d: %s" % (a.label, a.workload, parts, [p for p, _ in r.cu_partitions()], a.frame_mode, a.pool, "; ".join(out)), flush=True)
    r.close()


if __name__ == "__main__":
    main()
