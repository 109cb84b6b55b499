#!/usr/bin/env python3
"""usage (GPU box): python tools/inflight_probe.py [frames_in_flight=4] [frames=240]

Where does a compositor's time go when other frames share the GPU?  The compositor's own probe (composite_kernel<.., PROBE = true>:
shader clocks per work item, and of those the clocks spent in the inner loops) read after (a) frames rendered one at a time and
(b) the same frames with N contexts in flight.  An item's clocks outside its inner loops are list / record fetches, the exact
tile test, the LDS compaction and barriers (memory latency + issue); the inner loop is pure VALU issue.  Prints per mode the sums
over the work items of the last frame and the stretch (b) / (a) of each part: the stall attribution rocprofv3's counters cannot
give for overlapped dispatches (it serialises them)."""
import math
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splatapult_amd import SplatRenderer, camera, synthetic  # noqa: E402


def run(P, frames, cloud, W, H, probe=True, pool=None):
    dev = torch.device("cuda:0")
    r = SplatRenderer(device=0, fb_format="fp32", frames_in_flight=P, compositor_waves=pool)
    assert r.Init(cloud, False, False), r.last_error()
    fbs = [torch.zeros((((H + 31) // 32) * 32, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    if probe:
        r.set_tile_probe(True)
    import time
    t0 = None
    poses = [camera.orbit(7.0, 2.0 * math.pi * k / 64.0) for k in range(64)]      # (a pose costs the host more than a frame's enqueue)
    for s in range(frames):
        if s == frames // 2:
            r.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
        c = poses[s % 64]
        r.Sort(c, proj, vp, nf)
        r.Render(c, proj, vp, nf, out_ptr=fbs[r.frame_slot].data_ptr(), pitch_bytes=W * 16)
    r.synchronize(); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (frames - frames // 2)
    out = None
    if probe:
        pr = r.debug_tile_probe().astype(np.float64)
        out = dict(items=pr.shape[0], clocks=pr[:, 0].sum(), inner=pr[:, 3].sum(), batches=pr[:, 2].sum(), recs=pr[:, 1].sum(),
                   max_item=pr[:, 0].max(), mean_item=pr[:, 0].mean())
    r.close()
    return ms, out


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    W, H = 1920, 1080
    cloud = synthetic.make_cloud(1_000_000, seed=0x5EED1234, full_sh=True, pos_sigma=1.5)
    rows = []
    for label, p, env in (("serial, every item its own wave", 1, None), ("serial, pool of 1280 waves", 1, "1280"), ("%d in flight (pool 1280)" % P, P, None),
                          ("%d in flight, no probe" % P, P, "noprobe")):
        ms, o = run(p, frames, cloud, W, H, probe=(env != "noprobe"), pool=1280 if env == "1280" else None)
        rows.append((label, ms, o))
        if o:
            print("%-36s %.4f ms/frame   items %d  item clocks: sum %.4g (inner loops %.4g = %.1f%%, outside %.4g)  mean %.0f max %.0f  batches %d  records %d"
                  % (label, ms, o["items"], o["clocks"], o["inner"], 100 * o["inner"] / o["clocks"], o["clocks"] - o["inner"], o["mean_item"], o["max_item"],
                     o["batches"], o["recs"]))
        else:
            print("%-36s %.4f ms/frame" % (label, ms))
    a, b = rows[0][2], rows[2][2]
    print("stretch in flight / serial: whole items x%.2f, inner loops x%.2f, outside the inner loops x%.2f"
          % (b["clocks"] / a["clocks"], b["inner"] / a["inner"], (b["clocks"] - b["inner"]) / (a["clocks"] - a["inner"])))
    a = rows[1][2]
    print("stretch in flight / serial with the same pool: whole items x%.2f, inner loops x%.2f, outside x%.2f"
          % (b["clocks"] / a["clocks"], b["inner"] / a["inner"], (b["clocks"] - b["inner"]) / (a["clocks"] - a["inner"])))


if __name__ == "__main__":
    main()
