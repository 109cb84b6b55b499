#!/usr/bin/env python3
"""usage (GPU box): python tools/stream_priority_probe.py -- frames/s of BASELINE configs[1] with four frames in flight when the four
contexts' streams have DIFFERENT hardware-queue priorities (hipStreamCreateWithPriority through ctypes: -1 high, 0 normal, 1 low).
Question (r6): does a priority order between the streams turn the time-shared launch path into a first-come pipeline?"""
import ctypes as C
import math
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splatapult_amd import SplatRenderer, camera, synthetic  # noqa: E402


def main():
    hip = C.CDLL([ln.rsplit(" ", 1)[-1].strip() for ln in open("/proc/self/maps") if "libamdhip64" in ln][0])
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
    print("stream priority range: least %d, greatest %d" % (lo.value, hi.value))
    W, H, P = 1920, 1080, 4
    dev = torch.device("cuda:0")
    cloud = synthetic.make_cloud(1_000_000, seed=0x5EED1234, full_sh=True, pos_sigma=1.5)
    proj = camera.perspective(camera.FOVY, W / H)
    vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
    poses = [camera.orbit(7.0, 2.0 * math.pi * k / 64.0) for k in range(64)]
    fbs = [torch.zeros((1088, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
    for label, prios in (("library streams", None), ("0 0 0 0", [0, 0, 0, 0]), ("-1 -1 0 0", [-1, -1, 0, 0]), ("-1 0 0 1", [-1, 0, 0, 1]), ("-1 0 1 1", [-1, 0, 1, 1]),
                         ("-1 -1 -1 -1", [-1, -1, -1, -1]), ("library streams", None)):
        streams = None
        if prios is not None:
            streams = []
            for p in prios:
                h = C.c_void_p()
                assert hip.hipStreamCreateWithPriority(C.byref(h), 1, p) == 0          # hipStreamNonBlocking
                streams.append(h.value)
        r = SplatRenderer(device=0, fb_format="fp32", frames_in_flight=P, stream=streams)
        assert r.Init(cloud, False, False), r.last_error()

        def frames(n, first):
            for s in range(n):
                c = poses[(first + s) % 64]
                r.Sort(c, proj, vp, nf)
                r.Render(c, proj, vp, nf, out_ptr=fbs[r.frame_slot].data_ptr(), pitch_bytes=W * 16)
        frames(400, 0)
        r.synchronize(); torch.cuda.synchronize()
        res = []
        for steps in (20, 200):
            blocks = []
            for b in range(40 if steps == 20 else 8):
                t0 = time.perf_counter()
                frames(steps, b * steps)
                r.synchronize(); torch.cuda.synchronize()
                blocks.append(time.perf_counter() - t0)
            res.append(steps / float(np.median(blocks)))
        print("%-16s %6.0f frames/s in 20-frame blocks, %6.0f in 200-frame blocks" % (label, res[0], res[1]))
        r.close()
        if streams:
            for h in streams:
                hip.hipStreamDestroy(C.c_void_p(h))


if __name__ == "__main__":
    main()
