#!/usr/bin/env python
"""tools/stream_offset_probe.py <frames in flight> <dummy streams> -- frames/s of config 2 with that many HIP streams created
(and used once) BEFORE the renderer's own streams: shifts which hardware queues the frames' streams are mapped onto."""
import math, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from splatapult_amd import SplatRenderer, camera, synthetic

P, K = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
dummies = [torch.cuda.Stream(device=dev) for _ in range(K)]
x = torch.zeros(1024, device=dev)
for st in dummies:
    with torch.cuda.stream(st):
        x.add_(1.0)
torch.cuda.synchronize()
wl = bench.WORKLOADS["cfg2"]
W, H = wl["W"], wl["H"]
cloud = synthetic.make_cloud(wl["n"], seed=wl["seed"], full_sh=True, pos_sigma=wl["pos_sigma"])
r = SplatRenderer(device=0, fb_format="fp32", frames_in_flight=P)
assert r.Init(cloud, False, False)
fbs = [torch.zeros((1088, W, 4), device=dev) for _ in range(P)]
proj = camera.perspective(camera.FOVY, W / H)
vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
def frame(s):
    cam = camera.orbit(wl["cam_z"], 2.0 * math.pi * (s % 64) / 64.0)
    r.Sort(cam, proj, vp, nf)
    r.Render(cam, proj, vp, nf, out_ptr=fbs[r.frame_slot % P].data_ptr(), pitch_bytes=W * 16)
for s in range(500):
    frame(s)
torch.cuda.synchronize()
rates = []
for rep in range(3):
    t0 = time.perf_counter()
    for s in range(400):
        frame(s)
    torch.cuda.synchronize()
    rates.append(400 / (time.perf_counter() - t0))
print("P=%d dummies=%d  fps %.0f" % (P, K, sorted(rates)[1]))
