import math, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from splatapult_amd import SplatRenderer, camera, synthetic
dev = torch.device("cuda:0")
n, W, H = 1000000, 1920, 1080
cloud = synthetic.make_cloud(n, seed=0x5EED1234, full_sh=True)
Hpad = (H + 31) // 32 * 32
proj = camera.perspective(camera.FOVY, W / H)
vp, nf = [0, 0, W, H], [camera.Z_NEAR, camera.Z_FAR]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
r = SplatRenderer(device=0, frames_in_flight=P)
assert r.Init(cloud, False, False)
fbs = [torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
cams = [camera.orbit(7.0, 2.0 * math.pi * k / 64.0) for k in range(64)]
def frame(k):
    c = cams[k % 64]
    r.Sort(c, proj, vp, nf)
    r.Render(c, proj, vp, nf, out_ptr=fbs[r.frame_slot].data_ptr(), pitch_bytes=W * 16)
SYNC_AT = int(sys.argv[2]) if len(sys.argv) > 2 else -1
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
evs = []
T0 = time.perf_counter()
for k in range(2000):
    if k == SYNC_AT:
        torch.cuda.synchronize()
    frame(k)
    r.wait_on_stream(side.cuda_stream)
    e = torch.cuda.Event(enable_timing=True)
    e.record(side)
    evs.append(e)
torch.cuda.synchronize()
t = [evs[0].elapsed_time(e) for e in evs]
print("P=%d sync@%d GPU completion ms/frame per 50:" % (P, SYNC_AT), " ".join("%.3f" % ((t[i + 50] - t[i]) / 50) for i in range(0, 1950, 50)))
