"""feasibility probe: do two frames on two streams overlap?  two independent contexts, alternate frames"""
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
for nctx in (1, 2, 3):
    for cw in (8192, 4096, 2048):
        os.environ["MSPLAT_COMP_WAVES"] = str(cw)
        streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
        rs, fbs = [], []
        for s in streams:
            r = SplatRenderer(device=0, stream=s.cuda_stream)
            assert r.Init(cloud, False, False)
            rs.append(r); fbs.append(torch.zeros((Hpad, W, 4), dtype=torch.float32, device=dev))
        def frame(k):
            c = camera.orbit(7.0, 2.0 * math.pi * (k % 64) / 64.0)
            r = rs[k % nctx]
            r.Sort(c, proj, vp, nf)
            r.Render(c, proj, vp, nf, out_ptr=fbs[k % nctx].data_ptr(), pitch_bytes=W * 16)
        for k in range(12): frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 300
        for k in range(K): frame(12 + k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("ctx %d comp_waves %d: %.1f fps  %.4f ms/frame" % (nctx, cw, K / dt, 1e3 * dt / K), flush=True)
        for r in rs: r.close()
